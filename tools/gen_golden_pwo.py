#!/usr/bin/env python3
"""Golden set G2: the REAL Pathway Ontology of the reference's data directory
(/root/reference/onto_vae/data/pw.obo + gene_term_mapping.txt), trimmed at 1000 / 30 as the vignette does, and the
reference OntoVAE trained on it (BASELINE.json configs[0]: N = 1000, batch 128, 5 epochs).

How the ontology gets here.  onto_vae/ontobj.py imports goatools, which this image lacks, so the module cannot be
imported and nothing here stands in for goatools inside it.  Instead:

  * `read_obo` below is this repository's own OBO reader (id / name / is_a / alt_id / is_obsolete of [Term] stanzas,
    depth = longest is_a path to a root, which is what goatools' GOTerm.depth holds);
  * `initialize_dag` below restates Ontobj.initialize_dag (ontobj.py:129-223) on that reader's output: the annotation
    table sorted by (depth, ID), descendant-gene counts, removal of terms without descendant genes, graph_base;
  * Ontobj.trim_dag, Ontobj.create_masks and Ontobj._decoder_masks (ontobj.py:226-392) are the REFERENCE's own code:
    lifted out of the file by their AST nodes and executed unchanged on a plain object (they only need pandas / numpy
    and onto_vae.utils, which imports normally), as tools/gen_golden_masks.py does for the synthetic graphs.

What pins the restated part: the reference's own recorded outputs.  vignette.ipynb cell adca4442 prints ten rows of the
result table with their row index in annot (sorted by depth, then ID), term ID and depth; all ten must come out, and the
counts SURVEY.md 0.3-0.5 recorded from the reference (657 terms, 5 892 genes, terms per depth, mask shapes and ones).

The fixture (tests/golden/pwo_g2.npz) holds data only: term IDs / depths, gene names, the trimmed graph as index pairs,
the reference's masks as packed bits, and what the reference model produced (per-step rec / kl, epoch losses, in-mask
decoder weights, small parameters, rows of W0, activities of 32 samples before and after training).  Inputs are regenerated from numpy seeds.

Usage: python tools/gen_golden_pwo.py [--out tests/golden]
"""
import argparse
import ast
import contextlib
import io
import itertools
import os
import sys
import tempfile
from collections import Counter

import numpy as np
import pandas as pd
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import onto_vae.utils as ref_utils                                  # noqa: E402  (the reference)
from onto_vae.vae_model import OntoVAE as RefOntoVAE               # noqa: E402  (the reference)
from onto_vae_amd.synthetic import StubOntobj, random_state_dict   # noqa: E402
from gen_golden import EpsQueue, RunRecorder, quiet                 # noqa: E402

DATA = "/root/reference/onto_vae/data"
# vignette.ipynb cell adca4442 (recorded output of the reference): row index in annot['1000_30'], ID, depth
VIGNETTE_ROWS = [(426, "PW:0000817", 5), (425, "PW:0000816", 5), (298, "PW:0000925", 4), (424, "PW:0000814", 5),
                 (45, "PW:0000818", 0), (6, "PW:0000024", 0), (557, "PW:0000233", 6), (78, "PW:0000926", 1),
                 (562, "PW:0000516", 6), (571, "PW:0000883", 6)]


def read_obo(path):
    """{id: {'name', 'parents' (is_a ids, file order)}} of the non-obsolete [Term] stanzas, plus depth."""
    terms, cur, in_term = {}, None, False
    for line in open(path, encoding="utf-8"):
        line = line.rstrip("\n")
        if line.startswith("["):
            if cur is not None and not cur.pop("obsolete"):
                terms[cur["id"]] = cur
            in_term = line.strip() == "[Term]"
            cur = {"id": None, "name": "", "parents": [], "obsolete": False} if in_term else None
            continue
        if not in_term or cur is None or ":" not in line:
            continue
        key, val = line.split(":", 1)
        val = val.split(" ! ")[0].strip()
        if key == "id":
            cur["id"] = val
        elif key == "name":
            cur["name"] = val
        elif key == "is_a":
            cur["parents"].append(val)
        elif key == "is_obsolete" and val == "true":
            cur["obsolete"] = True
    if cur is not None and not cur.pop("obsolete"):
        terms[cur["id"]] = cur
    for t in terms.values():
        t["parents"] = [p for p in t["parents"] if p in terms]
    depth = {}

    def d(t):
        if t not in depth:
            ps = terms[t]["parents"]
            depth[t] = 0 if not ps else 1 + max(d(p) for p in ps)
        return depth[t]

    sys.setrecursionlimit(10000)
    for t in terms:
        terms[t]["depth"] = d(t)
    children = Counter(p for t in terms.values() for p in t["parents"])
    for t in terms:
        terms[t]["children"] = children.get(t, 0)
    return terms


def initialize_dag(obo, gene_annot_path):
    """Restatement of Ontobj.initialize_dag (ontobj.py:129-223): returns (annot_base, genes_base, graph_base)."""
    dag = read_obo(obo)
    gene_annot = pd.read_csv(gene_annot_path, sep="\t", header=None)
    gene_annot.columns = ["Gene", "ID"]
    ids = list(dag)
    annot = pd.DataFrame({"ID": ids, "Name": [dag[t]["name"] for t in ids], "depth": [dag[t]["depth"] for t in ids],
                          "children": [dag[t]["children"] for t in ids], "parents": [len(dag[t]["parents"]) for t in ids]})
    annot = annot.sort_values(["depth", "ID"])
    gene_term = {a: b["ID"].tolist() for a, b in gene_annot.groupby("Gene")}
    idset = set(annot.ID)
    term_term = {t: [p for p in dag[t]["parents"] if p in idset] for t in annot[annot.depth > 0].ID}
    kids = ref_utils.reverse_graph(term_term)
    genes_of = ref_utils.reverse_graph(gene_term)
    n_desc, n_genes = [], []
    for t in annot.ID:
        desc = set(ref_utils.get_descendants(kids, t))
        n_desc.append(len(desc) - 1)
        n_genes.append(len({g for u in desc for g in genes_of.get(u, ())}))
    annot["descendants"], annot["desc_genes"] = n_desc, n_genes
    upd = annot[annot.desc_genes > 0].sort_values(["depth", "ID"]).reset_index(drop=True)
    keep = set(upd.ID)
    term_dict = {t: [p for p in dag[t]["parents"] if p in keep] for t in upd[upd.depth > 0].ID}
    size = gene_annot["ID"].value_counts()
    upd["genes"] = [float(size.get(t, 0)) for t in upd.ID]
    n_child = Counter(p for ps in term_dict.values() for p in ps)
    upd["children"] = [n_child.get(t, 0) for t in upd.ID]
    kids = ref_utils.reverse_graph(term_dict)
    upd["descendants"] = [len(set(ref_utils.get_descendants(kids, t))) - 1 for t in upd.ID]
    graph = dict(term_dict)
    graph.update(gene_term)
    return upd, sorted(set(gene_annot.Gene.tolist())), graph


def reference_methods(names):
    src = open("/root/reference/onto_vae/ontobj.py").read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "Ontobj")
    wanted = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(wanted) == len(names)
    mod = ast.Module(body=[ast.ClassDef(name="RefOntobjCode", bases=[], keywords=[], body=wanted, decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(mod)
    env = {"np": np, "pd": pd, "itertools": itertools, "contextlib": contextlib, "io": io}
    env.update({k: getattr(ref_utils, k) for k in dir(ref_utils) if not k.startswith("_")})
    exec(compile(mod, "ontobj.py (trim_dag, create_masks)", "exec"), env)
    return env["RefOntobjCode"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    torch.set_num_threads(8)
    annot_base, genes_base, graph_base = initialize_dag(os.path.join(DATA, "pw.obo"), os.path.join(DATA, "gene_term_mapping.txt"))
    obj = reference_methods(("trim_dag", "create_masks", "_decoder_masks"))()
    obj.annot_base, obj.genes_base, obj.graph_base = annot_base, genes_base, graph_base
    obj.annot, obj.graph, obj.genes, obj.desc_genes, obj.masks = {}, {}, {}, {}, {}
    obj.trim_dag(1000, 30)
    obj.create_masks(1000, 30)
    key = "1000_30"
    annot, genes, graph, masks = obj.annot[key], obj.genes[key], obj.graph[key], obj.masks[key]["decoder"]
    # ---- pinned by the reference's recorded outputs
    for row, tid, depth in VIGNETTE_ROWS:
        assert annot.ID[row] == tid and int(annot.depth[row]) == depth, (row, tid, depth, annot.ID[row], annot.depth[row])
    assert len(annot) == 657 and len(genes) == 5892
    per_depth = [int((annot.depth == d).sum()) for d in range(8)]
    assert per_depth == [76, 3, 12, 131, 169, 160, 101, 5], per_depth
    shapes = [m.shape for m in masks]
    ones = [int(np.asarray(m).sum()) for m in masks]
    assert shapes == [(3, 76), (12, 79), (131, 91), (169, 222), (160, 391), (101, 551), (5, 652), (5892, 657)], shapes
    assert ones == [3, 12, 144, 200, 186, 121, 7, 24409], ones
    print("PWO 1000_30:", len(annot), "terms,", len(genes), "genes, per depth", per_depth, "ones", ones)

    ids = annot.ID.tolist()
    index = {t: i for i, t in enumerate(ids)}
    index.update({g: len(ids) + i for i, g in enumerate(genes)})
    child, parent, dropped = [], [], 0
    for c, ps in graph.items():
        for p in ps:
            if c in index and p in index:
                child.append(index[c]); parent.append(index[p])
            else:
                dropped += 1
    fx = {"term_ids": np.array(ids), "term_names": np.array(annot.Name.tolist()), "term_depth": annot.depth.to_numpy().astype(np.int64),
          "genes": np.array(genes), "link_child": np.array(child, np.int32), "link_parent": np.array(parent, np.int32),
          "links_outside": np.int64(dropped), "nmask": np.int64(len(masks)),
          "vignette_rows": np.array([r for r, _, _ in VIGNETTE_ROWS], np.int64),
          "vignette_ids": np.array([t for _, t, _ in VIGNETTE_ROWS]), "vignette_depths": np.array([d for _, _, d in VIGNETTE_ROWS], np.int64)}
    for i, m in enumerate(masks):
        m = np.asarray(m)
        assert m.dtype == np.int64 and set(np.unique(m)) <= {0, 1}
        fx[f"mask_shape/{i}"] = np.array(m.shape, np.int64)
        fx[f"mask_bits/{i}"] = np.packbits(m.astype(np.uint8), axis=None)

    # ---- configs[0] on the real masks, by the reference: train_model(batch_size=128, epochs=5), N = 1000
    nu, N, B, epochs, lr, kc = 3, 1000, 128, 5, 1e-4, 1e-4
    G, L = len(genes), 76 * nu
    masks64 = [np.asarray(m).astype(np.int64) for m in masks]
    sd0 = random_state_dict(masks64, nu, 2)
    rng = np.random.RandomState(1)
    X = np.log1p(rng.poisson(0.3, size=(N, G))).astype(np.float64)
    n_tr = round(N * 0.8)
    sizes = ([B] * (n_tr // B) + ([n_tr % B] if n_tr % B else []), [B] * ((N - n_tr) // B) + ([(N - n_tr) % B] if (N - n_tr) % B else []))
    onto = StubOntobj(masks64, data={"d": X}, description="PWO", gene_names=genes, term_ids=ids)
    torch.manual_seed(0)
    model = RefOntoVAE(onto, "d", neuronnum=nu, drop=0.0, z_drop=0.0)
    assert model.latent_dim == 228 and model.decoder.layer_shapes == [(228, 9), (237, 36), (273, 393), (666, 507), (1173, 480),
                                                                        (1653, 303), (1956, 15), (1971, 5892)]
    assert sum(int(m.sum()) for m in model.decoder.masks) == 79284
    model.load_state_dict({k: torch.tensor(v) for k, v in sd0.items()})
    epsq = EpsQueue()
    model.reparameterize = epsq
    sched_eps = [rng.standard_normal((b, L)).astype(np.float32) for ep in range(epochs) for b in sizes[0] + sizes[1]]
    n_inf = 32
    e1 = rng.standard_normal((n_inf, L)).astype(np.float32)
    e2 = rng.standard_normal((n_inf, L)).astype(np.float32)
    # activities of the first 32 samples on the injected weights (get_pathway_activities, vae_model.py:330-361)
    onto.add_dataset("inf", X[:n_inf])
    epsq.push(e1); epsq.push(e2)
    fx["g2/act0"] = model.get_pathway_activities(onto, "inf")
    epsq.push(e1); epsq.push(e2)
    fx["g2/rec0"] = model.get_reconstructed_values(onto, "inf", rec_genes=genes[:64])
    for e in sched_eps:
        epsq.push(e)
    run = RunRecorder()
    torch.manual_seed(7)
    with tempfile.TemporaryDirectory() as tmp:
        quiet(model.train_model, os.path.join(tmp, "m.pt"), lr=lr, kl_coeff=kc, batch_size=B, epochs=epochs, run=run)
    assert not epsq.q
    fx.update({"g2/seed_weights": np.int64(2), "g2/seed_data": np.int64(1), "g2/seed_torch": np.int64(7), "g2/N": np.int64(N),
               "g2/B": np.int64(B), "g2/epochs": np.int64(epochs), "g2/lr": np.float64(lr), "g2/kl_coeff": np.float64(kc)})
    for k in ("metrics/train/rec_loss", "metrics/train/kl_loss", "metrics/val/rec_loss", "metrics/val/kl_loss",
              "metrics/train/loss", "metrics/val/loss"):
        fx["g2/" + k] = np.array(run[k], np.float64)
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    for i, m in enumerate(model.decoder.masks):
        nz = np.nonzero(m.numpy())
        fx[f"g2/dec_vals/{i}"] = sd[f"decoder.decoder.{i}.0.weight"][nz]            # in-mask weights, row-major order of the mask
        fx[f"g2/dec_bias/{i}"] = sd[f"decoder.decoder.{i}.0.bias"]
        assert float(np.abs(sd[f"decoder.decoder.{i}.0.weight"] * (1 - m.numpy())).max()) == 0.0
    for k in ("encoder.encoder.0.0.bias", "encoder.encoder.0.1.weight", "encoder.encoder.0.1.bias", "encoder.encoder.0.1.running_mean",
              "encoder.encoder.0.1.running_var", "encoder.encoder.0.1.num_batches_tracked", "encoder.mu.0.weight", "encoder.mu.0.bias",
              "encoder.logvar.0.weight", "encoder.logvar.0.bias"):
        fx["g2/sdT/" + k] = sd[k]
    fx["g2/W0_rows"] = np.array([0, 113, 227], np.int64)
    fx["g2/W0_rows_values"] = sd["encoder.encoder.0.0.weight"][[0, 113, 227]]
    # the same on the trained weights
    epsq.push(e1); epsq.push(e2)
    fx["g2/act"] = model.get_pathway_activities(onto, "inf")
    epsq.push(e1); epsq.push(e2)
    fx["g2/act_terms"] = model.get_pathway_activities(onto, "inf", terms=[t for _, t, _ in VIGNETTE_ROWS])
    assert fx["g2/act"].shape == (n_inf, 657)
    path = os.path.join(a.out, "pwo_g2.npz")
    np.savez_compressed(path, **fx)
    print(f"-> {path} ({os.path.getsize(path) / 1e3:.0f} kB); train losses per epoch {run['metrics/train/loss']}")


if __name__ == "__main__":
    main()
