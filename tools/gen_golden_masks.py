#!/usr/bin/env python3
"""Golden vectors for onto_vae_amd/mask_builder.py, made by the REFERENCE's own mask code.

onto_vae/ontobj.py cannot be imported here (it imports goatools at module level, which this image lacks), so the two
methods that build the masks, Ontobj.create_masks, Ontobj._decoder_masks and Ontobj._encoder_masks, are lifted out of the reference file by
their AST nodes and executed unchanged on a minimal object; utils.create_binary_matrix is imported normally.  Nothing
of the reference is written into the repository: the fixture holds the synthetic graph (inputs) and the masks the
reference code produced (expected outputs).

Usage: python tools/gen_golden_masks.py [--out tests/golden]
"""
import argparse
import ast
import itertools
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from onto_vae.utils import create_binary_matrix          # noqa: E402  (the reference)


def reference_mask_methods():
    src = open("/root/reference/onto_vae/ontobj.py").read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Ontobj")
    wanted = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("create_masks", "_decoder_masks", "_encoder_masks")]
    mod = ast.Module(body=[ast.ClassDef(name="RefMaskCode", bases=[], keywords=[], body=wanted, decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(mod)
    env = {"np": np, "pd": pd, "itertools": itertools, "create_binary_matrix": create_binary_matrix}
    exec(compile(mod, "ontobj.py (mask methods)", "exec"), env)
    return env["RefMaskCode"]


def synthetic_graph(seed, n_terms, n_genes, max_depth):
    """A random DAG with string IDs whose sorted order differs from creation order, depths = longest path to a root."""
    rng = np.random.RandomState(seed)
    names = [f"T:{rng.randint(0, 10**6):06d}_{i}" for i in range(n_terms)]
    parents = {}
    depth = {}
    for i, t in enumerate(names):
        if i < 3 or rng.rand() < 0.1:
            parents[t] = []
            depth[t] = 0
            continue
        cands = [u for u in names[:i] if depth[u] < max_depth]
        k = 1 + rng.poisson(0.7)
        ps = list(dict.fromkeys(cands[j] for j in rng.randint(0, len(cands), size=k)))
        parents[t] = ps
        depth[t] = 1 + max(depth[u] for u in ps)
    genes = sorted(f"G{rng.randint(0, 10**5):05d}_{i}" for i in range(n_genes))
    for g in genes:
        k = 1 + rng.poisson(1.5)
        parents[g] = list(dict.fromkeys(names[j] for j in rng.randint(0, n_terms, size=k)))
    # a parent the trimming removed, and a duplicated parent entry: both occur in real graphs
    parents[genes[0]] = parents[genes[0]] + ["T:removed"]
    parents[genes[1]] = parents[genes[1]] + parents[genes[1]][:1]
    annot = pd.DataFrame({"ID": names, "depth": [depth[t] for t in names]})
    return parents, annot, genes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    Ref = reference_mask_methods()
    for name, (seed, nt, ng, md) in {"graph_small": (3, 40, 25, 4), "graph_deep": (7, 120, 60, 7)}.items():
        graph, annot, genes = synthetic_graph(seed, nt, ng, md)
        obj = Ref()
        obj.graph, obj.annot, obj.genes, obj.masks = {"1000_30": graph}, {"1000_30": annot}, {"1000_30": genes}, {}
        obj.create_masks(1000, 30, "decoder")
        masks = obj.masks["1000_30"]["decoder"]
        fx = {"graph_json": np.frombuffer(json.dumps(graph).encode(), dtype=np.uint8),
              "annot_ids": np.array(annot["ID"].tolist()), "annot_depth": annot["depth"].to_numpy().astype(np.int64),
              "genes": np.array(genes), "nmask": np.int64(len(masks))}
        for i, m in enumerate(masks):
            fx[f"mask{i}"] = np.asarray(m).astype(np.uint8)
        # the encoder masks of OntoEncVAE from the same graph (Ontobj.create_masks' else branch, _encoder_masks)
        obj.create_masks(1000, 30, "encoder")
        emasks = obj.masks["1000_30"]["encoder"]
        for i, m in enumerate(emasks):
            fx[f"emask{i}"] = np.asarray(m).astype(np.uint8)
        print("   encoder:", [np.asarray(m).shape for m in emasks])
        path = os.path.join(a.out, f"masks_{name}.npz")
        np.savez_compressed(path, **fx)
        print(path, [m.shape for m in masks], int(sum(np.asarray(m).sum() for m in masks)), "links")


if __name__ == "__main__":
    main()
