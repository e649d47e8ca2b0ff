#!/usr/bin/env python3
"""Golden set for OntoEncVAE on the real Pathway Ontology: tests/golden/pwo_encvae.npz.

The encoder masks come from the REFERENCE's own Ontobj.create_masks(module='encoder') / _encoder_masks (ontobj.py:327-404),
lifted by AST and executed unchanged on the trimmed PWO DAG of tools/gen_golden_pwo.py; the model is the reference OntoEncVAE
(vae_model.py:468-905) with `drop` / `z_drop` injected as module globals (tools/gen_golden_encvae.py), neuronnum = 1, weights and
eps injected, dropout off: eval forward, three train_round bodies at batch 128, a val_round body, embedding.
Fixtures hold data only (mask bits, in-mask weights, inputs, outputs).  Runs only in the build container.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, "/root/reference")

import onto_vae.modules as ref_modules                          # noqa: E402
from onto_vae.vae_model import OntoEncVAE as RefOntoEncVAE      # noqa: E402
from gen_golden import EpsQueue, RunRecorder, quiet             # noqa: E402
from gen_golden_pwo import DATA, initialize_dag, reference_methods   # noqa: E402
from gen_golden_encvae import StubOntobj                        # noqa: E402


def main():
    torch.set_num_threads(8)
    annot_base, genes_base, graph_base = initialize_dag(os.path.join(DATA, "pw.obo"), os.path.join(DATA, "gene_term_mapping.txt"))
    obj = reference_methods(("trim_dag", "create_masks", "_decoder_masks", "_encoder_masks"))()
    obj.annot_base, obj.genes_base, obj.graph_base = annot_base, genes_base, graph_base
    obj.annot, obj.graph, obj.genes, obj.desc_genes, obj.masks = {}, {}, {}, {}, {}
    obj.trim_dag(1000, 30)
    obj.create_masks(1000, 30, module="encoder")
    masks = [np.asarray(m).astype(np.int64) for m in obj.masks["1000_30"]["encoder"]]
    G = len(obj.genes["1000_30"])
    shapes = [m.shape for m in masks]
    print("PWO encoder masks:", shapes, "ones", [int(m.sum()) for m in masks])
    assert shapes[0][1] == G == 5892 and sum(s[0] for s in shapes) == 657
    Z = shapes[-1][0]
    fx = {"nmask": np.int64(len(masks)), "G": np.int64(G), "Z": np.int64(Z)}
    for i, m in enumerate(masks):
        fx[f"mask_shape/{i}"] = np.array(m.shape, np.int64)
        fx[f"mask_bits/{i}"] = np.packbits(m.astype(np.uint8), axis=None)
    rng = np.random.RandomState(5)
    B, steps, lr, kc = 128, 3, 1e-3, 1e-4
    sd0 = {}
    for i, m in enumerate(masks[:-1]):
        b = 1.0 / np.sqrt(max(m.sum(axis=1).mean(), 1.0))
        sd0[f"encoder.encoder.{i}.0.weight"] = (rng.uniform(-b, b, m.shape) * m).astype(np.float32)
        sd0[f"encoder.encoder.{i}.0.bias"] = rng.uniform(-0.1, 0.1, m.shape[0]).astype(np.float32)
    m = masks[-1]
    b = 1.0 / np.sqrt(max(m.sum(axis=1).mean(), 1.0))
    for h in ("mu", "logvar"):
        sd0[f"encoder.{h}.0.weight"] = (rng.uniform(-b, b, m.shape) * m * 0.3).astype(np.float32)
        sd0[f"encoder.{h}.0.bias"] = rng.uniform(-0.1, 0.1, Z).astype(np.float32)
    sd0["decoder.decoder.0.0.weight"] = rng.uniform(-0.1, 0.1, (Z, Z)).astype(np.float32)
    sd0["decoder.decoder.0.0.bias"] = rng.uniform(-0.1, 0.1, Z).astype(np.float32)
    sd0["decoder.decoder.0.1.weight"] = (1 + 0.1 * rng.standard_normal(Z)).astype(np.float32)
    sd0["decoder.decoder.0.1.bias"] = (0.1 * rng.standard_normal(Z)).astype(np.float32)
    sd0["decoder.decoder.0.1.running_mean"], sd0["decoder.decoder.0.1.running_var"] = np.zeros(Z, np.float32), np.ones(Z, np.float32)
    sd0["decoder.decoder.0.1.num_batches_tracked"] = np.int64(0)
    sd0["decoder.decoder.1.0.weight"] = rng.uniform(-0.1, 0.1, (G, Z)).astype(np.float32)
    sd0["decoder.decoder.1.0.bias"] = rng.uniform(-0.1, 0.1, G).astype(np.float32)

    def pack(sd, prefix):
        """in-mask values only for the masked weights (row-major order of the mask), everything else as it is"""
        names = [f"encoder.encoder.{i}.0.weight" for i in range(len(masks) - 1)] + ["encoder.mu.0.weight", "encoder.logvar.0.weight"]
        mk = dict(zip(names, masks[:-1] + [masks[-1], masks[-1]]))
        for k, v in sd.items():
            v = np.asarray(v)
            if k in mk:
                assert not v[mk[k] == 0].any(), k
                fx[prefix + k] = v[mk[k] != 0]
            else:
                fx[prefix + k] = v

    pack(sd0, "sd0/")
    xs = [np.log1p(rng.poisson(0.3, size=(B, G))).astype(np.float32) for _ in range(steps)]
    epss = [rng.standard_normal(size=(B, Z)).astype(np.float32) for _ in range(steps)]
    for s in range(steps):
        fx[f"x/{s}"], fx[f"eps/{s}"] = xs[s], epss[s]
    ref_modules.drop, ref_modules.z_drop = 0.0, 0.0
    torch.manual_seed(0)
    onto = StubOntobj(masks, np.zeros((4, G), np.float32))
    model = RefOntoEncVAE(onto, "d", neuronnum=1, drop=0.0, z_drop=0.0)
    model.load_state_dict({k: torch.tensor(v) for k, v in sd0.items()})
    epsq = EpsQueue()
    model.reparameterize = epsq
    model.eval()
    epsq.push(epss[0])
    with torch.no_grad():
        recon, mu, lv = model.forward(torch.tensor(xs[0]))
    fx["fwd/mu"], fx["fwd/log_var"], fx["fwd/recon_cols"] = mu.numpy(), lv.numpy(), recon.numpy()[:, ::97]
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    run = RunRecorder()
    losses = []
    for s in range(steps):
        epsq.push(epss[s])
        losses.append(quiet(model.train_round, [(torch.tensor(xs[s]),)], lr, kc, opt, run))
    fx["train/loss"], fx["train/rec"], fx["train/kl"] = np.array(losses), np.array(run["metrics/train/rec_loss"]), np.array(run["metrics/train/kl_loss"])
    pack({k: v.detach().numpy() for k, v in model.state_dict().items() if not k.startswith("decoder.decoder.1.0.weight")}, "sdT/")
    fx["sdT_rows/decoder.decoder.1.0.weight"] = model.state_dict()["decoder.decoder.1.0.weight"].numpy()[::97]
    epsq.push(epss[1])
    runv = RunRecorder()
    vloss = quiet(model.val_round, [(torch.tensor(xs[1]),)], kc, runv)
    fx["val/rec_kl_loss"] = np.array([runv["metrics/val/rec_loss"][0], runv["metrics/val/kl_loss"][0], vloss])
    fx["lr"], fx["kl_coeff"], fx["steps"] = np.float64(lr), np.float64(kc), np.int64(steps)
    path = os.path.join(ROOT, "tests", "golden", "pwo_encvae.npz")
    np.savez_compressed(path, **fx)
    print(f"-> {path} ({os.path.getsize(path)/1e6:.2f} MB); losses", losses)


if __name__ == "__main__":
    main()
