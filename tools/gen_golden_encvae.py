#!/usr/bin/env python3
"""Generate tests/golden/encvae_*.npz by running the REFERENCE OntoEncVAE (imported from /root/reference,
onto_vae/vae_model.py:468-905) on seeded inputs.  Runs only in the build container; fixtures hold data only.

The reference class does not construct as it stands: OntoEncoder.__init__ reads `drop` and `z_drop`, which are neither
arguments nor attributes (onto_vae/modules.py:190-191).  The two names are injected as module globals of onto_vae.modules
before the unmodified class is instantiated; neuronnum = 1 is the one value for which the encoder's latent width
(modules.py:189) and the Decoder's (vae_model.py:501) agree.  Dropout is off (0, 0); weights and eps are injected as in
tools/gen_golden.py.

Usage: python tools/gen_golden_encvae.py [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, "/root/reference")

import onto_vae.modules as ref_modules                          # noqa: E402  (the reference)
from onto_vae.vae_model import OntoEncVAE as RefOntoEncVAE      # noqa: E402
from gen_golden import EpsQueue, RunRecorder, quiet             # noqa: E402
from onto_vae_amd.synthetic import synthetic_links             # noqa: E402


def encoder_masks(counts, G, seed):
    """Encoder masks in the layout of Ontobj._encoder_masks (ontobj.py:394-404) for a synthetic DAG: level 0 of the encoder is
    the genes, then the terms from the deepest depth up to the roots; mask j has the terms of encoder level j + 1 as rows and
    [level j | ... | level 0] as columns, 1 where the column node has the row term as a parent."""
    rng = np.random.RandomState(seed)
    off, term_parents, gene_parents = synthetic_links(counts, G, rng)
    K = len(counts)
    levels = [("g", list(range(G)))] + [("t", list(range(off[d], off[d + 1]))) for d in range(K - 1, -1, -1)]
    parents = lambda kind, i: gene_parents[i] if kind == "g" else term_parents[i]
    masks = []
    for j in range(K):
        kind_r, rows = levels[j + 1]
        rpos = {t: r for r, t in enumerate(rows)}
        blocks = []
        for lv in range(j, -1, -1):
            kind_c, cols = levels[lv]
            m = np.zeros((len(rows), len(cols)), dtype=np.int64)
            for c, node in enumerate(cols):
                for p in parents(kind_c, node):
                    if p in rpos:
                        m[rpos[p], c] = 1
            blocks.append(m)
        masks.append(np.hstack(blocks))
    return masks


class StubOntobj:
    def __init__(self, masks, X):
        G = masks[0].shape[1]
        self.description = "synthetic"
        self.genes = {"1000_30": [f"G{i:06d}" for i in range(G)]}
        self.masks = {"1000_30": {"encoder": masks}}
        self.data = {"1000_30": {"d": X}}


def random_sd(masks, dec, seed):
    rng = np.random.RandomState(seed)
    G, Z = masks[0].shape[1], masks[-1].shape[0]
    sd = {}
    for i, m in enumerate(masks[:-1]):
        b = 1.0 / np.sqrt(m.shape[1])
        sd[f"encoder.encoder.{i}.0.weight"] = (rng.uniform(-b, b, m.shape) * m).astype(np.float32)      # signed: no clamp at construction
        sd[f"encoder.encoder.{i}.0.bias"] = rng.uniform(-b, b, m.shape[0]).astype(np.float32)
    m = masks[-1]
    b = 1.0 / np.sqrt(m.shape[1])
    for h in ("mu", "logvar"):
        sd[f"encoder.{h}.0.weight"] = (rng.uniform(-b, b, m.shape) * m).astype(np.float32)
        sd[f"encoder.{h}.0.bias"] = rng.uniform(-b, b, m.shape[0]).astype(np.float32)
    cin = Z
    for i, w in enumerate(dec):
        b = 1.0 / np.sqrt(cin)
        sd[f"decoder.decoder.{i}.0.weight"] = rng.uniform(-b, b, (w, cin)).astype(np.float32)
        sd[f"decoder.decoder.{i}.0.bias"] = rng.uniform(-b, b, w).astype(np.float32)
        sd[f"decoder.decoder.{i}.1.weight"] = (1 + 0.1 * rng.standard_normal(w)).astype(np.float32)
        sd[f"decoder.decoder.{i}.1.bias"] = (0.1 * rng.standard_normal(w)).astype(np.float32)
        sd[f"decoder.decoder.{i}.1.running_mean"] = np.zeros(w, np.float32)
        sd[f"decoder.decoder.{i}.1.running_var"] = np.ones(w, np.float32)
        sd[f"decoder.decoder.{i}.1.num_batches_tracked"] = np.int64(0)
        cin = w
    n = len(dec)
    b = 1.0 / np.sqrt(cin)
    sd[f"decoder.decoder.{n}.0.weight"] = rng.uniform(-b, b, (G, cin)).astype(np.float32)
    sd[f"decoder.decoder.{n}.0.bias"] = rng.uniform(-b, b, G).astype(np.float32)
    return sd


def gen_case(name, counts, G, seed, B, steps, out_dir, lr=1e-3, kl_coeff=1e-4):
    masks = encoder_masks(counts, G, seed)
    Z = masks[-1].shape[0]
    rng = np.random.RandomState(seed + 7)
    sd0 = random_sd(masks, [Z], seed + 1)
    xs = [np.log1p(rng.poisson(0.8, size=(B, G))).astype(np.float32) for _ in range(steps)]
    epss = [rng.standard_normal(size=(B, Z)).astype(np.float32) for _ in range(steps)]
    fx = {"counts": np.array(counts, np.int64), "G": np.int64(G), "Z": np.int64(Z), "lr": np.float64(lr), "kl_coeff": np.float64(kl_coeff),
          "steps": np.int64(steps), "n_masks": np.int64(len(masks))}
    for i, m in enumerate(masks):
        fx[f"mask/{i}"] = np.packbits(m.astype(np.uint8), axis=1)
        fx[f"mask_shape/{i}"] = np.array(m.shape, np.int64)
    for k, v in sd0.items():
        fx[f"sd0/{k}"] = v
    for s in range(steps):
        fx[f"x/{s}"], fx[f"eps/{s}"] = xs[s], epss[s]

    ref_modules.drop, ref_modules.z_drop = 0.0, 0.0          # the two free names of OntoEncoder.__init__ (modules.py:190-191)

    def build():
        torch.manual_seed(0)
        onto = StubOntobj(masks, np.zeros((4, G), np.float32))
        m = RefOntoEncVAE(onto, "d", neuronnum=1, drop=0.0, z_drop=0.0)
        m.load_state_dict({k: torch.tensor(v) for k, v in sd0.items()})
        q = EpsQueue()
        m.reparameterize = q
        return m, onto, q

    model, onto, epsq = build()
    assert sorted(model.state_dict().keys()) == sorted(sd0.keys()), sorted(set(model.state_dict().keys()) ^ set(sd0.keys()))
    model.eval()
    epsq.push(epss[0])
    with torch.no_grad():
        x0 = torch.tensor(xs[0])
        recon, mu, lv = model.forward(x0)
    fx["fwd/recon"], fx["fwd/mu"], fx["fwd/log_var"] = recon.numpy(), mu.numpy(), lv.numpy()

    model, onto, epsq = build()
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    run = RunRecorder()
    losses = []
    for s in range(steps):
        epsq.push(epss[s])
        losses.append(quiet(model.train_round, [(torch.tensor(xs[s]),)], lr, kl_coeff, opt, run))
        if s == 0:
            for n, p_ in model.named_parameters():
                fx[f"grad0/{n}"] = p_.grad.detach().numpy().copy()
    fx["train/loss"] = np.array(losses)
    fx["train/rec"] = np.array(run["metrics/train/rec_loss"])
    fx["train/kl"] = np.array(run["metrics/train/kl_loss"])
    for k, v in model.state_dict().items():
        fx[f"sdT/{k}"] = v.detach().numpy().copy()
    xv = np.log1p(rng.poisson(0.8, size=(9, G))).astype(np.float32)
    ev = rng.standard_normal(size=(9, Z)).astype(np.float32)
    ev2 = rng.standard_normal(size=(9, Z)).astype(np.float32)
    epsq.push(ev)
    runv = RunRecorder()
    vloss = quiet(model.val_round, [(torch.tensor(xv),)], kl_coeff, runv)
    fx["val/x"], fx["val/eps"], fx["val/eps2"] = xv, ev, ev2
    fx["val/rec_kl_loss"] = np.array([runv["metrics/val/rec_loss"][0], runv["metrics/val/kl_loss"][0], vloss])
    model.eval()
    epsq.push(ev)
    with torch.no_grad():
        fx["val/embedding"] = model.get_embedding(torch.tensor(xv)).numpy()
    # retrieval on the trained weights: _pass_data draws eps twice (get_embedding, then forward)
    onto.data["1000_30"]["v"] = xv
    epsq.push(ev); epsq.push(ev2)
    fx["val/act"] = model.get_pathway_activities(onto, "v")
    epsq.push(ev); epsq.push(ev2)
    fx["val/rec"] = model.get_reconstructed_values(onto, "v")
    path = os.path.join(out_dir, f"{name}.npz")
    np.savez_compressed(path, **fx)
    print(f"{name}: OntoEncVAE counts={counts} G={G} Z={Z} masks={[m.shape for m in masks]} steps={steps} -> {path} ({os.path.getsize(path)/1e3:.0f} kB)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    gen_case("encvae_small", (6, 14, 25), 90, 3, 24, 6, a.out)
    gen_case("encvae_deep", (5, 9, 17, 30, 44), 260, 4, 40, 4, a.out)
