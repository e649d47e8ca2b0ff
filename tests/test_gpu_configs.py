"""GPU parity at the sizes the benchmarks run (BASELINE.json configs[0..3]): the HIP path through the
C ABI against the dense CPU oracle on the same seeded inputs.

  configs[0]  PWO-shaped model (G = 5892, T = 657, K = 8), train_model(batch_size=128, epochs=5), per-step losses
  configs[1]  GOBP-syn, B = 128: tests/test_gpu_parity.py::test_gobp_syn_full_size_vs_oracle
  configs[2]  GOBP-syn train step at B = 1024 / 4096 / 8192 (per-GPU and global batch sizes of the data-parallel runs),
              and the two-process data-parallel step at global batches that select each exchange form by the heuristic
  configs[3]  GOBP-syn get_pathway_activities / get_reconstructed_values in 4096- and 16384-row batches

Tolerances: losses 1e-4 relative (north_star); raw gradients 2e-5 of each tensor's scale; activities per
conftest.act_close (floor 1e-2 of scale, from the float64 evidence in the golden fixtures).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import act_close
from test_gpu_parity import REL, check_params, feed, hip_state, make_model, medium

pytestmark = pytest.mark.gpu

_GOBP = {}


def gobp():
    """GOBP-syn masks, state_dict and 2048 expression rows (SURVEY.md 8(d) generator), built once per session."""
    if not _GOBP:
        from onto_vae_amd.synthetic import synthetic_masks, synthetic_expression, random_state_dict
        masks, rng = synthetic_masks(return_rng=True)
        _GOBP["masks"] = masks
        _GOBP["X"] = synthetic_expression(rng, 2048, 20000).astype(np.float32)
        _GOBP["sd"] = random_state_dict(masks, 3, 77, dense=False)
    return _GOBP["masks"], _GOBP["sd"], _GOBP["X"]


def grads_of(e):
    G = e.Gr
    got = {"encoder.encoder.0.0.weight": e.view("W0", G), "encoder.encoder.0.1.weight": e.view("gamma", G),
           "encoder.encoder.0.1.bias": e.view("beta", G), "encoder.mu.0.weight": e.view("Wmu", G),
           "encoder.mu.0.bias": e.view("bmu", G), "encoder.logvar.0.weight": e.view("Wlv", G),
           "encoder.logvar.0.bias": e.view("blv", G)}
    for i in range(e.plan.K):
        got[f"decoder.decoder.{i}.0.bias"] = e.layer_bias(i, G)
    return got


def packed_layer(e, i, dense):
    """The in-mask entries of a dense [out, in] decoder matrix, in packed order."""
    k0, k1 = e.plan.layer_entries[i]
    return dense[e.plan.entry_row[k0:k1], e.plan.entry_col[k0:k1]]


def step_vs_oracle(masks, sd, nu, x, eps, lr=1e-4, kc=1e-4):
    """One training step at batch len(x): raw gradients (apply_update = 0) against the oracle's hand-written backward, then
    the unfused AdamW and, on a second model, the fused step, both against the oracle's parameters.

    Gradient criterion.  At these batch sizes a gradient element is a sum of thousands of products and two correct float32
    evaluations differ by more than the 2e-5 of scale that holds at B <= 200 (the oracle's own numpy reductions over the
    batch axis are sequential).  So the float64 instance of the oracle is the yardstick: per tensor, the HIP path's worst
    error against float64 may be at most twice the float32 oracle's (floor 2e-6 of the tensor's scale), and never above
    1e-4 of scale unless the float32 oracle itself is beyond 5e-5 (dW0 at B = 8192 on the medium model: the BatchNorm backward
    makes every row of dh sum to zero, the gradient is what survives the cancellation, and both float32 evaluations sit at
    6e-3 of its scale)."""
    from ontovae_oracle import OracleOntoVAE
    from onto_vae_amd.vae_model import FusedAdamW
    o = OracleOntoVAE(masks, nu)
    o.load_state_dict(sd)
    rec, kl, loss, g = o.gradients(x, eps, kc)
    o64 = OracleOntoVAE(masks, nu, dtype=np.float64)
    o64.load_state_dict(sd)
    _, _, loss64, g64 = o64.gradients(x, eps, kc)
    del o64
    xt, et = torch.as_tensor(x), torch.as_tensor(eps)
    model, _ = make_model(masks, nu, sd)
    e = model._engine
    out = torch.zeros(3, device="cuda")
    e.train_step(xt, None, et, kc, None, out, apply_update=False)
    np.testing.assert_allclose(out.cpu().numpy(), [rec, kl, loss], rtol=REL)
    assert abs(float(out[2]) - loss64) <= max(2 * abs(float(loss) - loss64), 1e-6 * abs(loss64))

    def check(name, got, want32, want64):
        scale = np.abs(want64).max()
        e_hip, e_ref = np.abs(got - want64).max() / scale, np.abs(want32 - want64).max() / scale
        assert e_hip <= max(2 * e_ref, 2e-6) and (e_hip <= 1e-4 or e_ref > 5e-5), (name, e_hip, e_ref)

    for k, v in grads_of(e).items():
        check(k, v.cpu().numpy(), g[k], g64[k])
    for i in range(e.plan.K):
        k = f"decoder.decoder.{i}.0.weight"
        check(k, e.layer_vals(i, e.Gr).cpu().numpy(), packed_layer(e, i, g[k]), packed_layer(e, i, g64[k]))
    del g64
    assert np.abs(e.view("b0", e.Gr).cpu().numpy()).max() <= 1e-4 * np.abs(g["encoder.encoder.0.1.bias"]).max()
    o.apply_step(g, lr)
    want_sd = o.state_dict()
    fused, _ = make_model(masks, nu, sd)
    fout = torch.zeros(3, device="cuda")
    fused._engine.train_step(xt, None, et, kc, FusedAdamW(fused, lr=lr).next_hyper(), fout)
    np.testing.assert_allclose(fout.cpu().numpy(), [rec, kl, loss], rtol=REL)
    e.adamw_apply(FusedAdamW(model, lr=lr).next_hyper())
    for m in (model, fused):
        me = m._engine
        got = {"W0": me.view("W0").cpu().numpy(), "Wmu": me.view("Wmu").cpu().numpy(), "gamma": me.view("gamma").cpu().numpy(),
               "rvar": me.rvar.cpu().numpy(), "rmean": me.rmean.cpu().numpy()}
        want = {"W0": want_sd["encoder.encoder.0.0.weight"], "Wmu": want_sd["encoder.mu.0.weight"],
                "gamma": want_sd["encoder.encoder.0.1.weight"], "rvar": want_sd["encoder.encoder.0.1.running_var"],
                "rmean": want_sd["encoder.encoder.0.1.running_mean"]}
        for i in range(me.plan.K):
            got[f"dec{i}"] = me.layer_vals(i).cpu().numpy()
            want[f"dec{i}"] = packed_layer(me, i, want_sd[f"decoder.decoder.{i}.0.weight"])
            assert got[f"dec{i}"].min() >= 0.0
        # one step: running statistics are exact arithmetic of this batch (strict), Adam's first step is lr * sign(g)
        np.testing.assert_allclose(got.pop("rmean"), want.pop("rmean"), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(got.pop("rvar"), want.pop("rvar"), rtol=1e-5, atol=1e-6)
        check_params(got, want, lr, rtol=2e-4, atol=2e-6, outliers=0.01)
        b0 = me.view("b0").cpu().numpy()
        assert np.abs(b0 - sd["encoder.encoder.0.0.bias"] * (1 - lr * 1e-2)).max() <= lr * 1.001


@pytest.mark.parametrize("B", [1024, 4096, 8192])
def test_gobp_syn_train_step_large_batches_vs_oracle(B):
    """BASELINE.json configs[2]: the per-GPU (1024, 4096) and global (8192) batch sizes of the data-parallel runs on the
    full-size model: dW0 with a K loop, 4-column lanes, 256 column blocks of weight-gradient partials."""
    masks, sd, X = gobp()
    rng = np.random.RandomState(B)
    x = X[rng.randint(0, X.shape[0], size=B)]
    eps = rng.standard_normal((B, 360)).astype(np.float32)
    step_vs_oracle(masks, sd, 3, x, eps)


@pytest.mark.parametrize("B", [1024, 2048, 8192])
def test_medium_train_step_large_batches_vs_oracle(B):
    masks, sd = medium()
    rng = np.random.RandomState(B + 1)
    x = np.log1p(rng.poisson(0.5, size=(B, 1000))).astype(np.float32)
    eps = rng.standard_normal((B, 60)).astype(np.float32)
    step_vs_oracle(masks, sd, 3, x, eps, lr=1e-3)


# ---------------------------------------------------------------------------------------------- configs[3]
@pytest.mark.parametrize("batch", [4096, 16384])
def test_gobp_syn_inference_vs_oracle(batch):
    """get_pathway_activities and get_reconstructed_values on GOBP-syn with `batch` rows per launch set (the kernels that
    only inference batches select: elementwise eval BatchNorm, LDS-staged heads GEMM, per-level term launches, 64 x 64
    output transposition) against oracle.pass_data.  The oracle evaluates every row for the activities and the first and
    last 1024 rows for the reconstruction (each row is independent in eval mode)."""
    from ontovae_oracle import OracleOntoVAE
    masks, sd, X = gobp()
    N = batch
    rng = np.random.RandomState(batch)
    rows = rng.randint(0, X.shape[0], size=N)
    Xd = (X[rows] * rng.uniform(0.5, 1.5, size=(N, 1))).astype(np.float32)      # distinct rows
    e1 = rng.standard_normal((N, 360)).astype(np.float32)
    e2 = rng.standard_normal((N, 360)).astype(np.float32)
    model, onto = make_model(masks, 3, sd, X=Xd)
    model.inference_batch = batch
    o = OracleOntoVAE(masks, 3)
    o.load_state_dict(sd)
    feed(model, [e1, e2])
    act = model.get_pathway_activities(onto, "d")
    assert act.shape == (N, 3000) and act.dtype == np.float32
    want = o.pass_data(Xd, e1, e2, "act_only")
    assert act_close(act, want)
    feed(model, [e1, e2])
    rec = model.get_reconstructed_values(onto, "d")
    assert rec.shape == (N, 20000)
    sel = np.r_[0:1024, N - 1024:N]
    want = o.pass_data(Xd[sel], e1[sel], e2[sel], "rec")
    assert act_close(rec[sel], want)
    # the decoder alone on a supplied z (ontovae_decoder_forward, modules.py:303-313)
    z = rng.standard_normal((2048, 360)).astype(np.float32)
    a_out = torch.empty(2048, 3000, device="cuda")
    r_out = torch.empty(2048, 20000, device="cuda")
    model._engine.decoder_forward(torch.as_tensor(z), act_out=a_out, rec_out=r_out)
    recon, _, cs = o.decoder_forward(z, collect=True)
    from ontovae_oracle import nu_mean
    want_act = np.hstack((nu_mean(z, 3), nu_mean(np.concatenate(cs, axis=1), 3)))
    assert act_close(a_out.cpu().numpy(), want_act) and act_close(r_out.cpu().numpy(), recon)


def test_gobp_syn_full_size_properties():
    """configs[3] at a size no oracle pass covers (65536 rows of GOBP-syn in batches of 16384): properties that do not depend on
    the size.  (1) The decoder is linear (modules.py:303-313 has no activation): dec(z1 + z2) - dec(0) = (dec(z1) - dec(0)) +
    (dec(z2) - dec(0)).  (2) Rows are independent in eval mode: permuting the dataset permutes the activities, and the batch
    size does not matter.  (3) Perturbing a gene that no row expresses to 0 changes nothing; the rank-1 screen pass of an empty
    perturbation list equals the plain pass."""
    masks, sd, X = gobp()
    N = 65536
    rng = np.random.RandomState(65536)
    rows = rng.randint(0, X.shape[0], size=N)
    Xd = torch.as_tensor(X[rows] * rng.uniform(0.5, 1.5, size=(N, 1)).astype(np.float32)).cuda()
    model, onto = make_model(masks, 3, sd, X=np.zeros((4, 20000), np.float32))
    model.eval()
    e = model._engine
    # (1) linearity of the decoder, activities and reconstruction
    n = 16384
    z1 = torch.as_tensor(rng.standard_normal((n, 360)).astype(np.float32)).cuda()
    z2 = torch.as_tensor(rng.standard_normal((n, 360)).astype(np.float32)).cuda()

    def dec(z):
        a, r = torch.empty(n, 3000, device="cuda"), torch.empty(n, 20000, device="cuda")
        e.decoder_forward(z, act_out=a, rec_out=r)
        return a, r
    a0, r0 = dec(torch.zeros_like(z1))
    a1, r1 = dec(z1)
    a2, r2 = dec(z2)
    a12, r12 = dec(z1 + z2)
    for lhs, rhs in (((a12 - a0), (a1 - a0) + (a2 - a0)), ((r12 - r0), (r1 - r0) + (r2 - r0))):
        scale = float(rhs.abs().max())
        assert float((lhs - rhs).abs().max()) <= 2e-5 * scale
    del a0, r0, a1, r1, a2, r2, a12, r12
    # (2) row independence: permutation equivariance and batch-size invariance, with eps = 0 (the draws are keyed by position)
    zero = lambda B: torch.zeros(B, 360)
    model.eps_source = zero
    model.inference_batch = 16384
    act = torch.as_tensor(model._pass_data(Xd, 'act'))
    assert act.shape == (N, 3000) and bool(torch.isfinite(act).all())
    perm = torch.as_tensor(rng.permutation(N))
    act_p = torch.as_tensor(model._pass_data(Xd[perm.cuda()].contiguous(), 'act'))
    assert float((act_p - act[perm]).abs().max()) <= 2e-5 * float(act.abs().max())
    model.inference_batch = 4096
    act_s = torch.as_tensor(model._pass_data(Xd[:8192], 'act'))
    assert float((act_s - act[:8192]).abs().max()) <= 2e-5 * float(act.abs().max())
    # (3) the rank-1 screen pass with an empty perturbation list is the plain pass
    hpre = torch.empty(360, 4096, device="cuda")
    xb = Xd[:4096]
    e.encode_pre(xb, None, hpre, 0)
    out = torch.empty(4096, 3000, device="cuda")
    e.pass_data_perturbed(xb, None, hpre, 0, np.zeros(0, np.int32), np.zeros(0, np.float32), zero(4096), zero(4096), act_out=out)
    assert float((out.cpu() - act[:4096]).abs().max()) <= 2e-5 * float(act.abs().max())


def test_inference_one_million_rows():
    """BASELINE.json configs[3] at its stated size: get_pathway_activities-equivalent over
    1 000 000 GOBP-syn samples generated ON THE DEVICE in chunks from a seeded generator (80 GB of X, resident), streamed in
    batches of 16384 rows into a resident [1 M, 3000] output (12 GB) by OntoVAE._pass_data.  The reference's path
    (vae_model.py:330-361) moves the dataset to the device in one tensor; no oracle pass covers 1 M rows, so: rows sampled
    from the first, a middle and the last (partial) batch against oracle.pass_data with the same eps, and the size-independent
    properties -- every row finite, rows independent of their batch (windows re-run at another batch size), and of their
    position (a window re-run on its own)."""
    from ontovae_oracle import OracleOntoVAE
    masks, sd, _ = gobp()
    N, G, L, T, step = 1_000_000, 20000, 360, 3000, 16384
    model, _ = make_model(masks, 3, sd, X=np.zeros((4, G), np.float32))
    model.eval()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20250117)
    X = torch.empty(N, G, dtype=torch.float32, device="cuda")
    for i in range(0, N, 8192):
        n = min(8192, N - i)
        X[i:i + n] = torch.log1p(torch.poisson(torch.full((n, G), 0.3, device="cuda"), generator=gen))
    # eps per batch from a seeded device generator; the batches the oracle will look at keep theirs
    nb = (N + step - 1) // step
    watch = {0: None, nb // 2: None, nb - 1: None}
    egen = torch.Generator(device="cuda")
    egen.manual_seed(7)
    calls = [0]

    def eps_source(B):
        e = torch.randn(B, L, generator=egen, device="cuda")
        b, which = divmod(calls[0], 2)
        if b in watch:
            watch[b] = (watch[b] or ()) + (e[:48].cpu().numpy(),)
        calls[0] += 1
        return e

    model.eps_source = eps_source
    model.inference_batch = step
    act = model._pass_data(X, 'act')
    assert calls[0] == 2 * nb and N % step == 576
    assert act.shape == (N, T) and act.dtype == np.float32
    # every row finite (checked in slabs: the array is 12 GB)
    for i in range(0, N, 100000):
        assert np.isfinite(act[i:i + 100000]).all()
    o = OracleOntoVAE(masks, 3)
    o.load_state_dict(sd)
    for b, (e1, e2) in watch.items():
        rows = slice(b * step, b * step + 48)
        want = o.pass_data(X[rows].cpu().numpy(), e1, e2, "act_only")
        assert act_close(act[rows], want), b
    # rows do not depend on their batch or their position: windows re-run alone, in batches of 4096 rows, with eps = 0 on both
    # sides (the first pass's draws are keyed by call order); the full-size pass is repeated with eps = 0 for that
    zero = torch.zeros(step, L, device="cuda")
    model.eps_source = lambda B: zero[:B]
    act0 = model._pass_data(X, 'act')
    scale = float(np.abs(act0[:step]).max())
    model.inference_batch = 4096
    for lo in (0, (nb // 2) * step + 1000, N - 5000):
        win = model._pass_data(X[lo:lo + 5000], 'act')
        assert np.abs(win - act0[lo:lo + 5000]).max() <= 2e-5 * scale, lo
    # and the draws matter: the eps = 0 pass differs from the first one in the latent columns
    assert np.abs(act0[:64, :120] - act[:64, :120]).max() > 1e-3


# ---------------------------------------------------------------------------------------------- configs[0]
PWO_COUNTS = [76, 3, 12, 131, 169, 160, 101, 5]      # terms per depth of the trimmed Pathway Ontology


def test_pwo_shaped_train_model_vs_oracle(tmp_path, capsys):
    """BASELINE.json configs[0] (vignette-sized; HPO is not available offline, so the shape of the in-repo Pathway
    Ontology: G = 5892, T = 657, K = 8, N = 1000): train_model(batch_size=128, epochs=5) = 7 train + 2 val iterations per
    epoch, per-step rec / kl and per-epoch losses against the oracle walking the same split, shuffle and eps; the printed
    lines are the reference's (vae_model.py:254, 263, 272-273)."""
    from ontovae_oracle import OracleOntoVAE, loader_batches, train_val_split
    from onto_vae_amd.synthetic import synthetic_masks, random_state_dict
    G, nu, N, B, epochs, lr, kc = 5892, 3, 1000, 128, 5, 1e-4, 1e-4
    masks = synthetic_masks(PWO_COUNTS, G, 1)
    assert [m.shape[0] for m in masks[:-1]] == PWO_COUNTS[1:] and masks[-1].shape == (G, 657)
    sd = random_state_dict(masks, nu, 2)
    rng = np.random.RandomState(1)
    X = np.log1p(rng.poisson(0.3, size=(N, G))).astype(np.float64)
    L = PWO_COUNTS[0] * nu
    tr, va = train_val_split(N)
    torch.manual_seed(7)
    perms = [torch.randperm(len(tr)).numpy() for _ in range(epochs)]
    train_batches = loader_batches(len(tr), B, perms)
    val_batches = loader_batches(len(va), B, [None])[0]
    assert len(train_batches[0]) == 7 and len(val_batches) == 2
    eps_list, sched = [], []
    for ep in range(epochs):
        for b in train_batches[ep]:
            eps_list.append(rng.standard_normal((len(b), L)).astype(np.float32))
            sched.append(("train", tr[b], eps_list[-1]))
        for b in val_batches:
            eps_list.append(rng.standard_normal((len(b), L)).astype(np.float32))
            sched.append(("val", va[b], eps_list[-1]))
    o = OracleOntoVAE(masks, nu)
    o.load_state_dict(sd)
    want = {"train": [], "val": []}
    Xf = X.astype(np.float32)
    for mode, rows, eps in sched:
        if mode == "train":
            want["train"].append(o.train_step(Xf[rows], eps, lr, kc))
        else:
            want["val"].append(o.val_step(Xf[rows], eps, kc))

    class Run(dict):
        class S(list):
            def log(self, v):
                self.append(float(v))

        def __missing__(self, k):
            self[k] = Run.S()
            return self[k]

    model, _ = make_model(masks, nu, sd, X=X)
    feed(model, eps_list)
    run = Run()
    torch.manual_seed(7)
    path = tmp_path / "best.pt"
    model.train_model(str(path), lr=lr, kl_coeff=kc, batch_size=B, epochs=epochs, run=run)
    printed = capsys.readouterr().out.splitlines()
    wt, wv = np.array(want["train"], dtype=np.float64), np.array(want["val"], dtype=np.float64)
    np.testing.assert_allclose(run["metrics/train/rec_loss"], wt[:, 0], rtol=REL)
    np.testing.assert_allclose(run["metrics/train/kl_loss"], wt[:, 1], rtol=REL)
    # Validation inside train_model runs on trained weights, and eval-mode BatchNorm sees the Linear bias in front of it
    # through (b0 - running_mean).  That bias has an analytically zero gradient; Adam turns its rounding noise into
    # +-lr steps (NOISE_KEYS), so after n steps two correct float32 implementations -- the reference on
    # two machines included -- differ in b0 - running_mean by up to ~9 lr (the EMA lag) and in the validation losses by a
    # few 1e-4.  Held to 2e-3 here, and to 1e-4 below on IDENTICAL weights.
    VAL = 2e-3
    np.testing.assert_allclose(run["metrics/val/rec_loss"], wv[:, 0], rtol=VAL)
    np.testing.assert_allclose(run["metrics/val/kl_loss"], wv[:, 1], rtol=VAL)
    ep_train = wt[:, 2].reshape(epochs, 7).mean(1)
    ep_val = wv[:, 2].reshape(epochs, 2).mean(1)
    np.testing.assert_allclose(run["metrics/train/loss"], ep_train, rtol=REL)
    np.testing.assert_allclose(run["metrics/val/loss"], ep_val, rtol=VAL)
    # printed protocol: "Epoch i of n", "New best model!" when the validation loss improves, then the two loss lines
    got_val = run["metrics/val/loss"]
    best, it = float("inf"), iter(printed)
    for ep in range(epochs):
        assert next(it) == f"Epoch {ep + 1} of {epochs}"
        if got_val[ep] < best:
            assert next(it) == "New best model!"
            best = got_val[ep]
        t, v = next(it), next(it)
        assert t.startswith("Train Loss: ") and v.startswith("Val Loss: ")
        np.testing.assert_allclose(float(t.split(": ")[1]), ep_train[ep], rtol=REL, atol=1e-4)
        np.testing.assert_allclose(float(v.split(": ")[1]), got_val[ep], rtol=1e-6, atol=1e-4)
        assert len(t.split(".")[-1]) == 4
    assert next(it, None) is None
    # val_round at this shape on identical weights: the oracle's trained state loaded into a fresh model
    vmodel, _ = make_model(masks, nu, {k: np.asarray(v) for k, v in o.state_dict().items()}, X=X)
    veps = [rng.standard_normal((len(b), L)).astype(np.float32) for b in val_batches]
    feed(vmodel, list(veps))
    vrun = Run()
    vloss = vmodel.val_round([(torch.as_tensor(Xf[va[b]]),) for b in val_batches], kc, vrun)
    vwant = np.array([o.val_step(Xf[va[b]], e_, kc) for b, e_ in zip(val_batches, veps)], dtype=np.float64)
    np.testing.assert_allclose(vrun["metrics/val/rec_loss"], vwant[:, 0], rtol=REL)
    np.testing.assert_allclose(vrun["metrics/val/kl_loss"], vwant[:, 1], rtol=REL)
    np.testing.assert_allclose(vloss, vwant[:, 2].mean(), rtol=REL)
    ck = torch.load(str(path), weights_only=True)
    assert set(ck) >= {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}
    np.testing.assert_allclose(ck["loss"], min(got_val), rtol=1e-6)
    steps = epochs * 7
    check_params(hip_state(model), o.state_dict(), steps * lr, rtol=2e-4, atol=2e-6, outliers=0.01)


# ---------------------------------------------------------------------------------------------- configs[2], data parallel
def test_dp_allreduce_form_gobp_syn_global_8192():
    """configs[2] proper: global batch 8192 on the full-size model in the all-reduce form that use_gathered_dh selects there
    (2, 4 and 8 ranks), as two simulated ranks of 4096 rows on one GPU with the collectives done by hand
    (ontovae_train_step_dp phases 0 / 1 with all-gathered statistics / 2, SUM of the flat gradient buffers, flat AdamW):
    must land where the single-device step on the global batch lands (held to the oracle above), replicas identical.
    The same exchange through onto_vae_amd.parallel.dp_train_step and real collectives runs in
    test_gpu_parity.py::test_dp_two_processes_one_gpu (medium model) and, in both forms at full size, in two processes below."""
    from onto_vae_amd.parallel import use_gathered_dh
    from onto_vae_amd.vae_model import FusedAdamW
    for world in (2, 4, 8):
        assert not use_gathered_dh(8192, world, 360, 20000)
        assert use_gathered_dh(128 * world, world, 360, 20000)
    masks, sd, X = gobp()
    B, L, kc, lr = 8192, 360, 1e-4, 1e-4
    rng = np.random.RandomState(8192 + 5)
    Xd = torch.as_tensor(X).cuda()
    idx = torch.as_tensor(rng.randint(0, X.shape[0], size=B).astype(np.int32)).cuda()
    eps = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).cuda()
    ref, _ = make_model(masks, 3, sd)
    rout = torch.zeros(3, device="cuda")
    ref._engine.train_step(Xd, idx, eps, kc, FusedAdamW(ref, lr=lr).next_hyper(), rout)
    ranks = [make_model(masks, 3, sd)[0] for _ in range(2)]
    parts = [(0, 4096), (4096, 8192)]
    stats = [torch.zeros(2 * L + 1, device="cuda") for _ in ranks]
    bstats = [torch.zeros(2, L, device="cuda") for _ in ranks]
    louts = [torch.zeros(3, device="cuda") for _ in ranks]
    for r, (m, (a, b)) in enumerate(zip(ranks, parts)):
        stats[r][2 * L] = float(b - a)
        m._engine.train_step_dp(0, Xd, idx[a:b], B, eps[a:b], kc, stats[r][:2 * L].view(2, L), bstats[r], louts[r], 1)
    gathered = torch.cat(stats)
    for r, (m, (a, b)) in enumerate(zip(ranks, parts)):
        m._engine.train_step_dp(16 + 2, Xd, idx[a:b], B, eps[a:b], kc, gathered, bstats[r], louts[r], 1)
    bsum = bstats[0] + bstats[1]
    for r, (m, (a, b)) in enumerate(zip(ranks, parts)):
        bstats[r].copy_(bsum)
        m._engine.train_step_dp(2, Xd, idx[a:b], B, eps[a:b], kc, stats[r][:2 * L].view(2, L), bstats[r], louts[r], 1)
    gsum = ranks[0]._engine.Gr + ranks[1]._engine.Gr
    np.testing.assert_allclose((louts[0] + louts[1])[:2].cpu().numpy(), rout[:2].cpu().numpy(), rtol=1e-5)
    for m in ranks:
        m._engine.Gr.copy_(gsum)
        m._engine.adamw_apply(FusedAdamW(m, lr=lr).next_hyper())
    assert torch.equal(ranks[0]._engine.P, ranks[1]._engine.P)
    e = ref._engine
    want, got = e.P.cpu().numpy(), ranks[0]._engine.P.cpu().numpy()
    keep = np.ones(want.shape, bool)
    keep[int(e.plan.layout.b0):int(e.plan.layout.b0) + L] = False
    diff = np.abs(got - want)[keep]
    assert (diff > 1e-6 + 1e-4 * np.abs(want[keep])).mean() <= 0.01 and diff.max() <= 2 * lr


@pytest.mark.parametrize("B,form", [(1024, "gathered"), (8192, "allreduce")])
def test_dp_two_processes_gobp_syn_heuristic_forms(tmp_path, B, form):
    """onto_vae_amd.parallel.dp_train_step on the full-size model, two processes on GPU 0 over gloo, exchange form chosen by
    use_gathered_dh itself: global batch 1024 -> all-gathered dh and dW0 over 8 chunks of the K loop on every rank; global
    batch 8192 (configs[2]) -> all-reduce of the first layer's gradient in row panels.  One step must land where the single-process step on the global batch lands (itself held to the oracle above), replicas
    bitwise identical."""
    from onto_vae_amd.parallel import use_gathered_dh
    from onto_vae_amd.vae_model import FusedAdamW
    assert use_gathered_dh(B, 2, 360, 20000) == (form == "gathered")
    masks, sd, X = gobp()
    rng = np.random.RandomState(B + 5)
    idx = rng.randint(0, X.shape[0], size=B).astype(np.int32)
    eps = rng.standard_normal((1, B, 360)).astype(np.float32)
    path = str(tmp_path / "dp_gobp.npz")
    np.savez(path, gobp=1, nu=3, idx=idx, eps=eps, B=B, lr=1e-4, use_idx=1)
    ref, _ = make_model(masks, 3, sd)
    rl = torch.zeros(1, 3, device="cuda")
    Xd = torch.as_tensor(X).cuda()
    ref._engine.train_step(Xd, torch.as_tensor(idx).cuda(), torch.as_tensor(eps[0]), 1e-4, FusedAdamW(ref, lr=1e-4).next_hyper(), rl[0])
    want = ref._engine.P.cpu().numpy()
    lay, Lw = ref._engine.plan.layout, ref._engine.plan.L
    del ref, Xd
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items() if k not in ("ONTOVAE_DP_ALLREDUCE", "ONTOVAE_DP_TWO_EXCHANGES")}
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_worker.py")
    port = str(29900 + (os.getpid() % 90))
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, path, "1"], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=170) == 0
    got = [np.load(path + f".rank{r}.npz") for r in range(2)]
    assert np.array_equal(got[0]["P"], got[1]["P"])
    np.testing.assert_allclose(got[0]["losses"], rl.cpu().numpy(), rtol=2e-5)
    mask = np.ones(want.shape, bool)
    mask[int(lay.b0):int(lay.b0) + Lw] = False
    diff = np.abs(got[0]["P"] - want)[mask]
    assert (diff > 1e-6 + 1e-4 * np.abs(want[mask])).mean() <= 0.01 and diff.max() <= 2 * 1e-4


def test_dp_async_handles_stay_outstanding_under_the_panel_products(tmp_path):
    """The asynchronous control flow of the all-reduce form against a real peer: two processes on GPU 0, gloo behind the
    stream-ordered staging of onto_vae_amd.parallel._HostStaged (a collective is issued against what is on the compute stream,
    runs on a worker thread while the caller keeps launching, completes for the stream at wait()).  Two steps at global batch
    8192 on the full-size model.  Checked: (1) the order of issue / launch / wait in a step -- the [heads | decoder] bucket is
    issued before the first row panel of dW0, panel p's all-reduce right after its product and before panel p + 1's product,
    no wait before the last product is launched, AdamW of a range only after that range's wait; (2) the result: replicas
    bitwise identical and equal to the single-process steps -- a buffer overwritten while its all-reduce was outstanding, or
    an update applied before its exchange completed, lands somewhere else."""
    from onto_vae_amd.parallel import use_gathered_dh, w0_row_panels
    from onto_vae_amd.vae_model import FusedAdamW
    B, steps = 8192, 2
    assert not use_gathered_dh(B, 2, 360, 20000)
    masks, sd, X = gobp()
    rng = np.random.RandomState(4242)
    idx = rng.randint(0, X.shape[0], size=B * steps).astype(np.int32)
    eps = rng.standard_normal((steps, B, 360)).astype(np.float32)
    path = str(tmp_path / "dp_async.npz")
    np.savez(path, gobp=1, nu=3, idx=idx, eps=eps, B=B, lr=1e-4, use_idx=1)
    ref, _ = make_model(masks, 3, sd)
    ropt = FusedAdamW(ref, lr=1e-4)
    rl = torch.zeros(steps, 3, device="cuda")
    Xd = torch.as_tensor(X).cuda()
    for s in range(steps):
        ref._engine.train_step(Xd, torch.as_tensor(idx[s * B:(s + 1) * B]).cuda(), torch.as_tensor(eps[s]), 1e-4, ropt.next_hyper(), rl[s])
    want = ref._engine.P.cpu().numpy()
    lay, Lw, G = ref._engine.plan.layout, ref._engine.plan.L, ref._engine.plan.G
    del ref, Xd
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items() if k not in ("ONTOVAE_DP_ALLREDUCE", "ONTOVAE_DP_TWO_EXCHANGES", "ONTOVAE_DP_W0_PANELS")}
    env["ONTOVAE_DP_WORKER_CALLS"] = "1"
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_worker.py")
    port = str(29800 + (os.getpid() % 90))
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, path, str(steps)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=170) == 0
    got = [np.load(path + f".rank{r}.npz") for r in range(2)]
    assert np.array_equal(got[0]["P"], got[1]["P"])
    np.testing.assert_allclose(got[0]["losses"], rl.cpu().numpy(), rtol=2e-5)
    mask = np.ones(want.shape, bool)
    mask[int(lay.b0):int(lay.b0) + Lw] = False
    diff = np.abs(got[0]["P"] - want)[mask]
    assert (diff > 1e-6 + 1e-4 * np.abs(want[mask])).mean() <= 0.01 and diff.max() <= 2 * steps * 1e-4
    panels = w0_row_panels(Lw, 2)
    assert len(panels) == 4
    early = int(lay.total) - int(lay.Wh)
    small = int(lay.Wh) - int(lay.b0)
    for r in range(2):
        calls = [str(c).replace("wait(pending) ", "wait ") for c in got[r]["calls"]]
        per_step = len(calls) // steps
        assert per_step * steps == len(calls)
        for s in range(steps):
            c = calls[s * per_step:(s + 1) * per_step]
            want_seq = [f"issue all_gather:{2 * Lw + 1}", f"wait all_gather:{2 * Lw + 1}",          # SyncBN forward statistics
                        f"issue all_reduce:{2 * Lw}", f"wait all_reduce:{2 * Lw}",                    # SyncBN backward sums
                        f"issue all_reduce:{early}"]                                                 # [heads | decoder], left outstanding
            for l0, l1 in panels:
                want_seq += [f"dw0_rows {l0}:{l1}", f"issue all_reduce:{(l1 - l0) * G}"]
            want_seq += [f"issue all_reduce:{small}"]
            for l0, l1 in panels:
                want_seq += [f"wait all_reduce:{(l1 - l0) * G}", f"adamw {int(lay.W0) + l0 * G}:{int(lay.W0) + l1 * G}"]
            want_seq += [f"wait all_reduce:{small}", f"wait all_reduce:{early}", f"adamw {int(lay.b0)}:{int(lay.total)}",
                         "issue all_reduce:3", "wait all_reduce:3"]
            assert c == want_seq, (r, s, c)
    # at least one wait found its collective still running when it was called: the handles are real
    assert any(str(c).startswith("wait(pending)") for r in range(2) for c in got[r]["calls"])


# ---------------------------------------------------------------------------------------------- configs[4], bf16 variant
BF16_LOSS_REL = 2e-2        # stated tolerance of the bf16-MFMA variant against the float32 oracle: loss per step
BF16_ACT_COS = 0.999        # pathway activities / reconstructions: cosine per sample


def _cos_rows(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-300)


@pytest.mark.parametrize("which", ["medium", "fullgo"])
def test_bf16_variant_vs_oracle(which):
    """BASELINE.json configs[4]: the bf16-MFMA variant (x W0^T and dW0 = dh x with bf16 inputs, fp32 accumulation, fp32 master
    weights and AdamW) against the float32 oracle, at the tolerance the variant states: loss 2e-2 relative, activities cosine
    >= 0.999; plus the gradient it changes (dW0: cosine and relative L2 error) and three AdamW steps.  'fullgo' is the deeper
    Full-GO-syn ontology of SURVEY.md 8(d) config 5 (T = 4500, K = 14, L = 600: the panel takes 2 columns per workgroup)."""
    from ontovae_oracle import OracleOntoVAE
    from onto_vae_amd.synthetic import synthetic_masks, random_state_dict, FULLGO_SYN_COUNTS, FULLGO_SYN_SEED
    from onto_vae_amd.vae_model import FusedAdamW
    if which == "medium":
        masks, sd = medium()
        G, nu, B, lr = 1000, 3, 512, 1e-3
    else:
        masks = synthetic_masks(FULLGO_SYN_COUNTS, 20000, FULLGO_SYN_SEED)
        sd = random_state_dict(masks, 3, 78, dense=False)
        G, nu, B, lr = 20000, 3, 1024, 1e-4
    L = masks[0].shape[1] * nu
    rng = np.random.RandomState(31)
    x = np.log1p(rng.poisson(0.3, size=(B, G))).astype(np.float32)
    eps = rng.standard_normal((B, L)).astype(np.float32)
    o = OracleOntoVAE(masks, nu)
    o.load_state_dict(sd)
    rec, kl, loss, g = o.gradients(x, eps, 1e-4)
    model, onto = make_model(masks, nu, sd, X=x)
    model.set_precision("bf16")
    e = model._engine
    out = torch.zeros(3, device="cuda")
    e.train_step(torch.as_tensor(x), None, torch.as_tensor(eps), 1e-4, None, out, apply_update=False)
    got = out.cpu().numpy()
    np.testing.assert_allclose(got, [rec, kl, loss], rtol=BF16_LOSS_REL)
    assert abs(got[2] - loss) > 1e-7 * abs(loss)                 # it IS another arithmetic (the flag is not ignored)
    gw, ww = e.view("W0", e.Gr).cpu().numpy().ravel(), g["encoder.encoder.0.0.weight"].ravel()
    cos = float(gw @ ww / (np.linalg.norm(gw) * np.linalg.norm(ww)))
    # dh is a cancelling quantity (every row sums to zero after the BatchNorm backward): bf16 rounding of dh and x leaves 3-5 % of
    # relative L2 error in dW0
    assert cos >= 0.995 and np.linalg.norm(gw - ww) <= 0.1 * np.linalg.norm(ww), (cos, np.linalg.norm(gw - ww) / np.linalg.norm(ww))
    # inference in the variant: activities and reconstruction per sample
    e1, e2 = rng.standard_normal((B, L)).astype(np.float32), rng.standard_normal((B, L)).astype(np.float32)
    o2 = OracleOntoVAE(masks, nu)
    o2.load_state_dict(sd)
    imodel, ionto = make_model(masks, nu, sd, X=x)
    imodel.set_precision("bf16")
    feed(imodel, [e1, e2])
    act = imodel.get_pathway_activities(ionto, "d")
    assert _cos_rows(act, o2.pass_data(x, e1, e2, "act_only")).min() >= BF16_ACT_COS
    # three fused steps: losses inside the stated tolerance, weights move the way the oracle's do
    opt = FusedAdamW(model, lr=lr)
    o3 = OracleOntoVAE(masks, nu)
    o3.load_state_dict(sd)
    m3, _ = make_model(masks, nu, sd)
    m3.set_precision("bf16")
    opt3 = FusedAdamW(m3, lr=lr)
    outs = torch.zeros(3, 3, device="cuda")
    want = []
    for s_ in range(3):
        xs = np.log1p(rng.poisson(0.3, size=(B, G))).astype(np.float32)
        es = rng.standard_normal((B, L)).astype(np.float32)
        want.append(o3.train_step(xs, es, lr, 1e-4))
        m3._engine.train_step(torch.as_tensor(xs), None, torch.as_tensor(es), 1e-4, opt3.next_hyper(), outs[s_])
    np.testing.assert_allclose(outs.cpu().numpy(), np.array(want), rtol=BF16_LOSS_REL)
    w_hip, w_ref, w0 = m3._engine.view("W0").cpu().numpy(), o3.p["encoder.encoder.0.0.weight"], np.asarray(sd["encoder.encoder.0.0.weight"])
    du_h, du_r = (w_hip - w0).ravel(), (w_ref - w0).ravel()
    assert float(du_h @ du_r / (np.linalg.norm(du_h) * np.linalg.norm(du_r))) >= 0.9      # Adam's sign-like steps agree on most weights


@pytest.mark.parametrize("size,form", [("medium", "allreduce"), ("medium", "gathered"), ("gobp", "allreduce")])
def test_bf16_variant_data_parallel_two_processes(tmp_path, size, form):
    """BASELINE.json configs[4] as stated: the bf16-MFMA variant UNDER data parallel.  onto_vae_amd.parallel.dp_train_step in two
    processes on GPU 0 with set_precision('bf16'); the all-reduce form forms dW0 in row panels through ontovae_dw0_rows' bf16
    branch (the transposed bf16 copy of x that phase 0 left behind), the gathered form keeps dW0 in fp32 and takes only the
    forward product in bf16.  Replicas must stay bitwise identical and land where the single-process bf16 steps on the global
    batch land (itself held to the oracle at the variant's stated tolerance in test_bf16_variant_vs_oracle): the partition
    changes only fp32 summation order, not which values are rounded to bf16."""
    from onto_vae_amd.vae_model import FusedAdamW
    if size == "medium":
        masks, sd = medium()
        rng = np.random.RandomState(88)
        B, steps, lr, L = 512, 2, 1e-3, 60
        X = np.log1p(rng.poisson(0.5, size=(2048, 1000))).astype(np.float32)
        idx = rng.permutation(2048)[:B * steps].astype(np.int32)
        eps = rng.standard_normal((steps, B, L)).astype(np.float32)
        path = str(tmp_path / "dp_bf16.npz")
        np.savez(path, nmask=len(masks), nu=3, X=X, idx=idx, eps=eps, B=B, lr=lr, use_idx=1,
                 **{f"mask{i}": m for i, m in enumerate(masks)}, **{"sd:" + k: np.asarray(v) for k, v in sd.items()})
    else:
        masks, sd, X = gobp()
        rng = np.random.RandomState(8192 + 16)
        B, steps, lr, L = 8192, 1, 1e-4, 360
        idx = rng.randint(0, X.shape[0], size=B).astype(np.int32)
        eps = rng.standard_normal((steps, B, L)).astype(np.float32)
        path = str(tmp_path / "dp_bf16_gobp.npz")
        np.savez(path, gobp=1, nu=3, idx=idx, eps=eps, B=B, lr=lr, use_idx=1)
    ref, _ = make_model(masks, 3, sd)
    ref.set_precision("bf16")
    ropt = FusedAdamW(ref, lr=lr)
    rl = torch.zeros(steps, 3, device="cuda")
    Xd = torch.as_tensor(X).cuda()
    for s in range(steps):
        ref._engine.train_step(Xd, torch.as_tensor(idx[s * B:(s + 1) * B]).cuda(), torch.as_tensor(eps[s]), 1e-4, ropt.next_hyper(), rl[s])
    want = ref._engine.P.cpu().numpy()
    lay, Lw = ref._engine.plan.layout, ref._engine.plan.L
    w0_before = np.asarray(sd["encoder.encoder.0.0.weight"], np.float32).ravel()
    # the flag is live: the same first step in float32 moves W0 differently (Adam's first step is lr * sign(g): every
    # gradient element whose sign the bf16 rounding flips shows)
    f32, _ = make_model(masks, 3, sd)
    fl = torch.zeros(3, device="cuda")
    f32._engine.train_step(Xd, torch.as_tensor(idx[:B]).cuda(), torch.as_tensor(eps[0]), 1e-4, FusedAdamW(f32, lr=lr).next_hyper(), fl)
    if steps == 1:
        assert not torch.equal(f32._engine.view("W0"), ref._engine.view("W0"))
    else:
        assert abs(float(fl[2]) - float(rl[0, 2])) > 1e-7 * abs(float(fl[2]))
    del ref, f32, Xd
    torch.cuda.empty_cache()
    env = dict(os.environ, ONTOVAE_DP_ALLREDUCE="1" if form == "allreduce" else "0", ONTOVAE_DP_WORKER_PRECISION="bf16")
    env.pop("ONTOVAE_DP_TWO_EXCHANGES", None)
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_worker.py")
    port = str(29500 + (os.getpid() % 90))
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, path, str(steps)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=170) == 0
    got = [np.load(path + f".rank{r}.npz") for r in range(2)]
    assert np.array_equal(got[0]["P"], got[1]["P"])
    np.testing.assert_allclose(got[0]["losses"], rl.cpu().numpy(), rtol=2e-5)
    mask = np.ones(want.shape, bool)
    mask[int(lay.b0):int(lay.b0) + Lw] = False
    w0 = slice(int(lay.W0), int(lay.W0) + Lw * masks[-1].shape[0])
    if form == "gathered":
        # this form keeps dW0 = dh x in float32 on every rank (DESIGN.md section 3, bf16 variant) while the single-process
        # bf16 step forms it from bf16 operands: W0 moves the same way (Adam's sign-like steps) inside the Adam bound, every
        # other parameter sees identical arithmetic and is held tightly
        du_g, du_w = (got[0]["P"][w0] - w0_before).astype(np.float64), (want[w0] - w0_before).astype(np.float64)
        assert np.abs(got[0]["P"][w0] - want[w0]).max() <= 2 * steps * lr
        assert float(du_g @ du_w / (np.linalg.norm(du_g) * np.linalg.norm(du_w))) >= 0.9
        mask[w0] = False
    diff = np.abs(got[0]["P"] - want)[mask]
    assert (diff > 1e-6 + 1e-4 * np.abs(want[mask])).mean() <= 0.01 and diff.max() <= 2 * steps * lr


@pytest.mark.parametrize("args", [[], ["--global-batch", "1024"], ["--global-batch", "2048", "--dtype", "bf16"]])
def test_bench_line_is_self_consistent(args):
    """bench.py's JSON line on short runs: every launch of the step has a roofline entry whose fraction of peak is in (0, 1]
    (the cost model charges no launch more than the hardware could move, at any batch regime or dtype), the dominant kernel's
    entry carries both the raw event time and the one net of the bracket cost, and the two differ by less than 10 %."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "10", "--no-extras", "--no-cpu-baseline"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=160, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    names = {k["kernel"] for k in d["roofline_kernels"]}
    assert {"enc_gemm", "bn_fwd", "heads", "reparam", "dec_fwd_terms", "dec_fwd_gene", "dec_bwd_gene", "dec_bwd", "latent_bwd",
            "da_gemm", "dwh", "dw0"} <= names
    for k in d["roofline_kernels"]:
        assert 0.0 < k["frac"] <= 1.0 and "exceeds_peak" not in k, k
        assert k["frac_events"] <= k["frac"]
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and 0.0 < roof["frac"] <= 1.0
    assert roof["kernel_ms_per_step"] <= roof["kernel_ms_per_step_events"] <= 1.1 * roof["kernel_ms_per_step"]
    if roof["kernel"] == "dw0":
        # differential timing: alternate steps bracket (dWh + BN backward) and (dWh + BN backward, dW0); both halves were taken
        p = roof["differential"]
        assert p["brackets"] == [15, 15] and p["front_launch_plus_dw0_ms"] > p["front_launch_ms"] > 0.0
        assert abs((p["front_launch_plus_dw0_ms"] - p["front_launch_ms"]) - roof["kernel_ms_per_step"]) < 1e-4
    assert d["dtype"] == ("bf16" if "bf16" in args else "f32") and 0.0 < d["step_roofline_frac"] <= 1.0
    if "bf16" in args:
        assert any(k["kernel"] == "enc_gemm" and (k["bound"] == "hbm" or k["peak"] == 2.5e6) for k in d["roofline_kernels"])


def test_bench_multi_rank_code_path_rehearsal():
    """The command line the driver uses for N > 1 (torch.distributed.run, one rank per GPU, configs[2]) with both ranks on this
    box's one GPU and gloo in place of RCCL (ONTOVAE_BENCH_SHARED_GPU=1): the line must carry configs[2]'s workload, strong
    scaling, the one-GPU time of the same global batch, and a finite loss.  Its throughput means nothing."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ONTOVAE_BENCH_SHARED_GPU="1")
    port = str(29700 + (os.getpid() % 200))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--rows", "8192"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=160, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 8192 and d["config"]["batch_per_gpu"] == 4096
    assert "configs[2]" in d["config"]["workload"] and d["config"]["parallelism"] == "dp2"
    assert d["single_gpu_same_global_batch"]["n_gpus"] == 1 and d["single_gpu_same_global_batch"]["value"] > 0
    assert np.isfinite(d["final_loss"]) and d["value"] > 0 and "rehearsal" in d
