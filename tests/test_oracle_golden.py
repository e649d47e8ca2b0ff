"""The CPU oracle against fixtures produced by the imported reference (tools/gen_golden.py)."""
import numpy as np

from conftest import NOISE_KEYS, act_close, sub
from ontovae_oracle import OracleOntoVAE, loader_batches, train_val_split, expand_masks, nu_mean

REL = 1e-4  # north_star tolerance for loss in fp32


def fresh(fx, masks, which="sd0/"):
    o = OracleOntoVAE(masks, int(fx["nu"]))
    o.load_state_dict(sub(fx, which))
    return o


def test_forward_and_loss(case):
    _, fx, masks = case
    o = fresh(fx, masks)
    recon, mu, lv = o.forward(fx["x/0"], fx["eps/0"], train=True)
    assert act_close(recon, fx["fwd/recon"]) and act_close(mu, fx["fwd/mu"]) and act_close(lv, fx["fwd/log_var"])
    rec, kl, loss = o.vae_loss(recon, mu, lv, fx["x/0"], float(fx["kl_coeff"]))
    np.testing.assert_allclose([rec, kl, loss], fx["fwd/rec_kl_loss"], rtol=REL)


def test_train_steps(case):
    _, fx, masks = case
    o = fresh(fx, masks)
    for s in range(int(fx["steps"])):
        rec, kl, loss = o.train_step(fx[f"x/{s}"], fx[f"eps/{s}"], float(fx["lr"]), float(fx["kl_coeff"]))
        np.testing.assert_allclose(loss, fx["train/loss"][s], rtol=REL)
        np.testing.assert_allclose(rec, fx["train/rec"][s], rtol=REL)
        np.testing.assert_allclose(kl, fx["train/kl"][s], rtol=REL)
    sd = o.state_dict()
    steps_lr = int(fx["steps"]) * float(fx["lr"])
    for k, v in sub(fx, "sdT/").items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v)
        elif k in NOISE_KEYS:
            assert np.abs(sd[k] - v).max() <= 2.1 * steps_lr
        else:
            np.testing.assert_allclose(sd[k], v, rtol=1e-4, atol=1e-6, err_msg=k)
    for n in o.param_names():
        if n in NOISE_KEYS:
            continue
        np.testing.assert_allclose(o.opt["m"][n], fx["adam_m/" + n], rtol=1e-3, atol=1e-5, err_msg=n)
        np.testing.assert_allclose(o.opt["v"][n], fx["adam_v/" + n], rtol=1e-3, atol=1e-5, err_msg=n)
    for i, m in enumerate(o.masks):  # off-mask weights stay exactly zero
        assert np.abs(sd[f"decoder.decoder.{i}.0.weight"] * (1 - m)).max() == 0.0


def test_val_and_inference(case):
    _, fx, masks = case
    o = fresh(fx, masks, "sdT/")
    np.testing.assert_allclose(o.val_step(fx["val/x"], fx["val/eps"], float(fx["kl_coeff"])),
                               fx["val/rec_kl_loss"], rtol=REL)
    act = o.pass_data(fx["inf/X"], fx["inf/eps1"], fx["inf/eps2"], "act")
    rec = o.pass_data(fx["inf/X"], fx["inf/eps1"], fx["inf/eps2"], "rec")
    assert act.dtype == np.float32 and act.shape == fx["inf/act"].shape
    assert act_close(act, fx["inf/act"]) and act_close(rec, fx["inf/rec"])
    Xp = fx["inf/X"].copy()
    for gi, val in zip(fx["inf/pert_gene_idx"], fx["inf/pert_values"]):
        Xp[:, gi] = val
    assert act_close(o.pass_data(Xp, fx["inf/eps1"], fx["inf/eps2"], "act"), fx["inf/pert_act"])
    assert act_close(o.pass_data(Xp, fx["inf/eps1"], fx["inf/eps2"], "rec"), fx["inf/pert_rec"])


def test_loader_order_and_split():
    import os
    import torch
    from conftest import GOLDEN
    fx = dict(np.load(os.path.join(GOLDEN, "loader.npz")))
    N, B, seed, epochs = (int(fx[k]) for k in ("N", "B", "seed", "epochs"))
    torch.manual_seed(seed)
    perms = [torch.randperm(N).numpy() for _ in range(epochs)]
    got = loader_batches(N, B, perms)
    for e in range(epochs):
        assert np.array_equal(np.concatenate(got[e]), fx["order"][e])
        assert [len(b) for b in got[e]] == [16, 16, 16, 5]
    assert np.array_equal(np.concatenate(loader_batches(N, B, [None])[0]), fx["order_noshuffle"])
    for n in (10, 1000, 1001):
        tr, va = train_val_split(n)
        assert np.array_equal(tr, fx[f"split_train/{n}"]) and np.array_equal(va, fx[f"split_val/{n}"])


def test_expand_and_numean():
    m = [np.array([[1, 0], [0, 1], [1, 1]]), np.array([[1, 0, 0, 1, 1]])]
    e = expand_masks(m, 2)
    assert e[0].shape == (6, 4) and e[1].shape == (1, 10)
    assert e[0][0].tolist() == [1, 1, 0, 0] and e[0][5].tolist() == [1, 1, 1, 1]
    a = np.arange(12, dtype=np.float32).reshape(2, 6)
    ref = np.array(np.split(a, 3, axis=1)).mean(axis=2).T
    assert np.array_equal(nu_mean(a, 2), ref)
