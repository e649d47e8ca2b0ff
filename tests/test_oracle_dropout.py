"""Dropout-on semantics of the CPU oracle against the fixture produced by the imported reference with its Dropout modules
fed fixed keep-scales (tools/gen_golden.py::gen_dropout_case), the float64 instance of the oracle against the reference
modules run in float64, and the host Philox restatement against Random123's known-answer vectors."""
import os

import numpy as np

from conftest import GOLDEN, NOISE_KEYS, sub
from ontovae_oracle import OracleOntoVAE
import philox_oracle as ph

REL = 1e-4


def load_drop():
    fx = dict(np.load(os.path.join(GOLDEN, "g1_drop.npz")))
    masks = [fx[f"mask/{i}"].astype(np.int64) for i in range(int(fx["K"]))]
    return fx, masks


def test_oracle_dropout_steps_match_reference():
    fx, masks = load_drop()
    o = OracleOntoVAE(masks, int(fx["nu"]))
    o.load_state_dict(sub(fx, "sd0/"))
    lr, kc = float(fx["lr"]), float(fx["kl_coeff"])
    sc = lambda s: (fx[f"drop_scale/{s}"], fx[f"zmu_scale/{s}"], fx[f"zlv_scale/{s}"])
    # keep-scales are {0, 1/(1-p)} at the reference's default rates
    assert set(np.unique(fx["drop_scale/0"])) == {np.float32(0), np.float32(1 / 0.8)}
    assert set(np.unique(fx["zmu_scale/0"])) == {np.float32(0), np.float32(2)}
    rec, kl, loss, g = o.gradients(fx["x/0"], fx["eps/0"], kc, sc(0))
    o.buf = {"running_mean": np.asarray(fx["sd0/encoder.encoder.0.1.running_mean"]).copy(),
             "running_var": np.asarray(fx["sd0/encoder.encoder.0.1.running_var"]).copy(), "num_batches_tracked": 0}
    for k, want in sub(fx, "grad0/").items():
        got = g[k]
        if k.startswith("decoder") and k.endswith("weight"):
            got = got * o.masks[int(k.split(".")[2])]
        if k == "encoder.encoder.0.0.bias":          # analytically zero: rounding noise on both sides
            assert np.abs(got).max() <= 1e-4 * np.abs(g["encoder.encoder.0.1.bias"]).max()
            continue
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), k
    for s in range(int(fx["steps"])):
        rec, kl, loss = o.train_step(fx[f"x/{s}"], fx[f"eps/{s}"], lr, kc, sc(s))
        np.testing.assert_allclose([rec, kl, loss], [fx["train/rec"][s], fx["train/kl"][s], fx["train/loss"][s]], rtol=REL)
    sd = o.state_dict()
    for k, v in sub(fx, "sdT/").items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v)
        elif k in NOISE_KEYS:
            assert np.abs(sd[k] - v).max() <= 2.1 * int(fx["steps"]) * lr
        else:
            np.testing.assert_allclose(sd[k], v, rtol=1e-4, atol=1e-6, err_msg=k)
    # eval mode: dropout is the identity (on the reference's trained weights: the noise keys feed running_mean)
    o = OracleOntoVAE(masks, int(fx["nu"]))
    o.load_state_dict(sub(fx, "sdT/"))
    np.testing.assert_allclose(o.val_step(fx["val/x"], fx["val/eps"], kc), fx["val/rec_kl_loss"], rtol=REL)


def test_float64_oracle_matches_reference_float64(case):
    """The float64 yardstick: the oracle in double against the reference modules in double, to 1e-12 of scale."""
    _, fx, masks = case
    o = OracleOntoVAE(masks, int(fx["nu"]), dtype=np.float64)
    o.load_state_dict(sub(fx, "sd0/"))
    recon, mu, lv = o.forward(fx["x/0"], fx["eps/0"], train=True)
    for got, key in ((recon, "fwd64/recon"), (mu, "fwd64/mu"), (lv, "fwd64/log_var")):
        assert got.dtype == np.float64 and np.abs(got - fx[key]).max() <= 1e-12 * np.abs(fx[key]).max(), key
    o = OracleOntoVAE(masks, int(fx["nu"]), dtype=np.float64)
    o.load_state_dict(sub(fx, "sdT/"))
    for what in ("act", "rec"):
        got = o.pass_data(fx["inf/X"], fx["inf/eps1"], fx["inf/eps2"], what)
        assert np.abs(got - fx[f"inf64/{what}"]).max() <= 1e-12 * np.abs(fx[f"inf64/{what}"]).max(), what


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 with 10 rounds."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = ph.philox4x32_10([np.uint64(c) for c in ctr], key)
        assert tuple(int(v) for v in got) == want


def test_dropout_scale_statistics():
    s = ph.dropout_scale(0.2, 12345, 7, ph.STREAM_DROP, 64, 4096)
    assert set(np.unique(s)) == {np.float32(0), np.float32(1.0) / (np.float32(1.0) - np.float32(0.2))}
    n = s.size
    keep = (s > 0).mean()
    assert abs(keep - 0.8) <= 3 * np.sqrt(0.8 * 0.2 / n)
    # streams and offsets decorrelate
    s2 = ph.dropout_scale(0.2, 12345, 8, ph.STREAM_DROP, 64, 4096)
    assert abs(((s > 0) == (s2 > 0)).mean() - (0.64 + 0.04)) < 0.01
    e = ph.randn(1, 1, ph.STREAM_EPS, 32, 8192)
    assert abs(e.mean()) < 0.01 and abs(e.std() - 1) < 0.01
