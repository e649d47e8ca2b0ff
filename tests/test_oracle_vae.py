"""The dense VAE oracle (oracle/vae_oracle.py) against fixtures produced by the imported reference VAE (tools/gen_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, act_close, sub
from vae_oracle import OracleVAE

REL = 1e-4
VAE_CASES = ["vae_small", "vae_deep"]


def load(name):
    fx = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    o = OracleVAE(int(fx["G"]), fx["enc"], fx["dec"], int(fx["Z"]))
    return fx, o


def is_noise(k, fx):
    """Linear bias in front of a train-mode BatchNorm (analytically zero gradient) and the running mean that contains it."""
    n_dec = len(fx["dec"])
    if k.endswith("running_mean"):
        return True
    if k.endswith(".0.bias"):
        return not (k.startswith("encoder.mu") or k.startswith("encoder.logvar") or k.startswith(f"decoder.decoder.{n_dec}."))
    return False


@pytest.mark.parametrize("name", VAE_CASES)
def test_vae_oracle_forward_gradients_steps(name):
    fx, o = load(name)
    o.load_state_dict(sub(fx, "sd0/"))
    kc, lr = float(fx["kl_coeff"]), float(fx["lr"])
    recon, mu, lv, _ = o.forward(fx["x/0"], fx["eps/0"], True)
    assert act_close(recon, fx["fwd/recon"]) and act_close(mu, fx["fwd/mu"]) and act_close(lv, fx["fwd/log_var"])
    np.testing.assert_allclose(o.vae_loss(recon, mu, lv, fx["x/0"], kc), fx["fwd/rec_kl_loss"], rtol=REL)
    o.load_state_dict(sub(fx, "sd0/"))
    _, _, _, g = o.gradients(fx["x/0"], fx["eps/0"], kc)
    # absolute floor: 1e-6 of the largest gradient entry of the model.  Sums over the batch that cancel (mu bias: 1e-3 out of
    # terms of order 10; Linear biases in front of a train-mode BatchNorm: analytically zero) carry the fp32 noise of their
    # terms, in the reference's own evaluation as much as in the oracle's (a float64 oracle differs from the fixture by the same).
    floor = 1e-6 * max(np.abs(v).max() for v in sub(fx, "grad0/").values())
    for k, want in sub(fx, "grad0/").items():
        if is_noise(k, fx):
            assert np.abs(g[k]).max() <= floor, k
            continue
        assert np.abs(g[k] - want).max() <= 2e-5 * np.abs(want).max() + floor, k
    o.load_state_dict(sub(fx, "sd0/"))
    for s in range(int(fx["steps"])):
        got = o.train_step(fx[f"x/{s}"], fx[f"eps/{s}"], lr, kc)
        np.testing.assert_allclose(got, [fx["train/rec"][s], fx["train/kl"][s], fx["train/loss"][s]], rtol=REL)
    sd = o.state_dict()
    steps_lr = int(fx["steps"]) * lr
    for k, v in sub(fx, "sdT/").items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v)
        elif is_noise(k, fx):
            assert np.abs(sd[k] - v).max() <= 2.1 * steps_lr
        else:
            d = np.abs(sd[k] - v)
            bad = d > 1e-6 + 1e-4 * np.abs(v)
            # AdamW normalises each entry by its own gradient scale: a key whose gradient is mostly cancellation noise
            # (mu bias, see above) moves by a noisy fraction of lr per step and is held to the step bound only
            noisy = "grad0/" + k in fx and np.abs(fx["grad0/" + k]).max() <= 100 * floor
            assert (noisy or bad.mean() <= 0.01) and d.max() <= 2.1 * steps_lr, (k, float(bad.mean()), float(d.max()))
    o2 = load(name)[1]
    o2.load_state_dict(sub(fx, "sdT/"))
    np.testing.assert_allclose(o2.val_step(fx["val/x"], fx["val/eps"], kc), fx["val/rec_kl_loss"], rtol=REL)
    _, mu, lv, f = o2.forward(fx["val/x"], fx["val/eps"], False)
    assert act_close(f["z"], fx["val/embedding"])
