"""The plain VAE sibling (reference onto_vae/vae_model.py:913-1140) on the HIP path, through onto_vae_amd.VAE ->
ontovae_vae_* of the C ABI: against the fixtures of the imported reference (tests/golden/vae_*.npz) and, at sizes the
fixtures do not reach, against the numpy oracle (oracle/vae_oracle.py) with its float64 instance as the yardstick."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, act_close, sub
from test_oracle_vae import is_noise

pytestmark = pytest.mark.gpu
REL = 1e-4


def fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def build(fx, drop=0.0, z_drop=0.0, sd="sd0/"):
    from onto_vae_amd import VAE
    m = VAE(np.zeros((4, int(fx["G"])), np.float32), [int(v) for v in fx["enc"]], [int(v) for v in fx["dec"]], int(fx["Z"]), drop, z_drop)
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sub(fx, sd).items()})
    return m


def feed(m, eps):
    m.eps_source = lambda B: torch.as_tensor(eps[:B])


def grads_by_name(m):
    """The gradient buffer after train_step(apply_update=False), keyed like the reference's parameters."""
    e = m._engine
    g = {}
    for side, pre, n in (("enc", "encoder.encoder.%d", len(e.enc)), ("dec", "decoder.decoder.%d", len(e.dec))):
        for i in range(n):
            W, b, ga, be = e.block_views(side, i, e.Gr)
            p = pre % i
            g[p + ".0.weight"], g[p + ".0.bias"], g[p + ".1.weight"], g[p + ".1.bias"] = W, b, ga, be
    g["encoder.mu.0.weight"], g["encoder.mu.0.bias"], g["encoder.logvar.0.weight"], g["encoder.logvar.0.bias"] = e.head_views(e.Gr)
    g["decoder.decoder.%d.0.weight" % len(e.dec)], g["decoder.decoder.%d.0.bias" % len(e.dec)] = e.out_views(e.Gr)
    return {k: v.detach().cpu().numpy() for k, v in g.items()}


@pytest.mark.parametrize("name", ["vae_small", "vae_deep"])
def test_vae_matches_reference_fixture(name):
    from onto_vae_amd.vae_model import FusedAdamW
    fx = fixture(name)
    kc, lr, steps = float(fx["kl_coeff"]), float(fx["lr"]), int(fx["steps"])
    # state_dict round trip: the reference's keys and shapes
    m = build(fx)
    sd = m.state_dict()
    assert set(sd.keys()) == set(sub(fx, "sd0/").keys())
    for k, v in sub(fx, "sd0/").items():
        assert tuple(sd[k].shape) == tuple(np.asarray(v).shape), k
        np.testing.assert_array_equal(sd[k].cpu().numpy(), v)
    # train-mode forward
    m.train()
    feed(m, fx["eps/0"])
    recon, mu, lv = m(torch.as_tensor(fx["x/0"]))
    assert act_close(recon.cpu().numpy(), fx["fwd/recon"]) and act_close(mu.cpu().numpy(), fx["fwd/mu"])
    assert act_close(lv.cpu().numpy(), fx["fwd/log_var"])
    loss = m.vae_loss(recon, mu, lv, torch.as_tensor(fx["x/0"]).to(recon.device), kc)
    np.testing.assert_allclose(float(loss), fx["fwd/rec_kl_loss"][2], rtol=REL)
    # gradients of the first step
    m = build(fx)
    e = m._engine
    out = torch.zeros(3, device=e.device)
    e.train_step(torch.as_tensor(fx["x/0"]), None, torch.as_tensor(fx["eps/0"]), kc, None, out, apply_update=False)
    np.testing.assert_allclose(out.cpu().numpy(), [fx["train/rec"][0], fx["train/kl"][0], fx["train/loss"][0]], rtol=REL)
    g = grads_by_name(m)
    floor = 1e-6 * max(np.abs(v).max() for v in sub(fx, "grad0/").values())        # see tests/test_oracle_vae.py
    for k, want in sub(fx, "grad0/").items():
        if is_noise(k, fx):
            assert np.abs(g[k]).max() <= floor, k
        else:
            assert np.abs(g[k] - want).max() <= 2e-5 * np.abs(want).max() + floor, k
    # fused steps: losses and the state after them
    m = build(fx)
    m.train()
    opt = FusedAdamW(m, lr=lr)
    out = torch.zeros(steps, 3, device=m._engine.device)
    for s in range(steps):
        m._engine.train_step(torch.as_tensor(fx[f"x/{s}"]), None, torch.as_tensor(fx[f"eps/{s}"]), kc, opt.next_hyper(), out[s])
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:, 0], fx["train/rec"], rtol=REL)
    np.testing.assert_allclose(got[:, 1], fx["train/kl"], rtol=REL)
    np.testing.assert_allclose(got[:, 2], fx["train/loss"], rtol=REL)
    sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    for k, v in sub(fx, "sdT/").items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v), k
            continue
        d = np.abs(sd[k] - v)
        assert d.max() <= 2.1 * steps * lr, k
        if not is_noise(k, fx):
            noisy = "grad0/" + k in fx and np.abs(fx["grad0/" + k]).max() <= 100 * floor
            bad = d > 1e-6 + 1e-4 * np.abs(v)
            assert noisy or bad.mean() <= 0.01, (k, float(bad.mean()))
    # validation body and embedding on the reference's final weights
    m = build(fx, sd="sdT/")
    m.eval()
    feed(m, fx["val/eps"])
    out = torch.zeros(3, device=m._engine.device)
    m._engine.eval_step(torch.as_tensor(fx["val/x"]), None, torch.as_tensor(fx["val/eps"]), kc, out)
    np.testing.assert_allclose(out.cpu().numpy(), fx["val/rec_kl_loss"], rtol=REL)
    assert act_close(m.get_embedding(torch.as_tensor(fx["val/x"])).cpu().numpy(), fx["val/embedding"])
    mu, lv = m.encoder(torch.as_tensor(fx["val/x"]))
    assert mu.shape == (fx["val/x"].shape[0], int(fx["Z"]))


def random_state(o, rng):
    sd = {}
    for pre, cin, w in o.blocks():
        sd[pre + ".0.weight"] = rng.uniform(-1, 1, (w, cin)) / np.sqrt(cin)
        sd[pre + ".0.bias"] = rng.uniform(-1, 1, w) / np.sqrt(cin)
        sd[pre + ".1.weight"] = rng.uniform(0.5, 1.5, w)
        sd[pre + ".1.bias"] = rng.uniform(-0.3, 0.3, w)
        sd[pre + ".1.running_mean"] = rng.normal(0, 0.2, w)
        sd[pre + ".1.running_var"] = rng.uniform(0.5, 1.5, w)
        sd[pre + ".1.num_batches_tracked"] = np.int64(3)
    H, Hd, n = o.enc[-1], o.dec[-1], len(o.dec)
    for h in ("mu", "logvar"):
        sd[f"encoder.{h}.0.weight"] = rng.uniform(-1, 1, (o.Z, H)) / np.sqrt(H) * (0.3 if h == "logvar" else 1.0)
        sd[f"encoder.{h}.0.bias"] = rng.uniform(-1, 1, o.Z) / np.sqrt(H)
    sd[f"decoder.decoder.{n}.0.weight"] = rng.uniform(-1, 1, (o.G, Hd)) / np.sqrt(Hd)
    sd[f"decoder.decoder.{n}.0.bias"] = rng.uniform(-1, 1, o.G) / np.sqrt(Hd)
    return {k: np.asarray(v, np.float32) if not k.endswith("tracked") else v for k, v in sd.items()}


SHAPES = [
    # G, enc, dec, Z, B: the reference's defaults (vae_model.py:931) at a reduced gene count, then a deep ragged one
    (2000, [1000], [1000], 128, 128),
    (1237, [300, 171], [96, 203, 77], 50, 333),
]


def hip_model(o, sd, drop=0.0, z_drop=0.0):
    from onto_vae_amd import VAE
    m = VAE(np.zeros((4, o.G), np.float32), o.enc, o.dec, o.Z, drop, z_drop)
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    return m


@pytest.mark.parametrize("G,enc,dec,Z,B", SHAPES)
def test_vae_step_vs_oracle(G, enc, dec, Z, B):
    from onto_vae_amd.vae_model import FusedAdamW
    from vae_oracle import OracleVAE
    rng = np.random.default_rng(G + B)
    o32, o64 = OracleVAE(G, enc, dec, Z), OracleVAE(G, enc, dec, Z, dtype=np.float64)
    sd = random_state(o32, rng)
    x = rng.gamma(2.0, 1.0, (B, G)).astype(np.float32)
    eps = rng.standard_normal((B, Z)).astype(np.float32)
    kc, lr = 1e-4, 1e-3
    o32.load_state_dict(sd); o64.load_state_dict(sd)
    r32 = o32.gradients(x, eps, kc)
    r64 = o64.gradients(x, eps, kc)
    m = hip_model(o32, sd)
    e = m._engine
    out = torch.zeros(3, device=e.device)
    e.train_step(torch.as_tensor(x), None, torch.as_tensor(eps), kc, None, out, apply_update=False)
    np.testing.assert_allclose(out.cpu().numpy(), [float(v) for v in r64[:3]], rtol=REL)
    g = grads_by_name(m)
    gmax = max(np.abs(v).max() for v in r64[3].values())
    for k, want in r64[3].items():
        # the HIP gradient may miss the float64 one by twice what the float32 oracle misses it by (floor 2e-6 of the key's
        # scale + 1e-6 of the model's largest gradient entry, for cancelling sums)
        scale = np.abs(want).max() + 1e-30
        e_hip, e_ref = np.abs(g[k] - want).max(), np.abs(r32[3][k] - want).max()
        assert e_hip <= 2 * e_ref + 2e-6 * scale + 1e-6 * gmax, (k, e_hip / scale, e_ref / scale)
    # BatchNorm running statistics after that one train-mode pass
    sd_h = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    for k, v in o64.state_dict().items():
        if k.endswith("running_var") or k.endswith("running_mean"):
            np.testing.assert_allclose(sd_h[k], v, rtol=1e-4, atol=1e-5)
        elif k.endswith("tracked"):
            assert int(sd_h[k]) == int(v)
    # three fused steps against the oracle's
    o32.load_state_dict(sd); o32.opt = None
    m = hip_model(o32, sd)
    m.train()
    opt = FusedAdamW(m, lr=lr)
    out = torch.zeros(3, 3, device=m._engine.device)
    want = []
    for s in range(3):
        xs = np.roll(x, s, axis=0)
        m._engine.train_step(torch.as_tensor(xs), None, torch.as_tensor(eps), kc, opt.next_hyper(), out[s])
        want.append([float(v) for v in o32.train_step(xs, eps, lr, kc)])
    np.testing.assert_allclose(out.cpu().numpy(), np.asarray(want), rtol=2e-4)
    sd_h = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    for k, v in o32.state_dict().items():
        if k.endswith("tracked"):
            continue
        d = np.abs(sd_h[k] - v)
        assert d.max() <= 2.1 * 3 * lr + 1e-4 * np.abs(v).max(), (k, float(d.max()))
    # eval-mode pass on the moved weights
    m.eval()
    feed(m, eps)
    recon, mu, lv = m(torch.as_tensor(x))
    o32b = OracleVAE(G, enc, dec, Z)
    o32b.load_state_dict(sd_h)
    r, mu_o, lv_o, _ = o32b.forward(x, eps, False)
    assert act_close(recon.cpu().numpy(), r) and act_close(mu.cpu().numpy(), mu_o) and act_close(lv.cpu().numpy(), lv_o)


def test_vae_dropout_on_gradients_match_oracle_with_same_masks():
    """drop / z_drop > 0: the oracle takes the keep-scales of the HIP path's Philox stream (oracle/philox_oracle.py), block k
    keyed by rng_offset ^ (k + 1) << 48, the heads by rng_offset itself."""
    import philox_oracle as ph
    from vae_oracle import OracleVAE
    G, enc, dec, Z, B = 310, [64, 40], [48], 24, 96
    drop, z_drop = 0.2, 0.5
    rng = np.random.default_rng(5)
    o = OracleVAE(G, enc, dec, Z)
    sd = random_state(o, rng)
    o.load_state_dict(sd)
    x = rng.gamma(2.0, 1.0, (B, G)).astype(np.float32)
    eps = rng.standard_normal((B, Z)).astype(np.float32)
    m = hip_model(o, sd, drop, z_drop)
    e = m._engine
    off = e.rng_offset + 1
    scales = {"mu": ph.dropout_scale(z_drop, e.seed, off, ph.STREAM_ZDROP_MU, Z, B),
              "logvar": ph.dropout_scale(z_drop, e.seed, off, ph.STREAM_ZDROP_LV, Z, B)}
    for k, (pre, _, w) in enumerate(o.blocks()):
        scales[pre] = ph.dropout_scale(drop, e.seed, off ^ ((k + 1) << 48), ph.STREAM_DROP, w, B)
    assert 0.1 < (scales["encoder.encoder.0"] == 0).mean() < 0.3 and 0.35 < (scales["mu"] == 0).mean() < 0.65
    want = o.gradients(x, eps, 1e-3, scales)
    out = torch.zeros(3, device=e.device)
    e.train_step(torch.as_tensor(x), None, torch.as_tensor(eps), 1e-3, None, out, apply_update=False)
    np.testing.assert_allclose(out.cpu().numpy(), [float(v) for v in want[:3]], rtol=REL)
    g = grads_by_name(m)
    gmax = max(np.abs(v).max() for v in want[3].values())
    for k, w in want[3].items():
        assert np.abs(g[k] - w).max() <= 5e-5 * np.abs(w).max() + 2e-6 * gmax, k
    # eval mode ignores both rates (fresh model: the pass above moved the running statistics)
    m = hip_model(o, sd, drop, z_drop)
    m.eval()
    feed(m, eps)
    m2 = hip_model(o, sd)
    m2.eval()
    feed(m2, eps)
    a, b = m(torch.as_tensor(x)), m2(torch.as_tensor(x))
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_vae_gradients_large_ragged_batch_with_dropout():
    """1100 rows (1152 padded columns): the 16-B-per-lane passes of the BatchNorm backward, which the dense blocks share with the
    OntoVAE encoder, with dropout masks on and padded columns in play: loss and every gradient against the oracle."""
    import philox_oracle as ph
    from vae_oracle import OracleVAE
    G, enc, dec, Z, B = 310, [64, 40], [48], 24, 1100
    drop, z_drop = 0.2, 0.5
    rng = np.random.default_rng(15)
    o = OracleVAE(G, enc, dec, Z)
    sd = random_state(o, rng)
    o.load_state_dict(sd)
    x = rng.gamma(2.0, 1.0, (B, G)).astype(np.float32)
    eps = rng.standard_normal((B, Z)).astype(np.float32)
    m = hip_model(o, sd, drop, z_drop)
    e = m._engine
    off = e.rng_offset + 1
    scales = {"mu": ph.dropout_scale(z_drop, e.seed, off, ph.STREAM_ZDROP_MU, Z, B),
              "logvar": ph.dropout_scale(z_drop, e.seed, off, ph.STREAM_ZDROP_LV, Z, B)}
    for k, (pre, _, w) in enumerate(o.blocks()):
        scales[pre] = ph.dropout_scale(drop, e.seed, off ^ ((k + 1) << 48), ph.STREAM_DROP, w, B)
    want = o.gradients(x, eps, 1e-3, scales)
    out = torch.zeros(3, device=e.device)
    e.train_step(torch.as_tensor(x), None, torch.as_tensor(eps), 1e-3, None, out, apply_update=False)
    np.testing.assert_allclose(out.cpu().numpy(), [float(v) for v in want[:3]], rtol=REL)
    g = grads_by_name(m)
    gmax = max(np.abs(v).max() for v in want[3].values())
    for k, w in want[3].items():
        assert np.abs(g[k] - w).max() <= 5e-5 * np.abs(w).max() + 2e-6 * gmax, k


def test_vae_train_model_protocol(tmp_path, capsys):
    """train_model (vae_model.py:1082-1140): split, loaders, printed lines, best-model checkpoint in the reference's format."""
    from onto_vae_amd import VAE
    rng = np.random.default_rng(11)
    X = rng.gamma(2.0, 1.0, (300, 200)).astype(np.float32)
    torch.manual_seed(3)
    m = VAE(X, [64], [64], 16, drop=0.2, z_drop=0.5)
    path = str(tmp_path / "best.pt")
    m.train_model(path, lr=1e-3, kl_coeff=1e-4, batch_size=64, epochs=3)
    lines = capsys.readouterr().out.splitlines()
    assert lines[0] == "Epoch 1 of 3" and any(l.startswith("Train Loss: ") for l in lines) and "New best model!" in lines
    tr = [float(l.split(": ")[1]) for l in lines if l.startswith("Train Loss")]
    assert tr[-1] < tr[0]
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}
    m2 = VAE(X, [64], [64], 16, drop=0.2, z_drop=0.5)
    m2.load_state_dict(ck["model_state_dict"])
    dl_x = torch.as_tensor(X[:50])
    m.eval(); m2.eval()
    if ck["epoch"] == 2:
        eps = torch.randn(50, 16)
        feed(m, eps.numpy()); feed(m2, eps.numpy())
        assert torch.allclose(m(dl_x)[0], m2(dl_x)[0])


def test_vae_rejects_bad_arguments():
    from onto_vae_amd import VAE
    m = VAE(np.zeros((4, 30), np.float32), [8], [8], 4, 0.0, 0.0)
    with pytest.raises(ValueError):
        m(torch.zeros(5, 31))
    m.train()
    with pytest.raises(ValueError, match="more than 1 sample"):
        m(torch.zeros(1, 30))
    with pytest.raises(ValueError):
        VAE(np.zeros((4, 30), np.float32), [8] * 9, [8], 4)


def test_vae_get_embedding_leaves_the_decoder_alone():
    """VAE.get_embedding and Encoder.forward run the encoder only (vae_model.py:970-981, modules.py:74-84): in train mode the
    encoder's BatchNorm buffers move, the decoder's running_mean / running_var / num_batches_tracked must not; the values equal
    those of the full forward."""
    fx = fixture("vae_deep")
    m = build(fx)
    m.train()
    x = torch.as_tensor(fx["x/0"])
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    feed(m, fx["eps/0"])
    z = m.get_embedding(x)
    mu, lv = m.encoder(x)
    after = m.state_dict()
    for k, v in before.items():
        if k.startswith("decoder."):
            assert torch.equal(after[k], v), k
    assert int(after["encoder.encoder.0.1.num_batches_tracked"]) == int(before["encoder.encoder.0.1.num_batches_tracked"]) + 2
    assert not torch.equal(after["encoder.encoder.0.1.running_mean"], before["encoder.encoder.0.1.running_mean"])
    m2 = build(fx)
    m2.train()
    feed(m2, fx["eps/0"])
    _, mu2, lv2 = m2.forward(x)
    assert torch.equal(mu2, mu) or torch.allclose(mu2, mu, rtol=1e-5, atol=1e-6)
    want_z = mu2 + torch.as_tensor(fx["eps/0"]).cuda() * torch.exp(0.5 * lv2)
    assert torch.allclose(z, want_z, rtol=1e-5, atol=1e-6)
