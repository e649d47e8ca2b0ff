"""OntoEncVAE (reference onto_vae/vae_model.py:468-905: ontology-structured encoder, dense decoder) on the HIP path, through
onto_vae_amd.OntoEncVAE -> ontovae_decoder_forward / ontovae_vae_decode_step / ontovae_decoder_backward of the C ABI: against the
fixtures of the imported reference (tests/golden/encvae_*.npz, tools/gen_golden_encvae.py) and, at a size the fixtures do not
reach, against the numpy oracle (oracle/encvae_oracle.py), with the heads' and the decoder's dropout on under shared masks."""
import numpy as np
import pytest
import torch

from conftest import act_close, sub
from test_oracle_encvae import ENCVAE_CASES, is_noise, load_encvae

pytestmark = pytest.mark.gpu
REL = 1e-4


class Stub:
    def __init__(self, masks, X, names=None):
        G = masks[0].shape[1]
        self.description = "synthetic"
        self.genes = {"1000_30": [f"G{i:06d}" for i in range(G)]}
        self.masks = {"1000_30": {"encoder": masks}}
        self.data = {"1000_30": {"d": X}}


def build(masks, sd, drop=0.0, z_drop=0.0, X=None):
    from onto_vae_amd import OntoEncVAE
    G = masks[0].shape[1]
    onto = Stub(masks, X if X is not None else np.zeros((4, G), np.float32))
    m = OntoEncVAE(onto, "d", neuronnum=1, drop=drop, z_drop=z_drop)
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    return m, onto


def feed(m, *eps):
    q = [torch.as_tensor(e) for e in eps]
    m.eps_source = lambda B: q.pop(0)[:B] if len(q) > 1 else q[0][:B]


def grads_by_name(m):
    """The gradient buffers after train_step(apply_update=False), keyed like the reference's parameters."""
    e, d = m.encoder._engine, m._dense
    K, Z = e.plan.K, m.latent_dim
    g = {}
    for i in range(K - 1):
        g[f"encoder.encoder.{i}.0.weight"] = e.layer_dense(i, e.Gr)
        g[f"encoder.encoder.{i}.0.bias"] = e.layer_bias(i, e.Gr)
    W, b = e.layer_dense(K - 1, e.Gr), e.layer_bias(K - 1, e.Gr)
    g["encoder.mu.0.weight"], g["encoder.logvar.0.weight"] = W[:Z], W[Z:]
    g["encoder.mu.0.bias"], g["encoder.logvar.0.bias"] = b[:Z], b[Z:]
    for i in range(len(d.dec)):
        Wd, bd, ga, be = d.block_views("dec", i, d.Gr)
        p = f"decoder.decoder.{i}"
        g[p + ".0.weight"], g[p + ".0.bias"], g[p + ".1.weight"], g[p + ".1.bias"] = Wd, bd, ga, be
    g[f"decoder.decoder.{len(d.dec)}.0.weight"], g[f"decoder.decoder.{len(d.dec)}.0.bias"] = d.out_views(d.Gr)
    return {k: v.detach().cpu().numpy() for k, v in g.items()}


@pytest.mark.parametrize("name", ENCVAE_CASES)
def test_encvae_matches_reference_fixture(name):
    from onto_vae_amd.enc_vae_model import _EncVaeAdamW
    fx, masks, _ = load_encvae(name)
    kc, lr, steps = float(fx["kl_coeff"]), float(fx["lr"]), int(fx["steps"])
    # state_dict round trip: the reference's keys, shapes and values
    m, _ = build(masks, sub(fx, "sd0/"))
    sd = m.state_dict()
    assert set(sd.keys()) == set(sub(fx, "sd0/").keys())
    for k, v in sub(fx, "sd0/").items():
        assert tuple(sd[k].shape) == tuple(np.asarray(v).shape), k
        np.testing.assert_array_equal(sd[k].cpu().numpy(), v)
    # eval-mode forward
    m.eval()
    feed(m, fx["eps/0"])
    recon, mu, lv = m(torch.as_tensor(fx["x/0"]))
    assert act_close(recon.cpu().numpy(), fx["fwd/recon"]) and act_close(mu.cpu().numpy(), fx["fwd/mu"])
    assert act_close(lv.cpu().numpy(), fx["fwd/log_var"])
    # gradients of the first step
    m, _ = build(masks, sub(fx, "sd0/"))
    m.train()
    out = torch.zeros(3, device="cuda")
    m.train_step(torch.as_tensor(fx["x/0"]), None, kc, None, out, eps=torch.as_tensor(fx["eps/0"]), apply_update=False)
    np.testing.assert_allclose(out.cpu().numpy(), [fx["train/rec"][0], fx["train/kl"][0], fx["train/loss"][0]], rtol=REL)
    g = grads_by_name(m)
    floor = 1e-6 * max(np.abs(v).max() for v in sub(fx, "grad0/").values())
    for k, want in sub(fx, "grad0/").items():
        if is_noise(k):
            assert np.abs(g[k]).max() <= floor, k
        else:
            assert np.abs(g[k] - want).max() <= 2e-5 * np.abs(want).max() + floor, k
    # fused steps: losses and the state after them
    m, _ = build(masks, sub(fx, "sd0/"))
    m.train()
    opt = _EncVaeAdamW(m, lr=lr)
    out = torch.zeros(steps, 3, device="cuda")
    for s in range(steps):
        m.train_step(torch.as_tensor(fx[f"x/{s}"]), None, kc, opt.next_hyper(), out[s], eps=torch.as_tensor(fx[f"eps/{s}"]))
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:, 0], fx["train/rec"], rtol=REL)
    np.testing.assert_allclose(got[:1, 1], fx["train/kl"][:1], rtol=REL)
    np.testing.assert_allclose(got[:, 1], fx["train/kl"], rtol=1e-3)        # see tests/test_oracle_encvae.py
    np.testing.assert_allclose(got[:, 2], fx["train/loss"], rtol=REL)
    sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    for k, v in sub(fx, "sdT/").items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v), k
            continue
        d = np.abs(sd[k] - v)
        assert d.max() <= 2.1 * steps * lr, k
        if not is_noise(k):
            noisy = "grad0/" + k in fx and np.abs(fx["grad0/" + k]).max() <= 100 * floor
            bad = d > 1e-6 + 1e-4 * np.abs(v)
            assert noisy or bad.mean() <= 0.01, (k, float(bad.mean()))
    for k in ["encoder.mu.0.weight", "encoder.logvar.0.weight"] + [f"encoder.encoder.{i}.0.weight" for i in range(len(masks) - 1)]:
        assert (sd[k] >= 0).all(), k                                         # the clamp (vae_model.py:622-625)
    # validation body, embedding and retrieval on the reference's final weights
    m, onto = build(masks, sub(fx, "sdT/"))
    feed(m, fx["val/eps"])
    np.testing.assert_allclose(m.val_round([(torch.as_tensor(fx["val/x"]),)], kc), fx["val/rec_kl_loss"][2], rtol=REL)
    m.eval()
    assert act_close(m.get_embedding(torch.as_tensor(fx["val/x"])).cpu().numpy(), fx["val/embedding"])
    onto.data["1000_30"]["v"] = fx["val/x"]
    feed(m, fx["val/eps"], fx["val/eps2"])
    assert act_close(m.get_pathway_activities(onto, "v"), fx["val/act"])
    feed(m, fx["val/eps"], fx["val/eps2"])
    assert act_close(m.get_reconstructed_values(onto, "v"), fx["val/rec"])
    genes = [onto.genes["1000_30"][3], onto.genes["1000_30"][7]]
    feed(m, fx["val/eps"], fx["val/eps2"])
    sub_rec = m.get_reconstructed_values(onto, "v", rec_genes=genes)
    np.testing.assert_array_equal(sub_rec, m_rec_cols(m, onto, fx, [3, 7]))


def m_rec_cols(m, onto, fx, cols):
    feed(m, fx["val/eps"], fx["val/eps2"])
    return m.get_reconstructed_values(onto, "v")[:, cols]


def test_encvae_step_vs_oracle_with_dropout_on_heads_and_decoder():
    """A larger random model (2 000 genes, 4 levels, 300 rows) with the reference's default rates, drop = 0.2 behind every
    masked encoder block and in the Decoder, z_drop = 0.5 on the heads: loss and every gradient against the oracle fed with the
    keep-scales of the HIP path's Philox streams; then three fused steps with dropout off against the oracle's."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import philox_oracle as ph
    from encvae_oracle import OracleOntoEncVAE
    from onto_vae_amd.enc_vae_model import _EncVaeAdamW
    from onto_vae_amd.synthetic import synthetic_links
    rng = np.random.RandomState(11)
    counts, G, B = (12, 40, 120, 300), 2000, 300
    off, term_parents, gene_parents = synthetic_links(counts, G, rng)
    K = len(counts)
    levels = [("g", list(range(G)))] + [("t", list(range(off[d], off[d + 1]))) for d in range(K - 1, -1, -1)]
    masks = []
    for j in range(K):
        rows = levels[j + 1][1]
        rpos = {t: r for r, t in enumerate(rows)}
        blocks = []
        for lv in range(j, -1, -1):
            kind, cols = levels[lv]
            mm = np.zeros((len(rows), len(cols)), np.uint8)
            for c, node in enumerate(cols):
                for p in (gene_parents[node] if kind == "g" else term_parents[node]):
                    if p in rpos:
                        mm[rpos[p], c] = 1
            blocks.append(mm)
        masks.append(np.hstack(blocks))
    Z = masks[-1].shape[0]
    o = OracleOntoEncVAE(masks)
    sd = {}
    for i, mk in enumerate(masks[:-1]):
        b = 1.0 / np.sqrt(max(mk.sum(axis=1).mean(), 1.0))
        sd[f"encoder.encoder.{i}.0.weight"] = (rng.uniform(-b, b, mk.shape) * mk).astype(np.float32)
        sd[f"encoder.encoder.{i}.0.bias"] = rng.uniform(-0.1, 0.1, mk.shape[0]).astype(np.float32)
    b = 1.0 / np.sqrt(max(masks[-1].sum(axis=1).mean(), 1.0))
    for h in ("mu", "logvar"):
        sd[f"encoder.{h}.0.weight"] = (rng.uniform(-b, b, masks[-1].shape) * masks[-1] * 0.3).astype(np.float32)
        sd[f"encoder.{h}.0.bias"] = rng.uniform(-0.1, 0.1, Z).astype(np.float32)
    sd["decoder.decoder.0.0.weight"] = rng.uniform(-0.3, 0.3, (Z, Z)).astype(np.float32)
    sd["decoder.decoder.0.0.bias"] = rng.uniform(-0.1, 0.1, Z).astype(np.float32)
    sd["decoder.decoder.0.1.weight"] = (1 + 0.1 * rng.standard_normal(Z)).astype(np.float32)
    sd["decoder.decoder.0.1.bias"] = (0.1 * rng.standard_normal(Z)).astype(np.float32)
    sd["decoder.decoder.0.1.running_mean"], sd["decoder.decoder.0.1.running_var"] = np.zeros(Z, np.float32), np.ones(Z, np.float32)
    sd["decoder.decoder.0.1.num_batches_tracked"] = np.int64(0)
    sd["decoder.decoder.1.0.weight"] = rng.uniform(-0.3, 0.3, (G, Z)).astype(np.float32)
    sd["decoder.decoder.1.0.bias"] = rng.uniform(-0.1, 0.1, G).astype(np.float32)
    o.load_state_dict(sd)
    x = np.log1p(rng.poisson(0.8, size=(B, G))).astype(np.float32)
    eps = rng.standard_normal((B, Z)).astype(np.float32)
    drop, z_drop, kc, lr = 0.2, 0.5, 1e-3, 1e-3
    m, _ = build(masks, sd, drop, z_drop)
    m.train()
    d, e = m._dense, m.encoder._engine
    off_ = d.rng_offset + 1
    scales = {"mu": ph.dropout_scale(z_drop, d.seed, off_, ph.STREAM_ZDROP_MU, Z, B),
              "logvar": ph.dropout_scale(z_drop, d.seed, off_, ph.STREAM_ZDROP_LV, Z, B),
              # the Decoder's block is block 1 of the dense engine (behind its one-unit dummy encoder block)
              "decoder.decoder.0": ph.dropout_scale(drop, d.seed, off_ ^ (2 << 48), ph.STREAM_DROP, Z, B)}
    # the masked encoder's blocks: one Philox stream over the operator's term nodes, keyed by (node, batch column)
    nterm = e.plan.C - e.plan.L
    full = ph.dropout_scale(drop, e.seed, e.rng_offset + 1, ph.STREAM_DROP, nterm, B)
    for i in range(e.plan.K - 1):
        n0, no = e.plan.layer_nodes[i]
        scales[f"encoder.encoder.{i}"] = full[:, n0:n0 + no]
    assert 0.35 < (scales["mu"] == 0).mean() < 0.65 and 0.1 < (full == 0).mean() < 0.3
    want = o.gradients(x, eps, kc, scales)
    out = torch.zeros(3, device="cuda")
    m.train_step(torch.as_tensor(x), None, kc, None, out, eps=torch.as_tensor(eps), apply_update=False)
    np.testing.assert_allclose(out.cpu().numpy(), [float(v) for v in want[:3]], rtol=REL)
    g = grads_by_name(m)
    gmax = max(np.abs(v).max() for v in want[3].values())
    for k, w in want[3].items():
        assert np.abs(g[k] - w).max() <= 5e-5 * np.abs(w).max() + 2e-6 * gmax, k
    # three fused steps from an index list into an HBM-resident dataset (the loader's zero-copy protocol), dropout off
    o.load_state_dict(sd)
    m, _ = build(masks, sd)
    m.train()
    X = torch.as_tensor(np.vstack([x, x[::-1]])).cuda().contiguous()
    idx = torch.arange(B, dtype=torch.int32, device="cuda")
    opt = _EncVaeAdamW(m, lr=lr)
    out = torch.zeros(3, 3, device="cuda")
    want_l = []
    for s in range(3):
        e_s = rng.standard_normal((B, Z)).astype(np.float32)
        want_l.append([float(v) for v in o.train_step(x, e_s, lr, kc)])
        m.train_step(X, idx, kc, opt.next_hyper(), out[s], eps=torch.as_tensor(e_s))
    np.testing.assert_allclose(out.cpu().numpy()[:, 0], np.array(want_l)[:, 0], rtol=REL)
    np.testing.assert_allclose(out.cpu().numpy()[:, 1], np.array(want_l)[:, 1], rtol=1e-3)
    sdo, sdm = o.state_dict(), {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    for k in o.param_names():
        dd = np.abs(sdm[k] - sdo[k])
        assert dd.max() <= 2.1 * 3 * lr, k


def test_encvae_train_model_protocol_and_refusals(tmp_path, capsys):
    """train_model's protocol (vae_model.py:662-720): split, printed lines, checkpoint keys, reload; the refusals."""
    from onto_vae_amd import OntoEncVAE
    fx, masks, _ = load_encvae("encvae_small")
    G = masks[0].shape[1]
    rng = np.random.RandomState(3)
    X = np.log1p(rng.poisson(0.8, size=(200, G))).astype(np.float32)
    torch.manual_seed(5)
    m, onto = build(masks, sub(fx, "sd0/"), X=X)
    path = str(tmp_path / "best.pt")
    m.train_model(path, lr=1e-3, kl_coeff=1e-4, batch_size=64, epochs=3)
    outp = capsys.readouterr().out
    assert outp.count("Epoch ") == 3 and "Train Loss:" in outp and "Val Loss:" in outp and "New best model!" in outp
    ck = torch.load(path, weights_only=True)
    assert set(ck.keys()) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}
    m2, _ = build(masks, sub(fx, "sd0/"), X=X)
    m2.load_state_dict(ck["model_state_dict"])
    with pytest.raises(ValueError):
        OntoEncVAE(onto, "d", neuronnum=3)
    with pytest.raises(ValueError):
        OntoEncVAE(onto, "nope")
    # the reference's default rates: a round trains, and eval mode ignores every Dropout
    m3, _ = build(masks, sub(fx, "sd0/"), 0.2, 0.5, X=X)
    assert np.isfinite(m3.train_round([(torch.as_tensor(X[:64]),)], 1e-3, 1e-4, None))
    m4, _ = build(masks, sub(fx, "sd0/"), 0.2, 0.5, X=X)
    m5, _ = build(masks, sub(fx, "sd0/"), X=X)
    m4.eval(); m5.eval()
    eps = torch.zeros(8, m4.latent_dim)
    feed(m4, eps); feed(m5, eps)
    for u, v in zip(m4(torch.as_tensor(X[:8])), m5(torch.as_tensor(X[:8]))):
        assert torch.equal(u, v)
