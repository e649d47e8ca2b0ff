"""Device statistics of a perturbation screen (ontovae_wilcoxon_zsplit, ontovae_fdr_bh, ontovae_to_feature_major) through
the C ABI against oracle/stats_oracle.py, which tests/test_stats_oracle.py pins against scipy (onto_vae/ontobj.py:673).
Bar: the statistic (a sum of half-integer ranks) bit-exact; p- and q-values, float64, to 1e-11 relative."""
import numpy as np
import pytest
import torch

from oracle import stats_oracle as so
from onto_vae_amd import stats as dstats

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
P_RTOL = 1e-11


def _fm(arr):
    return dstats._feature_major_copy(arr, torch.device(DEV))


def _pair(rng, n, c, kind):
    x = rng.standard_normal((n, c)).astype(np.float32)
    y = (x + 0.3 * rng.standard_normal((n, c)) - 0.02).astype(np.float32)
    if kind == "ties":
        x, y = (np.round(x * 2) / 2).astype(np.float32), (np.round(y * 2) / 2).astype(np.float32)
    elif kind == "zeros":
        y[: max(1, n // 3)] = x[: max(1, n // 3)]
    return x, y


def _check(x, y, alternative):
    a, n = _fm(x)
    b, _ = _fm(y)
    stat, pval = dstats.wilcoxon_zsplit(a, b, n, alternative)
    s_ref, p_ref = so.wilcoxon_zsplit(x, y, alternative)
    stat, pval = stat.cpu().numpy(), pval.cpu().numpy()
    np.testing.assert_array_equal(stat, s_ref)
    np.testing.assert_allclose(pval, p_ref, rtol=P_RTOL, atol=0)


@pytest.mark.parametrize("alternative", ["greater", "less", "two-sided"])
@pytest.mark.parametrize("kind", ["plain", "ties", "zeros"])
def test_wilcoxon_small_and_lds_sizes(kind, alternative):
    """Every method branch (exact, 2^n enumeration, normal approximation) and both workgroup sizes of the sort kernel."""
    rng = np.random.default_rng(11)
    for n in (2, 5, 13, 14, 30, 50, 51, 64, 65, 200, 1000, 2048, 2049, 5000):
        x, y = _pair(rng, n, 7, kind)
        _check(x, y, alternative)


def test_wilcoxon_single_observation():
    x = np.array([[1.0, 2.0]], dtype=np.float32)
    y = np.array([[0.5, 3.0]], dtype=np.float32)
    _check(x, y, "greater")
    _check(x, y, "less")


def test_wilcoxon_at_lds_limit_and_beyond():
    """32768 samples still rank in LDS; 40000 and 70000 go through the workspace (sorted runs + merge passes)."""
    rng = np.random.default_rng(5)
    for n, kind in ((32768, "plain"), (40000, "ties"), (70000, "zeros")):
        x, y = _pair(rng, n, 3, kind)
        _check(x, y, "greater")


def test_wilcoxon_nan_identical_and_empty():
    rng = np.random.default_rng(2)
    x, y = _pair(rng, 300, 4, "plain")
    y[17, 1] = np.nan
    y[:, 2] = x[:, 2]                                              # all differences zero
    a, n = _fm(x)
    b, _ = _fm(y)
    stat, pval = dstats.wilcoxon_zsplit(a, b, n, "greater")
    s_ref, p_ref = so.wilcoxon_zsplit(x, y, "greater")
    np.testing.assert_array_equal(stat.cpu().numpy(), s_ref)       # NaN == NaN under assert_array_equal
    np.testing.assert_allclose(pval.cpu().numpy(), p_ref, rtol=P_RTOL)
    assert np.isnan(s_ref[1]) and p_ref[2] == pytest.approx(0.5)
    stat, pval = dstats.wilcoxon_zsplit(a, b, 0, "less")           # scipy: empty input -> NaN, NaN
    assert torch.isnan(stat).all() and torch.isnan(pval).all()
    with pytest.raises(ValueError):
        dstats.wilcoxon_zsplit(a, b, n, "upper")


def test_wilcoxon_go_scale_columns():
    """3000 terms x 4096 samples, the shape of one perturbation of the GO-BP screen."""
    rng = np.random.default_rng(9)
    x, y = _pair(rng, 4096, 3000, "plain")
    a, n = _fm(x)
    b, _ = _fm(y)
    stat, pval = dstats.wilcoxon_zsplit(a, b, n, "greater")
    cols = np.r_[0:40, 2990:3000]
    s_ref, p_ref = so.wilcoxon_zsplit(x[:, cols], y[:, cols], "greater")
    np.testing.assert_array_equal(stat.cpu().numpy()[cols], s_ref)
    np.testing.assert_allclose(pval.cpu().numpy()[cols], p_ref, rtol=P_RTOL)
    # size-independent property: swapping the arguments mirrors the statistic, r_plus + r_plus' = n (n + 1) / 2
    stat2, pval2 = dstats.wilcoxon_zsplit(b, a, n, "less")
    np.testing.assert_array_equal((stat + stat2).cpu().numpy(), np.full(3000, n * (n + 1) / 2))
    np.testing.assert_allclose(pval2.cpu().numpy(), pval.cpu().numpy(), rtol=1e-9)


def test_fdr_bh_sets_ties_nan():
    rng = np.random.default_rng(4)
    for n in (1, 2, 7, 1024, 1025, 3000, 20000):
        p = rng.uniform(size=(3, n)) ** 3
        if n >= 10:
            p[0, : n // 5] = p[0, n // 5: 2 * (n // 5)]
            p[1, -3:] = 1.0
        q = dstats.fdr_bh(torch.as_tensor(p, device=DEV)).cpu().numpy()
        for s in range(3):
            np.testing.assert_allclose(q[s], so.fdr_bh(p[s]), rtol=1e-13)
        q1 = dstats.fdr_bh(torch.as_tensor(p[2], device=DEV)).cpu().numpy()
        np.testing.assert_array_equal(q1, q[2])
    p = rng.uniform(size=(2, 50))
    p[1, 7] = np.nan
    q = dstats.fdr_bh(torch.as_tensor(p, device=DEV)).cpu().numpy()
    np.testing.assert_allclose(q[0], so.fdr_bh(p[0]), rtol=1e-13)
    assert np.isnan(q[1]).all()


def test_to_feature_major_blocks():
    rng = np.random.default_rng(1)
    src = torch.as_tensor(rng.standard_normal((77, 45)).astype(np.float32), device=DEV)
    dst = torch.full((45, 200), -1.0, device=DEV)
    dstats.to_feature_major(src, dst, col0=100)
    ref = torch.full((45, 200), -1.0, device=DEV)
    ref[:, 100:177] = src.t()
    assert torch.equal(dst, ref)
    dst2 = torch.zeros(40, 80, device=DEV)
    dstats.to_feature_major(src[:, :40], dst2)                     # strided source rows
    assert torch.equal(dst2[:, :77], src[:, :40].t()) and not dst2[:, 77:].any()
    with pytest.raises(ValueError):
        dstats.to_feature_major(src, dst, col0=150)


def test_wilcox_arrays_matches_reference_recipe():
    """ontobj.py:671-676 end to end on [N, C] arrays: stat, pval, qval."""
    rng = np.random.default_rng(8)
    ctrl, pert = _pair(rng, 500, 60, "plain")
    stat, p, q = dstats.wilcox_arrays(ctrl, pert, direction="up", device=DEV)
    s_ref, p_ref = so.wilcoxon_zsplit(pert, ctrl, "greater")
    np.testing.assert_array_equal(stat, s_ref)
    np.testing.assert_allclose(p, p_ref, rtol=P_RTOL)
    np.testing.assert_allclose(q, so.fdr_bh(p_ref), rtol=1e-10)
    stat_d, p_d, _ = dstats.wilcox_arrays(ctrl, pert, direction="down", device=DEV)
    np.testing.assert_allclose(p_d, so.wilcoxon_zsplit(pert, ctrl, "less")[1], rtol=P_RTOL)


# ------------------------------------------------------------------------------ perturbation screen
def _screen_model(N=300, seed=21):
    from test_gpu_parity import make_model, medium
    masks, sd = medium()
    rng = np.random.default_rng(seed)
    G = masks[-1].shape[0]
    X = np.abs(rng.standard_normal((N, G))).astype(np.float32)
    model, onto = make_model(masks, 3, sd, X=X)
    L = model._engine.plan.L
    eps = rng.standard_normal((64, N, L)).astype(np.float32)
    return model, onto, X, eps


def _replay(model, eps, N, step):
    """eps_source that hands every pass over the dataset the same two draws per batch, pass after pass."""
    state = {"call": 0}
    nb = (N + step - 1) // step

    def src(B):
        c = state["call"]
        state["call"] += 1
        pass_no, within = divmod(c, 2 * nb)
        b, which = divmod(within, 2)
        return torch.as_tensor(eps[2 * pass_no + which, b * step: b * step + B])
    model.eps_source = src
    return state


def test_pass_data_perturbed_equals_pass_over_overwritten_data():
    """The rank-1 form of the first layer against the reference's recipe (overwrite the columns, run the pass):
    activities and reconstruction to 1e-4 relative."""
    from conftest import act_close
    model, onto, X, eps = _screen_model()
    e = model._engine
    N, G = X.shape
    model.eval()
    Xd = torch.as_tensor(X, device=DEV)
    hpre = torch.empty(e.plan.L, N + 4, device=DEV)
    e.encode_pre(Xd, None, hpre, 0)
    for genes, vals in (([], []), ([5], [0.0]), ([0, 17, 999], [3.0, 0.0, -1.5]), (list(range(100, 164)), [0.25] * 64)):
        Xp = Xd.clone()
        if genes:
            Xp[:, genes] = torch.as_tensor(vals, device=DEV)
        ref_act = torch.empty(N, e.plan.T, device=DEV)
        ref_rec = torch.empty(N, G, device=DEV)
        e.pass_data(Xp, None, eps[0], eps[1], act_out=ref_act, rec_out=ref_rec)
        act = torch.empty(N, e.plan.T, device=DEV)
        rec = torch.empty(N, G, device=DEV)
        e.pass_data_perturbed(Xd, None, hpre, 0, genes, vals, eps[0], eps[1], act_out=act, rec_out=rec)
        assert act_close(act.cpu().numpy(), ref_act.cpu().numpy())
        assert act_close(rec.cpu().numpy(), ref_rec.cpu().numpy())
        # the same activities written feature-major into columns [4, 4 + N) of a wider matrix
        fm = torch.full((e.plan.T, N + 9), -7.0, device=DEV)
        e.pass_data_perturbed(Xd, None, hpre, 0, genes, vals, eps[0], eps[1], act_fm=fm, act_fm_col0=4)
        assert torch.equal(fm[:, 4:4 + N], act.t()) and (fm[:, :4] == -7.0).all() and (fm[:, 4 + N:] == -7.0).all()
    with pytest.raises(ValueError):
        e.pass_data_perturbed(Xd, None, hpre, 0, [G], [1.0], eps[0], eps[1], act_out=act)
    with pytest.raises(ValueError):
        e.pass_data_perturbed(Xd, None, hpre, 0, [3, 3], [1.0, 2.0], eps[0], eps[1], act_out=act)
    with pytest.raises(ValueError):
        e.pass_data_perturbed(Xd, None, hpre, 0, list(range(65)), [0.0] * 65, eps[0], eps[1], act_out=act)


@pytest.mark.parametrize("output", ["terms", "genes"])
def test_screen_equals_perturbation_then_wilcox_test(output):
    """perturbation_screen with the full pass per perturbation == the reference's loop: perturbation(...) per entry,
    then scipy's test and the FDR on the host arrays (oracle), with the same eps per pass.  Statistic exact."""
    model, onto, X, eps = _screen_model()
    N = X.shape[0]
    model.inference_batch = 128
    names = model.genes
    perts = [([names[3]], [0.0]), ([names[10], names[500], names[10]], [1.0, 0.0, 4.0]), ([names[i] for i in range(200, 270)], [0.0] * 70)]
    model.screen_rank1 = False
    _replay(model, eps, N, 128)
    res = model.perturbation_screen(onto, "d", perts, output=output, direction="up")
    _replay(model, eps, N, 128)
    ctrl = model.get_pathway_activities(onto, "d") if output == "terms" else model.get_reconstructed_values(onto, "d")
    for p, (genes, vals) in enumerate(perts):
        arr = model.perturbation(onto, "d", genes, vals, output=output)
        s_ref, p_ref = so.wilcoxon_zsplit(arr, ctrl, "greater")
        np.testing.assert_array_equal(res["stat"][p], s_ref)
        np.testing.assert_allclose(res["pval"][p], p_ref, rtol=P_RTOL)
        np.testing.assert_allclose(res["qval"][p], so.fdr_bh(p_ref), rtol=1e-10)


def test_screen_rank1_close_to_full_pass_and_subsets():
    """The default (cached first layer + rank-1 terms) changes activities by float32 rounding only, so the rank sums
    move by a few swaps of nearly tied differences at most; `terms` restricts the columns before the FDR."""
    model, onto, X, eps = _screen_model()
    N = X.shape[0]
    names = model.genes
    perts = [([names[3]], [0.0]), ([names[10], names[500]], [4.0, 0.0])]
    out = {}
    for rank1 in (False, True):
        model.screen_rank1 = rank1
        _replay(model, eps, N, model.inference_batch)
        out[rank1] = model.perturbation_screen(onto, "d", perts, output="terms", direction="down")
    total = N * (N + 1) / 2
    assert np.abs(out[True]["stat"] - out[False]["stat"]).max() <= 0.002 * total
    np.testing.assert_allclose(out[True]["pval"], out[False]["pval"], atol=0.02)
    ids = list(onto.annot["1000_30"].ID[[1, 5, 40]])
    _replay(model, eps, N, model.inference_batch)
    sub_res = model.perturbation_screen(onto, "d", perts, output="terms", direction="down", terms=ids)
    assert sub_res["pval"].shape == (2, 3)
    np.testing.assert_array_equal(sub_res["pval"], out[True]["pval"][:, [1, 5, 40]])
    for p in range(2):
        np.testing.assert_allclose(sub_res["qval"][p], so.fdr_bh(sub_res["pval"][p]), rtol=1e-10)
    with pytest.raises(ValueError):
        model.perturbation_screen(onto, "d", [(["no such gene"], [0.0])])


def test_wilcox_test_dataframe_like_the_reference():
    """Ontobj.wilcox_test (ontobj.py:678-701): columns, row order (sorted by pval) and values of the result frame."""
    import pandas as pd

    class Onto:          # the two accessors wilcox_test calls on the reference's Ontobj
        def __init__(self, n):
            self._annot = pd.DataFrame({"ID": [f"T:{i}" for i in range(n)], "Name": [f"term {i}" for i in range(n)], "depth": [i % 5 for i in range(n)]})

        def extract_annot(self, top_thresh=1000, bottom_thresh=30):
            return self._annot

        def extract_genes(self, top_thresh=1000, bottom_thresh=30):
            return [f"G{i}" for i in range(len(self._annot))]

    rng = np.random.default_rng(12)
    ctrl, pert = _pair(rng, 120, 25, "plain")
    s_ref, p_ref = so.wilcoxon_zsplit(pert, ctrl, "greater")
    q_ref = so.fdr_bh(p_ref)
    order = np.argsort(p_ref, kind="stable")
    res = dstats.wilcox_test(Onto(25), ctrl, pert, direction="up", option="terms", device=DEV)
    assert list(res.columns) == ["id", "term", "depth", "stat", "pval", "qval"]
    assert list(res["id"]) == [f"T:{i}" for i in order]
    np.testing.assert_array_equal(res["stat"].to_numpy(), s_ref[order])
    np.testing.assert_allclose(res["pval"].to_numpy(), p_ref[order], rtol=P_RTOL)
    np.testing.assert_allclose(res["qval"].to_numpy(), q_ref[order], rtol=1e-10)
    res_g = dstats.wilcox_test(Onto(25), ctrl, pert, direction="down", option="genes", device=DEV)
    assert list(res_g.columns) == ["gene", "stat", "pval", "qval"] and len(res_g) == 25
    assert (np.diff(res_g["pval"].to_numpy()) >= 0).all()


def test_perturbed_pass_feature_major_at_inference_batch():
    """From 2048 columns on the term levels run one launch per level and, for a feature-major output, write the nu-means
    themselves (one wave per term): against the sample-major output of the same pass, which takes the row-per-wave
    kernel and the separate output kernel (same summation order; last-bit differences from fused multiply-adds)."""
    model, onto, X, eps = _screen_model(N=2100, seed=5)
    e = model._engine
    N = X.shape[0]
    model.eval()
    Xd = torch.as_tensor(X, device=DEV)
    hpre = torch.empty(e.plan.L, N, device=DEV)
    e.encode_pre(Xd, None, hpre, 0)
    genes, vals = [7, 300], [0.0, 2.5]
    act = torch.empty(N, e.plan.T, device=DEV)
    e.pass_data_perturbed(Xd, None, hpre, 0, genes, vals, eps[0], eps[1], act_out=act)
    fm = torch.full((e.plan.T, N + 5), -7.0, device=DEV)
    e.pass_data_perturbed(Xd, None, hpre, 0, genes, vals, eps[0], eps[1], act_fm=fm, act_fm_col0=2)
    got, want = fm[:, 2:2 + N].t().cpu().numpy(), act.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * np.abs(want).max())
    assert (fm[:, :2] == -7.0).all() and (fm[:, 2 + N:] == -7.0).all()
    # and the screen built on it agrees with the full-pass recipe within what float32 rounding does to rank sums
    names = model.genes
    perts = [([names[7], names[300]], vals)]
    out = {}
    for rank1 in (False, True):
        model.screen_rank1 = rank1
        _replay(model, eps, N, model.inference_batch)
        out[rank1] = model.perturbation_screen(onto, "d", perts, output="terms", direction="up")
    total = N * (N + 1) / 2
    assert np.abs(out[True]["stat"] - out[False]["stat"]).max() <= 0.002 * total


def test_screen_edge_cases():
    """No perturbations, a dataset of one and of five samples (exact and enumeration branches of the test), a gene list of
    exactly 64 and of 65 entries (rank-1 limit), repeated genes (the last value stays, as sequential overwriting gives)."""
    model, onto, X, eps = _screen_model(N=5, seed=9)
    T = model._engine.plan.T
    names = model.genes
    res = model.perturbation_screen(onto, "d", [])
    assert res["stat"].shape == (0, T) and res["qval"].shape == (0, T)
    many = [names[i] for i in range(65)]
    _replay(model, eps, 5, model.inference_batch)
    res = model.perturbation_screen(onto, "d", [(many[:64], [0.0] * 64), (many, [0.0] * 65), ([names[3], names[3]], [9.0, 0.0]), ([names[3]], [0.0])])
    assert res["pval"].shape == (4, T) and np.isfinite(res["pval"]).all() and (res["qval"] <= 1).all()
    assert ((res["stat"] >= 0) & (res["stat"] <= 15)).all()                      # n (n + 1) / 2 = 15
    # a repeated gene keeps its last value: the same perturbation as listing it once, given the same eps draws
    model.eps_source = lambda B: torch.as_tensor(eps[0, :B])
    rep = model.perturbation_screen(onto, "d", [([names[3], names[3]], [9.0, 0.0]), ([names[3]], [0.0])])
    np.testing.assert_array_equal(rep["stat"][0], rep["stat"][1])
    onto.add_dataset("one", X[:1])
    res1 = model.perturbation_screen(onto, "one", [([names[0]], [1.0])])
    assert res1["pval"].shape == (1, T) and set(np.unique(res1["pval"])) <= {0.5, 1.0}
