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
