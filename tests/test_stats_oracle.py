"""Pins oracle/stats_oracle.py (the checker of the device statistics) against scipy itself, which is what the reference
calls at onto_vae/ontobj.py:673, on every branch of scipy's method='auto' logic, and the Benjamini-Hochberg restatement
against scipy.stats.false_discovery_control and a hand-worked vector (statsmodels, ontobj.py:676, is not installed)."""
import os
import sys
import warnings

import numpy as np
import pytest
from scipy import stats

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import stats_oracle as so  # noqa: E402


def _cases():
    rng = np.random.default_rng(7)
    out = []
    for n in (1, 2, 5, 13, 14, 30, 50, 51, 200, 1500):
        x = rng.standard_normal((n, 4)).astype(np.float32)
        y = (x + 0.3 * rng.standard_normal((n, 4))).astype(np.float32)
        out.append((f"plain{n}", x, y))
        if n == 1:          # scipy's permutation branch refuses a single observation (ValueError); nothing to pin there
            continue
        xq = np.round(x * 2) / 2                                   # heavy ties and zeros
        yq = np.round(y * 2) / 2
        out.append((f"ties{n}", xq.astype(np.float32), yq.astype(np.float32)))
        yz = y.copy()
        yz[: max(1, n // 3)] = x[: max(1, n // 3)]                 # zeros, no other ties
        out.append((f"zeros{n}", x, yz))
    return out


@pytest.mark.parametrize("alternative", ["greater", "less", "two-sided"])
def test_wilcoxon_oracle_equals_scipy(alternative):
    for name, x, y in _cases():
        s, p = so.wilcoxon_zsplit(x, y, alternative)
        for i in range(x.shape[1]):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = stats.wilcoxon(x[:, i], y[:, i], zero_method="zsplit", alternative=alternative)
            assert s[i] == ref.statistic, (name, i)
            np.testing.assert_allclose(p[i], ref.pvalue, rtol=1e-12, atol=0, err_msg=f"{name} col {i}")


def test_wilcoxon_oracle_nan_and_identical():
    x = np.arange(20, dtype=np.float32)[:, None].repeat(2, 1)
    y = x.copy()
    y[3, 0] = np.nan
    s, p = so.wilcoxon_zsplit(x, y, "greater")
    assert np.isnan(s[0]) and np.isnan(p[0])
    ref = stats.wilcoxon(x[:, 1], y[:, 1], zero_method="zsplit", alternative="greater")       # all differences zero
    assert s[1] == ref.statistic and p[1] == pytest.approx(ref.pvalue, rel=1e-12)


def test_fdr_oracle():
    p = np.array([0.01, 0.04, 0.03, 0.005])
    np.testing.assert_allclose(so.fdr_bh(p), [0.02, 0.04, 0.04, 0.02], rtol=1e-15)
    rng = np.random.default_rng(3)
    for n in (1, 2, 17, 3000):
        p = rng.uniform(size=n) ** 3
        p[: n // 5] = p[n // 5: 2 * (n // 5)][: n // 5]            # some exact ties
        np.testing.assert_allclose(so.fdr_bh(p), stats.false_discovery_control(p, method="bh"), rtol=1e-13)
    assert (so.fdr_bh(np.array([0.9, 0.95, 1.0])) <= 1).all()
