"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the statistics the reference runs after a perturbation
(onto_vae/ontobj.py:651-701, Ontobj.wilcox_test).  Nothing under onto_vae_amd/ may import this file.

The reference calls two third-party routines that are not part of its tree and are not pinned by it:
  * scipy.stats.wilcoxon(perturbed[:, i], control[:, i], zero_method='zsplit', alternative=...)  (ontobj.py:673)
  * statsmodels.stats.multitest.fdrcorrection(pvals)                                             (ontobj.py:676)
scipy 1.15.3 is importable in this image: tests/test_stats_oracle.py pins wilcoxon_zsplit below against it, column by
column, on every branch of its method='auto' logic.  statsmodels is absent: fdr_bh restates the published
Benjamini-Hochberg step-up rule as fdrcorrection(method='indep') applies it and is pinned by
scipy.stats.false_discovery_control(method='bh') and a hand-worked vector ("parity unpinned" against statsmodels
itself).
"""
import numpy as np
from scipy import special


def _rank_average(absd):
    """Average ranks (1-based) of a 1-D array and the tie-group sizes (scipy _rankdata(..., 'average'))."""
    order = np.argsort(absd, kind="stable")
    s = absd[order]
    starts = np.flatnonzero(np.r_[True, s[1:] != s[:-1]])
    sizes = np.diff(np.r_[starts, len(s)])
    avg = starts + (sizes + 1) / 2.0
    r = np.empty(len(s), dtype=np.float64)
    r[order] = np.repeat(avg, sizes)
    return r, sizes


def _exact_dist(n):
    """P(r_plus = k), k = 0 .. n (n + 1) / 2 (scipy _get_wilcoxon_distr)."""
    c = np.ones(1)
    for k in range(1, n + 1):
        prev = c
        c = np.zeros(k * (k + 1) // 2 + 1)
        c[:len(prev)] = prev * 0.5
        c[-len(prev):] += prev * 0.5
    return c


def wilcoxon_zsplit_1d(x, y, alternative="greater"):
    """(statistic, pvalue) of scipy.stats.wilcoxon(x, y, zero_method='zsplit', alternative=alternative) with the
    defaults correction=False, method='auto' (scipy/stats/_wilcoxon.py:116-259)."""
    d = np.asarray(x) - np.asarray(y)            # float32 inputs: float32 difference, as numpy gives the reference
    n = d.shape[0]
    if n == 0 or np.isnan(d).any():
        return np.nan, np.nan
    r, sizes = _rank_average(np.abs(d))
    zeros = d == 0
    r_plus = r[d > 0].sum() + r[zeros].sum() / 2
    r_minus = n * (n + 1) / 2 - r_plus
    has_ties = (sizes > 1).any()
    if n <= 50 and not has_ties and not zeros.any():
        pmf = _exact_dist(n)
        k = int(r_plus)
        mn = n * (n + 1) / 4
        sf = pmf[k:].sum() if k <= mn else 1 - pmf[:k].sum()
        cdf = pmf[:k + 1].sum() if k <= mn else 1 - pmf[k + 1:].sum()
        p = {"greater": sf, "less": cdf, "two-sided": min(max(2 * min(sf, cdf), 0.0), 1.0)}[alternative]
    elif n <= 13:
        nz = ~zeros
        rz = r[zeros].sum() / 2
        m = np.arange(1 << n)[:, None] >> np.arange(n)[None, :] & 1
        null = (m * (r * nz)[None, :]).sum(1) + rz
        pg, pl = (null >= r_plus).mean(), (null <= r_plus).mean()
        p = {"greater": pg, "less": pl, "two-sided": min(2 * min(pg, pl), 1.0)}[alternative]
    else:
        mn = n * (n + 1.) * 0.25
        se = n * (n + 1.) * (2. * n + 1.) - (sizes.astype(np.float64) ** 3 - sizes).sum() / 2
        z = (r_plus - mn) / np.sqrt(se / 24)
        p = {"greater": special.ndtr(-z), "less": special.ndtr(z), "two-sided": 2 * special.ndtr(-abs(z))}[alternative]
    return (min(r_plus, r_minus) if alternative == "two-sided" else r_plus), float(p)


def wilcoxon_zsplit(perturbed, control, alternative="greater"):
    """Column-wise test of two [N, C] arrays, as the list comprehension at ontobj.py:673: (stat [C], pval [C])."""
    res = [wilcoxon_zsplit_1d(perturbed[:, i], control[:, i], alternative) for i in range(control.shape[1])]
    return np.array([r[0] for r in res], dtype=np.float64), np.array([r[1] for r in res], dtype=np.float64)


def fdr_bh(pvals):
    """q-values of statsmodels fdrcorrection(pvals, method='indep')[1]: sort, p_(i) n / i, running minimum from the
    largest down, cap at 1, back to input order."""
    p = np.asarray(pvals, dtype=np.float64)
    n = p.shape[0]
    order = np.argsort(p)
    raw = p[order] / (np.arange(1, n + 1) / float(n))
    q = np.minimum.accumulate(raw[::-1])[::-1]
    q[q > 1] = 1
    out = np.empty(n)
    out[order] = q
    return out
