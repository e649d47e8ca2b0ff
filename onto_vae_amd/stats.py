"""Statistics after a perturbation, on the device: the mirror of Ontobj.wilcox_test (onto_vae/ontobj.py:651-701).

The reference loops over the terms with scipy.stats.wilcoxon(..., zero_method='zsplit') and corrects the p-values with
statsmodels' fdrcorrection; here both run in libontovae_hip.so (ontovae_wilcoxon_zsplit, ontovae_fdr_bh) on
feature-major device matrices, so that a screen of many perturbations never brings activities back to the host.
torch only allocates the buffers.  There is no CPU path: without the library or a GPU these functions raise.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

ALTERNATIVES = {"greater": 0, "less": 1, "two-sided": 2}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _need_gpu(t, what):
    if t.device.type != "cuda":
        raise _lib.OntoVAEHipError(f"{what} runs on an MI355X through libontovae_hip.so; there is no CPU path (tensor is on {t.device})")


def _workspace(nbytes, device):
    if nbytes <= 0:
        return None, C.c_void_p(0), 0
    buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    base = buf.data_ptr()
    aligned = (base + 255) // 256 * 256
    return buf, C.c_void_p(aligned), nbytes


def to_feature_major(src, dst, col0=0):
    """dst[:, col0:col0 + B] = src.T for a sample-major float32 [B, F] device block (rows may be strided)."""
    _need_gpu(src, "to_feature_major")
    if src.dim() != 2 or dst.dim() != 2 or src.dtype != torch.float32 or dst.dtype != torch.float32 or src.stride(1) != 1 or dst.stride(1) != 1:
        raise ValueError("to_feature_major: float32 matrices with unit inner stride expected")
    B, F = src.shape
    if dst.shape[0] != F or col0 < 0 or col0 + B > dst.shape[1] or dst.device != src.device:
        raise ValueError(f"to_feature_major: destination {tuple(dst.shape)} does not hold columns [{col0}, {col0 + B}) of {F} features")
    _lib.check(_lib.lib().ontovae_to_feature_major(_ptr(src), src.stride(0), B, F, C.c_void_p(dst.data_ptr() + 4 * col0),
                                                   dst.stride(0), _stream(src.device)), "to_feature_major")
    return dst


def wilcoxon_zsplit(perturbed_fm, control_fm, n=None, alternative="greater", out=None):
    """Paired signed-rank test of every row of two feature-major float32 [C, ld] device matrices over their first n
    columns (samples).  Returns (stat, pval), float64 [C] on the device (written into `out` = (stat, pval) if given)."""
    _need_gpu(perturbed_fm, "wilcoxon_zsplit")
    a, b = perturbed_fm, control_fm
    if a.dim() != 2 or a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32 or a.device != b.device:
        raise ValueError("wilcoxon_zsplit: two float32 device matrices of the same [columns, samples] shape expected")
    if a.stride(1) != 1 or b.stride(1) != 1 or a.stride(0) != b.stride(0):
        raise ValueError("wilcoxon_zsplit: both matrices must share one row stride and have unit inner stride")
    if alternative not in ALTERNATIVES:
        raise ValueError(f"`alternative` must be one of {set(ALTERNATIVES)}.")
    ncols = a.shape[0]
    n = a.shape[1] if n is None else int(n)
    if n < 0 or n > a.shape[1]:
        raise ValueError(f"wilcoxon_zsplit: n = {n} outside [0, {a.shape[1]}]")
    if out is None:
        stat = torch.empty(ncols, dtype=torch.float64, device=a.device)
        pval = torch.empty(ncols, dtype=torch.float64, device=a.device)
    else:
        stat, pval = out
        for t in (stat, pval):
            if t.dtype != torch.float64 or t.shape != (ncols,) or not t.is_contiguous() or t.device != a.device:
                raise ValueError(f"wilcoxon_zsplit: out tensors must be contiguous float64 [{ncols}] on {a.device}")
    L = _lib.lib()
    keep, ws, nbytes = _workspace(L.ontovae_wilcoxon_workspace_bytes(n, ncols), a.device)
    _lib.check(L.ontovae_wilcoxon_zsplit(_ptr(a), _ptr(b), a.stride(0) if ncols > 1 else max(a.stride(0), n), ncols, n,
                                         ALTERNATIVES[alternative], _ptr(stat), _ptr(pval), ws, nbytes, _stream(a.device)),
               "wilcoxon_zsplit")
    if keep is not None:
        keep.record_stream(torch.cuda.current_stream(a.device))
    return stat, pval


def fdr_bh(pvals):
    """Benjamini-Hochberg q-values (fdrcorrection(...)[1]) of a float64 device tensor [n] or [sets, n], per set."""
    _need_gpu(pvals, "fdr_bh")
    if pvals.dtype != torch.float64 or pvals.dim() not in (1, 2) or not pvals.is_contiguous():
        raise ValueError("fdr_bh: contiguous float64 [n] or [sets, n] expected")
    n = pvals.shape[-1]
    nsets = 1 if pvals.dim() == 1 else pvals.shape[0]
    q = torch.empty_like(pvals)
    L = _lib.lib()
    keep, ws, nbytes = _workspace(L.ontovae_fdr_workspace_bytes(n, nsets), pvals.device)
    _lib.check(L.ontovae_fdr_bh(_ptr(pvals), n, nsets, _ptr(q), ws, nbytes, _stream(pvals.device)), "fdr_bh")
    if keep is not None:
        keep.record_stream(torch.cuda.current_stream(pvals.device))
    return q


def _feature_major_copy(arr, device):
    """[N, C] host or device array -> feature-major [C, ld] float32 on `device` (ld = N rounded up to 4)."""
    src = torch.as_tensor(arr)
    if src.dim() != 2:
        raise ValueError("expected a 2-D [samples, columns] array")
    src = src.to(device=device, dtype=torch.float32).contiguous()
    N, Cn = src.shape
    dst = torch.zeros(Cn, max((N + 3) // 4 * 4, 4), dtype=torch.float32, device=device)
    if N and Cn:
        to_feature_major(src, dst)
    return dst, N


def wilcox_arrays(control, perturbed, direction="up", device="cuda"):
    """The numeric core of Ontobj.wilcox_test (ontobj.py:671-676) for [N, C] arrays: numpy (stat, pvals, qvals)."""
    if np.shape(control) != np.shape(perturbed):
        raise ValueError("control and perturbed must have the same shape")
    alternative = "greater" if direction == "up" else "less"
    device = torch.device(device)
    ctrl, N = _feature_major_copy(control, device)
    pert, _ = _feature_major_copy(perturbed, device)
    stat, pval = wilcoxon_zsplit(pert, ctrl, N, alternative)
    qval = fdr_bh(pval)
    return stat.cpu().numpy(), pval.cpu().numpy(), qval.cpu().numpy()


def wilcox_test(ontobj, control, perturbed, direction="up", option="terms", top_thresh=1000, bottom_thresh=30, device="cuda"):
    """Ontobj.wilcox_test (ontobj.py:651-701) with the tests on the device: the same DataFrame (columns id / term /
    depth / stat / pval / qval for option='terms', gene / stat / pval / qval for 'genes'), sorted by pval."""
    import pandas as pd
    stat, pvals, qvals = wilcox_arrays(control, perturbed, direction, device)
    if option == "terms":
        annot = ontobj.extract_annot(top_thresh=top_thresh, bottom_thresh=bottom_thresh)
        res = pd.DataFrame({"id": annot.ID.tolist(), "term": annot.Name.tolist(), "depth": annot.depth.tolist(),
                            "stat": stat, "pval": pvals, "qval": qvals})
    else:
        genes = ontobj.extract_genes(top_thresh=top_thresh, bottom_thresh=bottom_thresh)
        res = pd.DataFrame({"gene": genes, "stat": stat, "pval": pvals, "qval": qvals})
    return res.sort_values("pval").reset_index(drop=True)
