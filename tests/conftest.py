import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["g1_tiny", "g1_odd", "g1_nu1"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_case(name):
    fx = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    K = int(fx["K"])
    masks = [fx[f"mask/{i}"].astype(np.int64) for i in range(K)]
    return fx, masks


def sub(fx, prefix):
    return {k[len(prefix):]: v for k, v in fx.items() if k.startswith(prefix)}


@pytest.fixture(params=CASES)
def case(request):
    fx, masks = load_case(request.param)
    return request.param, fx, masks


# keys whose trajectory is rounding noise amplified by Adam (analytically-zero gradient of the
# Linear bias in front of a train-mode BatchNorm), see DESIGN.md "reference quirks"
NOISE_KEYS = ("encoder.encoder.0.0.bias", "encoder.encoder.0.1.running_mean")


def scaled_err(a, b, tau_frac=1e-3):
    """max |a - b| / max(|b|, tau), tau = tau_frac * max|b|: the activity metric of SURVEY.md 8(d)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    tau = tau_frac * np.abs(b).max()
    return float((np.abs(a - b) / np.maximum(np.abs(b), tau)).max())


def act_close(a, b, rel=1e-4, floor=1e-2, l2=2e-5):
    """Activity / reconstruction parity between two float32 evaluations: |a-b| <= rel * max(|b|, tau),
    tau = floor * max|b|, and relative L2 error <= l2.

    Where the floor comes from (measured, not assumed): tests/golden/*.npz hold the reference's outputs in float32
    AND the same reference modules evaluated in float64 (tools/gen_golden.py, keys fwd64/ inf64/).  Under SURVEY.md
    8(d)'s rule (tau = 1e-3 * max|b|, rel 1e-4) the reference's OWN float32 outputs miss their float64 values by up to
    1.36e-4 (g1_tiny log_var; 9.0e-5 on recon): elements near zero are cancelling sums of O(max|b|) terms, so their
    absolute error is ~3e-7 * max|b| whatever their size.  That rule therefore cannot separate two correct fp32
    evaluations.  At tau = 1e-2 * max|b| the reference's worst scaled error is 1.7e-5, a 6x margin under rel = 1e-4 for the
    sum of two such errors; that is the floor used here (round 1 used 1e-1).  test_gpu_parity.py::test_error_vs_float64_*
    additionally hold the HIP path to 2x the reference's own error against float64 under the SURVEY rule itself.
    """
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    tau = floor * np.abs(b).max()
    ok = bool(np.all(np.abs(a - b) <= rel * np.maximum(np.abs(b), tau)))
    return ok and float(np.linalg.norm(a - b)) <= l2 * float(np.linalg.norm(b)) + 1e-30


def close_vs_float64(got, ref32, ref64, factor=2.0, tau_frac=1e-3, rel=1e-4):
    """Evidence-based parity: `got` (HIP, fp32) may miss the float64 value by at most `factor` times what the reference's
    own float32 evaluation `ref32` misses it by, in SURVEY.md 8(d)'s scaled metric (tau = 1e-3 * max); and wherever the
    reference itself meets SURVEY's 1e-4 under that metric, so must `got`.  Returns (ok, got_err, ref_err)."""
    e_got, e_ref = scaled_err(got, ref64, tau_frac), scaled_err(ref32, ref64, tau_frac)
    ok = e_got <= factor * e_ref + 1e-7
    if e_ref <= rel:
        ok = ok and e_got <= rel
    return ok, e_got, e_ref


@pytest.fixture(autouse=True)
def _poisoned_device_pool(request):
    """Before every GPU test: fill torch's caching-allocator pool with NaN and hand it back, so that the workspaces of the
    test (torch.empty: whatever the pool holds) start as NaN instead of the zeros of fresh device memory.  A kernel that reads
    a workspace cell nothing wrote -- a padded column, a fallback row of an empty list gathered with weight 0 -- then
    shows up as NaN in the result instead of passing by luck (found that way: the backward gene kernels' empty lists used
    to gather panel row 0 of D, which the same launch writes)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        nan = float("nan")
        big = [torch.full((1 << 26,), nan, device="cuda") for _ in range(8)]                      # 8 x 256 MB
        small = [torch.full((n,), nan, device="cuda") for n in (1 << 8, 1 << 11, 1 << 14, 1 << 17, 1 << 20, 1 << 23) for _ in range(6)]
        torch.cuda.synchronize()
        del big, small
    yield
