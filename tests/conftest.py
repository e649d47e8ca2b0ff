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


def act_close(a, b, rel=1e-4, floor=0.1, l2=2e-5):
    """Activity / reconstruction parity: |a-b| <= rel * max(|b|, tau), tau = floor * max|b|,
    and relative L2 error <= l2.

    SURVEY.md 8(d) proposed tau = 1e-3 * max|b|; that is below the reference's own fp32 noise
    (outputs near zero are cancelling sums of O(max|b|) terms, so two correct fp32 evaluations
    in different summation orders already differ by ~3e-7 * max|b|: the oracle-vs-reference
    fixtures show 1.5e-4 under the tau = 1e-3 rule from reordering alone).  tau = 0.1 * max|b|
    holds every element >= 10 % of the scale to 1e-4 relative and the rest to 1e-5 of scale.
    """
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    tau = floor * np.abs(b).max()
    ok = bool(np.all(np.abs(a - b) <= rel * np.maximum(np.abs(b), tau)))
    return ok and float(np.linalg.norm(a - b)) <= l2 * float(np.linalg.norm(b)) + 1e-30
