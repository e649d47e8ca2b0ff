"""ontovae_fdr_bh_static, the entry point that allocates nothing (its scratch is a file-scope device array of the
library): called through the C ABI and compared with ontovae_fdr_bh (caller workspace) bit for bit, on two calls in a
row, the second with a smaller input, so that stale contents of the shared array would show."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(p):
    from onto_vae_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    pd = torch.as_tensor(p, device=dev).contiguous()
    nsets, n = (1, pd.shape[0]) if pd.dim() == 1 else pd.shape
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    # existing entry point: caller-provided workspace
    nbytes = L.ontovae_fdr_workspace_bytes(n, nsets)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    base = (ws.data_ptr() + 255) // 256 * 256
    q_ws = torch.full_like(pd, -7.0)
    _lib.check(L.ontovae_fdr_bh(C.c_void_p(pd.data_ptr()), n, nsets, C.c_void_p(q_ws.data_ptr()), C.c_void_p(base), nbytes, st), "fdr_bh")
    # the allocation-free one: outputs preallocated, nothing else
    q_st = torch.full_like(pd, -9.0)
    # a first (cold) call resolves the scratch array's address once per device (hipGetSymbolAddress: call it once outside
    # any graph capture); the check is on the warm call.  torch's counters only see torch's allocator, so the device's own
    # free-memory figure is compared as well: a hipMalloc inside the library would show there.
    _lib.check(L.ontovae_fdr_bh_static(C.c_void_p(pd.data_ptr()), n, nsets, C.c_void_p(q_st.data_ptr()), st), "fdr_bh_static")
    torch.cuda.synchronize(dev)
    q_st.fill_(-9.0)
    before, free_before = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
    _lib.check(L.ontovae_fdr_bh_static(C.c_void_p(pd.data_ptr()), n, nsets, C.c_void_p(q_st.data_ptr()), st), "fdr_bh_static")
    torch.cuda.synchronize(dev)
    assert torch.cuda.memory_allocated(dev) == before and torch.cuda.mem_get_info(dev)[0] == free_before
    return q_ws.cpu().numpy(), q_st.cpu().numpy()


def test_fdr_bh_static_equals_workspace_version_twice():
    import stats_oracle as so
    rng = np.random.RandomState(5)
    big = rng.uniform(size=(7, 3000)) ** 3
    big[2, 17] = np.nan              # a NaN set: its flag must not leak into the next call
    big[4, :40] = big[4, 40:80]      # ties
    a, b = _both(big)
    assert np.array_equal(a, b, equal_nan=True)
    for s in (0, 4, 6):
        np.testing.assert_allclose(b[s], so.fdr_bh(big[s]), rtol=1e-13)
    assert np.isnan(b[2]).all()
    small = rng.uniform(size=(5, 211))
    a, b = _both(small)
    assert np.array_equal(a, b) and np.isfinite(b).all()
    for s in range(5):
        np.testing.assert_allclose(b[s], so.fdr_bh(small[s]), rtol=1e-13)
    one = rng.uniform(size=37)
    a, b = _both(one)
    assert np.array_equal(a, b)
    np.testing.assert_allclose(b, so.fdr_bh(one), rtol=1e-13)


def test_fdr_bh_static_rejects_oversize():
    from onto_vae_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    p = torch.zeros(8, dtype=torch.float64, device=dev)
    q = torch.zeros(8, dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    # sizes beyond the static array are refused before anything is touched (the pointers are never dereferenced)
    assert L.ontovae_fdr_bh_static(C.c_void_p(p.data_ptr()), (1 << 18) + 1, 1, C.c_void_p(q.data_ptr()), st) == -1
    assert b"static scratch" in L.ontovae_last_error()
    assert L.ontovae_fdr_bh_static(C.c_void_p(p.data_ptr()), 1, 4097, C.c_void_p(q.data_ptr()), st) == -1
    assert L.ontovae_fdr_bh_static(C.c_void_p(p.data_ptr()), 0, 5, C.c_void_p(q.data_ptr()), st) == 0
    # q must not alias p: other blocks still read the p-values while q is written
    assert L.ontovae_fdr_bh_static(C.c_void_p(p.data_ptr()), 8, 1, C.c_void_p(p.data_ptr()), st) == -1
    assert b"alias" in L.ontovae_last_error()
