"""Host restatement of the random streams of libontovae_hip.so.  TEST INFRASTRUCTURE ONLY.

The reference draws dropout masks and eps from torch's generator (nn.Dropout, torch.randn_like:
onto_vae/modules.py:48, 58, 63; onto_vae/vae_model.py:92); the HIP path draws them from Philox4x32-10
(Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 v1.14 philox.h) keyed by
(seed, rng_offset, stream, element), see onto_vae_amd/csrc/ontovae_device.h.  This module restates that keying in
numpy so that a test can hand the SAME keep-scales to the CPU oracle (oracle/ontovae_oracle.py) and compare
gradients with dropout on.  Pinned by Random123's known-answer vectors (tests/test_oracle_dropout.py).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

STREAM_DROP, STREAM_ZDROP_MU, STREAM_ZDROP_LV, STREAM_EPS, STREAM_EPS2 = 1, 2, 3, 4, 5


def philox4x32_10(counter, key):
    """counter: 4 uint32 arrays (broadcastable), key: 2 ints.  Returns 4 uint32 arrays."""
    c = [np.asarray(v, dtype=np.uint64) & MASK for v in counter]
    c = list(np.broadcast_arrays(*c))
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def _draw(seed, offset, stream, L, B):
    """Philox outputs for element (feature l, batch column b): counter = (b, l ^ stream << 24, offset lo, offset hi),
    key = (seed lo, seed hi)  (ov_philox with idx = l * 2^32 + b).  Arrays are [B, L] (sample-major)."""
    b = np.arange(B, dtype=np.uint64)[:, None]
    l = np.arange(L, dtype=np.uint64)[None, :]
    c1 = l ^ np.uint64((int(stream) << 24) & 0xFFFFFFFF)
    off = int(offset)
    return philox4x32_10((b, c1, np.uint64(off & 0xFFFFFFFF), np.uint64((off >> 32) & 0xFFFFFFFF)),
                         (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF))


def u01(u):
    """ov_u01: ((u >> 8) + 0.5) / 2^24 in float32, in (0, 1)."""
    return ((u >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def dropout_scale(p, seed, offset, stream, L, B):
    """Keep-scales [B, L] of inverted dropout as ov_dropout_scale computes them: 1/(1-p) where u >= p, else 0."""
    p = np.float32(p)
    if p <= 0:
        return np.ones((B, L), np.float32)
    if p >= 1:
        return np.zeros((B, L), np.float32)
    r = _draw(seed, offset, stream, L, B)
    keep = u01(r[0]) >= p
    return np.where(keep, np.float32(1.0) / (np.float32(1.0) - p), np.float32(0.0)).astype(np.float32)


def scales(drop, z_drop, seed, offset, L, B):
    """(hidden, mu, log_var) keep-scales of one training step, the tuple OracleOntoVAE.gradients takes."""
    return (dropout_scale(drop, seed, offset, STREAM_DROP, L, B), dropout_scale(z_drop, seed, offset, STREAM_ZDROP_MU, L, B),
            dropout_scale(z_drop, seed, offset, STREAM_ZDROP_LV, L, B))


def randn(seed, offset, stream, L, B):
    """eps [B, L] as ov_randn draws it (Box-Muller on the first two outputs); float32 libm differences allowed for."""
    r = _draw(seed, offset, stream, L, B)
    u1, u2 = u01(r[0]), u01(r[1])
    return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.28318530717958647692) * u2)).astype(np.float32)
