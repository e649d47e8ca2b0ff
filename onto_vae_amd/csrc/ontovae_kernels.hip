// HIP kernels + C ABI entry points of the OntoVAE hot path (gfx950 / MI355X only).
//
// Layout rule: inside the library every activation is FEATURE-MAJOR, [features, ldb] fp32 with
// the batch contiguous (ldb = B rounded up to 64).  The ontology decoder is then a gather-SpMM
// whose every access is a coalesced row read: Y[n, :] = sum_k W[n, c_k] * ACT[c_k, :].
// The dense encoder runs on fp32 MFMA (ontovae_gemm.h).
#include "ontovae_internal.h"
#include "ontovae_device.h"
#include "ontovae_gemm.h"
#include "ontovae_bf16.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>

#define OV_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            ontovae_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ONTOVAE_EHIP;                                                                 \
        }                                                                                        \
    } while (0)

#define OV_LAUNCH_CHECK() OV_HIP_CHECK(hipGetLastError())

#define OV_KNOWN_FLAGS (ONTOVAE_FLAG_LEVEL_LAUNCHES | ONTOVAE_FLAG_ROW_GENE_LAYER | ONTOVAE_FLAG_SIDE_STREAM | ONTOVAE_FLAG_WIDE_GENE_TILES | ONTOVAE_FLAG_NO_RIDERS | ONTOVAE_FLAG_BF16_GEMM | ONTOVAE_FLAG_CB_GENE_FWD | ONTOVAE_FLAG_PER_BLOCK_WGRAD)

// batch columns per wave pass (64 lanes x VEC) and the padded row length, a multiple of it
// -------------------------------------------------------------------------------------------------
// Optional per-phase HIP-event timing (bench.py's roofline leg).  Off by default; when a phase is
// enabled, its launches are bracketed by two events on the launch stream.
// -------------------------------------------------------------------------------------------------
struct OvProfState {
    uint32_t mask = 0;
    int cap = 0;
    int n[ONTOVAE_PHASE_COUNT] = {0};
    hipEvent_t* ev[ONTOVAE_PHASE_COUNT] = {nullptr};   // 2 * cap events per enabled phase
};
static OvProfState g_prof;

// Differential timing of the step's last kernel (dW0), free of what a bracket adds to the launch inside it: with phase
// ONTOVAE_PHASE_PAIR_A enabled, even-numbered training steps bracket the launch before dW0 alone (dWh + BatchNorm backward) and
// odd-numbered ones that launch AND dW0 in ONE bracket, with no event between the two kernels, as in an un-instrumented step
// (slots PAIR_A / PAIR_AB).  mean(AB) - mean(A) is dW0's time in the pipeline, launch gap included.
static int g_prof_pair_step = 0;
static bool g_prof_pair_open = false;
static inline bool ov_prof_pair_on() { return (g_prof.mask >> ONTOVAE_PHASE_PAIR_A) & 1u; }
static void ov_prof_pair_begin(hipStream_t st) {
    if (!ov_prof_pair_on()) return;
    const int ph = (g_prof_pair_step & 1) ? ONTOVAE_PHASE_PAIR_AB : ONTOVAE_PHASE_PAIR_A;
    g_prof_pair_open = g_prof.ev[ph] && g_prof.n[ph] < g_prof.cap;
    if (g_prof_pair_open) (void)hipEventRecord(g_prof.ev[ph][2 * g_prof.n[ph]], st);
}
static void ov_prof_pair_close(hipStream_t st, int ph) {
    (void)hipEventRecord(g_prof.ev[ph][2 * g_prof.n[ph] + 1], st);
    g_prof.n[ph]++;
    g_prof_pair_open = false;
}
static void ov_prof_pair_mid(hipStream_t st) {           // after the launch in front of dW0
    if (g_prof_pair_open && !(g_prof_pair_step & 1)) ov_prof_pair_close(st, ONTOVAE_PHASE_PAIR_A);
}
static void ov_prof_pair_end(hipStream_t st) {           // after dW0
    if (!ov_prof_pair_on()) return;
    if (g_prof_pair_open && (g_prof_pair_step & 1)) ov_prof_pair_close(st, ONTOVAE_PHASE_PAIR_AB);
    g_prof_pair_open = false;
    g_prof_pair_step++;
}

struct OvProfScope {
    int phase; hipStream_t st; bool on;
    OvProfScope(int ph, hipStream_t s) : phase(ph), st(s), on(false) {
        if ((g_prof.mask >> ph) & 1u) {
            if (g_prof.n[ph] < g_prof.cap) {
                on = true;
                (void)hipEventRecord(g_prof.ev[ph][2 * g_prof.n[ph]], st);
            }
        }
    }
    ~OvProfScope() {
        if (on) {
            (void)hipEventRecord(g_prof.ev[phase][2 * g_prof.n[phase] + 1], st);
            g_prof.n[phase]++;
        }
    }
};

extern "C" int ontovae_profile_end(void) {
    for (int ph = 0; ph < ONTOVAE_PHASE_COUNT; ++ph) {
        if (g_prof.ev[ph]) {
            for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[ph][i]);
            delete[] g_prof.ev[ph];
            g_prof.ev[ph] = nullptr;
        }
        g_prof.n[ph] = 0;
    }
    g_prof.mask = 0;
    g_prof.cap = 0;
    return ONTOVAE_OK;
}

extern "C" int ontovae_profile_begin(uint32_t phase_mask, int32_t capacity) {
    ontovae_profile_end();
    if (capacity < 1 || capacity > (1 << 20)) { ontovae_set_error("profile_begin: bad capacity %d", capacity); return ONTOVAE_EINVAL; }
    g_prof.cap = capacity;
    for (int ph = 0; ph < ONTOVAE_PHASE_COUNT; ++ph) {
        if (!((phase_mask >> ph) & 1u)) continue;
        g_prof.ev[ph] = new hipEvent_t[2 * capacity];
        for (int i = 0; i < 2 * capacity; ++i) OV_HIP_CHECK(hipEventCreateWithFlags(&g_prof.ev[ph][i], hipEventDisableSystemFence));      // no system-scope release (L2 write-back) charged to the bracketed kernel
    }
    if ((phase_mask >> ONTOVAE_PHASE_PAIR_A) & 1u) {       // the pair mode needs both slots
        if (!g_prof.ev[ONTOVAE_PHASE_PAIR_AB]) {
            g_prof.ev[ONTOVAE_PHASE_PAIR_AB] = new hipEvent_t[2 * capacity];
            for (int i = 0; i < 2 * capacity; ++i) OV_HIP_CHECK(hipEventCreateWithFlags(&g_prof.ev[ONTOVAE_PHASE_PAIR_AB][i], hipEventDisableSystemFence));
        }
        g_prof_pair_step = 0;
        g_prof_pair_open = false;
    }
    g_prof.mask = phase_mask;
    return ONTOVAE_OK;
}

extern "C" int ontovae_profile_read(int32_t phase, float* ms, int32_t cap, int32_t* n_out) {
    if (phase < 0 || phase >= ONTOVAE_PHASE_COUNT || !ms || !n_out) { ontovae_set_error("profile_read: bad arguments"); return ONTOVAE_EINVAL; }
    const int n = std::min(g_prof.n[phase], (int)cap);
    for (int i = 0; i < n; ++i) {
        OV_HIP_CHECK(hipEventSynchronize(g_prof.ev[phase][2 * i + 1]));
        OV_HIP_CHECK(hipEventElapsedTime(&ms[i], g_prof.ev[phase][2 * i], g_prof.ev[phase][2 * i + 1]));
    }
    *n_out = n;
    g_prof.n[phase] = 0;
    return ONTOVAE_OK;
}

__global__ void ov_empty_kernel() {}

extern "C" int ontovae_profile_overhead(void* stream, int32_t n, float* null_ms, float* empty_ms) {
    if (n < 1 || n > 4096 || !null_ms || !empty_ms) { ontovae_set_error("profile_overhead: bad arguments"); return ONTOVAE_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t* ev = new hipEvent_t[4 * n];
    for (int i = 0; i < 4 * n; ++i) {
        hipError_t e_ = hipEventCreateWithFlags(&ev[i], hipEventDisableSystemFence);
        if (e_ != hipSuccess) { for (int j = 0; j < i; ++j) (void)hipEventDestroy(ev[j]); delete[] ev; ontovae_set_error("profile_overhead: hipEventCreateWithFlags failed: %s", hipGetErrorString(e_)); return ONTOVAE_EHIP; }
    }
    for (int i = 0; i < n; ++i) {
        (void)hipEventRecord(ev[4 * i], st);
        (void)hipEventRecord(ev[4 * i + 1], st);
        (void)hipEventRecord(ev[4 * i + 2], st);
        hipLaunchKernelGGL(ov_empty_kernel, dim3(1), dim3(64), 0, st);
        (void)hipEventRecord(ev[4 * i + 3], st);
    }
    hipError_t err = hipStreamSynchronize(st);
    double a = 0.0, b = 0.0;
    for (int i = 0; i < n && err == hipSuccess; ++i) {
        float t = 0.f;
        err = hipEventElapsedTime(&t, ev[4 * i], ev[4 * i + 1]);
        a += t;
        if (err == hipSuccess) { err = hipEventElapsedTime(&t, ev[4 * i + 2], ev[4 * i + 3]); b += t; }
    }
    for (int i = 0; i < 4 * n; ++i) (void)hipEventDestroy(ev[i]);
    delete[] ev;
    if (err != hipSuccess) { ontovae_set_error("profile_overhead: %s", hipGetErrorString(err)); return ONTOVAE_EHIP; }
    *null_ms = (float)(a / n);
    *empty_ms = (float)(b / n);
    return ONTOVAE_OK;
}

static inline int ov_vec(int B) { return B <= 64 ? 1 : (B <= 128 ? 2 : 4); }
static inline int ov_ldb(int B) {
    const int q = 64 * ov_vec(B);
    int ldb = (B + q - 1) / q * q;
    // Large batches: an odd multiple of 256 floats, so that the rows of a feature-major matrix do not sit a power of two
    // apart.  A 32-column block reads one 128-B line of every row; at a row stride of 16 or 32 KB those lines fell on a
    // few memory channels and the gene layer ran 1.7x slower per column at 8192 columns than at 128 (measured: forward
    // gene layer 2250 -> 1290 us at B = 8192, whole step 9.8 -> 8.5 ms).  At most 6 % of padding columns.
    if (ldb >= 4096 && ((ldb >> 8) & 1) == 0) ldb += 256;
    return ldb;
}

// =================================================================================================
// Workspace
// =================================================================================================
#define OV_DWH_SMAX 16
struct OvWs {
    float *slab, *xhat, *a, *mulv, *epsw, *act, *d, *dblk, *dmulv, *da, *dh;
    float *bn_mean, *bn_invstd, *bn_bwd_local;
    float *rec_part, *kl_part;
    float *wpart;              // [ldb / 32][nnz + C + G] weight-gradient partials of the 32-column blocks
    float *wgene;              // [nnz - nnz_term] gene-layer weights in CSR order, refreshed from P at the start of every pass
    float *whT;                // [L][2 L] transposed copy of the head weights (da at large batches)
    float *whslab;             // [OV_DWH_SMAX][2 L][L] split-K slabs of dWh at large batches
    float *gb4;                // [ldb / 4][lanes of the gene layer's SELL form] gene bias gradient partials of ov_gene_lds_kernel
    int64_t slab_floats;
    int npart_max;
};

// Encoder forward GEMM (M = L, N = B, K = G): wave tile 96 x (32 NT), four waves of a workgroup split
// its K range, and enough workgroups along K to give every CU one (they each keep one wave per SIMD)
#define OV_ENC_TILES2 16
static int ov_enc_nt(int B) { return B <= 32 ? 1 : (B <= 64 ? 2 : 4); }
static int ov_enc_splitk(int L, int B, int G) {
    const int tiles = ((L + 95) / 96) * ((B + 32 * ov_enc_nt(B) - 1) / (32 * ov_enc_nt(B)));
    // one workgroup per CU while the split is what fills the chip (training batches); with many output tiles (inference
    // batches) two per CU hide each other's latency: 523 instead of 587 us at B = 4096
    const int target = tiles >= OV_ENC_TILES2 ? 512 : 256;
    int s = std::max(1, target / tiles);          // never more workgroups than the target: the extra ones would be a second round
    const int smax = std::max(1, G / 256);
    return std::max(1, std::min(std::min(s, 128), smax));
}

// lanes of the gene layer's SELL form (ontovae_gsell_t): every window of 512 genes rounds up to whole slices of 64
static inline int64_t ov_gsell_lanes_max(int64_t G) { return (G + 511) / 512 * 512; }

// rows a training entry point may take: ctx->max_train_batch, or max_batch if that is 0
static inline int ov_train_cap(const ontovae_ctx_t* c) {
    return c->max_train_batch > 0 ? std::min(c->max_train_batch, c->max_batch) : c->max_batch;
}

// train_batch: the buffers only a training pass touches (eps, the column-group-major D panel, the backward's small
// matrices, the weight-gradient partials: 128 to 512 copies of the packed decoder at inference batch sizes) are sized by it
static int64_t ov_carve(const ontovae_dims_t& d, int max_batch, int train_batch, void* base, OvWs* ws) {
    const int64_t ldb = ov_ldb(max_batch);
    const int64_t tldb = ov_ldb(std::min(train_batch, max_batch));
    const int64_t L = d.L, C = d.C, G = d.G;
    int64_t slab = 0;
    for (int b = 64; b <= ldb; b += 64)
        slab = std::max<int64_t>(slab, (int64_t)ov_enc_splitk(d.L, b, d.G) * L * ov_ldb(b));
    // the heads GEMM leaves its [2L, ldb] output (1 to 3 split-K slabs) in the same buffer (ov_heads_fwd)
    slab = std::max<int64_t>(slab, 2 * L * ldb);
    const int64_t npart = ((G + 3) / 4) * (ldb / 64) + 1024;
    int64_t off = 0;
    auto take = [&](int64_t n) {
        float* p = base ? reinterpret_cast<float*>(base) + off : nullptr;
        off += (n + 63) / 64 * 64;
        return p;
    };
    OvWs w;
    w.slab = take(slab);
    w.xhat = take(L * ldb);
    w.a = take(L * ldb);
    w.mulv = take(2 * L * ldb);
    w.epsw = take(L * tldb);
    w.act = take(C * ldb);
    w.d = take((C + G) * ldb);
    w.dblk = take(C * tldb);     // panel rows of D in column-group-major order [ldb / 4][C][4] (backward chain input)
    w.dmulv = take(2 * L * tldb);
    w.da = take(L * tldb);
    w.dh = take(L * tldb);
    w.bn_mean = take(L);
    w.bn_invstd = take(L);
    w.bn_bwd_local = take(2 * L);
    w.rec_part = take(npart);
    w.kl_part = take(npart);
    w.wpart = take((tldb / 32) * ((int64_t)d.nnz + C + G));
    w.wgene = take((int64_t)d.nnz - d.nnz_term);
    w.gb4 = take((tldb / 4) * ov_gsell_lanes_max(d.G));
    w.whT = take(2 * L * L);
    w.whslab = take((int64_t)OV_DWH_SMAX * 2 * L * L);
    w.slab_floats = slab;
    w.npart_max = (int)npart;
    if (ws) *ws = w;
    return off * (int64_t)sizeof(float);
}

extern "C" int64_t ontovae_workspace_bytes(const ontovae_dims_t* dims, int32_t max_batch) {
    if (!dims || max_batch < 1) return -1;
    return ov_carve(*dims, max_batch, max_batch, nullptr, nullptr);
}
extern "C" int64_t ontovae_workspace_bytes_split(const ontovae_dims_t* dims, int32_t max_batch, int32_t max_train_batch) {
    if (!dims || max_batch < 1 || max_train_batch < 1) return -1;
    return ov_carve(*dims, max_batch, max_train_batch, nullptr, nullptr);
}

// Operand copies of the bf16-MFMA variant (ONTOVAE_FLAG_BF16_GEMM): xb [max_batch, G], xT [G, ldb(max_batch)], wb [L, G],
// dhb [L, ldb(max_batch)], 16-bit elements, 256-B aligned pieces of ctx->bf16_ws
struct OvBfWs { uint16_t *xb, *xT, *wb, *dhb; };
static int64_t ov_bf16_carve(const ontovae_dims_t& d, int max_batch, void* base, OvBfWs* w) {
    const int64_t ldb = ov_ldb(max_batch);
    int64_t off = 0;
    auto take = [&](int64_t n) {
        uint16_t* p = base ? reinterpret_cast<uint16_t*>(base) + off : nullptr;
        off += (n + 127) / 128 * 128;
        return p;
    };
    OvBfWs r;
    r.xb = take((int64_t)max_batch * d.G);
    r.xT = take((int64_t)d.G * ldb);
    r.wb = take((int64_t)d.L * d.G);
    r.dhb = take((int64_t)d.L * ldb);
    if (w) *w = r;
    return off * (int64_t)sizeof(uint16_t);
}

extern "C" int64_t ontovae_bf16_workspace_bytes(const ontovae_dims_t* dims, int32_t max_batch) {
    if (!dims || max_batch < 1) return -1;
    return ov_bf16_carve(*dims, max_batch, nullptr, nullptr);
}

// =================================================================================================
// Encoder: BatchNorm (+ dropout + ReLU), feature-major, one wave per feature
// =================================================================================================
// h[l, b] = b0[l] + sum_s slab[s][l][b]                 (split-K combine of x W0^T, modules.py:46)
// train: batch statistics (biased variance), running buffers updated (modules.py:47)
// a = relu(dropout(gamma * xhat + beta))                (modules.py:47-49: BN, Dropout, ReLU)
// mode: 0 eval (running stats), 1 train fused, 2 DP phase 0 (h + local mean/M2 only),
//       3 DP phase 1 (normalise with the supplied global mean/var; h already in xhat)
//       5 DP phase 1 with the ranks' all-gathered partial statistics, combined here
__global__ __launch_bounds__(256) void ov_bn_fwd_kernel(const float* __restrict__ slab, int S, int64_t slab_stride,
                                                       const float* __restrict__ b0, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ rmean,
                                                       float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                                       float* __restrict__ xhat, float* __restrict__ a,
                                                       float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                       float* __restrict__ stats, int L, int B, int Bglobal, int ldb,
                                                       int mode, float drop, uint64_t seed, uint64_t rng_off) {
    // one workgroup per feature: the four waves share the split-K combine (each sums a quarter of the
    // slabs, 16 B per lane), wave 0 then owns the row
    __shared__ __attribute__((aligned(16))) float part[4][256];
    const int l = blockIdx.x;
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    float* xr = xhat + (int64_t)l * ldb;
    float* ar = a + (int64_t)l * ldb;
    float mean, var;
    if (mode != 3 && mode != 5) {
        const float bias = b0[l];
        float s = 0.f;
        for (int c0 = 0; c0 < ldb; c0 += 256) {
            const int b = c0 + lane * 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (b < ldb) {
                const float* src = slab + (int64_t)l * ldb + b;
                int z = wave;
                for (; z + 12 < S; z += 16) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (int64_t)z * slab_stride);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 4) * slab_stride);
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 8) * slab_stride);
                    const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 12) * slab_stride);
                    acc += (v0 + v1) + (v2 + v3);
                }
                for (; z < S; z += 4) acc += *reinterpret_cast<const f32x4*>(src + (int64_t)z * slab_stride);
            }
            *reinterpret_cast<f32x4*>(&part[wave][lane * 4]) = acc;
            __syncthreads();
            if (wave == 0 && b < ldb) {
                f32x4 h4 = (*reinterpret_cast<const f32x4*>(&part[0][lane * 4]) + *reinterpret_cast<const f32x4*>(&part[1][lane * 4])) +
                           (*reinterpret_cast<const f32x4*>(&part[2][lane * 4]) + *reinterpret_cast<const f32x4*>(&part[3][lane * 4]));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float h = b + j < B ? h4[j] + bias : 0.f;     // slab columns >= B were never written
                    h4[j] = h;
                    s += h;
                }
                *reinterpret_cast<f32x4*>(xr + b) = h4;
            }
            __syncthreads();
        }
        if (wave != 0) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the row just stored is re-read by other lanes below
        if (mode == 0) {
            mean = rmean[l];
            var = rvar[l];
        } else {
            mean = ov_wave_sum(s) / (float)B;
            float q = 0.f;
            for (int b = lane; b < B; b += 64) {
                const float dlt = xr[b] - mean;
                q += dlt * dlt;
            }
            q = ov_wave_sum(q);
            if (mode == 2) {
                if (lane == 0) {
                    stats[l] = mean;
                    stats[L + l] = q;
                }
                return;
            }
            var = q / (float)B;
        }
    } else if (mode == 5) {
        // the ranks' all-gathered (mean, M2, rows) blocks, S of them with stride slab_stride: Chan's combination
        if (wave != 0) return;
        float n = 0.f, m = 0.f;
        for (int r = 0; r < S; ++r) {
            const float nr = stats[r * slab_stride + 2 * L];
            n += nr;
            m += nr * stats[r * slab_stride + l];
        }
        mean = m / n;
        float m2 = 0.f;
        for (int r = 0; r < S; ++r) {
            const float nr = stats[r * slab_stride + 2 * L];
            const float dm = stats[r * slab_stride + l] - mean;
            m2 += stats[r * slab_stride + L + l] + nr * dm * dm;
        }
        var = m2 / n;
    } else {
        if (wave != 0) return;
        mean = stats[l];
        var = stats[L + l];
    }
    if (mode == 1 || mode == 3 || mode == 5) {
        if (lane == 0) {
            const float mom = 0.1f;
            const float nb = (float)Bglobal;
            rmean[l] = (1.f - mom) * rmean[l] + mom * mean;
            rvar[l] = (1.f - mom) * rvar[l] + mom * (var * (nb / (nb - 1.f)));
            if (l == 0) *nbt += 1;
        }
    }
    const float invstd = 1.0f / sqrtf(var + 1e-5f);
    if (lane == 0) {
        mean_out[l] = mean;
        invstd_out[l] = invstd;
    }
    const float g = gamma[l], be = beta[l];
    const bool dropping = (mode != 0) && drop > 0.f;
    for (int b = lane; b < ldb; b += 64) {
        float xh = 0.f, y = 0.f;
        if (b < B) {
            xh = (xr[b] - mean) * invstd;
            y = xh * g + be;
            if (dropping) y *= ov_dropout_scale(drop, seed, rng_off, OV_RNG_DROP, (uint64_t)l * 0x100000000ull + b);
            y = fmaxf(y, 0.f);
        }
        xr[b] = xh;
        ar[b] = y;
    }
}

// The same at 1024 columns and more (modes 1, 2, 3, 5): ov_bn_fwd_kernel is built for 64 slabs of 128 columns -- its waves
// split the slabs, then wave 0 alone walks the row twice in 4-B steps (87 us at 8192 columns).  Here the whole workgroup walks the
// row in every pass, 16 B per lane: slab combine (the order of ov_bn_eval_kernel: four interleaved partial sums, (p0 + p1) +
// (p2 + p3)) + bias + row sum; squared deviations from the mean; normalise / scale / dropout / ReLU.  A thread re-reads only
// what it stored itself, so the passes need no barrier except inside the two reductions (lane -> wave -> workgroup, fixed order).
__device__ __forceinline__ float ov_wg_sum256(float v, float* red) {
    v = ov_wave_sum(v);
    if (ov_lane() == 0) red[ov_wave_id()] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(256) void ov_bn_fwd_wide_kernel(const float* __restrict__ slab, int S, int64_t slab_stride,
                                                            const float* __restrict__ b0, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                                            float* __restrict__ xhat, float* __restrict__ a,
                                                            float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                            float* __restrict__ stats, int L, int B, int Bglobal, int ldb,
                                                            int mode, float drop, uint64_t seed, uint64_t rng_off) {
    __shared__ float red[4];
    const int l = blockIdx.x, tid = threadIdx.x;
    float* xr = xhat + (int64_t)l * ldb;
    float* ar = a + (int64_t)l * ldb;
    float mean, var;
    if (mode == 1 || mode == 2) {
        const float bias = b0[l];
        float sum = 0.f;
        for (int b = tid * 4; b < ldb; b += 1024) {
            const float* src = slab + (int64_t)l * ldb + b;
            f32x4 p[4];
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                p[z] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int q = z; q < S; q += 4) p[z] += *reinterpret_cast<const f32x4*>(src + (int64_t)q * slab_stride);
            }
            f32x4 h4 = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h4[j] = b + j < B ? h4[j] + bias : 0.f;          // slab columns >= B were never written
                sum += h4[j];
            }
            *reinterpret_cast<f32x4*>(xr + b) = h4;
        }
        mean = ov_wg_sum256(sum, red) / (float)B;
        float q = 0.f;
        for (int b = tid * 4; b < ldb; b += 1024) {
            const f32x4 h4 = *reinterpret_cast<const f32x4*>(xr + b);       // this thread's own stores
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dlt = h4[j] - mean;
                q += b + j < B ? dlt * dlt : 0.f;
            }
        }
        q = ov_wg_sum256(q, red);
        if (mode == 2) {
            if (tid == 0) { stats[l] = mean; stats[L + l] = q; }
            return;
        }
        var = q / (float)B;
    } else if (mode == 5) {
        // the ranks' all-gathered (mean, M2, rows) blocks, S of them with stride slab_stride: Chan's combination
        float n = 0.f, m = 0.f;
        for (int r = 0; r < S; ++r) {
            const float nr = stats[r * slab_stride + 2 * L];
            n += nr;
            m += nr * stats[r * slab_stride + l];
        }
        mean = m / n;
        float m2 = 0.f;
        for (int r = 0; r < S; ++r) {
            const float nr = stats[r * slab_stride + 2 * L];
            const float dm = stats[r * slab_stride + l] - mean;
            m2 += stats[r * slab_stride + L + l] + nr * dm * dm;
        }
        var = m2 / n;
    } else {                       // mode 3
        mean = stats[l];
        var = stats[L + l];
    }
    if (tid == 0) {
        const float mom = 0.1f;
        const float nb = (float)Bglobal;
        rmean[l] = (1.f - mom) * rmean[l] + mom * mean;
        rvar[l] = (1.f - mom) * rvar[l] + mom * (var * (nb / (nb - 1.f)));
        if (l == 0) *nbt += 1;
    }
    const float invstd = 1.0f / sqrtf(var + 1e-5f);
    if (tid == 0) {
        mean_out[l] = mean;
        invstd_out[l] = invstd;
    }
    const float g = gamma[l], be = beta[l];
    const bool dropping = drop > 0.f;
    for (int b = tid * 4; b < ldb; b += 1024) {
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(xr + b);
        f32x4 xh4, y4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float xh = 0.f, y = 0.f;
            if (b + j < B) {
                xh = (h4[j] - mean) * invstd;
                y = xh * g + be;
                if (dropping) y *= ov_dropout_scale(drop, seed, rng_off, OV_RNG_DROP, (uint64_t)l * 0x100000000ull + (b + j));
                y = fmaxf(y, 0.f);
            }
            xh4[j] = xh; y4[j] = y;
        }
        *reinterpret_cast<f32x4*>(xr + b) = xh4;
        *reinterpret_cast<f32x4*>(ar + b) = y4;
    }
}

// Eval mode (running statistics) at inference batch sizes: plain elementwise work over the whole [L, ldb] matrix instead of
// one wave walking a feature's row.  Same operations in the same order as mode 0 above (S <= 12: wave z there adds slabs
// z, z + 4, z + 8 one after the other).
__global__ __launch_bounds__(256) void ov_bn_eval_kernel(const float* __restrict__ slab, int S, int64_t slab_stride,
                                                        const float* __restrict__ b0, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ rmean,
                                                        const float* __restrict__ rvar, float* __restrict__ a, int B, int ldb) {
    const int l = blockIdx.y, b = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (b >= ldb) return;
    const float* src = slab + (int64_t)l * ldb + b;
    f32x4 p[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        p[z] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int q = z; q < S; q += 4) p[z] += *reinterpret_cast<const f32x4*>(src + (int64_t)q * slab_stride);
    }
    f32x4 h4 = (p[0] + p[1]) + (p[2] + p[3]);
    const float bias = b0[l], mean = rmean[l], invstd = 1.0f / sqrtf(rvar[l] + 1e-5f), g = gamma[l], be = beta[l];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float y = 0.f;
        if (b + j < B) {
            const float xh = ((h4[j] + bias) - mean) * invstd;
            y = fmaxf(xh * g + be, 0.f);
        }
        h4[j] = y;
    }
    *reinterpret_cast<f32x4*>(a + (int64_t)l * ldb + b) = h4;
}

// BN backward (+ ReLU, dropout), one wave per feature.  da = dL/da.
//   dy = da * [a > 0] * keep;  dgamma = sum dy*xhat;  dbeta = sum dy
//   dh = gamma * invstd / Bg * (Bg * dy - dbeta - xhat * dgamma)
// mode 1: fused single GPU (local sums are global), then AdamW on gamma, beta, b0 or grads to Gr
// mode 2: DP phase 1 tail: write dy into dh buffer and local sums to bwd_local/bwd_stats
// mode 3: DP phase 2 head: read dy from dh, global sums from bwd_stats; grads (LOCAL sums) to Gr
struct OvBnBwdArgs {
    const float* da; const float* a; const float* xhat; const float* invstd;
    float* P_gamma; float* P_beta; float* P_b0; float* M_gamma; float* M_beta; float* M_b0;
    float* V_gamma; float* V_beta; float* V_b0; float* G_gamma; float* G_beta; float* G_b0;
    float* dh; float* bwd_local; float* bwd_stats;
    int L, B, Bglobal, ldb, mode, apply_update; OvAdam adam; float drop; uint64_t seed, rng_off;
    int daS; int64_t dastride;
    float* send; int blm;          // mode 4: also [L][blm] dh + [L] local b0 gradient for the all-gather
};
// block: this workgroup's number inside the BatchNorm part = the feature.  The four waves take every fourth chunk of 64
// columns (a wave per feature walked 8192 columns in 128 dependent trips: 249 us at that batch); their sums are combined
// through LDS in a fixed order.
__device__ __forceinline__ void ov_bn_bwd_body(const OvBnBwdArgs& q, const int block) {
    __shared__ float bn_red[3][4];
    const float* __restrict__ da = q.da; const float* __restrict__ a = q.a; const float* __restrict__ xhat = q.xhat;
    const float* __restrict__ invstd = q.invstd;
    float* __restrict__ P_gamma = q.P_gamma; float* __restrict__ P_beta = q.P_beta; float* __restrict__ P_b0 = q.P_b0;
    float* __restrict__ M_gamma = q.M_gamma; float* __restrict__ M_beta = q.M_beta; float* __restrict__ M_b0 = q.M_b0;
    float* __restrict__ V_gamma = q.V_gamma; float* __restrict__ V_beta = q.V_beta; float* __restrict__ V_b0 = q.V_b0;
    float* __restrict__ G_gamma = q.G_gamma; float* __restrict__ G_beta = q.G_beta; float* __restrict__ G_b0 = q.G_b0;
    float* __restrict__ dh = q.dh; float* __restrict__ bwd_local = q.bwd_local; float* __restrict__ bwd_stats = q.bwd_stats;
    const int L = q.L, B = q.B, Bglobal = q.Bglobal, ldb = q.ldb, mode = q.mode, apply_update = q.apply_update;
    const OvAdam adam = q.adam; const float drop = q.drop; const uint64_t seed = q.seed, rng_off = q.rng_off;
    const int daS = q.daS; const int64_t dastride = q.dastride;
    const int l = block;
    if (l >= L) return;                            // workgroup-uniform
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    const int64_t ro = (int64_t)l * ldb;
    float sdy = 0.f, sdyx = 0.f;
    // from 1024 columns on the row is walked 16 B per lane (ldb is a multiple of 64 and the rows are 256-B aligned); a thread
    // still re-reads in the second pass only what it stored in the first
    const bool wide = ldb >= 1024;
    if (mode < 3 && wide) {
        for (int b = threadIdx.x * 4; b < ldb; b += 1024) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(a + ro + b);
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(xhat + ro + b);
            f32x4 dy4 = {0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < daS; ++z) dy4 += *reinterpret_cast<const f32x4*>(da + z * dastride + ro + b);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dy = (b + j < B && a4[j] > 0.f) ? dy4[j] : 0.f;
                if (drop > 0.f && b + j < B) dy *= ov_dropout_scale(drop, seed, rng_off, OV_RNG_DROP, (uint64_t)l * 0x100000000ull + (b + j));
                sdy += dy;
                sdyx += dy * (b + j < B ? x4[j] : 0.f);
                dy4[j] = dy;
            }
            *reinterpret_cast<f32x4*>(dh + ro + b) = dy4;
        }
    } else if (mode < 3) {
        for (int b = wave * 64 + lane; b < ldb; b += 256) {
            float dy = 0.f;
            if (b < B) {
                if (a[ro + b] > 0.f)
                    for (int q = 0; q < daS; ++q) dy += da[q * dastride + ro + b];      // split-K partials of the da GEMM
                if (drop > 0.f) dy *= ov_dropout_scale(drop, seed, rng_off, OV_RNG_DROP, (uint64_t)l * 0x100000000ull + b);
                sdy += dy;
                sdyx += dy * xhat[ro + b];
            }
            dh[ro + b] = dy;
        }
    }
    if (mode < 3) {
        sdy = ov_wave_sum(sdy);
        sdyx = ov_wave_sum(sdyx);
        if (lane == 0) { bn_red[0][wave] = sdy; bn_red[1][wave] = sdyx; }
        __syncthreads();
        sdy = (bn_red[0][0] + bn_red[0][1]) + (bn_red[0][2] + bn_red[0][3]);
        sdyx = (bn_red[1][0] + bn_red[1][1]) + (bn_red[1][2] + bn_red[1][3]);
        if (mode == 2) {
            if (threadIdx.x == 0) {
                bwd_local[l] = sdy; bwd_local[L + l] = sdyx;
                bwd_stats[l] = sdy; bwd_stats[L + l] = sdyx;
            }
            return;
        }
    }
    float gdy = sdy, gdyx = sdyx;          // global sums
    if (mode >= 3) {
        gdy = bwd_stats[l]; gdyx = bwd_stats[L + l];
        sdy = bwd_local[l]; sdyx = bwd_local[L + l];
        if (mode == 4) { sdy = gdy; sdyx = gdyx; }      // gradients of gamma / beta: the global sums, on every rank
    }
    const float g = P_gamma[l];
    const float is = invstd[l];
    const float nb = (float)Bglobal;
    const float scale = g * is / nb;
    float sdh = 0.f;
    if (wide && mode != 4) {
        for (int b = threadIdx.x * 4; b < ldb; b += 1024) {      // the columns this thread wrote above
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dh + ro + b);
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(xhat + ro + b);
            f32x4 v4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = b + j < B ? scale * (nb * d4[j] - gdy - x4[j] * gdyx) : 0.f;
                sdh += v;
                v4[j] = v;
            }
            *reinterpret_cast<f32x4*>(dh + ro + b) = v4;
        }
    } else
    for (int b = wave * 64 + lane; b < ldb; b += 256) {          // the columns this thread wrote above
        float v = 0.f;
        if (b < B) {
            v = scale * (nb * dh[ro + b] - gdy - xhat[ro + b] * gdyx);
            sdh += v;
        }
        dh[ro + b] = v;
        if (mode == 4 && b < q.blm) q.send[(int64_t)l * q.blm + b] = v;
    }
    sdh = ov_wave_sum(sdh);
    if (lane == 0) bn_red[2][wave] = sdh;
    __syncthreads();
    sdh = (bn_red[2][0] + bn_red[2][1]) + (bn_red[2][2] + bn_red[2][3]);
    if (mode == 4 && threadIdx.x == 0) q.send[(int64_t)L * q.blm + l] = sdh;
    if (threadIdx.x == 0) {
        if (apply_update) {
            float w, m, v;
            w = P_gamma[l]; m = M_gamma[l]; v = V_gamma[l]; ov_adamw(w, m, v, sdyx, adam, false); P_gamma[l] = w; M_gamma[l] = m; V_gamma[l] = v;
            w = P_beta[l];  m = M_beta[l];  v = V_beta[l];  ov_adamw(w, m, v, sdy, adam, false);  P_beta[l] = w;  M_beta[l] = m;  V_beta[l] = v;
            w = P_b0[l];    m = M_b0[l];    v = V_b0[l];    ov_adamw(w, m, v, sdh, adam, false);  P_b0[l] = w;    M_b0[l] = m;    V_b0[l] = v;
        } else {
            G_gamma[l] = sdyx; G_beta[l] = sdy; G_b0[l] = sdh;
        }
    }
}

__global__ __launch_bounds__(256) void ov_bn_bwd_kernel(const OvBnBwdArgs q) { ov_bn_bwd_body(q, blockIdx.x); }

// =================================================================================================
// Latent: z_drop dropout on the heads, reparameterisation, KL
// =================================================================================================
// mulv[0:L] = mu, mulv[L:2L] = log_var (feature-major, bias already added by the GEMM epilogue).
// z = mu + eps * exp(0.5 * log_var)     (vae_model.py:91-93)  -> panel rows of level 0
// kl = -0.5 * sum(1 + lv - mu^2 - exp(lv))   (vae_model.py:124), block partials in kl_part
// Blocks beyond nrep do an unrelated job that is due at the same point of a pass and would otherwise be a launch of
// its own: the refresh of the SELL-ordered weight copies of the term layers from the parameter buffer.
struct OvRefresh {
    const float* vals; const float* bias;
    const int32_t* kf; float* wf; int nf;
    const int32_t* nodef; float* biasf; int nbf;
    const int32_t* kb; float* wb; int nb;
    const int32_t* kg; float* wg; int ng;      // gene layer: weights in the order the pass's forward gene kernel walks (CSR, or the SELL form) <- packed order
};
__device__ __forceinline__ void ov_sell_refresh_part(const OvRefresh& r, int first, int stride) {
    for (int i = first; i < r.nf; i += stride) { const int k = r.kf[i]; r.wf[i] = k >= 0 ? r.vals[k] : 0.f; }
    for (int i = first; i < r.nbf; i += stride) { const int n = r.nodef[i]; r.biasf[i] = n >= 0 ? r.bias[n] : 0.f; }
    for (int i = first; i < r.nb; i += stride) { const int k = r.kb[i]; r.wb[i] = k >= 0 ? r.vals[k] : 0.f; }
    for (int i = first; i < r.ng; i += stride) { const int k = r.kg[i]; r.wg[i] = k >= 0 ? r.vals[k] : 0.f; }
}

__global__ __launch_bounds__(256) void ov_reparam_kernel(float* __restrict__ mulv, const float* __restrict__ eps,
                                                        float* __restrict__ epsw, float* __restrict__ zrows,
                                                        float* __restrict__ kl_part, int L, int B, int ldb, int train,
                                                        float z_drop, uint64_t seed, uint64_t rng_off, uint32_t eps_stream,
                                                        int write_kl, int nrep, const OvRefresh rf,
                                                        const float* __restrict__ hslab, int hS, int64_t hstride,
                                                        const float* __restrict__ bh) {
    if ((int)blockIdx.x >= nrep) {
        ov_sell_refresh_part(rf, ((int)blockIdx.x - nrep) * 256 + threadIdx.x, ((int)gridDim.x - nrep) * 256);
        return;
    }
    const int64_t n = (int64_t)L * ldb;
    float kl = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)nrep * 256) {
        // ldb is a multiple of 64: the split into (feature, column) needs no 64-bit division
        const uint32_t q = (uint32_t)(idx >> 6) / (uint32_t)(ldb >> 6);
        const int l = (int)q, b = (int)(idx - (int64_t)q * ldb);
        float z = 0.f, e = 0.f;
        if (b < B) {
            float mu, lv;
            if (hslab) {          // heads GEMM left split-K partials: mu / log_var = bias + sum of the slabs (modules.py:57, 62)
                mu = bh[l];
                lv = bh[L + l];
                for (int q = 0; q < hS; ++q) { mu += hslab[q * hstride + idx]; lv += hslab[q * hstride + n + idx]; }
            } else {
                mu = mulv[idx];
                lv = mulv[n + idx];
            }
            if (train && z_drop > 0.f) {
                mu *= ov_dropout_scale(z_drop, seed, rng_off, OV_RNG_ZDROP_MU, (uint64_t)l * 0x100000000ull + b);
                lv *= ov_dropout_scale(z_drop, seed, rng_off, OV_RNG_ZDROP_LV, (uint64_t)l * 0x100000000ull + b);
            }
            if (hslab || (train && z_drop > 0.f)) {
                mulv[idx] = mu;
                mulv[n + idx] = lv;
            }
            e = eps ? eps[(int64_t)b * L + l] : ov_randn(seed, rng_off, eps_stream, (uint64_t)l * 0x100000000ull + b);
            z = mu + e * expf(0.5f * lv);
            kl += 1.f + lv - mu * mu - expf(lv);
        } else {
            mulv[idx] = 0.f;
            mulv[n + idx] = 0.f;
        }
        if (epsw) epsw[idx] = e;
        zrows[idx] = z;
    }
    if (write_kl) {
        __shared__ float red[4];
        kl = ov_wave_sum(kl);
        if (ov_lane() == 0) red[threadIdx.x >> 6] = kl;
        __syncthreads();
        if (threadIdx.x == 0) kl_part[blockIdx.x] = -0.5f * (red[0] + red[1] + red[2] + red[3]);
    }
}

// dmu = dz + kc*mu ; dlv = dz * eps * 0.5 * sigma + 0.5*kc*(exp(lv) - 1), then back through z_drop.
// One wave per latent feature l (rows l and L + l of dmulv).  Two small jobs that are due at the same point ride
// along instead of being launches of their own: the row sums of dmulv are the gradient of the head biases
// (AdamW applied here, or stored when apply_update == 0), and the LAST block does the final reduction of the loss
// partials (loss = rec + kl_coeff * kl, vae_model.py:129).
struct OvLossArgs { const float* rec_part; int nrec; const float* kl_part; int nkl; float kc; float* out; };
struct OvLatentArgs {
    const float* dz; const float* mulv; const float* epsw; float* dmulv;
    int L, B, ldb; float kc, z_drop; uint64_t seed, rng_off;
    float* bh; float* mbh; float* vbh; float* gbh; int apply_update; OvAdam adam;
    OvLossArgs la;
    int nblocks;          // L feature blocks + 1 loss block
};
// block: this workgroup's number inside the latent part, in [0, nblocks)
__device__ __forceinline__ void ov_latent_bwd_body(const OvLatentArgs& a, const int block) {
    const int lane = ov_lane();
    if (block == a.nblocks - 1) {
        __shared__ double red[2][4];
        const OvLossArgs& la = a.la;
        double r = 0.0, k = 0.0;
        // one workgroup sums every partial of the step (160 k of them at 8192 rows): eight loads in flight per thread, or the
        // loop is one memory round trip per element (268 us at that batch)
        auto total = [&](const float* __restrict__ p, int n) {
            double acc = 0.0;
            int i = threadIdx.x;
            for (; i + 7 * 256 < n; i += 8 * 256) {
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = p[i + j * 256];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (double)t[j];
            }
            for (; i < n; i += 256) acc += (double)p[i];
            return acc;
        };
        r = total(la.rec_part, la.nrec);
        k = total(la.kl_part, la.nkl);
        r = ov_wave_sum_d(r);
        k = ov_wave_sum_d(k);
        if (lane == 0) { red[0][threadIdx.x >> 6] = r; red[1][threadIdx.x >> 6] = k; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float rf = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
            const float kf = (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
            la.out[0] = rf;
            la.out[1] = kf;
            la.out[2] = rf + la.kc * kf;
        }
        return;
    }
    // one workgroup per latent feature, the four waves take every fourth chunk of 64 columns (see ov_bn_bwd_body)
    __shared__ float lat_red[2][4];
    const int L = a.L, B = a.B, ldb = a.ldb;
    const int l = block;
    const int wave = ov_wave_id();
    if (l >= L) return;                            // workgroup-uniform
    const int64_t n = (int64_t)L * ldb;
    float smu = 0.f, slv = 0.f;
    for (int b = wave * 64 + lane; b < ldb; b += 256) {
        const int64_t idx = (int64_t)l * ldb + b;
        float dmu = 0.f, dlv = 0.f;
        if (b < B) {
            const float mu = a.mulv[idx], lv = a.mulv[n + idx], g = a.dz[idx], e = a.epsw[idx];
            dmu = g + a.kc * mu;
            dlv = g * e * 0.5f * expf(0.5f * lv) + 0.5f * a.kc * (expf(lv) - 1.f);
            if (a.z_drop > 0.f) {
                dmu *= ov_dropout_scale(a.z_drop, a.seed, a.rng_off, OV_RNG_ZDROP_MU, (uint64_t)l * 0x100000000ull + b);
                dlv *= ov_dropout_scale(a.z_drop, a.seed, a.rng_off, OV_RNG_ZDROP_LV, (uint64_t)l * 0x100000000ull + b);
            }
        }
        a.dmulv[idx] = dmu;
        a.dmulv[n + idx] = dlv;
        smu += dmu;
        slv += dlv;
    }
    smu = ov_wave_sum(smu);
    slv = ov_wave_sum(slv);
    if (lane == 0) { lat_red[0][wave] = smu; lat_red[1][wave] = slv; }
    __syncthreads();
    smu = (lat_red[0][0] + lat_red[0][1]) + (lat_red[0][2] + lat_red[0][3]);
    slv = (lat_red[1][0] + lat_red[1][1]) + (lat_red[1][2] + lat_red[1][3]);
    if (threadIdx.x < 2) {              // thread 0: bias of mu[l]; thread 1: bias of log_var[l]
        const int r = threadIdx.x == 0 ? l : L + l;
        const float g = threadIdx.x == 0 ? smu : slv;
        if (a.apply_update) {
            float ww = a.bh[r], mm = a.mbh[r], vv = a.vbh[r];
            ov_adamw(ww, mm, vv, g, a.adam, false);
            a.bh[r] = ww; a.mbh[r] = mm; a.vbh[r] = vv;
        } else {
            a.gbh[r] = g;
        }
    }
}

// =================================================================================================
// Ontology decoder: level-scheduled gather-SpMM over the packed masks
// =================================================================================================
// Forward, one wave per (node, 64*VEC batch columns):
//   ACT[n, :] = bias[n] + sum_k vals[k] * ACT[colidx[k], :]           (modules.py:309, 311)
// GENE variant (the last layer): the reconstruction never goes to HBM in training; the epilogue
// forms d = recon - x, writes dL/drecon = 2 d into D[n, :] and block partials of sum d^2
// (vae_model.py:125), or stores recon sample-major for inference.
template <int VEC>
struct OvVecT;
template <> struct OvVecT<1> { typedef float T; };
template <> struct OvVecT<2> { typedef float __attribute__((ext_vector_type(2))) T; };
template <> struct OvVecT<4> { typedef f32x4 T; };

template <int VEC>
__device__ __forceinline__ void ov_ldv(const float* p, float (&v)[VEC]) {
    typedef typename OvVecT<VEC>::T VT;
    const VT t = *reinterpret_cast<const VT*>(p);
    if constexpr (VEC == 1) v[0] = t;
    else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = t[j];
    }
}
template <int VEC>
__device__ __forceinline__ void ov_stv(float* p, const float (&v)[VEC]) {
    typedef typename OvVecT<VEC>::T VT;
    VT t;
    if constexpr (VEC == 1) t = v[0];
    else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) t[j] = v[j];
    }
    *reinterpret_cast<VT*>(p) = t;
}

// Term level of a screen pass (inference batch, activities wanted feature-major): one wave per TERM.  The NU neurons of a
// term have the same child list (every term-to-term link is a dense NU x NU block, modules.py:268-275), so the wave
// reads each child row once for NU accumulators, stores the NU rows of the panel, and writes their mean
// (vae_model.py:317) straight to the activity matrix: the separate output kernel, which re-reads the whole panel, is not
// needed.  Same summation order per row as ov_dec_fwd_kernel.  A term whose rows do not share one list (never
// produced by the packer; checked from the index lists) is done row by row.
template <int VEC, int NU>
__global__ __launch_bounds__(256) void ov_dec_fwd_term_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                             const float* __restrict__ vals, const float* __restrict__ bias,
                                                             float* __restrict__ act, int node_begin, int node_end, int B, int ldb,
                                                             float* __restrict__ fm_out, int64_t fm_ld, int c_base) {
    const int term = blockIdx.x * 4 + ov_wave_id();
    const int node = node_begin + term * NU;
    const int b0 = blockIdx.y * (64 * VEC) + ov_lane() * VEC;
    if (node >= node_end) return;
    int k0[NU];
    const int len = rowptr[node + 1] - rowptr[node];
    bool same = true;
#pragma unroll
    for (int a = 0; a < NU; ++a) {
        k0[a] = rowptr[node + a];
        same = same && (rowptr[node + a + 1] - k0[a] == len);
    }
    if (same)
        for (int i = 0; i < len; ++i)
#pragma unroll
            for (int a = 1; a < NU; ++a) same = same && (colidx[k0[a] + i] == colidx[k0[0] + i]);
    float acc[NU][VEC];
#pragma unroll
    for (int a = 0; a < NU; ++a) {
        const float bv = bias[node + a];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[a][j] = bv;
    }
    if (same) {
        int i = 0;
        for (; i + 2 <= len; i += 2) {
            float v0[VEC], v1[VEC];
            ov_ldv<VEC>(act + (int64_t)colidx[k0[0] + i] * ldb + b0, v0);
            ov_ldv<VEC>(act + (int64_t)colidx[k0[0] + i + 1] * ldb + b0, v1);
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                const float w0 = vals[k0[a] + i], w1 = vals[k0[a] + i + 1];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    acc[a][j] += w0 * v0[j];
                    acc[a][j] += w1 * v1[j];
                }
            }
        }
        if (i < len) {
            float v0[VEC];
            ov_ldv<VEC>(act + (int64_t)colidx[k0[0] + i] * ldb + b0, v0);
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                const float w0 = vals[k0[a] + i];
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[a][j] += w0 * v0[j];
            }
        }
    } else {
#pragma unroll
        for (int a = 0; a < NU; ++a)
            for (int k = k0[a]; k < rowptr[node + a + 1]; ++k) {
                float v0[VEC];
                const float w0 = vals[k];
                ov_ldv<VEC>(act + (int64_t)colidx[k] * ldb + b0, v0);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[a][j] += w0 * v0[j];
            }
    }
#pragma unroll
    for (int a = 0; a < NU; ++a) ov_stv<VEC>(act + (int64_t)(node + a) * ldb + b0, acc[a]);
    float* dst = fm_out + (int64_t)(c_base + term) * fm_ld + b0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        float m = 0.f;
#pragma unroll
        for (int a = 0; a < NU; ++a) m += acc[a][j];          // the order of ov_act_fm_levels_kernel
        if (b0 + j < B) dst[j] = m / (float)NU;
    }
}

// Dropout on the outputs of the term layers (OntoEncoder's [Linear, Dropout] blocks, modules.py:219-223; OntoDecoder has none):
// rate 0 = off.  The keep-scale of (node, batch column) comes from the Philox stream OV_RNG_DROP of (seed, rng_off); the forward
// level kernel multiplies a node's output by it, the backward level kernel the finished gradient row of the node, so that
// everything below (the children's gradients, the node's own weight gradient) sees the gradient at the Linear's output.
struct OvTermDrop { float rate; uint64_t seed, rng_off; };

// GENE: 0 = term level (write ACT), 1 = gene level training (D + loss partials), 2 = gene level
// inference (recon sample-major to rec_out)
template <int VEC, int GENE>
__global__ __launch_bounds__(256) void ov_dec_fwd_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                        const float* __restrict__ vals, const float* __restrict__ bias,
                                                        float* __restrict__ act, int node_begin, int node_end, int B, int ldb,
                                                        const float* __restrict__ x, const int32_t* __restrict__ xidx, int G,
                                                        float* __restrict__ d, float* __restrict__ rec_part,
                                                        float* __restrict__ rec_out, int64_t rec_ld, int C, const OvTermDrop dr) {
    const int node = node_begin + blockIdx.x * 4 + ov_wave_id();
    const int lane = ov_lane();
    const int b0 = blockIdx.y * (64 * VEC) + lane * VEC;
    float sq = 0.f;
    if (node < node_end) {
        float acc[VEC];
        const float bv = bias[node];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = bv;
        const int k0 = rowptr[node], k1 = rowptr[node + 1];
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
            const int c0 = colidx[k], c1 = colidx[k + 1], c2 = colidx[k + 2], c3 = colidx[k + 3];
            const float w0 = vals[k], w1 = vals[k + 1], w2 = vals[k + 2], w3 = vals[k + 3];
            ov_ldv<VEC>(act + (int64_t)c0 * ldb + b0, v0);
            ov_ldv<VEC>(act + (int64_t)c1 * ldb + b0, v1);
            ov_ldv<VEC>(act + (int64_t)c2 * ldb + b0, v2);
            ov_ldv<VEC>(act + (int64_t)c3 * ldb + b0, v3);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                acc[j] += w0 * v0[j];
                acc[j] += w1 * v1[j];
                acc[j] += w2 * v2[j];
                acc[j] += w3 * v3[j];
            }
        }
        for (; k < k1; ++k) {
            float v0[VEC];
            const int c0 = colidx[k];
            const float w0 = vals[k];
            ov_ldv<VEC>(act + (int64_t)c0 * ldb + b0, v0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += w0 * v0[j];
        }
        if (GENE == 0) {
            if (dr.rate > 0.f) {
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    acc[j] *= ov_dropout_scale(dr.rate, dr.seed, dr.rng_off, OV_RNG_DROP, (uint64_t)node * 0x100000000ull + (uint64_t)(b0 + j));
            }
            ov_stv<VEC>(act + (int64_t)node * ldb + b0, acc);
        } else if (GENE == 1) {
            const int g = node - C;
            float o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int b = b0 + j;
                float dv = 0.f;
                if (b < B) {
                    const int64_t r = xidx ? (int64_t)xidx[b] : (int64_t)b;
                    dv = acc[j] - x[r * G + g];
                    sq += dv * dv;
                }
                o[j] = 2.f * dv;
            }
            ov_stv<VEC>(d + (int64_t)node * ldb + b0, o);
        } else {
            const int g = node - C;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int b = b0 + j;
                if (b < B) rec_out[(int64_t)b * rec_ld + g] = acc[j];
            }
        }
    }
    if (GENE == 1) {
        __shared__ float red[4];
        sq = ov_wave_sum(sq);
        if (lane == 0) red[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) rec_part[blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}


// -------------------------------------------------------------------------------------------------
// Fused term levels.  The K-1 term layers are tiny (41 k non-zeros at GO-BP scale) and strictly
// sequential, so as separate launches they are pure launch + latency cost.  One workgroup takes BT
// batch columns, keeps the whole activation panel of those columns in LDS ([C][BT] floats, 144 KB
// at C = 9000, BT = 4) and walks the levels with a workgroup barrier in between; every level's
// rows are spread over the 1024 threads.  Results also go to the feature-major panel in HBM for
// the gene layer and the weight gradient.
// -------------------------------------------------------------------------------------------------
#define OV_MAX_LEVELS 64
// The chain as a flat list of steps that every wave walks, one packed word per step: in step q a wave owns slice
// base + wave if wave < count (count = 1 .. 16 slices of the step, bits 24-29; base = bits 0-23); a step that opens a new
// level is marked in bit 30 (workgroup barrier first).  The list lives in the kernel arguments, so a step's word is a scalar
// load with a memory round trip behind it: the kernel requests step q + 1's word while it works on step q (two dependent
// scalar loads at the head of every step, base and end in separate arrays, were 2 of the chain's 20 us at 128 columns).
#define OV_MAX_STEPS 480
struct OvSteps {
    int n;
    int last_slice;            // n_slices - 1: where requests beyond the schedule are clamped to
    int pk[OV_MAX_STEPS];
};
#define OV_STEP_NEW_LEVEL (1 << 30)
#define OV_STEP_BASE(w) ((w) & 0xFFFFFF)
#define OV_STEP_COUNT(w) (((w) >> 24) & 63)
// false if the schedule does not fit (the caller then launches level by level)
static bool ov_make_steps(const ontovae_sell_t& f, OvSteps* st) {
    st->n = 0;
    st->last_slice = std::max(f.n_slices - 1, 0);
    for (int li = 0; li < f.n_levels; ++li) {
        const int first = f.level_first[li], count = f.level_count[li];
        for (int j = 0; 16 * j < count; ++j) {
            if (st->n >= OV_MAX_STEPS) return false;
            if (first + 16 * j > 0xFFFFFF) return false;
            st->pk[st->n] = (first + 16 * j) | (std::min(16, count - 16 * j) << 24) | ((j == 0 && st->n > 0) ? OV_STEP_NEW_LEVEL : 0);
            ++st->n;
        }
    }
    return true;
}

// sum over aligned groups of 2^LG lanes; the LAST lane of every group ends up with the group total
template <int LG>
__device__ __forceinline__ float ov_group_sum(float v) {
    if (LG >= 1) v += ov_dpp<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    if (LG >= 2) v += ov_dpp<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    if (LG >= 3) v += ov_dpp<0x141, 0xF>(v);    // row_half_mirror
    if (LG >= 4) v += ov_dpp<0x140, 0xF>(v);    // row_mirror
    if (LG >= 5) v += ov_dpp<0x142, 0xA>(v);    // row_bcast15: rows 1, 3 hold 32-lane sums
    if (LG >= 6) v += ov_dpp<0x143, 0xC>(v);    // row_bcast31: row 3 holds the wave sum
    return v;
}

template <int BT>
__device__ __forceinline__ void ov_group_sum_rt(float (&acc)[BT], int lg) {
#define OV_GS_CASE(N) case N: _Pragma("unroll") for (int j = 0; j < BT; ++j) acc[j] = ov_group_sum<N>(acc[j]); break;
    switch (lg) {                          // wave-uniform; ONE branch tree for the BT columns
        case 0: break;
        OV_GS_CASE(1) OV_GS_CASE(2) OV_GS_CASE(3) OV_GS_CASE(4) OV_GS_CASE(5)
        default: _Pragma("unroll") for (int j = 0; j < BT; ++j) acc[j] = ov_group_sum<6>(acc[j]); break;
    }
#undef OV_GS_CASE
}

// Refresh of the SELL-ordered weight copies from the parameter buffer (start of every pass).
__global__ __launch_bounds__(256) void ov_sell_refresh_kernel(const OvRefresh rf) {
    ov_sell_refresh_part(rf, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// -------------------------------------------------------------------------------------------------
// The term-level chain, forward or backward, as ONE kernel.
//
// The K-1 term layers are tiny (41 k non-zeros at GO-BP scale) and strictly sequential, so as
// separate launches they are pure launch + latency cost.  One workgroup of 16 waves takes BT batch
// columns and keeps the whole activation panel of those columns in LDS ([C][BT] fp32: 144 KB at
// C = 9000, BT = 4).  Levels are walked with a workgroup barrier in between; inside a level every
// wave takes slices of the level-scheduled SELL-64 structure (ontovae_sell_t): per slice ONE round of
// contiguous loads (lane meta, 8 gather indices, 8 weights), 8 LDS gathers, a DPP sum across the
// lanes that share a node, one store to LDS and one to the feature-major matrix in HBM.
//   FWD: out = bias + sum w * panel[idx]      BWD: out = D (gene part, from HBM) + sum w * dpanel[idx]
// -------------------------------------------------------------------------------------------------
#ifndef OV_CHAIN_DEPTH
#define OV_CHAIN_DEPTH 1     // records in flight per wave beyond the current one (19 registers each; 16 waves per CU leave 128 per lane).
                             // Measured, forward + backward chain: 47.9 us at depth 3, 46.3 at 2, 44.9 at 1, 48.9 at 4 (128 columns); 70.4 / 69.4 /
                             // 68.9 / 72.2 at 1024 columns: the records are L2 hits, and the registers of a deeper ring cost more than its latency cover buys
#endif
struct OvRec {          // chunk 0 of one lane of a slice
    int node, info;
    int id[8];
    float w[8];
    float bias;
};

template <bool BWD>
__device__ __forceinline__ void ov_rec_load(OvRec& r, int sl, int lane, const int32_t* __restrict__ snode,
                                            const int32_t* __restrict__ sinfo, const int32_t* __restrict__ chunk0,
                                            const int32_t* __restrict__ sidx, const float* __restrict__ sw,
                                            const float* __restrict__ sbias, int single) {
    const int c0 = single ? sl : chunk0[sl];
    const int base = c0 * 512 + lane;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#ifdef OV_CHAIN_HALF
        if (e >= 4) { r.id[e] = r.id[e - 4]; r.w[e] = r.w[e - 4]; continue; }
#endif
        r.id[e] = sidx[base + e * 64];
        r.w[e] = sw[base + e * 64];
    }
    r.node = snode[sl * 64 + lane];
    r.info = sinfo[sl * 64 + lane];
    r.bias = BWD ? 0.f : sbias[sl * 64 + lane];
}

struct OvChainArgs {
    const int32_t* snode; const int32_t* sinfo; const int32_t* chunk0; const int32_t* sidx;
    const float* sw; const float* sbias;
    float* glob; const float* inblk;
    int ldb, in0, in1, out0, out1, dbg;
    int nrep;      // workgroups per column group (see ov_chain_replicas)
};
template <int BT, bool BWD, bool SINGLE>
__device__ __forceinline__ void ov_dec_chain_body(const OvChainArgs& ca, const OvSteps& st, const int wg) {
    // Workgroup -> (column group, replica).  A column group's panel lives in ONE workgroup's LDS, and handing its C rows to
    // the feature-major matrix in HBM is C separate 16-B writes from that CU, about 1.7 clocks each: 7.4 of the kernel's 30 us
    // at C = 9000, whatever the batch size.  While the launch leaves CUs idle (32 column groups at batch 128), nrep workgroups
    // walk the SAME chain for a column group (the records come from the L2 they share, the arithmetic is free) and each
    // writes only the rows of every nrep-th slice.  Ids are dealt round-robin to the 8 XCDs; an XCD's workgroups take adjacent
    // column groups and all replicas of a group sit on one XCD.  ldb / BT is a multiple of 8.
    const int gpx = ca.ldb / BT / 8;                    // column groups per XCD
    const int block = (wg & 7) * gpx + (wg >> 3) % gpx;
    const int rep = (wg >> 3) / gpx;
    const int32_t* __restrict__ snode = ca.snode;
    const int32_t* __restrict__ sinfo = ca.sinfo;
    const int32_t* __restrict__ chunk0 = ca.chunk0;
    const int32_t* __restrict__ sidx = ca.sidx;
    const float* __restrict__ sw = ca.sw;
    const float* __restrict__ sbias = ca.sbias;
    float* __restrict__ glob = ca.glob;
    const float* __restrict__ inblk = ca.inblk;
    const int ldb = ca.ldb, in0 = ca.in0, in1 = ca.in1, out0 = ca.out0, out1 = ca.out1, dbg = OV_DBG(ca.dbg);
    extern __shared__ __attribute__((aligned(16))) float ov_panel[];
    const int b0 = block * BT;
    const int lane = ov_lane();
    const int wave = ov_wave_id();

    // Records are requested OV_CHAIN_DEPTH steps ahead, across level boundaries: a step's own work (a few LDS
    // gathers) is far shorter than the memory round trip of its record.  Every request is unconditional (a
    // wave without a slice in some step, or past the end of the schedule, re-reads a clamped slice) so that
    // the number of loads between a request and its use is the same on every path and the waits stay exact.
    static_assert(OV_CHAIN_DEPTH == 1, "the step loop below keeps one record in flight");
    auto request = [&](OvRec& r, int word) {      // word: the packed step the record is for
        const int sl = min(OV_STEP_BASE(word) + min(wave, OV_STEP_COUNT(word) - 1), st.last_slice);
        ov_rec_load<BWD>(r, sl, lane, snode, sinfo, chunk0, sidx, sw, sbias, SINGLE ? 1 : 0);
    };
    OvRec rec;
    int next_word = st.pk[0];
    request(rec, next_word);

    // prologue: the rows the chain starts from, HBM -> LDS.  FWD: level 0 (z).  BWD: every panel row
    // (the gene layer's contribution, written by the gene backward kernel).
    // inblk (BT == 4 only): the same rows in column-group-major order, one contiguous piece per workgroup
    const float* src = (BT == 4 && inblk) ? inblk + (int64_t)block * out1 * 4 : glob + b0;
    const int64_t src_ld = (BT == 4 && inblk) ? 4 : ldb;
    for (int p = in0 + threadIdx.x; p < in1; p += 8 * 1024) {
        float v[8][BT];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p + u * 1024 < in1) ov_ldv<BT>(src + (int64_t)(p + u * 1024) * src_ld, v[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p + u * 1024 < in1) ov_stv<BT>(ov_panel + (int64_t)(p + u * 1024) * BT, v[u]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int q = 0; q < st.n; ++q) {
        {
            {
                const int bq = next_word;
                next_word = st.pk[min(q + 1, st.n - 1)];           // scalar load; consumed by the request at the end of this step
                if ((bq & OV_STEP_NEW_LEVEL) && !(dbg & 4)) {
                    // LDS writes of the level must land before the next level reads them
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                const int sl = OV_STEP_BASE(bq) + wave;
                if (wave < OV_STEP_COUNT(bq) && !(dbg & 2)) {      // uniform over the wave
                    const OvRec& cur = rec;
                    const int cnt = cur.info & 0xFFFF;
                    const int lg = __builtin_amdgcn_readfirstlane((cur.info >> 16) & 7);
                    // all 8 entries of the chunk whatever the slice's width (padding slots: weight 0, a row the slice reads anyway): the
                    // eight gathers and FMAs behind wave-uniform `e < width` branches were slower than the LDS reads they saved
                    // (forward + backward chain 45.2 -> 41.9 us at 128 columns, 68.8 -> 65.3 at 1024)
                    constexpr int width = 8;
                    const bool writer = (cur.info >> 24) & 1;
                    float acc[BT];
#pragma unroll
                    for (int j = 0; j < BT; ++j) acc[j] = 0.f;
                    {
                        // padding slots have weight 0 and point at a row this slice really reads: plain FMAs
                        float v[8][BT];
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e < width) ov_ldv<BT>(ov_panel + (int64_t)cur.id[e] * BT, v[e]);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e < width) {
#pragma unroll
                                for (int j = 0; j < BT; ++j) acc[j] += cur.w[e] * v[e][j];
                            }
                    }
                    if (!SINGLE) {          // lanes with more than 8 entries: further chunks, loaded on demand
                        const int c0 = chunk0[sl], R = chunk0[sl + 1] - c0;
                        for (int r = 1; r < R; ++r) {
                            const int base = (c0 + r) * 512 + lane;
                            const int rem = cnt - r * 8;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float v0[BT];
                                const float w0 = sw[base + e * 64];
                                ov_ldv<BT>(ov_panel + (int64_t)sidx[base + e * 64] * BT, v0);
#pragma unroll
                                for (int j = 0; j < BT; ++j) acc[j] += e < rem ? w0 * v0[j] : 0.f;
                            }
                        }
                    }
                    ov_group_sum_rt<BT>(acc, lg);
                    if (writer) {
                        if (BWD) {
                            float g0[BT];
                            ov_ldv<BT>(ov_panel + (int64_t)cur.node * BT, g0);
#pragma unroll
                            for (int j = 0; j < BT; ++j) acc[j] += g0[j];
                        } else {
#pragma unroll
                            for (int j = 0; j < BT; ++j) acc[j] += cur.bias;
                        }
                        ov_stv<BT>(ov_panel + (int64_t)cur.node * BT, acc);
                        // and to the feature-major matrix in HBM, for the gene layer and the weight gradient: issued here, the
                        // 16-B pieces drain under the rest of the chain (as one pass at the end they were 7.5 of the kernel's 28 us)
                        if (!(dbg & 8) && (sl & (ca.nrep - 1)) == rep) ov_stv<BT>(glob + (int64_t)cur.node * ldb + b0, acc);
                    }
                }
                if (!(dbg & 1)) request(rec, next_word);     // the next step's record
            }
        }
    }
    // No epilogue: every row the chain produces went to HBM when its writer lane stored it.  (Backward: rows without term
    // children have no writer; the gene backward kernel stores those feature-major itself.)
    (void)out0; (void)out1;
}

template <int BT, bool BWD, bool SINGLE>
__global__ __launch_bounds__(1024) void ov_dec_chain_kernel(const OvChainArgs ca, const OvSteps st) {
    ov_dec_chain_body<BT, BWD, SINGLE>(ca, st, blockIdx.x);
}
// forward chain + riders: workgroups beyond nchain refresh the CSR-ordered copy of the gene layer's weights, which the next
// launch (the gene layer) reads and nothing in this one does
template <int BT, bool SINGLE>
__global__ __launch_bounds__(1024) void ov_chain_fwd_refresh_kernel(const OvChainArgs ca, const OvSteps st, int nchain, const OvRefresh rf) {
    if ((int)blockIdx.x < nchain) { ov_dec_chain_body<BT, false, SINGLE>(ca, st, blockIdx.x); return; }
    ov_sell_refresh_part(rf, ((int)blockIdx.x - nchain) * 1024 + threadIdx.x, ((int)gridDim.x - nchain) * 1024);
}

// Cooperative entry prefetch for the row-per-wave gather kernels: the 64 lanes load up to 64
// (index, weight) pairs of the row in one coalesced pass; v_readlane then hands them out as scalars,
// so the row gathers that follow have no address dependency left and go out back to back.
#define OV_RL_I(v, l) __builtin_amdgcn_readlane((v), (l))
#define OV_RL_F(v, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (v)), (l)))

// acc[VEC] += sum over entries [k0, k1) of w * src[row(entry) * ldb + boff ..]
#define OV_GATHER_U 16      // row loads in flight per lane (64 % OV_GATHER_U == 0)
template <int VEC, bool INDIRECT>
__device__ __forceinline__ void ov_gather_rows_acc(float (&acc)[VEC], int k0, int k1, const int32_t* __restrict__ rowidx,
                                                   const int32_t* __restrict__ kidx, const float* __restrict__ vals,
                                                   const float* __restrict__ src, int ldb, int boff, int lane) {
    for (int base = k0; base < k1; base += 64) {
        const int cnt = min(64, k1 - base);
        // lanes >= cnt repeat the last entry's row with weight 0: padding that only touches a row the
        // list really names (never memory that another wave may be writing)
        const int my_r = rowidx[base + min(lane, cnt - 1)];
        float my_w = 0.f;
        if (lane < cnt) my_w = INDIRECT ? vals[kidx[base + lane]] : vals[base + lane];
        for (int u = 0; u < cnt; u += OV_GATHER_U) {
            float v[OV_GATHER_U][VEC], w[OV_GATHER_U];
#pragma unroll
            for (int t = 0; t < OV_GATHER_U; ++t) {
                const int r = OV_RL_I(my_r, u + t);
                w[t] = OV_RL_F(my_w, u + t);
                ov_ldv<VEC>(src + (int64_t)r * ldb + boff, v[t]);
            }
#pragma unroll
            for (int t = 0; t < OV_GATHER_U; ++t)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += w[t] * v[t][j];
        }
    }
}

// Gene part of the backward: for every panel node p, D[p, :] = sum over the genes g of column p of
// W[g, p] * D[g, :].  One wave per (node, 64*VEC columns).
template <int VEC>
__global__ __launch_bounds__(256) void ov_dec_bwd_gene_kernel(const int32_t* __restrict__ cscptr, const int32_t* __restrict__ csc_split,
                                                             const int32_t* __restrict__ csc_row, const int32_t* __restrict__ csc_k,
                                                             const float* __restrict__ vals, float* __restrict__ d, int C, int ldb) {
    const int node = blockIdx.x * 4 + ov_wave_id();
    if (node >= C) return;
    const int lane = ov_lane();
    const int b0 = blockIdx.y * (64 * VEC) + lane * VEC;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    ov_gather_rows_acc<VEC, true>(acc, csc_split[node], cscptr[node + 1], csc_row, csc_k, vals, d, ldb, b0, lane);
    ov_stv<VEC>(d + (int64_t)node * ldb + b0, acc);
}

// Gene layer, tiled: a workgroup of 8 waves takes 32 genes x (64*VEC) batch columns.  The x tile is
// read in full 128-B row segments (x is sample-major) and transposed through LDS, so the ELBO
// epilogue reads x feature-major like everything else; each wave then runs 4 gene rows.
// MODE 1: training (D = 2 (recon - x), partial sums of squares); MODE 2: inference (recon is
// transposed back through the same LDS tile and stored sample-major in 128-B segments).
#define OV_GT_GENES 32
template <int VEC, int MODE>
__global__ __launch_bounds__(512) void ov_dec_gene_tile_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                             const float* __restrict__ vals, const float* __restrict__ bias,
                                                             const float* __restrict__ act, int B, int ldb,
                                                             const float* __restrict__ x, const int32_t* __restrict__ xidx, int G,
                                                             float* __restrict__ d, float* __restrict__ rec_part,
                                                             float* __restrict__ rec_out, int64_t rec_ld, int C) {
    constexpr int NB = 64 * VEC;      // batch columns of this tile
    constexpr int LD = NB + 1;        // odd row length: conflict-free transposed writes
    constexpr int Q = OV_GT_GENES / 4;
    __shared__ float xs[OV_GT_GENES * LD];
    __shared__ float red[8];
    const int g0 = blockIdx.x * OV_GT_GENES;
    const int bt0 = blockIdx.y * NB;
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    if (MODE == 1) {
        for (int u = threadIdx.x; u < NB * Q; u += 512) {
            const int b = u / Q, j = u % Q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (bt0 + b < B && g0 + 4 * j < G) {
                const int64_t r = xidx ? (int64_t)xidx[bt0 + b] : (int64_t)(bt0 + b);
                v = *reinterpret_cast<const f32x4*>(x + r * G + g0 + 4 * j);
            }
            xs[(4 * j + 0) * LD + b] = v.x;
            xs[(4 * j + 1) * LD + b] = v.y;
            xs[(4 * j + 2) * LD + b] = v.z;
            xs[(4 * j + 3) * LD + b] = v.w;
        }
        __syncthreads();
    }
    float sq = 0.f;
    const int bl = lane * VEC;
    for (int gl = wave; gl < OV_GT_GENES; gl += 8) {
        const int g = g0 + gl;
        if (g >= G) break;                       // wave-uniform
        const int node = C + g;
        float acc[VEC];
        const float bv = bias[node];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = bv;
        ov_gather_rows_acc<VEC, false>(acc, rowptr[node], rowptr[node + 1], colidx, nullptr, vals, act, ldb, bt0 + bl, lane);
        if (MODE == 1) {
            float o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float dv = 0.f;
                if (bt0 + bl + j < B) {
                    dv = acc[j] - xs[gl * LD + bl + j];
                    sq += dv * dv;
                }
                o[j] = 2.f * dv;
            }
            ov_stv<VEC>(d + (int64_t)node * ldb + bt0 + bl, o);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) xs[gl * LD + bl + j] = acc[j];
        }
    }
    if (MODE == 1) {
        sq = ov_wave_sum(sq);
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int i = 0; i < 8; ++i) t += red[i];
            rec_part[blockIdx.y * gridDim.x + blockIdx.x] = t;
        }
    } else {
        __syncthreads();
        for (int u = threadIdx.x; u < NB * Q; u += 512) {
            const int b = u / Q, j = u % Q;
            if (bt0 + b < B && g0 + 4 * j < G) {
                f32x4 v;
                v.x = xs[(4 * j + 0) * LD + b];
                v.y = xs[(4 * j + 1) * LD + b];
                v.z = xs[(4 * j + 2) * LD + b];
                v.w = xs[(4 * j + 3) * LD + b];
                *reinterpret_cast<f32x4*>(rec_out + (int64_t)(bt0 + b) * rec_ld + g0 + 4 * j) = v;
            }
        }
    }
}

// =================================================================================================
// Column-block kernels of the gene layer.
//
// The gathers of the gene layer (forward: 24 panel rows per gene; backward: ~55 gene rows per panel node)
// move 512 B per non-zero through the L2 when a wave covers all 128 batch columns, and their working
// sets (activation panel 4.6 MB, gene rows of D 10 MB at GO-BP scale) do not stay in one XCD's 4 MB L2:
// measured hit rates were 40-68 % and the kernels ran at the pace of the fabric behind the L2.
// Here the batch is cut into blocks of 32 columns (one 128-B line per row) and every block is given to
// fixed XCDs (workgroup ids are dealt round-robin to the 8 XCDs), so an XCD's L2 only ever sees its own
// 32 columns of the panel (1.15 MB) and of D (2.5 MB).
//
// A wave is 8 lane groups of 8 lanes; lane (grp = lane >> 3, s = lane & 7) owns columns col0 + 4 s .. + 3
// of the row its group is working on, so one wave instruction gathers 8 different rows, 128 B each.
// The 64 (index, weight) pairs of a pass are loaded coalesced in the permuted order e = 8 s + grp, which
// puts the entry of group grp at step i into lane (grp, i): a ds_swizzle broadcast within the group
// hands it out.  The 8 group partials are summed at the end (swizzle xor 8, 16, bpermute xor 32).
// =================================================================================================
#ifndef OV_GENE_FWD_U
#define OV_GENE_FWD_U 1     // batches of 8 entries per lane group in flight, forward (lists of ~24)
#endif
#ifndef OV_GENE_BWD_U
#define OV_GENE_BWD_U 1     // backward (lists of ~55)
#endif
struct OvCbMap { int cb, tile; };
// workgroup id -> (column block, row tile); ncb is 1, 2, 4, 8 or a multiple of 8
__device__ __forceinline__ OvCbMap ov_cb_map(int id, int ncb, int ntiles) {
    const int xcd = id & 7, q = id >> 3;
    OvCbMap m;
    if (ncb <= 8) {
        const int xpc = 8 / ncb;              // XCDs per column block
        m.cb = xcd % ncb;
        m.tile = q * xpc + xcd / ncb;
    } else {
        // several column blocks per XCD, one after the other (block-major): at any time an XCD's L2 holds the rows of ONE
        // block (1.15 MB of panel, 2.5 MB of D at GO-BP scale), whatever the batch size.  Tile-major order made the
        // working set grow with the batch (4.6 MB at 1024 columns, 37 MB at 8192) and the gathers run at fabric pace.
        m.cb = xcd + 8 * (q / ntiles);
        m.tile = q % ntiles;
    }
    return m;
}
static int ov_cb_grid(int ncb, int ntiles) {
    if (ncb <= 8) { const int xpc = 8 / ncb; return 8 * ((ntiles + xpc - 1) / xpc); }
    return 8 * ntiles * (ncb >> 3);
}

template <int I>
__device__ __forceinline__ int ov_grp_bcast(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x18 | (I << 5)); }
template <int I>
__device__ __forceinline__ float ov_grp_bcast(float v) { return __builtin_bit_cast(float, ov_grp_bcast<I>(__builtin_bit_cast(int, v))); }

// sum of the 8 lane groups, per s; every lane ends with the total of its s
__device__ __forceinline__ float ov_grp_reduce(float v, int lane) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x1f | (8 << 10)));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x1f | (16 << 10)));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), __builtin_bit_cast(int, v)));
    return v;
}

// acc (this group's partial) += sum over its share of entries [k0, k1) of w * src[row(entry) * ldb + 0..3];
// srccol = src + col0 + 4 s.  k0, k1 are wave-uniform.
template <bool INDIRECT>
__device__ __forceinline__ void ov_gather_cb(f32x4& acc, int k0, int k1, const int32_t* __restrict__ rowidx,
                                             const int32_t* __restrict__ kidx, const float* __restrict__ vals,
                                             const float* __restrict__ srccol, int ldb, int lane) {
    const int e = 8 * (lane & 7) + (lane >> 3);
    for (int base = k0; base < k1; base += 64) {
        const int cnt = min(64, k1 - base);
        // lanes beyond cnt repeat the last entry's row with weight 0 (a row the list really names)
        const int my_r = rowidx[base + min(e, cnt - 1)];
        float my_w = 0.f;
        if (e < cnt) my_w = INDIRECT ? vals[kidx[base + e]] : vals[base + e];
        const int nsteps = (cnt + 7) >> 3;
        f32x4 v[8];
        float w[8];
#define OV_CB_STEP(I)                                                                                     \
        if (I < nsteps) {                                                                                 \
            const int r = ov_grp_bcast<I>(my_r);                                                          \
            w[I] = ov_grp_bcast<I>(my_w);                                                                 \
            v[I] = *reinterpret_cast<const f32x4*>(srccol + (int64_t)r * ldb);                            \
        }
        OV_CB_STEP(0) OV_CB_STEP(1) OV_CB_STEP(2) OV_CB_STEP(3) OV_CB_STEP(4) OV_CB_STEP(5) OV_CB_STEP(6) OV_CB_STEP(7)
#undef OV_CB_STEP
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nsteps) {
                acc.x += w[i] * v[i].x; acc.y += w[i] * v[i].y; acc.z += w[i] * v[i].z; acc.w += w[i] * v[i].w;
            }
    }
}

// One row per lane group: acc += sum over entries [k0, k0 + len) of w * src[row(entry) * ldb + 0..3], where k0 and len
// are per group.  Batches of 8 entries per group: the group's 8 lanes load 8 consecutive (index, weight) pairs, the
// next batch is requested before the current one's row gathers, and a ds_swizzle broadcast within the group hands
// entry i to all 8 lanes.  Groups whose list is shorter than the wave's longest re-read their last row with weight 0.
template <bool INDIRECT, int U>
__device__ __forceinline__ void ov_gather_grp(f32x4& acc, int k0, int len, const int32_t* __restrict__ rowidx,
                                              const int32_t* __restrict__ kidx, const float* __restrict__ vals,
                                              const float* __restrict__ srccol, int ldb, int lane, const int r_empty = 0) {
    // r_empty: a row of src that is always readable and finite, gathered with weight 0 by a group whose list is empty
    // (forward: panel row 0; backward: the first gene row of D -- the panel rows of D are what the kernel is writing)
    const int s = lane & 7;
    int maxlen = len;                    // wave maximum of len
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    auto fetch = [&](int e, int& r, float& w) {
        // entry e of this group's list; beyond the list: the last row (or r_empty for an empty list), weight 0
        const int ee = max(min(e, len - 1), 0);
        r = len > 0 ? rowidx[k0 + ee] : r_empty;
        w = 0.f;
        if (e < len) w = INDIRECT ? vals[kidx[k0 + e]] : vals[k0 + e];
    };
    int cur_r[U], nxt_r[U];
    float cur_w[U], nxt_w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { fetch(8 * u + s, cur_r[u], cur_w[u]); nxt_r[u] = 0; nxt_w[u] = 0.f; }
    for (int b0 = 0; b0 < maxlen; b0 += 8 * U) {
        if (b0 + 8 * U < maxlen) {
#pragma unroll
            for (int u = 0; u < U; ++u) fetch(b0 + 8 * U + 8 * u + s, nxt_r[u], nxt_w[u]);
        }
        f32x4 v[U][8];
        float w[U][8];
#define OV_GRP_STEP(I)                                                                                    \
        {                                                                                                 \
            const int r = ov_grp_bcast<I>(cur_r[u]);                                                      \
            w[u][I] = ov_grp_bcast<I>(cur_w[u]);                                                          \
            v[u][I] = *reinterpret_cast<const f32x4*>(srccol + (int64_t)r * ldb);                         \
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u == 0 || b0 + 8 * u < maxlen) {
                OV_GRP_STEP(0) OV_GRP_STEP(1) OV_GRP_STEP(2) OV_GRP_STEP(3) OV_GRP_STEP(4) OV_GRP_STEP(5) OV_GRP_STEP(6) OV_GRP_STEP(7)
            }
        }
#undef OV_GRP_STEP
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u == 0 || b0 + 8 * u < maxlen) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc.x += w[u][i] * v[u][i].x; acc.y += w[u][i] * v[u][i].y;
                    acc.z += w[u][i] * v[u][i].z; acc.w += w[u][i] * v[u][i].w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { cur_r[u] = nxt_r[u]; cur_w[u] = nxt_w[u]; }
    }
}

// ov_gather_grp<true, 1> for the gene part of the backward, with the weight gradient of the same links in the same pass.
// The group works on panel node p; its list names the genes g below p (rowidx = node of g, kidx = position k of the
// link in the packed weights).  Each gathered row D[g, cols] is used twice while it is in registers:
//   acc  += w[k] * D[g, cols]                          (data gradient of p, as before)
//   part[k] = sum over the block's columns of ACT[p, cols] * D[g, cols]    (weight gradient partial of link k)
// so the weight gradient of the gene layer needs no gathers of its own (it used to re-read one ACT row per link in the
// launches of the backward chain and of the latent backward).  The dot is reduced over the group's 8 lanes with 3 DPP
// adds; lane s keeps entry s of the batch of 8 and stores it at its link's packed position, which ascends by 1 along the
// list (ontovae_plan_row_k): the group's 8 lanes write 32 contiguous bytes, a node's list one contiguous run, and the
// weights are read the same way.
__device__ __forceinline__ void ov_gather_grp_wgrad(f32x4& acc, int k0, int len, const int32_t* __restrict__ rowidx,
                                                    const int32_t* __restrict__ kidx, const float* __restrict__ vals,
                                                    const float* __restrict__ srccol, int ldb, int lane, const f32x4 aseg,
                                                    float* __restrict__ part, const int kb_known, const int r_empty) {
    const int s = lane & 7;
    int maxlen = len;                    // wave maximum of len
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    // packed position of the list's first link; it ascends by 1 along the list (ontovae_plan_row_k), so the weight of entry e
    // is one load at a known address, not a load behind the load of its position (25.2 -> 22.1 us at 128 columns)
    const int kb = kb_known >= 0 ? kb_known : (len > 0 ? kidx[k0] : 0);
    auto fetch = [&](int e, int& r, float& w, int& k) {
        const int ee = max(min(e, len - 1), 0);
        r = len > 0 ? rowidx[k0 + ee] : r_empty;
        w = 0.f;
        k = -1;
        if (e < len) {
            k = kb + e;
            w = vals[k];
        }
    };
    int cur_r, nxt_r = 0, cur_k, nxt_k = -1;
    float cur_w, nxt_w = 0.f;
    fetch(s, cur_r, cur_w, cur_k);
    for (int b0 = 0; b0 < maxlen; b0 += 8) {
        if (b0 + 8 < maxlen) fetch(b0 + 8 + s, nxt_r, nxt_w, nxt_k);
        f32x4 v[8];
        float w[8];
#define OV_GW_STEP(I)                                                                                     \
        {                                                                                                 \
            const int r = ov_grp_bcast<I>(cur_r);                                                         \
            w[I] = ov_grp_bcast<I>(cur_w);                                                                \
            v[I] = *reinterpret_cast<const f32x4*>(srccol + (int64_t)r * ldb);                            \
        }
        OV_GW_STEP(0) OV_GW_STEP(1) OV_GW_STEP(2) OV_GW_STEP(3) OV_GW_STEP(4) OV_GW_STEP(5) OV_GW_STEP(6) OV_GW_STEP(7)
#undef OV_GW_STEP
        float mine = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc.x += w[i] * v[i].x; acc.y += w[i] * v[i].y;
            acc.z += w[i] * v[i].z; acc.w += w[i] * v[i].w;
            float t = v[i].x * aseg.x;
            t += v[i].y * aseg.y;
            t += v[i].z * aseg.z;
            t += v[i].w * aseg.w;
            t = ov_group_sum<3>(t);
            mine = s == i ? t : mine;
        }
        if (cur_k >= 0) part[cur_k] = mine;
        cur_r = nxt_r; cur_w = nxt_w; cur_k = nxt_k;
    }
}

// The 32 rows of a tile are dealt to the lane groups in order of decreasing list length (rank by counting in
// LDS), so that the 8 groups of a wave walk lists of similar length.  Call with all threads of the workgroup (tid is
// the thread's number inside its 256-thread tile team; every team has its own three arrays); returns the
// tile-local row of this thread's group and its list (k0, len).
__device__ __forceinline__ int ov_tile_balance(int my_k0, int my_len, int* sh_k0, int* sh_len, int* sh_ord, int& k0, int& len,
                                               int tid = threadIdx.x) {
    if (tid < 32) { sh_k0[tid] = my_k0; sh_len[tid] = my_len; }
    __syncthreads();
    if (tid < 32) {
        int rank = 0;
        for (int j = 0; j < 32; ++j) rank += (sh_len[j] > my_len || (sh_len[j] == my_len && j < tid)) ? 1 : 0;
        sh_ord[rank] = tid;
    }
    __syncthreads();
    const int row = sh_ord[tid >> 3];
    k0 = sh_k0[row];
    len = sh_len[row];
    return row;
}

// Gene layer forward on a (32 genes) x (32 columns) tile: 4 waves, one gene per lane group (8 genes per wave).
// MODE 1: training (D = 2 (recon - x), partial sums of squares); MODE 2: inference (recon stored sample-major).
// x (sample-major) and recon go through an LDS transpose so that both sides move whole 128-B segments.
template <int MODE>
__global__ __launch_bounds__(256) void ov_gene_cb_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                        const float* __restrict__ vals, const float* __restrict__ bias,
                                                        const float* __restrict__ act, int B, int ldb,
                                                        const float* __restrict__ x, const int32_t* __restrict__ xidx, int G,
                                                        float* __restrict__ d, float* __restrict__ rec_part,
                                                        float* __restrict__ rec_out, int64_t rec_ld, int C, int ncb, int ntiles,
                                                        float* __restrict__ bpart, int bp_rows, int bp_row0,
                                                        const int4* __restrict__ tiles) {
    constexpr int TG = 32, LD = 33;
    __shared__ float xs[TG * LD];
    __shared__ float red[4];
    __shared__ int sh_k0[32], sh_len[32], sh_ord[32];
    const OvCbMap map = ov_cb_map(blockIdx.x, ncb, ntiles);
    if (map.tile >= ntiles) {                      // workgroup-uniform
        if (MODE == 1 && threadIdx.x == 0) rec_part[blockIdx.x] = 0.f;
        return;
    }
    const int g0 = map.tile * TG, col0 = map.cb * 32;
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    const int tid = threadIdx.x;
    const int s4 = 4 * (lane & 7);
    // the lists first: everything else waits on them
    int k0, len, gl;
    if (tiles) {                                   // the tile's rows ranked by list length, laid down by the plan
        const int4 ti = tiles[map.tile * 32 + (tid >> 3)];
        k0 = ti.x; len = ti.y; gl = ti.z - g0;
    } else {
        int my_k0 = 0, my_len = 0;
        if (tid < 32 && g0 + tid < G) { my_k0 = rowptr[C + g0 + tid]; my_len = rowptr[C + g0 + tid + 1] - my_k0; }
        gl = ov_tile_balance(my_k0, my_len, sh_k0, sh_len, sh_ord, k0, len);
    }
    const int g = g0 + gl;
    const int node = C + min(g, G - 1);
    if (MODE == 1) {
        const int b = tid >> 3, j = tid & 7;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (col0 + b < B && g0 + 4 * j < G) {
            const int64_t r = xidx ? (int64_t)xidx[col0 + b] : (int64_t)(col0 + b);
            v = *reinterpret_cast<const f32x4*>(x + r * G + g0 + 4 * j);
        }
        xs[(4 * j + 0) * LD + b] = v.x;
        xs[(4 * j + 1) * LD + b] = v.y;
        xs[(4 * j + 2) * LD + b] = v.z;
        xs[(4 * j + 3) * LD + b] = v.w;
    }
    const float bv = bias[node];
    f32x4 acc = {bv, bv, bv, bv};
    ov_gather_grp<false, OV_GENE_FWD_U>(acc, k0, len, colidx, nullptr, vals, act + col0 + s4, ldb, lane);
    if (MODE == 1) {
        __syncthreads();
        float sq = 0.f;
        if (g < G) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float dv = 0.f;
                if (col0 + s4 + c < B) {
                    dv = acc[c] - xs[gl * LD + s4 + c];
                    sq += dv * dv;
                }
                o[c] = 2.f * dv;
            }
            *reinterpret_cast<f32x4*>(d + (int64_t)node * ldb + col0 + s4) = o;
            if (bpart) {          // gradient of the gene's bias: sum of D[g, :] over the block's columns (same order as ov_wgrad_cb_body)
                float gb = (o.x + o.y) + (o.z + o.w);
                gb = ov_group_sum<3>(gb);
                if ((lane & 7) == 7) bpart[(int64_t)map.cb * bp_rows + bp_row0 + g] = gb;
            }
        }
        sq = ov_wave_sum(sq);
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) rec_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) xs[gl * LD + s4 + c] = acc[c];
        __syncthreads();
        const int b = tid >> 3, j = tid & 7;
        if (col0 + b < B && g0 + 4 * j < G) {
            f32x4 v;
            v.x = xs[(4 * j + 0) * LD + b];
            v.y = xs[(4 * j + 1) * LD + b];
            v.z = xs[(4 * j + 2) * LD + b];
            v.w = xs[(4 * j + 3) * LD + b];
            *reinterpret_cast<f32x4*>(rec_out + (int64_t)(col0 + b) * rec_ld + g0 + 4 * j) = v;
        }
    }
}

// Gene layer forward of a training pass against an LDS panel (batches of 512 columns and more).  The 32-column-block kernel
// above gathers one 128-B row piece per link from the L2: 2 requests of 64 B per link and block, 261 M per pass at 8192
// columns of GO-BP, and that request rate (0.2 T/s, three quarters of what the 128 L2 channels take) is what bounds it.  Here a
// workgroup keeps the 4 columns of ALL C panel rows in LDS (C x 16 B, as the chain kernels do), walks the gene lists in
// SELL-64 order (ontovae_gsell_t: one gene per lane, 4 links per 8-B + 16-B load, every load of a wave one contiguous piece)
// and takes the panel rows from LDS: the L2 then sees 6.9 B per link and column group instead of 16 B x 8 lanes per link and
// block, 0.64x the requests in all (the 16-B pieces of D and of the panel load each cost a whole request).
// Workgroup = (column group, chunk of slices); the 8 column groups of a 128-B line of D sit on one XCD in consecutive
// workgroups and walk the slices in the same fixed rotation, so that their 16-B pieces meet in that L2 (with slices handed
// out by a shared counter the HBM saw 1.6 GB of writes instead of 0.9 GB).  Every per-slice result is independent of the
// wave that computed it, and the sums of squares are combined in slice order.
// gb4 != NULL: [column group][lane of the SELL form] sums of D over the group's 4 columns, the gene bias gradient's
// partials (ov_wgrad_finish_body adds the groups in order).
struct OvGeneLds {
    const int32_t* gene; const int32_t* quad0; const uint2* col; const f32x4* w; int n_slices;
    const float* bias; const float* act; float* d; int B, ldb, C;
    const float* x; const int32_t* xidx; int64_t xld;
    float* rec_part; float* gb4; int nl;
    int ncg, spc, nunits;        // column groups (ldb / 4), slices per chunk, (chunk, 8 column groups) units
};
__global__ __launch_bounds__(1024) void ov_gene_lds_kernel(const OvGeneLds a) {
    extern __shared__ __attribute__((aligned(16))) float ov_panel[];
    const int tid = threadIdx.x, lane = ov_lane(), wave = ov_wave_id();
    const int wg = blockIdx.x, xcd = wg & 7, idx = wg >> 3;
    const int unit = (idx >> 3) * 8 + xcd;
    if (unit >= a.nunits) {                        // workgroup-uniform
        if (tid == 0) a.rec_part[wg] = 0.f;
        return;
    }
    const int opc = a.ncg >> 3;
    const int chunk = unit / opc, cg = (unit - chunk * opc) * 8 + (idx & 7);
    const int col0 = cg * 4;
    const int s0 = chunk * a.spc, s1 = min(s0 + a.spc, a.n_slices);
    const bool active = col0 < a.B;                // else a group of padding columns: zero rows of D, nothing else
    f32x4* panel = reinterpret_cast<f32x4*>(ov_panel);
    float* sq_s = ov_panel + (size_t)a.C * 4;      // [spc] sums of squares per slice
    if (active) {
        int n = tid;
        for (; n + 3072 < a.C; n += 4096) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(a.act + (int64_t)n * a.ldb + col0);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(a.act + (int64_t)(n + 1024) * a.ldb + col0);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(a.act + (int64_t)(n + 2048) * a.ldb + col0);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(a.act + (int64_t)(n + 3072) * a.ldb + col0);
            panel[n] = v0; panel[n + 1024] = v1; panel[n + 2048] = v2; panel[n + 3072] = v3;
        }
        for (; n < a.C; n += 1024) panel[n] = *reinterpret_cast<const f32x4*>(a.act + (int64_t)n * a.ldb + col0);
    }
    int64_t xr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = col0 + c;
        xr[c] = col < a.B ? (a.xidx ? (int64_t)a.xidx[col] : (int64_t)col) * a.xld : -1;
    }
    __syncthreads();
    int iter = 0;
    for (int sl = s0 + wave; sl < s1; ++iter, sl = s0 + 16 * iter + ((wave + iter) & 15)) {
        const int g = a.gene[sl * 64 + lane];
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        float sq = 0.f;
        if (active) {
            float xv[4] = {0.f, 0.f, 0.f, 0.f};
            float bv = 0.f;
            if (g >= 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (xr[c] >= 0) xv[c] = a.x[xr[c] + g];
                bv = a.bias[a.C + g];
            }
            const int q0 = __builtin_amdgcn_readfirstlane(a.quad0[sl]);
            const int nq = __builtin_amdgcn_readfirstlane(a.quad0[sl + 1]) - q0;
            const uint2* cp = a.col + (int64_t)q0 * 64 + lane;
            const f32x4* wp = a.w + (int64_t)q0 * 64 + lane;
            f32x4 acc = {bv, bv, bv, bv};
            auto fetch = [&](int i, uint2& c, f32x4& w) {
                if (i < nq) { c = cp[i * 64]; w = wp[i * 64]; }
                else { c = make_uint2(0u, 0u); w = f32x4{0.f, 0.f, 0.f, 0.f}; }
            };
            auto use = [&](const uint2 c, const f32x4 w) {
                const f32x4 p0 = panel[c.x & 0xffffu], p1 = panel[c.x >> 16], p2 = panel[c.y & 0xffffu], p3 = panel[c.y >> 16];
                acc += w.x * p0;
                acc += w.y * p1;
                acc += w.z * p2;
                acc += w.w * p3;
            };
            uint2 c0, c1, nc0, nc1;
            f32x4 w0, w1, nw0, nw1;
            fetch(0, c0, w0);
            fetch(1, c1, w1);
            for (int i = 0; i < nq; i += 2) {
                fetch(i + 2, nc0, nw0);
                fetch(i + 3, nc1, nw1);
                use(c0, w0);
                use(c1, w1);
                c0 = nc0; w0 = nw0; c1 = nc1; w1 = nw1;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (g >= 0 && xr[c] >= 0) {
                    const float dv = acc[c] - xv[c];
                    sq += dv * dv;
                    o[c] = 2.f * dv;
                }
        }
        if (g >= 0) *reinterpret_cast<f32x4*>(a.d + (int64_t)(a.C + g) * a.ldb + col0) = o;
        if (active && a.gb4) a.gb4[(int64_t)cg * a.nl + sl * 64 + lane] = (o.x + o.y) + (o.z + o.w);
        sq = ov_wave_sum(sq);
        if (lane == 0) sq_s[sl - s0] = sq;
    }
    __syncthreads();
    if (wave == 0) {
        float t = 0.f;
        for (int i = lane; i < s1 - s0; i += 64) t += sq_s[i];
        t = ov_wave_sum(t);
        if (lane == 0) a.rec_part[wg] = t;
    }
}

// Gene part of the backward on 32-column blocks: D[p, cols] = sum over the genes g of column p of W[g, p] * D[g, cols].
// 4 waves, 32 panel nodes per workgroup, one node per lane group.
// wpart != NULL: also the weight-gradient partials of the gene layer's links, part[cb][k] (ov_gather_grp_wgrad).
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void ov_bwd_gene_cb_kernel(const int32_t* __restrict__ cscptr, const int32_t* __restrict__ csc_split,
                                                            const int32_t* __restrict__ csc_row, const int32_t* __restrict__ csc_k,
                                                            const float* __restrict__ vals, float* __restrict__ d,
                                                            float* __restrict__ dblk, int C, int ldb, int ncb, int ntiles,
                                                            const float* __restrict__ act, float* __restrict__ wpart, int64_t nnz,
                                                            const int4* __restrict__ tiles) {
    __shared__ int sh_k0[32], sh_len[32], sh_ord[32];
    const OvCbMap map = ov_cb_map(blockIdx.x, ncb, ntiles);
    if (map.tile >= ntiles) return;
    const int lane = ov_lane();
    const int col0 = map.cb * 32, s4 = 4 * (lane & 7);
    int k0, len, node, kb = -1;
    if (tiles) {                                   // the tile's nodes ranked by list length, laid down by the plan
        const int4 ti = tiles[map.tile * 32 + (threadIdx.x >> 3)];
        k0 = ti.x; len = ti.y; node = ti.z; kb = ti.w;
    } else {
        int my_k0 = 0, my_len = 0;
        if (threadIdx.x < 32) {
            const int n = map.tile * 32 + threadIdx.x;
            if (n < C) { my_k0 = csc_split[n]; my_len = cscptr[n + 1] - my_k0; }
        }
        node = map.tile * 32 + ov_tile_balance(my_k0, my_len, sh_k0, sh_len, sh_ord, k0, len);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (wpart) {
        const f32x4 aseg = *reinterpret_cast<const f32x4*>(act + (int64_t)min(node, C - 1) * ldb + col0 + s4);
        ov_gather_grp_wgrad(acc, k0, len, csc_row, csc_k, vals, d + col0 + s4, ldb, lane, aseg, wpart + (int64_t)map.cb * nnz, kb, C);
    } else {
        ov_gather_grp<true, OV_GENE_BWD_U>(acc, k0, len, csc_row, csc_k, vals, d + col0 + s4, ldb, lane, C);
    }
    if (node < C) {
        // dblk: column-group-major [ldb / 4][C][4], what a workgroup of the backward chain loads in one piece
        // feature-major: final for the nodes the backward chain does not touch (no term children), input of the latent
        // backward and the weight gradient
        *reinterpret_cast<f32x4*>(d + (int64_t)node * ldb + col0 + s4) = acc;
        if (dblk) *reinterpret_cast<f32x4*>(dblk + ((int64_t)(map.cb * 8 + (lane & 7)) * C + node) * 4) = acc;
    }
}

// The same on tiles of 32 TERMS (ontovae_plan_tile_info which = 3): the NU neurons of a term have the same genes below them,
// so a lane group takes a term, gathers each gene's D row ONCE and uses it for the NU nodes: NU data-gradient accumulators,
// NU weight-gradient dots.  A third of the gathers of ov_bwd_gene_cb_kernel at nu = 3 (the gathers, 128 B per link and block
// through the L2, are what that kernel is made of).  Entry e of node nu t + j has the packed position kb + j len + e.
template <int NU>
__device__ __forceinline__ void ov_gather_term_wgrad(f32x4 (&acc)[NU], int k0, int len, const int32_t* __restrict__ rowidx,
                                                     const float* __restrict__ vals, const float* __restrict__ srccol, int ldb, int lane,
                                                     const f32x4 (&aseg)[NU], float* __restrict__ part, int kb, const int r_empty) {
    const int s = lane & 7;
    int maxlen = len;                    // wave maximum of len
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    auto fetch = [&](int e, int& r, float (&w)[NU], int& k) {
        const int ee = max(min(e, len - 1), 0);
        r = len > 0 ? rowidx[k0 + ee] : r_empty;
        k = -1;
#pragma unroll
        for (int j = 0; j < NU; ++j) w[j] = 0.f;
        if (e < len) {
            k = kb + e;
#pragma unroll
            for (int j = 0; j < NU; ++j) w[j] = vals[k + j * len];
        }
    };
    int cur_r, nxt_r = 0, cur_k, nxt_k = -1;
    float cur_w[NU], nxt_w[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) nxt_w[j] = 0.f;
    fetch(s, cur_r, cur_w, cur_k);
    for (int b0 = 0; b0 < maxlen; b0 += 8) {
        if (b0 + 8 < maxlen) fetch(b0 + 8 + s, nxt_r, nxt_w, nxt_k);
        f32x4 v[8];
#define OV_GT_STEP(I) v[I] = *reinterpret_cast<const f32x4*>(srccol + (int64_t)ov_grp_bcast<I>(cur_r) * ldb);
        OV_GT_STEP(0) OV_GT_STEP(1) OV_GT_STEP(2) OV_GT_STEP(3) OV_GT_STEP(4) OV_GT_STEP(5) OV_GT_STEP(6) OV_GT_STEP(7)
#undef OV_GT_STEP
        float mine[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) mine[j] = 0.f;
#define OV_GT_USE(I)                                                                                      \
        {                                                                                                 \
            _Pragma("unroll") for (int j = 0; j < NU; ++j) {                                              \
                const float w = ov_grp_bcast<I>(cur_w[j]);                                                \
                acc[j].x += w * v[I].x; acc[j].y += w * v[I].y;                                           \
                acc[j].z += w * v[I].z; acc[j].w += w * v[I].w;                                           \
                float t = v[I].x * aseg[j].x;                                                             \
                t += v[I].y * aseg[j].y;                                                                  \
                t += v[I].z * aseg[j].z;                                                                  \
                t += v[I].w * aseg[j].w;                                                                  \
                t = ov_group_sum<3>(t);                                                                   \
                mine[j] = s == I ? t : mine[j];                                                           \
            }                                                                                             \
        }
        OV_GT_USE(0) OV_GT_USE(1) OV_GT_USE(2) OV_GT_USE(3) OV_GT_USE(4) OV_GT_USE(5) OV_GT_USE(6) OV_GT_USE(7)
#undef OV_GT_USE
        if (cur_k >= 0) {
#pragma unroll
            for (int j = 0; j < NU; ++j) part[cur_k + j * len] = mine[j];
        }
        cur_r = nxt_r; cur_k = nxt_k;
#pragma unroll
        for (int j = 0; j < NU; ++j) cur_w[j] = nxt_w[j];
    }
}

template <int NU, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void ov_bwd_gene_term_kernel(
        const int32_t* __restrict__ csc_row, const float* __restrict__ vals, float* __restrict__ d, float* __restrict__ dblk, int C,
        int ldb, int ncb, int ntiles, const float* __restrict__ act, float* __restrict__ wpart, int64_t nnz,
        const int4* __restrict__ tiles) {
    const OvCbMap map = ov_cb_map(blockIdx.x, ncb, ntiles);
    if (map.tile >= ntiles) return;
    const int lane = ov_lane();
    const int col0 = map.cb * 32, s4 = 4 * (lane & 7);
    const int4 ti = tiles[map.tile * 32 + (threadIdx.x >> 3)];
    const int k0 = ti.x, len = ti.y, node = ti.z, kb = ti.w;
    f32x4 acc[NU], aseg[NU];
    const int n0 = min(node, C - NU);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        aseg[j] = *reinterpret_cast<const f32x4*>(act + (int64_t)(n0 + j) * ldb + col0 + s4);
    }
    ov_gather_term_wgrad<NU>(acc, k0, len, csc_row, vals, d + col0 + s4, ldb, lane, aseg, wpart + (int64_t)map.cb * nnz, kb, C);
    if (node < C) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            *reinterpret_cast<f32x4*>(d + (int64_t)(node + j) * ldb + col0 + s4) = acc[j];
            if (dblk) *reinterpret_cast<f32x4*>(dblk + ((int64_t)(map.cb * 8 + (lane & 7)) * C + node + j) * 4) = acc[j];
        }
    }
}

// The term-tile kernel with the tile's lists in LDS and ALL column blocks of its XCD in one workgroup.  Counters of the two
// kernels above at 8192 columns (profiles/r03/bwd_gene_counters.txt): waves parked at s_waitcnt two thirds of the time, L1 miss
// queue full (TCP_PENDING_STALL), L2 hit rate 55-72 % although a block's D rows (2.5 MB) fit the XCD's 4 MB L2 -- every block
// also streams the tile lists and weights (3.8 MB of index + weight per block), the partials and the panel slice through it.
// Here a workgroup copies its tile's lists once into LDS as {gene row, NU weights} records and walks the blocks xcd, xcd + 8,
// xcd + 16, ... with them: per block the L2 sees the D gathers, one panel piece per node and the partials, nothing else, and the
// index / weight fetches and their ds_swizzle broadcasts leave the inner loop (a group's 8 lanes read one record: a broadcast).
template <int NU>
struct OvTermEnt { int32_t row; float w[NU]; };
// ACC: the weight-gradient dots of the workgroup's blocks are added up in LDS (one accumulator per link and neuron, touched by one
// lane only, blocks in ascending order: a fixed summation order without atomics) and written once, as partial copy cb0: the
// gene layer's links then have min(ncb, 8) partial copies instead of ncb (64 at 2048 columns) for the finish pass to add.
// acc_off: byte offset of the accumulators behind the records (the largest tile's record area).
template <int NU, int WPE, bool ACC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void ov_bwd_gene_term_lds_kernel(
        const int32_t* __restrict__ csc_row, const float* __restrict__ vals, float* __restrict__ d, float* __restrict__ dblk, int C,
        int ldb, int ncb, int ntiles, const float* __restrict__ act, float* __restrict__ wpart, int64_t nnz,
        const int4* __restrict__ tiles, int acc_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ov_tl_raw[];
    __shared__ int sh_off[33];
    OvTermEnt<NU>* ent = reinterpret_cast<OvTermEnt<NU>*>(ov_tl_raw);
    float* wacc = reinterpret_cast<float*>(ov_tl_raw + acc_off);
    // workgroup -> (first column block, tile): as ov_cb_map, but a workgroup keeps its tile for every block of its XCD
    const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
    int cb0, cbstep, tile;
    if (ncb <= 8) { const int xpc = 8 / ncb; cb0 = xcd % ncb; tile = q * xpc + xcd / ncb; cbstep = ncb; }
    else { cb0 = xcd; tile = q; cbstep = 8; }
    if (tile >= ntiles) return;
    const int lane = ov_lane(), grp = threadIdx.x >> 3, s = lane & 7, s4 = 4 * s;
    const int4 ti = tiles[tile * 32 + grp];
    const int k0 = ti.x, len = ti.y, node = ti.z, kb = ti.w;
    if (s == 0) sh_off[grp] = len;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int g = 0; g < 32; ++g) { const int l = sh_off[g]; sh_off[g] = run; run += l; }
        sh_off[32] = run;
    }
    __syncthreads();
    const int off = sh_off[grp];
    for (int e = s; e < len; e += 8) {
        OvTermEnt<NU> t;
        t.row = csc_row[k0 + e];
#pragma unroll
        for (int j = 0; j < NU; ++j) t.w[j] = vals[kb + j * len + e];
        ent[off + e] = t;
        if (ACC) {          // entry e belongs to lane e & 7 of this group from here to the end
#pragma unroll
            for (int j = 0; j < NU; ++j) wacc[(off + e) * NU + j] = 0.f;
        }
    }
    __syncthreads();
    int maxlen = len;                    // wave maximum of len
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    const int n0 = min(node, C - NU);
    const OvTermEnt<NU>* mine_ent = ent + off;
    for (int cb = cb0; cb < ncb; cb += cbstep) {
        const int col0 = cb * 32;
        const float* srccol = d + col0 + s4;
        float* part = wpart + (int64_t)cb * nnz;
        f32x4 acc[NU], aseg[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            aseg[j] = *reinterpret_cast<const f32x4*>(act + (int64_t)(n0 + j) * ldb + col0 + s4);
        }
        for (int b0 = 0; b0 < maxlen; b0 += 8) {
            f32x4 v[8];
            OvTermEnt<NU> en[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = b0 + i;
                en[i] = mine_ent[max(min(e, len - 1), 0)];          // an empty list reads record 0 of the tile's area (or stale LDS): weight forced to 0,
                if (len <= 0) en[i].row = C;                        // row = the first gene row of D (finite; the panel rows are being written here)
                if (e >= len) {
#pragma unroll
                    for (int j = 0; j < NU; ++j) en[i].w[j] = 0.f;
                }
                v[i] = *reinterpret_cast<const f32x4*>(srccol + (int64_t)en[i].row * ldb);
            }
            float mine[NU];
#pragma unroll
            for (int j = 0; j < NU; ++j) mine[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    const float w = en[i].w[j];
                    acc[j].x += w * v[i].x; acc[j].y += w * v[i].y;
                    acc[j].z += w * v[i].z; acc[j].w += w * v[i].w;
                    float t = v[i].x * aseg[j].x;
                    t += v[i].y * aseg[j].y;
                    t += v[i].z * aseg[j].z;
                    t += v[i].w * aseg[j].w;
                    t = ov_group_sum<3>(t);
                    mine[j] = s == i ? t : mine[j];
                }
            }
            if (b0 + s < len) {
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    if (ACC) wacc[(off + b0 + s) * NU + j] += mine[j];
                    else part[kb + j * len + b0 + s] = mine[j];
                }
            }
        }
        if (node < C) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                *reinterpret_cast<f32x4*>(d + (int64_t)(node + j) * ldb + col0 + s4) = acc[j];
                if (dblk) *reinterpret_cast<f32x4*>(dblk + ((int64_t)(cb * 8 + s) * C + node + j) * 4) = acc[j];
            }
        }
    }
    if (ACC) {
        float* part = wpart + (int64_t)cb0 * nnz;
        for (int e = s; e < len; e += 8) {
#pragma unroll
            for (int j = 0; j < NU; ++j) part[kb + j * len + e] = wacc[(off + e) * NU + j];
        }
    }
}

// Backward data, one wave per (panel node of the level, batch chunk), CSC gather:
//   D[p, :] = sum_q vals[csc_k[q]] * D[csc_row[q], :]
// csc_split != NULL: the term part of the lists only, added to what D[p, :] holds (the gene part, from ov_bwd_gene_cb_kernel)
// tw.part != NULL (training steps at large batches, no dropout): the wave has every D row of the node's term links in registers,
// so it also forms their weight-gradient dots with ACT[node] over its chunk of columns (partial copy blockIdx.y of the term
// layers' links: one per 64 VEC columns instead of one per 32 from riders that gathered the rows again) and the chunk's share of
// the bias gradient of its own row.
struct OvLvlWgrad { const float* act; float* part; float* bpart; int64_t nnz; int nrows, n_term_rows; };
template <int VEC>
__global__ __launch_bounds__(256) void ov_dec_bwd_kernel(const int32_t* __restrict__ cscptr, const int32_t* __restrict__ csc_row,
                                                        const int32_t* __restrict__ csc_k, const float* __restrict__ vals,
                                                        float* __restrict__ d, int node_begin, int node_end, int ldb,
                                                        const int32_t* __restrict__ csc_split, const OvTermDrop dr, int n_drop,
                                                        const OvLvlWgrad tw) {
    const int node = node_begin + blockIdx.x * 4 + ov_wave_id();
    if (node >= node_end) return;
    const int b0 = blockIdx.y * (64 * VEC) + ov_lane() * VEC;
    const bool wg = tw.part != nullptr;
    float acc[VEC], an[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc[j] = 0.f; an[j] = 0.f; }
    const int q0 = cscptr[node], q1 = csc_split ? csc_split[node] : cscptr[node + 1];
    if (csc_split) {
        if (q1 == q0 && !(wg && node < tw.n_term_rows)) return;          // no term children: the row is final
        ov_ldv<VEC>(d + (int64_t)node * ldb + b0, acc);
    }
    if (wg && q1 > q0) ov_ldv<VEC>(tw.act + (int64_t)node * ldb + b0, an);          // (a node without term children only has its bias partial to form)
    float* part = wg ? tw.part + (int64_t)blockIdx.y * tw.nnz : nullptr;
    auto dot = [&](const float (&v)[VEC]) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) t += v[j] * an[j];
        return t;
    };
    int q = q0;
    for (; q + 4 <= q1; q += 4) {
        float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
        const int r0 = csc_row[q], r1 = csc_row[q + 1], r2 = csc_row[q + 2], r3 = csc_row[q + 3];
        const int k0 = csc_k[q], k1 = csc_k[q + 1], k2 = csc_k[q + 2], k3 = csc_k[q + 3];
        const float w0 = vals[k0], w1 = vals[k1], w2 = vals[k2], w3 = vals[k3];
        ov_ldv<VEC>(d + (int64_t)r0 * ldb + b0, v0);
        ov_ldv<VEC>(d + (int64_t)r1 * ldb + b0, v1);
        ov_ldv<VEC>(d + (int64_t)r2 * ldb + b0, v2);
        ov_ldv<VEC>(d + (int64_t)r3 * ldb + b0, v3);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            acc[j] += w0 * v0[j];
            acc[j] += w1 * v1[j];
            acc[j] += w2 * v2[j];
            acc[j] += w3 * v3[j];
        }
        if (wg) {
            const float t0 = ov_wave_sum(dot(v0)), t1 = ov_wave_sum(dot(v1)), t2 = ov_wave_sum(dot(v2)), t3 = ov_wave_sum(dot(v3));
            if (ov_lane() == 0) { part[k0] = t0; part[k1] = t1; part[k2] = t2; part[k3] = t3; }
        }
    }
    for (; q < q1; ++q) {
        float v0[VEC];
        const int r0 = csc_row[q];
        const int k0 = csc_k[q];
        const float w0 = vals[k0];
        ov_ldv<VEC>(d + (int64_t)r0 * ldb + b0, v0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += w0 * v0[j];
        if (wg) {
            const float t0 = ov_wave_sum(dot(v0));
            if (ov_lane() == 0) part[k0] = t0;
        }
    }
    if (wg && node < tw.n_term_rows) {              // bias gradient of this row: its finished D over the chunk's columns
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) t += acc[j];
        t = ov_wave_sum(t);
        if (ov_lane() == 0) tw.bpart[(int64_t)blockIdx.y * tw.nrows + node] = t;
        if (q1 == q0) return;                       // D[node] is unchanged
    }
    if (dr.rate > 0.f && node < n_drop) {        // term nodes only: the rows of the operator's input carry no dropout
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            acc[j] *= ov_dropout_scale(dr.rate, dr.seed, dr.rng_off, OV_RNG_DROP, (uint64_t)node * 0x100000000ull + (uint64_t)(b0 + j));
    }
    ov_stv<VEC>(d + (int64_t)node * ldb + b0, acc);
}

// Weight gradient at mask positions (SDDMM) on 32-column blocks, one row per lane group:
//   part[cb][k] = sum over the block's columns of D[n, b] * ACT[colidx[k], b]        bpart[cb][r] = sum of D[n, b]
// The group's 8 lanes hold 4 columns each of D[n] in registers, gather the same 4 columns of ACT[p] (one 128-B
// line per group), and a dot needs 3 DPP adds within the group instead of a wave-wide reduction.  The column
// blocks are pinned to XCDs like the other gene-layer kernels, so the panel columns a workgroup gathers stay in its
// L2.  ov_wgrad_finish_kernel adds the blocks' partials in a fixed order and applies AdamW + clamp.
struct OvWgradArgs {
    const int32_t* rowptr; const int32_t* colidx; const int32_t* row_k; const float* act; const float* d;
    float* part; float* bpart;
    int64_t nnz;
    int row0, nrows_part, nrows;      // this launch covers rows [row0, row0 + nrows_part) of the nrows rows
    int first_term_node, n_term_rows, C, ldb, ncb, ntiles;
    const int4* tiles;                // the plan's records of the term tiles (launches that cover exactly the term rows), or NULL
};
// wg: virtual workgroup number of the 256-thread team, tid: thread number inside the team.  Every thread of the
// real workgroup must call this (one workgroup barrier pair inside).
__device__ __forceinline__ void ov_wgrad_cb_body(const OvWgradArgs& a, int wg, int tid, int* sh_k0, int* sh_len, int* sh_ord) {
    const OvCbMap map = ov_cb_map(wg, a.ncb, a.ntiles);
    const bool tile_on = map.tile < a.ntiles;
    const int lane = ov_lane();
    const int s = lane & 7;
    const int col0 = map.cb * 32, s4 = 4 * s;
    auto node_of = [&](int r) { return r < a.n_term_rows ? a.first_term_node + r : a.C + (r - a.n_term_rows); };
    const int rend = a.row0 + a.nrows_part;
    int k0, len, r;
    if (a.tiles) {                                   // the tile's rows ranked by list length, laid down by the plan
        if (!tile_on) return;
        const int4 ti = a.tiles[map.tile * 32 + (tid >> 3)];
        k0 = ti.x; len = ti.y; r = ti.z;             // an unused slot has r = n_term_rows = rend
    } else {
        int my_k0 = 0, my_len = 0;
        if (tile_on && tid < 32) {
            const int rr = a.row0 + map.tile * 32 + tid;
            if (rr < rend) { const int n = node_of(rr); my_k0 = a.rowptr[n]; my_len = a.rowptr[n + 1] - my_k0; }
        }
        r = a.row0 + map.tile * 32 + ov_tile_balance(my_k0, my_len, sh_k0, sh_len, sh_ord, k0, len, tid);
        if (!tile_on) return;                        // no barrier below
    }
    const bool row_on = r < rend;
    const int node = node_of(min(r, rend - 1));
    const f32x4 dseg = *reinterpret_cast<const f32x4*>(a.d + (int64_t)node * a.ldb + col0 + s4);
    {
        float gb = (dseg.x + dseg.y) + (dseg.z + dseg.w);
        gb = ov_group_sum<3>(gb);
        if (row_on && s == 7) a.bpart[(int64_t)map.cb * a.nrows + r] = gb;
    }
    int maxlen = len;                                // wave maximum
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    const float* actcol = a.act + col0 + s4;
    float* pblk = a.part + (int64_t)map.cb * a.nnz;
    // the link's column node and its packed position (where its partial goes)
    auto fetch = [&](int e, int& cnode, int& kpos) {
        const int j = k0 + max(min(e, len - 1), 0);
        cnode = len > 0 ? a.colidx[j] : 0;
        kpos = len > 0 ? a.row_k[j] : 0;
    };
    int cur_c, cur_k, nxt_c = 0, nxt_k = 0;
    fetch(s, cur_c, cur_k);
    for (int b0 = 0; b0 < maxlen; b0 += 8) {
        if (b0 + 8 < maxlen) fetch(b0 + 8 + s, nxt_c, nxt_k);
        f32x4 v[8];
#define OV_WG_LOAD(I) v[I] = *reinterpret_cast<const f32x4*>(actcol + (int64_t)ov_grp_bcast<I>(cur_c) * a.ldb);
        OV_WG_LOAD(0) OV_WG_LOAD(1) OV_WG_LOAD(2) OV_WG_LOAD(3) OV_WG_LOAD(4) OV_WG_LOAD(5) OV_WG_LOAD(6) OV_WG_LOAD(7)
#undef OV_WG_LOAD
        float mine = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = dseg.x * v[i].x;
            t += dseg.y * v[i].y;
            t += dseg.z * v[i].z;
            t += dseg.w * v[i].w;
            t = ov_group_sum<3>(t);                                   // xor 1, xor 2, half mirror: every lane of the group has the total
            mine = s == i ? t : mine;
        }
        if (row_on && b0 + s < len) pblk[cur_k] = mine;
        cur_c = nxt_c; cur_k = nxt_k;
    }
}

__global__ __launch_bounds__(256) void ov_wgrad_cb_kernel(const OvWgradArgs a) {
    __shared__ int sh_k0[32], sh_len[32], sh_ord[32];
    ov_wgrad_cb_body(a, blockIdx.x, threadIdx.x, sh_k0, sh_len, sh_ord);
}

// g_k = sum over column blocks of part[cb][k] (fixed order), then AdamW + clamp(0) (vae_model.py:172-176) or the
// plain gradient; the same for the biases of the rows.
struct OvFinishArgs {
    const float* part; const float* bpart; int ncb;
    int ncb_gene;       // partial copies of the GENE layer's links (k >= nnz_term): ncb, or fewer when the backward gene kernel added blocks up itself
    int ncb_term;       // ... of the term layers' links and of the term rows' bias partials: ncb, or one per chunk of columns from the per-level backward kernels
    int64_t nnz; int nrows, first_term_node, n_term_rows, C;
    float* vals; float* mvals; float* vvals; float* gvals;
    float* bias; float* mbias; float* vbias; float* gbias;
    int apply_update; OvAdam adam;
    int64_t nnz_term;
    int sel;        // 0: everything; 1: the gene layer's links and biases only; 2: the term layers' only (nnz_term must be set)
    // gene rows' bias partials as ov_gene_lds_kernel leaves them: gb4[group][lane] over ngrp column groups, the lane's gene in lane_gene
    const float* gb4; const int32_t* lane_gene; int nl, ngrp;
};
// first / stride: this thread's number and the thread count of the finish part
__device__ __forceinline__ void ov_wgrad_finish_body(const OvFinishArgs& a, const int64_t first, const int64_t stride) {
    auto update = [&](int64_t k, float g) {
        if (a.apply_update) {
            float w = a.vals[k], m = a.mvals[k], v = a.vvals[k];
            ov_adamw(w, m, v, g, a.adam, true);
            a.vals[k] = w; a.mvals[k] = m; a.vvals[k] = v;
        } else {
            a.gvals[k] = g;
        }
    };
    const bool terms = a.sel != 1, genes = a.sel != 2;
    const int64_t kb = terms ? 0 : a.nnz_term, ke = genes ? a.nnz : a.nnz_term;
    for (int64_t k = kb + first; k < ke; k += stride) {
        // the column blocks' partials in a fixed order; eight loads in flight (at 256 blocks this pass reads 0.5 GB)
        float g = 0.f;
        int cb = 0;
        const int ncopies = k >= a.nnz_term ? a.ncb_gene : a.ncb_term;
        for (; cb + 8 <= ncopies; cb += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = a.part[(int64_t)(cb + j) * a.nnz + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) g += t[j];
        }
        for (; cb < ncopies; ++cb) g += a.part[(int64_t)cb * a.nnz + k];
        update(k, g);
    }
    auto update_bias = [&](int node, float g) {
        if (a.apply_update) {
            float w = a.bias[node], m = a.mbias[node], v = a.vbias[node];
            ov_adamw(w, m, v, g, a.adam, false);
            a.bias[node] = w; a.mbias[node] = m; a.vbias[node] = v;
        } else {
            a.gbias[node] = g;
        }
    };
    if (genes && a.gb4) {
        // 16 lanes of the SELL form per wave, the column groups in 4 runs: every load of a wave is 4 pieces of 64 B, and four
        // times as many threads walk runs a quarter as long (2048 groups at 8192 columns); run sums are added in run order
        const int ii = (int)(first & 15), seg = (int)((first >> 4) & 3);
        const int per = (a.ngrp + 3) / 4, c0 = seg * per, c1 = min(c0 + per, a.ngrp);
        for (int64_t i0 = (first >> 6) * 16; i0 < a.nl; i0 += (stride >> 6) * 16) {        // wave-uniform trip count
            const int64_t i = i0 + ii;
            float g = 0.f;
            if (i < a.nl) {
                int cg = c0;
                for (; cg + 8 <= c1; cg += 8) {
                    float t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = a.gb4[(int64_t)(cg + j) * a.nl + i];
#pragma unroll
                    for (int j = 0; j < 8; ++j) g += t[j];
                }
                for (; cg < c1; ++cg) g += a.gb4[(int64_t)cg * a.nl + i];
            }
            const float g1 = __shfl(g, ii + 16, 64), g2 = __shfl(g, ii + 32, 64), g3 = __shfl(g, ii + 48, 64);
            if (seg == 0 && i < a.nl) {
                const int gene = a.lane_gene[i];
                if (gene >= 0) update_bias(a.C + gene, ((g + g1) + g2) + g3);
            }
        }
    }
    const int rb = terms ? 0 : a.n_term_rows, re = (genes && !a.gb4) ? a.nrows : a.n_term_rows;
    auto node_of_row = [&](int64_t r) { return r < a.n_term_rows ? a.first_term_node + (int)r : a.C + ((int)r - a.n_term_rows); };
    // copies of the rows this pass covers: one kind of row, or both kinds with the same count
    const bool gene_rows = genes && !a.gb4;
    const int nbr = (terms && !gene_rows) ? a.ncb_term : a.ncb;
    if (nbr >= 16 && (!terms || !gene_rows || a.ncb_term == a.ncb)) {
        // many column blocks: 16 rows per wave, the blocks in 4 runs of eight loads in flight, run sums added in run order
        // (the term layers' 8640 rows alone were 118 us at 8192 batch rows as one thread per row, 256 loads one behind the other)
        const int ii = (int)(first & 15), seg = (int)((first >> 4) & 3);
        const int per = (nbr + 3) / 4, c0 = seg * per, c1 = min(c0 + per, nbr);
        for (int64_t r0 = rb + (first >> 6) * 16; r0 < re; r0 += (stride >> 6) * 16) {        // wave-uniform trip count
            const int64_t r = r0 + ii;
            float g = 0.f;
            if (r < re) {
                int cb = c0;
                for (; cb + 8 <= c1; cb += 8) {
                    float t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = a.bpart[(int64_t)(cb + j) * a.nrows + r];
#pragma unroll
                    for (int j = 0; j < 8; ++j) g += t[j];
                }
                for (; cb < c1; ++cb) g += a.bpart[(int64_t)cb * a.nrows + r];
            }
            const float g1 = __shfl(g, ii + 16, 64), g2 = __shfl(g, ii + 32, 64), g3 = __shfl(g, ii + 48, 64);
            if (seg == 0 && r < re) update_bias(node_of_row(r), ((g + g1) + g2) + g3);
        }
        return;
    }
    for (int64_t r = rb + first; r < re; r += stride) {
        float g = 0.f;
        const int ncopies = r < a.n_term_rows ? a.ncb_term : a.ncb;
        for (int cb = 0; cb < ncopies; ++cb) g += a.bpart[(int64_t)cb * a.nrows + r];
        update_bias(node_of_row(r), g);
    }
}
__global__ __launch_bounds__(256) void ov_wgrad_finish_kernel(const OvFinishArgs a) {
    ov_wgrad_finish_body(a, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

// Backward term chain and the gene layer's share of the decoder update in ONE launch.  The chain occupies only ldb / BT
// workgroups (32 CUs at batch 128) for its whole, latency-bound duration.  Everything the update of the gene layer's
// weights and biases needs (the partials of the two gene-layer kernels) is final by then and nothing in this launch or
// after it reads those weights again, so its finish pass (sum of the column blocks' partials, AdamW, clamp) runs on the
// CUs the chain leaves idle: workgroups [0, nchain) are the chain, the others finish.  Low workgroup ids are dispatched
// first, so the chain starts at once.
template <int BT, bool SINGLE>
__global__ __launch_bounds__(1024) void ov_chain_bwd_finish_kernel(const OvChainArgs ca, const OvSteps st, int nchain,
                                                                 const OvFinishArgs fa, int fblocks) {
    if ((int)blockIdx.x < nchain) {
        ov_dec_chain_body<BT, true, SINGLE>(ca, st, blockIdx.x);
        return;
    }
    ov_wgrad_finish_body(fa, (int64_t)((int)blockIdx.x - nchain) * 1024 + threadIdx.x, (int64_t)fblocks * 1024);
}

// Two launches of the backward pass that carry unrelated, independent work of the decoder's weight update so that it
// costs no launch of its own (all parts are 256-thread workgroups):
//   latent backward (a few dozen workgroups)  +  weight-gradient partials of the rows the backward chain finished
//   da = Wh^T dmulv (split-K GEMM)             +  the finish pass over the partials (AdamW + clamp)
// fblocks > 0 (large batches): workgroups [nblocks, nblocks + fblocks) run the finish pass of the gene layer's update.  Its
// partials are final since the gene backward kernel and nothing reads those weights again in the step; it is a pure HBM
// stream (0.5 GB at 8192 rows), and the weight-gradient workgroups that share this launch are many, short and bound by their
// memory round trips.  They come first in the id order, so that they are resident while the others pass through.
__global__ __launch_bounds__(256) void ov_latent_bwd_kernel(const OvLatentArgs la, const OvWgradArgs wa, int wteams, const OvFinishArgs fa,
                                                           int fblocks) {
    __shared__ int sh_k0[32], sh_len[32], sh_ord[32];
    int b = blockIdx.x;
    if (b < la.nblocks) { ov_latent_bwd_body(la, b); return; }
    b -= la.nblocks;
    if (b < fblocks) { ov_wgrad_finish_body(fa, (int64_t)b * 256 + threadIdx.x, (int64_t)fblocks * 256); return; }
    b -= fblocks;
    if (b < wteams) ov_wgrad_cb_body(wa, b, threadIdx.x, sh_k0, sh_len, sh_ord);
}
// Large batches: da = Wh^T dmulv goes through the LDS-staged GEMM, which wants its A operand with k contiguous.  One small
// launch makes the transposed copy of Wh ([2 L, L] -> [L, 2 L], 32 x 32 tiles) and carries the finish pass that would
// otherwise ride with the da GEMM.
__global__ __launch_bounds__(256) void ov_whT_finish_kernel(const float* __restrict__ src, int R, int F, float* __restrict__ dst,
                                                           int ntx, int nty, const OvFinishArgs fa, int fblocks) {
    __shared__ float tile[32][33];
    const int b = blockIdx.x;
    if (b < ntx * nty) {
        const int f0 = (b % ntx) * 32, r0 = (b / ntx) * 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int j = ty; j < 32; j += 8) {
            const int r = r0 + j, f = f0 + tx;
            tile[j][tx] = (f < F && r < R) ? src[(int64_t)r * F + f] : 0.f;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int f = f0 + j, r = r0 + tx;
            if (f < F && r < R) dst[(int64_t)f * R + r] = tile[tx][j];
        }
        return;
    }
    ov_wgrad_finish_body(fa, (int64_t)(b - ntx * nty) * 256 + threadIdx.x, (int64_t)fblocks * 256);
}
template <bool VEC>
__global__ __launch_bounds__(256) void ov_da_gemm_kernel(const OvGemm g, int gx, int gy, int gz, const OvFinishArgs fa, int fblocks) {
    const int b = blockIdx.x;
    const int ngemm = gx * gy * gz;
    if (b < ngemm) {
        ov_gemm_body<1, 1, 4, 4, false, false, VEC, OV_EPI_STORE>(g, b % gx, (b / gx) % gy, b / (gx * gy));
        return;
    }
    ov_wgrad_finish_body(fa, (int64_t)(b - ngemm) * 256 + threadIdx.x, (int64_t)fblocks * 256);
}
// dWh = dmulv a^T (+ AdamW) and the BatchNorm backward are independent once da exists: one launch
template <int EPI>
__global__ __launch_bounds__(256) void ov_dwh_bn_kernel(const OvGemm g, int gx, int gy, const OvBnBwdArgs bn) {
    const int b = blockIdx.x;
    if (b < gx * gy) {
        ov_gemm_body<1, 1, 4, 2, true, true, true, EPI>(g, b % gx, b / gx, 0);
        return;
    }
    ov_bn_bwd_body(bn, b - gx * gy);
}

// The same launch from 1024 columns on: dWh as LDS-staged 96 x 128 tiles over split-K slabs of the batch (the 32 x 32 tiles above
// load MFMA fragments straight from global memory, 32 lines per wave instruction: 169 us at 8192 columns for 4.2 GFLOP), the
// BatchNorm backward beside it; ov_wh_combine_kernel then adds the slabs in slab order and applies AdamW.
__global__ __launch_bounds__(256) void ov_dwh_lds_bn_kernel(const OvGemm g, int gx, int gy, int gz, const OvBnBwdArgs bn) {
    const int b = blockIdx.x;
    const int ngemm = gx * gy * gz;
    if (b < ngemm) {
        ov_gemm_lds_body<1, 4, true, OV_EPI_STORE>(g, dim3(b % gx, (b / gx) % gy, b / (gx * gy)), dim3(gx, gy, gz));
        return;
    }
    ov_bn_bwd_body(bn, b - ngemm);
}
__global__ __launch_bounds__(256) void ov_wh_combine_kernel(const float* __restrict__ slabs, int S, int64_t n, float* __restrict__ P,
                                                           float* __restrict__ M, float* __restrict__ V, float* __restrict__ Gr,
                                                           int apply, OvAdam adam) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;                                        // n % 4 == 0 (host check)
    f32x4 acc = *reinterpret_cast<const f32x4*>(slabs + i);
    for (int z = 1; z < S; ++z) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * n + i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (!apply) { *reinterpret_cast<f32x4*>(Gr + i) = acc; return; }
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(P + i), m4 = *reinterpret_cast<const f32x4*>(M + i), v4 = *reinterpret_cast<const f32x4*>(V + i);
    float w[4] = {w4.x, w4.y, w4.z, w4.w}, m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
    const float g[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) ov_adamw(w[e], m[e], v[e], g[e], adam, false);
    *reinterpret_cast<f32x4*>(P + i) = f32x4{w[0], w[1], w[2], w[3]};
    *reinterpret_cast<f32x4*>(M + i) = f32x4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<f32x4*>(V + i) = f32x4{v[0], v[1], v[2], v[3]};
}

// Weight gradient at mask positions only (SDDMM) + fused AdamW + clamp(0), one wave per row node:
//   g_k = sum_b D[n, b] * ACT[colidx[k], b];   gb = sum_b D[n, b]
// The padded batch columns of D are zero by construction, so no masking is needed in the dots.
// The 64 lanes then update 64 consecutive packed weights at once (coalesced w, m, v traffic).
#define OV_WGRAD_U 16      // row dots in flight per wave (64 % OV_WGRAD_U == 0)
#define OV_WGRAD_NB 2      // 64-column blocks of the batch per trip
__global__ __launch_bounds__(256) void ov_dec_wgrad_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                          const int32_t* __restrict__ row_k, const float* __restrict__ act, const float* __restrict__ d,
                                                          float* __restrict__ vals, float* __restrict__ mvals,
                                                          float* __restrict__ vvals, float* __restrict__ gvals,
                                                          float* __restrict__ bias, float* __restrict__ mbias,
                                                          float* __restrict__ vbias, float* __restrict__ gbias, int nrows,
                                                          int first_term_node, int n_term_rows, int C, int ldb,
                                                          int apply_update, OvAdam adam) {
    const int r = blockIdx.x * 4 + ov_wave_id();
    if (r >= nrows) return;
    // rows are the term-feature nodes that have a layer above them, then the gene nodes
    const int node = r < n_term_rows ? first_term_node + r : C + (r - n_term_rows);
    const int lane = ov_lane();
    const float* dr = d + (int64_t)node * ldb;
    const int k0 = rowptr[node], k1 = rowptr[node + 1];
    float gb = 0.f;
    for (int b = lane; b < ldb; b += 64) gb += dr[b];
    gb = ov_wave_sum(gb);
    for (int kb = k0; kb < k1; kb += 64) {
        const int cnt = min(64, k1 - kb);
        // one coalesced index load per 64 entries; lanes beyond cnt repeat the last entry (its row is re-read from L1)
        const int my_c = colidx[kb + min(lane, cnt - 1)];
        const int k = row_k[kb + min(lane, cnt - 1)];      // packed position of the lane's link
        float w = 0.f, m = 0.f, v = 0.f;
        if (apply_update && lane < cnt) { w = vals[k]; m = mvals[k]; v = vvals[k]; }     // lands under the dots
        float mine = 0.f;
        for (int b0 = 0; b0 < ldb; b0 += 64 * OV_WGRAD_NB) {      // one trip for batches up to 256 columns
            float dv[OV_WGRAD_NB];
#pragma unroll
            for (int q = 0; q < OV_WGRAD_NB; ++q) dv[q] = b0 + 64 * q < ldb ? dr[b0 + 64 * q + lane] : 0.f;
            for (int j = 0; j < cnt; j += OV_WGRAD_U) {          // OV_WGRAD_U row dots in flight
                float av[OV_WGRAD_U][OV_WGRAD_NB], sacc[OV_WGRAD_U];
#pragma unroll
                for (int t = 0; t < OV_WGRAD_U; ++t) {
                    const float* ar = act + (int64_t)OV_RL_I(my_c, j + t) * ldb + b0 + lane;
#pragma unroll
                    for (int q = 0; q < OV_WGRAD_NB; ++q) av[t][q] = b0 + 64 * q < ldb ? ar[64 * q] : 0.f;
                }
#pragma unroll
                for (int t = 0; t < OV_WGRAD_U; ++t) {
                    sacc[t] = 0.f;
#pragma unroll
                    for (int q = 0; q < OV_WGRAD_NB; ++q) sacc[t] += dv[q] * av[t][q];
                }
#pragma unroll
                for (int t = 0; t < OV_WGRAD_U; ++t) {
                    const float tot = ov_wave_sum(sacc[t]);
                    mine += lane == j + t ? tot : 0.f;
                }
            }
        }
        if (lane < cnt) {
            if (apply_update) {
                ov_adamw(w, m, v, mine, adam, true);
                vals[k] = w; mvals[k] = m; vvals[k] = v;
            } else {
                gvals[k] = mine;
            }
        }
    }
    if (lane == 0) {
        if (apply_update) {
            float w = bias[node], m = mbias[node], v = vbias[node];
            ov_adamw(w, m, v, gb, adam, false);
            bias[node] = w; mbias[node] = m; vbias[node] = v;
        } else {
            gbias[node] = gb;
        }
    }
}

// =================================================================================================
// Small kernels
// =================================================================================================
// row sums of a feature-major matrix (+ AdamW): bias gradients of the heads
__global__ __launch_bounds__(256) void ov_rowsum_adamw_kernel(const float* __restrict__ src, int rows, int ldb,
                                                             float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                             float* __restrict__ g, int apply_update, OvAdam adam) {
    const int r = blockIdx.x * 4 + ov_wave_id();
    if (r >= rows) return;
    float s = 0.f;
    for (int b = ov_lane(); b < ldb; b += 64) s += src[(int64_t)r * ldb + b];
    s = ov_wave_sum(s);
    if (ov_lane() == 0) {
        if (apply_update) {
            float ww = w[r], mm = m[r], vv = v[r];
            ov_adamw(ww, mm, vv, s, adam, false);
            w[r] = ww; m[r] = mm; v[r] = vv;
        } else {
            g[r] = s;
        }
    }
}

// deterministic final reduction of the block partials; loss = rec + kl_coeff * kl (vae_model.py:129)
__global__ __launch_bounds__(1024) void ov_loss_finalize_kernel(const float* __restrict__ rec_part, int nrec,
                                                               const float* __restrict__ kl_part, int nkl, float kc,
                                                               float* __restrict__ out) {
    __shared__ double red[2][16];
    double r = 0.0, k = 0.0;
    auto total = [&](const float* __restrict__ p, int n) {          // eight loads in flight per thread
        double acc = 0.0;
        int i = threadIdx.x;
        for (; i + 7 * 1024 < n; i += 8 * 1024) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = p[i + j * 1024];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (double)t[j];
        }
        for (; i < n; i += 1024) acc += (double)p[i];
        return acc;
    };
    r = total(rec_part, nrec);
    k = total(kl_part, nkl);
    r = ov_wave_sum_d(r);
    k = ov_wave_sum_d(k);
    if (ov_lane() == 0) {
        red[0][threadIdx.x >> 6] = r;
        red[1][threadIdx.x >> 6] = k;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double rr = 0.0, kk = 0.0;
        for (int i = 0; i < 16; ++i) { rr += red[0][i]; kk += red[1][i]; }
        const float rf = (float)rr, kf = (float)kk;
        out[0] = rf;
        out[1] = kf;
        out[2] = rf + kc * kf;
    }
}

// flat AdamW over [0, n) with clamp(0) inside [c0, c1)
__global__ __launch_bounds__(256) void ov_adamw_flat_kernel(float* __restrict__ P, float* __restrict__ M, float* __restrict__ V,
                                                           const float* __restrict__ Gr, int64_t n, int64_t c0, int64_t c1,
                                                           OvAdam adam) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float w = P[i], m = M[i], v = V[i];
        ov_adamw(w, m, v, Gr[i], adam, i >= c0 && i < c1);
        P[i] = w; M[i] = m; V[i] = v;
    }
}

// feature-major [F, ldb] -> sample-major [B, ld_out] (32x32 LDS tile)
__global__ __launch_bounds__(256) void ov_transpose_out_kernel(const float* __restrict__ src, int F, int B, int ldb,
                                                              float* __restrict__ dst, int64_t ld_out) {
    __shared__ float tile[32][33];
    const int f0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int f = f0 + j, b = b0 + tx;
        tile[j][tx] = (f < F && b < B) ? src[(int64_t)f * ldb + b] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int b = b0 + j, f = f0 + tx;
        if (f < F && b < B) dst[(int64_t)b * ld_out + f] = tile[tx][j];
    }
}

// the same in 64 x 64 tiles (whole 256-byte rows per wave on both sides), for inference batches
__global__ __launch_bounds__(256) void ov_transpose_out64_kernel(const float* __restrict__ src, int F, int B, int ldb,
                                                                float* __restrict__ dst, int64_t ld_out) {
    __shared__ float tile[64][65];
    const int f0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int f = f0 + ty + 4 * q, b = b0 + tx;
        v[q] = (f < F && b < B) ? src[(int64_t)f * ldb + b] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[ty + 4 * q][tx] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int b = b0 + ty + 4 * q, f = f0 + tx;
        if (f < F && b < B) dst[(int64_t)b * ld_out + f] = tile[tx][ty + 4 * q];
    }
}

// sample-major z [B, L] -> feature-major panel rows
__global__ __launch_bounds__(256) void ov_transpose_in_kernel(const float* __restrict__ src, int F, int B, int ldb,
                                                             float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int f0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int b = b0 + j, f = f0 + tx;
        tile[j][tx] = (f < F && b < B) ? src[(int64_t)b * F + f] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int f = f0 + j, b = b0 + tx;
        if (f < F && b < ldb) dst[(int64_t)f * ldb + b] = tile[tx][j];
    }
}

// nu-mean of the panel rows of one level into sample-major activities (vae_model.py:297, 317):
// out[b, t0 + j] = (ACT[n0 + j*nu, b] + ... + ACT[n0 + j*nu + nu - 1, b]) / nu
__global__ __launch_bounds__(256) void ov_act_out_kernel(const float* __restrict__ act, int n0, int nterms, int nu, int B,
                                                        int ldb, float* __restrict__ out, int64_t ld_out, int t0) {
    __shared__ float tile[32][33];
    const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int jj = ty; jj < 32; jj += 8) {
        const int j = j0 + jj, b = b0 + tx;
        float s = 0.f;
        if (j < nterms && b < B) {
            for (int a = 0; a < nu; ++a) s += act[(int64_t)(n0 + j * nu + a) * ldb + b];
            s = s / (float)nu;
        }
        tile[jj][tx] = s;
    }
    __syncthreads();
    for (int jj = ty; jj < 32; jj += 8) {
        const int b = b0 + jj, j = j0 + tx;
        if (j < nterms && b < B) out[(int64_t)b * ld_out + t0 + j] = tile[tx][jj];
    }
}

// The same for a range of levels in one launch: output column c belongs to the level lv with tstart[lv] <= c < tstart[lv + 1]
struct OvActLevels { int n; int tstart[OV_MAX_LEVELS + 1]; int node0[OV_MAX_LEVELS]; };
__global__ __launch_bounds__(256) void ov_act_out_levels_kernel(const float* __restrict__ act, const OvActLevels lv, int c_begin,
                                                               int c_end, int nu, int B, int ldb, float* __restrict__ out,
                                                               int64_t ld_out) {
    __shared__ float tile[32][33];
    const int j0 = c_begin + blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int jj = ty; jj < 32; jj += 8) {
        const int c = j0 + jj, b = b0 + tx;
        float s = 0.f;
        if (c < c_end && b < B) {
            int l = 0;
            while (l + 1 < lv.n && c >= lv.tstart[l + 1]) ++l;
            const int row = lv.node0[l] + (c - lv.tstart[l]) * nu;
            for (int a = 0; a < nu; ++a) s += act[(int64_t)(row + a) * ldb + b];
            s = s / (float)nu;
        }
        tile[jj][tx] = s;
    }
    __syncthreads();
    for (int jj = ty; jj < 32; jj += 8) {
        const int b = b0 + jj, c = j0 + tx;
        if (c < c_end && b < B) out[(int64_t)b * ld_out + c] = tile[tx][jj];
    }
}

// The same, feature-major: out[c * ld_out + b] (the layout the Wilcoxon kernel reads), no transposition needed
__global__ __launch_bounds__(256) void ov_act_fm_levels_kernel(const float* __restrict__ act, const OvActLevels lv, int c_begin, int c_end,
                                                              int nu, int B, int ldb, float* __restrict__ out, int64_t ld_out) {
    const int c = c_begin + blockIdx.y, b = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= c_end || b >= B) return;
    int l = 0;
    while (l + 1 < lv.n && c >= lv.tstart[l + 1]) ++l;
    const float* src = act + (int64_t)(lv.node0[l] + (c - lv.tstart[l]) * nu) * ldb + b;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < nu; ++a) s += *reinterpret_cast<const f32x4*>(src + (int64_t)a * ldb);
    float* dst = out + (int64_t)c * ld_out + b;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (b + j < B) dst[j] = s[j] / (float)nu;
}

__global__ void ov_compose_order_kernel(const int32_t* __restrict__ in, const int64_t* __restrict__ perm,
                                        int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}

__global__ __launch_bounds__(256) void ov_gather_rows_kernel(const float* __restrict__ X, int64_t ld, const int32_t* __restrict__ idx,
                                                            int G, float* __restrict__ out) {
    const int b = blockIdx.x;
    const float* src = X + (int64_t)idx[b] * ld;
    float* dst = out + (int64_t)b * G;
    for (int g = threadIdx.x; g < G; g += 256) dst[g] = src[g];
}

__global__ void ov_vals_to_dense_kernel(const float* __restrict__ vals, const int32_t* __restrict__ er,
                                        const int32_t* __restrict__ ec, int64_t k0, int64_t k1, float* __restrict__ dense,
                                        int64_t ld) {
    const int64_t k = k0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < k1) dense[(int64_t)er[k] * ld + ec[k]] = vals[k];
}
__global__ void ov_dense_to_vals_kernel(const float* __restrict__ dense, int64_t ld, const int32_t* __restrict__ er,
                                        const int32_t* __restrict__ ec, int64_t k0, int64_t k1, float* __restrict__ vals) {
    const int64_t k = k0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < k1) vals[k] = dense[(int64_t)er[k] * ld + ec[k]];
}

// =================================================================================================
// Host-side launch helpers
// =================================================================================================
static inline OvAdam ov_adam_from(const ontovae_adamw_t* o) {
    OvAdam a{};
    if (o) {
        a.decay_mul = o->decay_mul; a.one_m_beta1 = o->one_m_beta1; a.beta2 = o->beta2; a.one_m_beta2 = o->one_m_beta2;
        a.step_size = o->step_size; a.bc2_sqrt = o->bc2_sqrt; a.eps = o->eps;
    }
    return a;
}

static int ov_check_ctx(const ontovae_ctx_t* c, int B, bool need_opt, bool train = false) {
    if (!c) { ontovae_set_error("ctx is NULL"); return ONTOVAE_EINVAL; }
    const ontovae_dims_t& d = c->dims;
    if (d.G < 1 || d.L < 1 || d.C < d.L || d.K < 1 || d.nu < 1 || d.nnz < 0) {
        ontovae_set_error("ctx: bad dims G=%d L=%d C=%d K=%d nu=%d", d.G, d.L, d.C, d.K, d.nu);
        return ONTOVAE_EINVAL;
    }
    if (!c->P || !c->rowptr || !c->colidx || !c->cscptr || !c->csc_row || !c->csc_k || !c->csc_split || !c->row_k || !c->level_begin || !c->level_end ||
        !c->workspace || !c->bn_rmean || !c->bn_rvar || !c->bn_nbt) {
        ontovae_set_error("ctx: a required pointer is NULL");
        return ONTOVAE_EINVAL;
    }
    if (need_opt && (!c->M || !c->V)) { ontovae_set_error("ctx: optimizer state (M, V) is NULL"); return ONTOVAE_EINVAL; }
    if (B < 1 || B > c->max_batch) {
        ontovae_set_error("batch size %d outside [1, max_batch=%d]", B, c->max_batch);
        return ONTOVAE_EINVAL;
    }
    if (train && B > ov_train_cap(c)) {
        ontovae_set_error("ctx: %d rows in a training pass, the workspace is laid out for %d (max_train_batch)", B, ov_train_cap(c));
        return ONTOVAE_EINVAL;
    }
    if (c->workspace_bytes < ov_carve(d, c->max_batch, ov_train_cap(c), nullptr, nullptr)) {
        ontovae_set_error("workspace too small: %lld < %lld bytes", (long long)c->workspace_bytes,
                          (long long)ov_carve(d, c->max_batch, ov_train_cap(c), nullptr, nullptr));
        return ONTOVAE_EINVAL;
    }
    if (c->flags & ONTOVAE_FLAG_BF16_GEMM) {
        if (d.G % 8) { ontovae_set_error("ctx: the bf16 variant needs a gene count that is a multiple of 8 (G = %d)", d.G); return ONTOVAE_EINVAL; }
        if (!c->bf16_ws || (reinterpret_cast<uintptr_t>(c->bf16_ws) & 255) || c->bf16_ws_bytes < ov_bf16_carve(d, c->max_batch, nullptr, nullptr)) {
            ontovae_set_error("ctx: ONTOVAE_FLAG_BF16_GEMM needs bf16_ws of %lld bytes, 256-B aligned (ontovae_bf16_workspace_bytes)",
                              (long long)ov_bf16_carve(d, c->max_batch, nullptr, nullptr));
            return ONTOVAE_EINVAL;
        }
    }
#ifndef OV_MEASUREMENT_SWITCHES
    if (c->flags & ~(uint32_t)OV_KNOWN_FLAGS) {
        ontovae_set_error("ctx: unknown flag bits 0x%x (known: 0x%x)", c->flags & ~(uint32_t)OV_KNOWN_FLAGS, (unsigned)OV_KNOWN_FLAGS);
        return ONTOVAE_EINVAL;
    }
#endif
    if ((reinterpret_cast<uintptr_t>(c->workspace) & 255) || (reinterpret_cast<uintptr_t>(c->P) & 15)) {
        ontovae_set_error("ctx: workspace must be 256-B aligned and P 16-B aligned");
        return ONTOVAE_EINVAL;
    }
    return ONTOVAE_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a driver call of several microseconds: remember, per kernel and
// device, the largest size that was already granted
static int ov_allow_lds(const void* kern, size_t bytes) {
    struct Key { const void* k; int dev; size_t bytes; };
    static thread_local Key cache[64];
    static thread_local int n = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < n; ++i)
        if (cache[i].k == kern && cache[i].dev == dev) {
            if (cache[i].bytes >= bytes) return ONTOVAE_OK;
            OV_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            cache[i].bytes = bytes;
            return ONTOVAE_OK;
        }
    OV_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (n < 64) cache[n++] = Key{kern, dev, bytes};
    return ONTOVAE_OK;
}

template <int MT, int NT, int KW, int UK, bool A_KC, bool B_KC, int EPI>
static int ov_launch_gemm(const OvGemm& g, int S, bool vec, hipStream_t st) {
    const int bm = KW == 1 ? 64 * MT : 32 * MT, bn = KW == 1 ? 64 * NT : 32 * NT;
    dim3 grid((g.N + bn - 1) / bn, (g.M + bm - 1) / bm, S);
    size_t lds = 0;
    if (KW == 4) lds = (size_t)4 * 16 * 64 * sizeof(float);
    if (!B_KC && g.bgather) lds = std::max(lds, (size_t)std::min(g.K, g.kchunk) * sizeof(int32_t));
    if (EPI == OV_EPI_ADAMW_T) {
        if (KW != 1) { ontovae_set_error("gemm: the transposed AdamW epilogue needs KW == 1"); return ONTOVAE_EINVAL; }
        const size_t rows_floats = (!B_KC && g.bgather) ? (size_t)((std::min(g.K, g.kchunk) + 3) / 4 * 4) : 0;
        lds = (rows_floats + (size_t)4 * (32 * MT) * (32 * NT + 4)) * sizeof(float);
    }
    if (KW == 4 && !B_KC && g.bgather) { ontovae_set_error("gemm: in-block split-K cannot be combined with a gathered K"); return ONTOVAE_EINVAL; }
    if (lds > 160 * 1024) { ontovae_set_error("gemm: %zu bytes of LDS needed", lds); return ONTOVAE_EINVAL; }
#define OV_GL(V)                                                                                                  \
    do {                                                                                                          \
        auto kern = ov_gemm_kernel<MT, NT, KW, UK, A_KC, B_KC, V, EPI>;                                            \
        if (lds > 48 * 1024)                                                                                      \
            { int rc_ = ov_allow_lds(reinterpret_cast<const void*>(kern), lds); if (rc_) return rc_; } \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g);                                                    \
    } while (0)
    if (vec) OV_GL(true); else OV_GL(false);
#undef OV_GL
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

template <int WM, int WN, bool B_KC, int EPI>
static int ov_launch_gemm_lds(const OvGemm& g, int S, hipStream_t st) {
    constexpr int RA = 96 * WM, RB = 32 * WN;
    dim3 grid((g.N + RB - 1) / RB, (g.M + RA - 1) / RA, S);
    size_t lds = (size_t)RA * 36 * sizeof(float) + (B_KC ? (size_t)RB * 36 : (size_t)32 * (RB + 4)) * sizeof(float);
    if (!B_KC) lds += (size_t)std::min(g.K, g.kchunk) * sizeof(int32_t);
    if (lds > 160 * 1024) { ontovae_set_error("gemm_lds: %zu bytes of LDS needed", lds); return ONTOVAE_EINVAL; }
    auto kern = ov_gemm_lds_kernel<WM, WN, B_KC, EPI>;
    if (lds > 48 * 1024)
        { int rc_ = ov_allow_lds(reinterpret_cast<const void*>(kern), lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// the two-stage form of the encoder-forward product (ov_gemm_lds2_kernel)
template <int EPI, int WM>
static int ov_launch_gemm_lds2(const OvGemm& g, int S, hipStream_t st) {
    dim3 grid((g.N + 127) / 128, (g.M + 96 * WM - 1) / (96 * WM), S);
    const size_t lds = (size_t)2 * (96 * WM + 128) * 36 * sizeof(float);
    auto kern = ov_gemm_lds2_kernel<EPI, WM>;
    { int rc_ = ov_allow_lds(reinterpret_cast<const void*>(kern), lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(kern, grid, dim3(256 * WM), lds, st, g);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

static inline bool ov_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// dW0 with the x fragment resident in registers (K <= 128, dh rows 16-B aligned)
template <int EPI>
static int ov_launch_dw0(const OvGemm& g, hipStream_t st) {
    constexpr int NW = 8;
    if ((g.lda & 3) || (g.a_blm & 3) || (g.a_rstride & 3) || !ov_al16(g.A) || (int64_t)g.M * g.ldc * 4 >= (int64_t)1 << 31) {
        ontovae_set_error("dw0: needs 16-B aligned dh rows and a parameter below 2 GiB");
        return ONTOVAE_EINVAL;
    }
    const int njobs = (g.M + 127) / 128;
    const int gxn = (g.N + 32 * NW - 1) / (32 * NW);
    const int blocks = (gxn + 7) / 8 * 8 * njobs;
    size_t lds = (size_t)128 * 132 * sizeof(float) + 128 * sizeof(int64_t);
    auto kern = (g.K <= 128 && g.a_blm == 0) ? ov_dw0_kernel<NW, EPI> : ov_dw0_multi_kernel<NW, EPI>;
    // more than one chunk of the batch: the pipelined kernel, while the whole batch's row indices fit LDS beside its two slabs
    const size_t lds_pipe = (size_t)2 * 128 * 68 * sizeof(float) + (size_t)((g.K + 63) / 64 * 64) * sizeof(int64_t);
    if (!(g.K <= 128 && g.a_blm == 0) && lds_pipe <= (size_t)160 * 1024 - 512) { kern = ov_dw0_pipe_kernel<NW, EPI>; lds = lds_pipe; }
    { int rc_ = ov_allow_lds(reinterpret_cast<const void*>(kern), lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), lds, st, g, njobs);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

struct OvStep {
    const ontovae_ctx_t* c;
    OvWs ws;
    int B, ldb, vec, nchunk;
    hipStream_t st;
    const float* x;
    const int32_t* xidx;
    int64_t xld;
    int heads_S;             // split-K slabs the heads GEMM left in ws.slab (0: mu / log_var are already in ws.mulv)
    struct OvSide* side;     // internal second stream of this device, or NULL (ONTOVAE_FLAG_SINGLE_STREAM / creation failed)
    float* fm_out;           // screen passes: feature-major activity matrix the term levels write their nu-means to, or NULL
    int64_t fm_ld;
    bool want_xT;            // bf16 variant: the encoder's conversion pass also writes the transposed copy of x (training steps)
    // training, column-block kernels: the gene layer's weight gradient is formed inside the gene-layer kernels themselves
    // (bias partials in the forward kernel, link partials in the backward kernel); NULL = not wanted
    float* gw_part; float* gw_bpart; int gw_rows, gw_row0;
    // ... and the term layers' inside the per-level backward kernels (large batches; set by ov_train_tail_a), one partial copy per chunk of columns
    float* tw_part; float* tw_bpart;
    bool gene_lds;           // the gene layer's forward of this pass is ov_gene_lds_kernel (entry points that compute the loss set it)
    OvTermDrop tdrop;        // dropout on the term layers' outputs (OntoEncoder training passes); rate 0 everywhere else
};

// Independent branches of a step run on a library-owned second stream, forked from and joined back into
// the caller's stream with events, so the caller still sees one stream-ordered operation:
//   forward : packed decoder weight refresh            || encoder
//   backward: loss, decoder weight gradient, dWh/dbh   || latent backward, da, BatchNorm backward, dW0
// One side stream and its events per device, created on first use and kept until the process ends.
struct OvSide {
    hipStream_t st = nullptr;
    hipEvent_t fork = nullptr, mid = nullptr, join = nullptr;
    bool tried = false, ok = false;
};
static OvSide g_side[64];

static OvSide* ov_side_get(const ontovae_ctx_t* c) {
    if (!(c->flags & ONTOVAE_FLAG_SIDE_STREAM)) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    OvSide& s = g_side[dev];
    static std::mutex init_mutex;
    std::lock_guard<std::mutex> lock(init_mutex);
    if (!s.tried) {
        s.tried = true;
        s.ok = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.mid, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess;
        if (!s.ok) (void)hipGetLastError();
    }
    return s.ok ? &s : nullptr;
}
// side stream starts after everything enqueued on the caller's stream so far
static int ov_side_fork(OvStep& s) {
    if (!s.side) return ONTOVAE_OK;
    OV_HIP_CHECK(hipEventRecord(s.side->fork, s.st));
    OV_HIP_CHECK(hipStreamWaitEvent(s.side->st, s.side->fork, 0));
    return ONTOVAE_OK;
}
// side stream additionally waits for what the caller's stream holds now
static int ov_side_sync_mid(OvStep& s) {
    if (!s.side) return ONTOVAE_OK;
    OV_HIP_CHECK(hipEventRecord(s.side->mid, s.st));
    OV_HIP_CHECK(hipStreamWaitEvent(s.side->st, s.side->mid, 0));
    return ONTOVAE_OK;
}
// caller's stream waits for the side stream's work
static int ov_side_join(OvStep& s) {
    if (!s.side) return ONTOVAE_OK;
    OV_HIP_CHECK(hipEventRecord(s.side->join, s.side->st));
    OV_HIP_CHECK(hipStreamWaitEvent(s.st, s.side->join, 0));
    return ONTOVAE_OK;
}
static inline hipStream_t ov_side_stream(const OvStep& s) { return s.side ? s.side->st : s.st; }

// adam != NULL: g.C holds the weights, Pm / Pv their moments, and the product is applied to them as a gradient (AdamW epilogue)
// two_stage: ov_bf16_gemm2_kernel (two workgroups per CU, loads two K-steps ahead) instead of ov_bf16_gemm_tn_kernel (one LDS stage, up to
// four workgroups per CU).  Measured at 8192 rows: the encoder-forward product (K = 20 000 genes, every workgroup streams x) 205 us with
// the first, 255 with the second -- its short 64-deep steps are bound by load latency, which more workgroups per CU hide better than two
// steps of prefetch; dW0 (K = batch) 205 + 33 us of AdamW pass with the first, 198 us with the second and its AdamW epilogue.
static int ov_bf16_gemm(const OvBfGemm& g, int S, hipStream_t st, const OvAdam* adam = nullptr, float* Pm = nullptr, float* Pv = nullptr,
                        bool two_stage = false) {
    if (g.M < 1 || g.N < 1 || g.K < 1) return ONTOVAE_OK;
    if ((g.K % 8) || (g.lda % 8) || (g.ldb % 8) || (g.kchunk % OV_BF_BK) || !ov_al16(g.A) || !ov_al16(g.B)) {
        ontovae_set_error("bf16 gemm: K = %d, lda = %lld, ldb = %lld, kchunk = %d or operand alignment not supported", g.K, (long long)g.lda,
                          (long long)g.ldb, g.kchunk);
        return ONTOVAE_EINVAL;
    }
    const OvBfGemm2 q{g, Pm, Pv, adam ? *adam : OvAdam{}};
    const size_t lds = (size_t)2 * (OV_BF_BM + OV_BF_BN) * OV_BF_LD * sizeof(uint16_t);
    const dim3 grid((g.N + OV_BF_BN - 1) / OV_BF_BN, (g.M + OV_BF_BM - 1) / OV_BF_BM, S);
    if (adam) {
        if (S != 1 || !Pm || !Pv) { ontovae_set_error("bf16 gemm: the AdamW epilogue needs one K slab and the moment buffers"); return ONTOVAE_EINVAL; }
        { int rc_ = ov_allow_lds(reinterpret_cast<const void*>(ov_bf16_gemm2_kernel<1>), lds); if (rc_) return rc_; }
        hipLaunchKernelGGL(ov_bf16_gemm2_kernel<1>, grid, dim3(256), lds, st, q);
    } else if (two_stage) {
        { int rc_ = ov_allow_lds(reinterpret_cast<const void*>(ov_bf16_gemm2_kernel<0>), lds); if (rc_) return rc_; }
        hipLaunchKernelGGL(ov_bf16_gemm2_kernel<0>, grid, dim3(256), lds, st, q);
    } else {
        hipLaunchKernelGGL(ov_bf16_gemm_tn_kernel, grid, dim3(256), 0, st, g);
    }
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// bf16 variant of ov_enc_gemm: W0 -> wb, batch rows of x -> xb (+ xT for the dW0 product of a training step), then the product
static int ov_enc_gemm_bf16(OvStep& s, int* S_out) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    OvBfWs bw;
    ov_bf16_carve(d, c->max_batch, c->bf16_ws, &bw);
    OvProfScope prof(ONTOVAE_PHASE_ENC_GEMM, s.st);
    const int64_t nw = (int64_t)d.L * d.G;
    hipLaunchKernelGGL(ov_bf16_convert_kernel, dim3((unsigned)std::min<int64_t>((nw / 8 + 255) / 256, 4096)), dim3(256), 0, s.st,
                       c->P + c->lay.W0, bw.wb, nw);
    OV_LAUNCH_CHECK();
    hipLaunchKernelGGL(ov_bf16_x_kernel, dim3((d.G + 63) / 64, s.ldb / 64), dim3(256), 0, s.st, s.x, s.xidx, s.xld, s.B, d.G, bw.xb,
                       s.want_xT ? bw.xT : (uint16_t*)nullptr, (int64_t)s.ldb);
    OV_LAUNCH_CHECK();
    const int tiles = ((d.L + OV_BF_BM - 1) / OV_BF_BM) * ((s.B + OV_BF_BN - 1) / OV_BF_BN);
    int S = std::max(1, std::min(std::min((512 + tiles - 1) / tiles, 64), std::max(1, d.G / 512)));
    // up to 4096 columns the two-stage kernel with about one workgroup per CU (two from 64 tiles on) is ahead of the one-stage
    // kernel at its four per CU (1024 columns: 36.7 against 46.8 us, 2048: 70.6 / 80.8, 4096: 115 / ~135); at 8192 it is behind
    // (242-257 against 228: see ov_bf16_gemm)
    const bool two_stage = tiles <= 96;
    if (two_stage) S = std::max(1, std::min((tiles < 64 ? 256 : 512) / tiles, std::max(1, d.G / 512)));
    S = (int)std::min<int64_t>(S, std::max<int64_t>(1, s.ws.slab_floats / ((int64_t)d.L * s.ldb)));
    const int kchunk = ((d.G + S - 1) / S + OV_BF_BK - 1) / OV_BF_BK * OV_BF_BK;
    const int Seff = (d.G + kchunk - 1) / kchunk;
    *S_out = Seff;
    const OvBfGemm g{bw.wb, d.G, bw.xb, d.G, s.ws.slab, s.ldb, (int64_t)d.L * s.ldb, d.L, s.B, d.G, kchunk};
    return ov_bf16_gemm(g, Seff, s.st, nullptr, nullptr, nullptr, two_stage);
}

// x W0^T -> split-K slabs  (modules.py:46)
static int ov_enc_gemm(OvStep& s, int* S_out) {
    if (s.c->flags & ONTOVAE_FLAG_BF16_GEMM) return ov_enc_gemm_bf16(s, S_out);
    const ontovae_dims_t& d = s.c->dims;
    const int S = ov_enc_splitk(d.L, s.B, d.G);
    int kchunk = (d.G + S - 1) / S;
    kchunk = (kchunk + 31) / 32 * 32;
    const int Seff = (d.G + kchunk - 1) / kchunk;
    OvGemm g{};
    g.M = d.L; g.N = s.B; g.K = d.G;
    g.A = s.c->P + s.c->lay.W0; g.lda = d.G;
    g.B = s.x; g.ldb = s.xld; g.bgather = s.xidx;
    g.C = s.ws.slab; g.ldc = s.ldb; g.slab_stride = (int64_t)d.L * s.ldb; g.kchunk = kchunk;
    const bool vec = (d.G % 4 == 0) && (s.xld % 4 == 0) && ov_al16(g.A) && ov_al16(g.B);
    *S_out = Seff;
    g.dbg = OV_DBG((s.c->flags >> 18) & 3);
    OvProfScope prof(ONTOVAE_PHASE_ENC_GEMM, s.st);
    const int nt = ov_enc_nt(s.B);
    // two LDS stages, one barrier per K-step (8192 columns: 1058 -> 1015 us, 1024: 166 -> 162, 512: 84 -> 82, 128: 26.9 -> 26.5, same box)
    // from 4096 columns (where the split-K count no longer has to make up for the halved number of workgroups): 192 x 128 tiles, 8 waves
    // sharing one stage of x (same box: 1047-1057 -> 1024-1030 us at 8192 columns, 2080-2112 -> 2020-2035 at 16384)
    if (vec && nt == 4 && s.ldb >= 4096 && !OV_DBG((s.c->flags >> 20) & 1)) return ov_launch_gemm_lds2<OV_EPI_STORE, 2>(g, Seff, s.st);
    if (vec && nt == 4 && !OV_DBG((s.c->flags >> 20) & 1)) return ov_launch_gemm_lds2<OV_EPI_STORE, 1>(g, Seff, s.st);
    if (nt == 1) return ov_launch_gemm<3, 1, 4, 2, true, true, OV_EPI_STORE>(g, Seff, vec, s.st);
    if (nt == 2) return ov_launch_gemm<3, 2, 4, 1, true, true, OV_EPI_STORE>(g, Seff, vec, s.st);
    if (OV_DBG((s.c->flags >> 14) & 1)) return ov_launch_gemm<3, 4, 4, 2, true, true, OV_EPI_STORE>(g, Seff, vec, s.st);
    return ov_launch_gemm<3, 4, 4, 1, true, true, OV_EPI_STORE>(g, Seff, vec, s.st);
}

// mode 5: S = number of rank blocks in `stats`, each 2 L + 1 floats
static int ov_bn_fwd(OvStep& s, int S, int mode, int Bglobal, float* stats, uint64_t rng_off) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    OvProfScope prof(ONTOVAE_PHASE_BN_FWD, s.st);
    if (mode == 0 && S <= 16 && s.ldb >= 1024) {
        hipLaunchKernelGGL(ov_bn_eval_kernel, dim3((s.ldb / 4 + 255) / 256, d.L), dim3(256), 0, s.st, s.ws.slab, S, (int64_t)d.L * s.ldb,
                           c->P + c->lay.b0, c->P + c->lay.gamma, c->P + c->lay.beta, c->bn_rmean, c->bn_rvar, s.ws.a, s.B, s.ldb);
        OV_LAUNCH_CHECK();
        return ONTOVAE_OK;
    }
    if (mode != 0 && s.ldb >= 1024 && (mode == 3 || mode == 5 || S <= 16)) {
        hipLaunchKernelGGL(ov_bn_fwd_wide_kernel, dim3(d.L), dim3(256), 0, s.st, s.ws.slab, S, mode == 5 ? (int64_t)(2 * d.L + 1) : (int64_t)d.L * s.ldb,
                           c->P + c->lay.b0, c->P + c->lay.gamma, c->P + c->lay.beta, c->bn_rmean, c->bn_rvar, c->bn_nbt,
                           s.ws.xhat, s.ws.a, s.ws.bn_mean, s.ws.bn_invstd, stats, d.L, s.B, Bglobal, s.ldb, mode, c->drop, c->seed,
                           rng_off);
        OV_LAUNCH_CHECK();
        return ONTOVAE_OK;
    }
    hipLaunchKernelGGL(ov_bn_fwd_kernel, dim3(d.L), dim3(256), 0, s.st, s.ws.slab, S, mode == 5 ? (int64_t)(2 * d.L + 1) : (int64_t)d.L * s.ldb,
                       c->P + c->lay.b0, c->P + c->lay.gamma, c->P + c->lay.beta, c->bn_rmean, c->bn_rvar, c->bn_nbt,
                       s.ws.xhat, s.ws.a, s.ws.bn_mean, s.ws.bn_invstd, stats, d.L, s.B, Bglobal, s.ldb, mode, c->drop, c->seed,
                       rng_off);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// heads: [mu; log_var] = Wh a + bh  (modules.py:57, 62), feature-major
static int ov_heads_fwd(OvStep& s) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    OvGemm g{};
    g.M = 2 * d.L; g.N = s.B; g.K = d.L;
    g.A = c->P + c->lay.Wh; g.lda = d.L;
    g.B = s.ws.a; g.ldb = s.ldb; g.bgather = nullptr;
    // split-K partials into the (now free) encoder slab buffer; ov_reparam adds them and the bias
    const int64_t one = (int64_t)2 * d.L * s.ldb;
    if (s.ws.slab_floats < one) {
        ontovae_set_error("heads GEMM: slab buffer holds %lld floats, [2L, ldb] needs %lld", (long long)s.ws.slab_floats, (long long)one);
        return ONTOVAE_EINVAL;
    }
    // training batches: 3 split-K slabs; inference batches (96 x 128 tiles of the LDS-staged kernel): as many slabs as it takes
    // to give every CU a workgroup (64 tiles at 1024 columns, 35 -> us), none once the tiles alone fill the chip
    const int tiles_lds = ((2 * d.L + 95) / 96) * ((s.B + 127) / 128);
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>(s.ldb >= 1024 ? std::min(4, 256 / std::max(tiles_lds, 1)) : 3,
                                                             s.ws.slab_floats / one));
    int kchunk = ((d.L + S - 1) / S + 31) / 32 * 32;
    g.C = s.ws.slab; g.ldc = s.ldb; g.slab_stride = one; g.kchunk = kchunk;
    s.heads_S = (d.L + kchunk - 1) / kchunk;
    const bool vec = (d.L % 4 == 0) && ov_al16(g.A);
    OvProfScope prof(ONTOVAE_PHASE_HEADS, s.st);
    // inference batches: LDS-staged 96 x 128 tiles (132 -> 98 us at 16384 columns, 46 -> 39 at 4096)
    if (s.ldb >= 1024 && vec && ov_al16(g.B)) return ov_launch_gemm_lds<1, 4, false, OV_EPI_STORE>(g, s.heads_S, s.st);
    return ov_launch_gemm<1, 1, 4, 4, true, false, OV_EPI_STORE>(g, s.heads_S, vec, s.st);
}

static int ov_fused_bt(const ontovae_ctx_t* c);
#define OV_CHAIN_MAX_LDB 2048            // backward, and forward passes whose level kernels write the activities
#define OV_CHAIN_FWD_MAX_LDB 4096        // every other forward pass
// below this many columns the gene backward also writes the panel rows of D column-group-major for the backward chain (us per step with /
// without that copy: 185.3 / 191.9 at 128 columns, 279.1 / 283.1 at 256, 451.0 / 448.6 at 512, 799.8 / 777.2 at 1024: the 16-B pieces
// cost the gene backward 23 us to write at 1024 columns and the chain 6 us to read from the feature-major matrix instead)
#define OV_BLK_MAX_LDB 512
// the forward pass of this step runs the fused term chain: the gene layer's copy is then refreshed by riders of that launch
// (25 us on a fraction of the chip) instead of in front of it
static bool ov_gene_refresh_rides(const OvStep& s) {
    const int bt = ov_fused_bt(s.c);
    if (s.tdrop.rate > 0.f) return false;       // term-layer dropout: per-level kernels, no chain launch to ride in
    return bt != 0 && s.ldb / bt <= 128;        // the chain and its replicas leave half of the CUs to riders
}
static void ov_gene_refresh_args(const OvStep& s, OvRefresh* r) {
    const ontovae_ctx_t* c = s.c;
    r->vals = c->P + c->lay.dec_vals;
    if (s.gene_lds) { r->kg = c->gsell.k; r->wg = c->gsell.w; r->ng = c->gsell.n_quads * 256; return; }
    r->kg = c->row_k + c->dims.nnz_term; r->wg = s.ws.wgene; r->ng = (int)(c->dims.nnz - c->dims.nnz_term);
}
// arguments of the refresh of the re-ordered weight copies: SELL order of the term layers (only if this model uses the fused
// chain), CSR order of the gene layer (every forward gene kernel reads it; here unless it rides in the chain's launch)
static int ov_refresh_args(const OvStep& s, OvRefresh* r) {
    const ontovae_ctx_t* c = s.c;
    *r = OvRefresh{};
    r->vals = c->P + c->lay.dec_vals; r->bias = c->P + c->lay.dec_bias;
    if (ov_fused_bt(c) != 0) {
        const ontovae_sell_t& f = c->sell_fwd;
        const ontovae_sell_t& b = c->sell_bwd;
        r->kf = f.k; r->wf = f.w; r->nf = f.n_chunks * 512;
        r->nodef = f.node; r->biasf = f.bias; r->nbf = f.n_slices * 64;
        r->kb = b.k; r->wb = b.w; r->nb = b.n_chunks * 512;
    }
    if (!ov_gene_refresh_rides(s)) ov_gene_refresh_args(s, r);
    return std::min(512, (std::max(std::max(r->nf, r->nb), r->ng) + 255) / 256);
}

// with_refresh: the launch also refreshes the packed term-layer weights (callers then skip ov_sell_refresh)
static int ov_reparam(OvStep& s, const float* eps, int train, uint64_t rng_off, uint32_t eps_stream, int write_kl, int* nkl,
                      bool with_refresh) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    const int64_t n = (int64_t)d.L * s.ldb;
    // partial KL sums go to ws.kl_part, one per block (ov_carve: G / 4 * ldb / 64 + 1024 floats)
    const int64_t npart = (int64_t)((d.G + 3) / 4) * (s.ldb / 64) + 1024;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, n >= (1 << 20) ? std::min<int64_t>(4096, npart) : 512);
    float* zrows = s.ws.act + (int64_t)(d.C - d.L) * s.ldb;
    OvRefresh rf{};
    const int rblocks = with_refresh ? ov_refresh_args(s, &rf) : 0;
    OvProfScope prof(ONTOVAE_PHASE_REPARAM, s.st);
    // eps is kept for the backward pass only
    hipLaunchKernelGGL(ov_reparam_kernel, dim3(blocks + rblocks), dim3(256), 0, s.st, s.ws.mulv, eps, train ? s.ws.epsw : nullptr, zrows, s.ws.kl_part,
                       d.L, s.B, s.ldb, train, c->z_drop, c->seed, rng_off, eps_stream, write_kl, blocks, rf,
                       s.heads_S > 0 ? s.ws.slab : (const float*)nullptr, s.heads_S, (int64_t)2 * d.L * s.ldb, c->P + c->lay.bh);
    OV_LAUNCH_CHECK();
    s.heads_S = 0;               // mu / log_var are in ws.mulv from here on
    if (nkl) *nkl = blocks;
    return ONTOVAE_OK;
}

template <int GENE>
static int ov_dec_fwd_level(OvStep& s, int nb, int ne, float* rec_out, int64_t rec_ld) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    if (ne <= nb) return ONTOVAE_OK;
    dim3 grid((ne - nb + 3) / 4, s.nchunk);
    // the gene rows' links are walked in CSR order: their weights come from the CSR-ordered copy (ov_refresh_args)
    const float* vals = GENE ? s.ws.wgene - d.nnz_term : c->P + c->lay.dec_vals;
    const float* bias = c->P + c->lay.dec_bias;
    OvProfScope prof(GENE ? ONTOVAE_PHASE_DEC_FWD_GENE : ONTOVAE_PHASE_DEC_FWD_TERMS, s.st);
    if (GENE == 0 && s.fm_out && s.tdrop.rate <= 0.f) {          // ov_fm_fused(s) held: see ontovae_pass_data_perturbed
        int c_base = 0;
        for (int lvl = 0; lvl < d.K && c->level_begin[lvl] != nb; ++lvl) c_base += (c->level_end[lvl] - c->level_begin[lvl]) / d.nu;
        const dim3 tgrid(((ne - nb) / d.nu + 3) / 4, s.nchunk);
#define OV_DT(NU) hipLaunchKernelGGL((ov_dec_fwd_term_kernel<4, NU>), tgrid, dim3(256), 0, s.st, c->rowptr, c->colidx, vals, bias, s.ws.act, nb, ne, s.B, s.ldb, s.fm_out, s.fm_ld, c_base)
        if (d.nu == 2) OV_DT(2); else if (d.nu == 3) OV_DT(3); else OV_DT(4);
#undef OV_DT
        OV_LAUNCH_CHECK();
        return ONTOVAE_OK;
    }
#define OV_DF(V)                                                                                                        \
    hipLaunchKernelGGL((ov_dec_fwd_kernel<V, GENE>), grid, dim3(256), 0, s.st, c->rowptr, c->colidx, vals, bias, s.ws.act, \
                       nb, ne, s.B, s.ldb, s.x, s.xidx, d.G, s.ws.d, s.ws.rec_part, rec_out, rec_ld, d.C, s.tdrop)
    if (s.vec == 1) OV_DF(1); else if (s.vec == 2) OV_DF(2); else OV_DF(4);
#undef OV_DF
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// batch columns per workgroup of the fused term-level kernels: the [C][BT] fp32 panel must fit LDS
static int ov_fused_bt(const ontovae_ctx_t* c) {
    const ontovae_dims_t& d = c->dims;
    if (d.K > OV_MAX_LEVELS || d.K < 2 || (c->flags & ONTOVAE_FLAG_LEVEL_LAUNCHES)) return 0;
    const ontovae_sell_t& f = c->sell_fwd;
    const ontovae_sell_t& b = c->sell_bwd;
    if (f.n_slices < 1 || b.n_slices < 1 || f.n_levels != d.K - 1 || b.n_levels != d.K || !f.node || !f.info || !f.chunk0 ||
        !f.idx || !f.k || !f.w || !f.bias || !b.node || !b.info || !b.chunk0 || !b.idx || !b.k || !b.w || !f.level_first ||
        !f.level_count || !b.level_first || !b.level_count)
        return 0;
    OvSteps tmp;
    if (!ov_make_steps(f, &tmp) || !ov_make_steps(b, &tmp)) return 0;
    const int64_t lds = 160 * 1024 - 2048;      // the panel, minus the static LDS of the kernels that carry the chain
    if ((int64_t)d.C * 16 <= lds) return 4;
    if ((int64_t)d.C * 8 <= lds) return 2;
    if ((int64_t)d.C * 4 <= lds) return 1;
    return 0;
}

template <typename KernelT>
static int ov_set_lds(KernelT kernel, size_t bytes) {
    return ov_allow_lds(reinterpret_cast<const void*>(kernel), bytes);
}

// SELL weight copies <- parameter buffer; called once per pass before the first chain kernel
static int ov_sell_refresh(OvStep& s, hipStream_t st) {
    OvRefresh rf{};
    const int blocks = ov_refresh_args(s, &rf);
    if (blocks == 0) return ONTOVAE_OK;
    OvProfScope prof(ONTOVAE_PHASE_SELL_REFRESH, st);
    hipLaunchKernelGGL(ov_sell_refresh_kernel, dim3(blocks), dim3(256), 0, st, rf);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// replicas of every column group's chain workgroup (ov_dec_chain_body): up to half of the CUs (the riders of the chain
// launches take the rest; every workgroup of such a launch reserves the panel's LDS, so it has a CU to itself), at most 4
// (8 buy nothing more)
// (a power of two: the kernel picks a replica's slices with a mask, not a division)
static int ov_chain_replicas(int nchain) { const int r = std::max(1, std::min(4, 128 / std::max(nchain, 1))); return r >= 4 ? 4 : (r >= 2 ? 2 : 1); }

// refreshed = the packed weight copies were already refreshed for this pass (on the side stream, joined)
static int ov_dec_forward_terms(OvStep& s, bool refreshed = false) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    // The chain kernel keeps a [C][bt] panel in LDS, so a workgroup carries 4 columns and a large batch becomes many rounds
    // of latency-bound workgroups; from 4096 columns on one launch per level over the whole batch is faster
    // (us, chain against levels: 31 / 75 at 1024 columns, 53 / 70 at 2048, 117 / 105 at 4096), from OV_CHAIN_MAX_LDB columns
    // when the level kernels also write the activities (ov_fm_fused)
    // (dropout on the term layers' outputs lives in the per-level kernels only)
    const int chain_max = s.fm_out ? OV_CHAIN_MAX_LDB : OV_CHAIN_FWD_MAX_LDB;
    const int bt = (s.ldb >= chain_max || s.tdrop.rate > 0.f) ? 0 : ov_fused_bt(c);
    int rc;
    if (!refreshed && (rc = ov_sell_refresh(s, s.st))) return rc;      // also the gene layer's CSR-ordered copy, whatever the path
    if (bt == 0) {
        for (int lvl = 1; lvl < d.K; ++lvl)
            if ((rc = ov_dec_fwd_level<0>(s, c->level_begin[lvl], c->level_end[lvl], nullptr, 0))) return rc;
        return ONTOVAE_OK;
    }
    const ontovae_sell_t& f = c->sell_fwd;
    OvSteps lv;
    ov_make_steps(f, &lv);
    const size_t lds = (size_t)d.C * bt * sizeof(float);
    OvProfScope prof(ONTOVAE_PHASE_DEC_FWD_TERMS, s.st);
    const int nrep = ov_chain_replicas(s.ldb / bt);
    const OvChainArgs ca{f.node, f.info, f.chunk0, f.idx, f.w, f.bias, s.ws.act, nullptr, s.ldb, d.C - d.L, d.C, 0, d.C - d.L,
                         (int)OV_DBG((c->flags >> 8) & 15), nrep};
    OvRefresh rg{};
    ov_gene_refresh_args(s, &rg);
    const int nchain = s.ldb / bt * nrep;
    const int rblocks = ov_gene_refresh_rides(s) ? std::max(1, std::min(96, (rg.ng + 4095) / 4096)) : 0;      // else done with the SELL copies
#define OV_TF2(BT, SG)                                                                                                \
    if ((rc = ov_set_lds(ov_chain_fwd_refresh_kernel<BT, SG>, lds))) return rc;                                        \
    hipLaunchKernelGGL((ov_chain_fwd_refresh_kernel<BT, SG>), dim3(nchain + rblocks), dim3(1024), lds, s.st, ca, lv, nchain, rg)
#define OV_TF(BT) do { if (f.all_single_chunk) { OV_TF2(BT, true); } else { OV_TF2(BT, false); } } while (0)
    if (bt == 4) { OV_TF(4); } else if (bt == 2) { OV_TF(2); } else { OV_TF(1); }
#undef OV_TF
#undef OV_TF2
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// geometry of ov_gene_lds_kernel for this pass
struct OvGeneLdsGeo { int ncg, nchunk, spc, nunits, grid; size_t lds; };
static OvGeneLdsGeo ov_gene_lds_geo(const OvStep& s) {
    const ontovae_gsell_t& g = s.c->gsell;
    OvGeneLdsGeo q;
    q.ncg = s.ldb / 4;
    // two chunks of slices where the column groups alone leave CUs idle (every chunk loads the whole panel)
    const int nchunk = (q.ncg < 192 && g.n_slices >= 128) ? 2 : 1;
    q.spc = (g.n_slices + nchunk - 1) / nchunk;
    q.nchunk = (g.n_slices + q.spc - 1) / q.spc;
    q.nunits = q.nchunk * (q.ncg / 8);
    q.grid = (q.nunits + 7) / 8 * 64;
    q.lds = (size_t)s.c->dims.C * 16 + (size_t)q.spc * sizeof(float);      // panel, sums of squares per slice
    return q;
}
// the loss-computing forward of the gene layer goes through the LDS-panel kernel
static bool ov_gene_cb_on(const OvStep& s);
static bool ov_gene_lds_on(const OvStep& s) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_gsell_t& g = c->gsell;
    if (c->flags & ONTOVAE_FLAG_CB_GENE_FWD) return false;
    if (!ov_gene_cb_on(s)) return false;
    if (g.n_slices < 1 || g.n_quads < 0 || !g.gene || !g.quad0 || !g.col || !g.k || !g.w) return false;
    if ((int64_t)g.n_slices * 64 > ov_gsell_lanes_max(c->dims.G) || c->dims.C > 65535) return false;
    // from 512 columns on: below, (column group, chunk) workgroups that each load the whole panel are too many for what they
    // then do, and the 32-column-block kernel is faster (27 against 23 us at 128 columns)
    if (s.ldb < 512) return false;
    return ov_gene_lds_geo(s).lds <= (size_t)160 * 1024 - 64;
}

// gene layer: tiled kernel when x can be read in aligned 16-B pieces, row-per-wave kernel otherwise
template <int MODE>
static int ov_dec_gene(OvStep& s, float* rec_out, int64_t rec_ld) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    if (MODE == 1 && s.gene_lds) {
        const ontovae_gsell_t& g = c->gsell;
        const OvGeneLdsGeo q = ov_gene_lds_geo(s);
        // the bias partials only where the training buffers hold this batch (a loss-only pass has no use for them)
        float* gb4 = s.gw_bpart ? s.ws.gb4 : nullptr;
        const OvGeneLds a{g.gene, g.quad0, reinterpret_cast<const uint2*>(g.col), reinterpret_cast<const f32x4*>(g.w), g.n_slices,
                          c->P + c->lay.dec_bias, s.ws.act, s.ws.d, s.B, s.ldb, d.C, s.x, s.xidx, s.xld, s.ws.rec_part, gb4,
                          g.n_slices * 64, q.ncg, q.spc, q.nunits};
        int rc;
        OvProfScope prof(ONTOVAE_PHASE_DEC_FWD_GENE, s.st);
        if ((rc = ov_set_lds(ov_gene_lds_kernel, q.lds))) return rc;
        hipLaunchKernelGGL(ov_gene_lds_kernel, dim3(q.grid), dim3(1024), q.lds, s.st, a);
        OV_LAUNCH_CHECK();
        return ONTOVAE_OK;
    }
    bool tiled = (d.G % 4 == 0) && !(c->flags & ONTOVAE_FLAG_ROW_GENE_LAYER);
    if (MODE == 1) tiled = tiled && ov_al16(s.x) && (s.xld % 4 == 0);
    if (MODE == 2) tiled = tiled && ov_al16(rec_out) && (rec_ld % 4 == 0);
    if (!tiled) return ov_dec_fwd_level<MODE>(s, d.C, d.C + d.G, rec_out, rec_ld);
    const float* vals = s.ws.wgene - d.nnz_term;            // CSR-ordered copy of the gene layer's weights (ov_refresh_args)
    const float* bias = c->P + c->lay.dec_bias;
    OvProfScope prof(ONTOVAE_PHASE_DEC_FWD_GENE, s.st);
    if (!(c->flags & ONTOVAE_FLAG_WIDE_GENE_TILES)) {
        const int ncb = s.ldb / 32, ntiles = (d.G + 31) / 32;
        hipLaunchKernelGGL((ov_gene_cb_kernel<MODE>), dim3(ov_cb_grid(ncb, ntiles)), dim3(256), 0, s.st, c->rowptr, c->colidx, vals, bias,
                           s.ws.act, s.B, s.ldb, s.x, s.xidx, d.G, s.ws.d, s.ws.rec_part, rec_out, rec_ld, d.C, ncb, ntiles,
                           MODE == 1 ? s.gw_bpart : (float*)nullptr, s.gw_rows, s.gw_row0,
                           // the plan's tile records up to 512 columns (21.2 against 22.9 us at 128); beyond, where every XCD works
                           // through many column blocks, the kernel is 3 to 8 % faster when its workgroups rank their rows themselves
                           s.ldb < 1024 ? reinterpret_cast<const int4*>(c->gene_tiles) : (const int4*)nullptr);
        OV_LAUNCH_CHECK();
        return ONTOVAE_OK;
    }
    dim3 grid((d.G + OV_GT_GENES - 1) / OV_GT_GENES, s.nchunk);
#define OV_GT(V)                                                                                                       \
    hipLaunchKernelGGL((ov_dec_gene_tile_kernel<V, MODE>), grid, dim3(512), 0, s.st, c->rowptr, c->colidx, vals, bias, s.ws.act, \
                       s.B, s.ldb, s.x, s.xidx, d.G, s.ws.d, s.ws.rec_part, rec_out, rec_ld, d.C)
    if (s.vec == 1) OV_GT(1); else if (s.vec == 2) OV_GT(2); else OV_GT(4);
#undef OV_GT
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// the training-mode gene layer goes through the column-block kernel (must match ov_dec_gene<1>)
static bool ov_gene_cb_on(const OvStep& s) {
    const ontovae_dims_t& d = s.c->dims;
    return (d.G % 4 == 0) && !(s.c->flags & (ONTOVAE_FLAG_ROW_GENE_LAYER | ONTOVAE_FLAG_WIDE_GENE_TILES)) && ov_al16(s.x) && (s.xld % 4 == 0);
}

// number of sum-of-squares partials the gene layer wrote (must match ov_dec_gene<1>)
static int ov_nrec(const OvStep& s) {
    const ontovae_dims_t& d = s.c->dims;
    if (s.gene_lds) return ov_gene_lds_geo(s).grid;
    const bool tiled = (d.G % 4 == 0) && !(s.c->flags & ONTOVAE_FLAG_ROW_GENE_LAYER) && ov_al16(s.x) && (s.xld % 4 == 0);
    if (tiled && !(s.c->flags & ONTOVAE_FLAG_WIDE_GENE_TILES)) return ov_cb_grid(s.ldb / 32, (d.G + 31) / 32);
    return (tiled ? (d.G + OV_GT_GENES - 1) / OV_GT_GENES : (d.G + 3) / 4) * s.nchunk;
}

static int ov_dec_bwd_level(OvStep& s, int nb, int ne, bool term_part_only = false) {
    const ontovae_ctx_t* c = s.c;
    if (ne <= nb) return ONTOVAE_OK;
    dim3 grid((ne - nb + 3) / 4, s.nchunk);
    const float* vals = c->P + c->lay.dec_vals;
    OvProfScope prof(ONTOVAE_PHASE_DEC_BWD, s.st);
    // the term layers' weight gradient rides here (s.tw_part, set by the training step from 4096 columns on): only with the term part of the lists and without dropout
    const bool tw_on = term_part_only && s.tw_part && s.tdrop.rate <= 0.f;
    const OvLvlWgrad tw{s.ws.act, tw_on ? s.tw_part : nullptr, s.tw_bpart, (int64_t)c->dims.nnz, s.gw_rows, c->dims.C - c->dims.L};
#define OV_DB(V) hipLaunchKernelGGL((ov_dec_bwd_kernel<V>), grid, dim3(256), 0, s.st, c->cscptr, c->csc_row, c->csc_k, vals, s.ws.d, nb, ne, s.ldb, \
                                    term_part_only ? c->csc_split : (const int32_t*)nullptr, s.tdrop, c->dims.C - c->dims.L, tw)
    if (s.vec == 1) OV_DB(1); else if (s.vec == 2) OV_DB(2); else OV_DB(4);
#undef OV_DB
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// the backward gene kernel of this pass is the term-tile form with its lists in LDS (needs s.gw_part: the weight gradient rides there)
static bool ov_bwd_gene_lds_form(const OvStep& s) {
    const ontovae_ctx_t* c = s.c;
    const int tp_mode = OV_DBG((c->flags >> 26) & 3);
    const size_t tl_lds = (size_t)c->tpanel_max_entries * (1 + c->dims.nu) * 4;
    if (c->flags & ONTOVAE_FLAG_WIDE_GENE_TILES) return false;
    if (s.tdrop.rate > 0.f || ov_fused_bt(c) == 0) return false;          // per-level path: ov_dec_backward returns before the gene kernels
    if (!c->tpanel_tiles || c->dims.nu < 1 || c->dims.nu > 3 || tp_mode == 1 || tp_mode == 2) return false;
    // without room for the accumulators (ov_bwd_gene_acc) only below 4096 columns: beyond, one partial copy per block made this form slower
    // than the block-major kernel (680 against 612 us at 4096 columns, 1386 / 1183 at 8192); with them it is 462 and 958 us
    const bool acc = tl_lds / (1 + c->dims.nu) * (1 + 2 * c->dims.nu) <= 52 * 1024 && !(c->flags & ONTOVAE_FLAG_PER_BLOCK_WGRAD);
    return c->tpanel_max_entries > 0 && tl_lds <= 48 * 1024 && ((s.ldb >= 512 && (s.ldb < 4096 || acc)) || tp_mode == 3);
}
// ... and adds the column blocks' weight-gradient dots up in LDS (three workgroups per CU must still fit: 160 KB)
static bool ov_bwd_gene_acc(const OvStep& s) {
    const size_t lds = (size_t)s.c->tpanel_max_entries * (1 + 2 * s.c->dims.nu) * 4;
    return ov_bwd_gene_lds_form(s) && lds <= 52 * 1024 && !(s.c->flags & ONTOVAE_FLAG_PER_BLOCK_WGRAD);
}
#define OV_RIDE_PCT 75      // share of the gene rows whose weight gradient rides in the backward chain's launch (flat from 55 to 100)
// wg: gene rows of the weight gradient to run in the same launch as the backward chain (fused path only), or NULL;
// *wg_done tells the caller whether that happened
struct OvFinishArgs;
// fin: finish pass of the gene layer's update to run in the backward chain's launch (fused path only), or NULL
static int ov_dec_backward(OvStep& s, const OvFinishArgs* fin = nullptr, int fblocks = 0) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    const int bt = s.tdrop.rate > 0.f ? 0 : ov_fused_bt(c);       // (term-layer dropout: per-level kernels only)
    int rc;
    if (bt == 0) {
        for (int lvl = d.K - 1; lvl >= 0; --lvl)
            if ((rc = ov_dec_bwd_level(s, c->level_begin[lvl], c->level_end[lvl]))) return rc;
        return ONTOVAE_OK;
    }
    const float* vals = c->P + c->lay.dec_vals;
    float* blk = nullptr;
    // as in the forward pass: from 4096 columns on, one launch per level over the whole batch instead of the LDS
    // chain (a workgroup of the chain carries 4 columns: many rounds of latency-bound workgroups)
    // (8192 columns: term part 270 -> 204 us; 2048: 77 us as a chain reading the feature-major D, 87 us per level)
    const bool per_level = s.ldb >= OV_CHAIN_FWD_MAX_LDB && !fin;
    if (!(c->flags & ONTOVAE_FLAG_WIDE_GENE_TILES)) {
        OvProfScope prof(ONTOVAE_PHASE_DEC_BWD_GENE, s.st);
        const int ncb = s.ldb / 32, ntiles = (d.C + 31) / 32;
        // the chain's workgroups take 4 columns each: hand them contiguous pieces while that copy is cheap to write
        if (bt == 4 && !per_level && s.ldb < OV_BLK_MAX_LDB) blk = s.ws.dblk;
        // waves per SIMD: the kernel needs 5 to cover its latencies while 4 column blocks are all there is (22.1 us at 128
        // columns against 24.7 with 4); once every XCD works through many blocks, fewer waves leave more of its L2 to each
        // (1024 columns: 170 us with 4 against 176 with 5; 8192: 1.40 ms with 3, 1.43 with 4, 1.52 with 5)
        const int wpe = s.ldb >= 4096 ? 3 : (s.ldb >= 1024 ? 4 : 5);
        // one lane group per TERM where the plan provides term tiles and the weight gradient rides here (the usual case):
        // a gene's D row is gathered once for the term's nu nodes
        const int tp_mode = OV_DBG((c->flags >> 26) & 3);       // measurement builds: 1 = per-node kernel even so
        // Which form (measured, GO-BP, us at 1024 / 2048 / 4096 / 8192 columns; per-node kernel 164 / 316 / 657 / 1263):
        //   term tiles, lists in LDS, a workgroup walks all blocks of its XCD : 158 / 291 / 680 / 1386  with one partial copy of the
        //                                                                      weight gradient per block,
        //                                                                      107 / 222 / 462 /  958  with the blocks added up in LDS
        //   term tiles, block-major as the per-node kernel                   :   - /   - / 612 / 1183
        // below 512 columns the per-node kernel stays (4 x more workgroups for the few blocks there are: 20.6 us at 128 columns).
        const size_t tl_lds = (size_t)c->tpanel_max_entries * (1 + d.nu) * 4;
        const bool term_ok = c->tpanel_tiles && s.gw_part && d.nu >= 1 && d.nu <= 3 && tp_mode != 1;
        if (term_ok && ov_bwd_gene_lds_form(s)) {
            const int tt = (d.C / d.nu + 31) / 32;
            const int4* tl = reinterpret_cast<const int4*>(c->tpanel_tiles);
            const int grid = ncb <= 8 ? ov_cb_grid(ncb, tt) : 8 * tt;
            const bool wacc = ov_bwd_gene_acc(s);
            const size_t lds = tl_lds + (wacc ? (size_t)c->tpanel_max_entries * d.nu * 4 : 0);
#define OV_BG_L2(N, W, A) do { if ((rc = ov_set_lds(ov_bwd_gene_term_lds_kernel<N, W, A>, lds))) return rc;                      \
                           hipLaunchKernelGGL((ov_bwd_gene_term_lds_kernel<N, W, A>), dim3(grid), dim3(256), lds, s.st, c->csc_row, vals, s.ws.d, blk, d.C, \
                           s.ldb, ncb, tt, s.ws.act, s.gw_part, (int64_t)d.nnz, tl, (int)tl_lds); } while (0)
#define OV_BG_L(N, W) do { if (wacc) OV_BG_L2(N, W, true); else OV_BG_L2(N, W, false); } while (0)
            if (d.nu == 1) OV_BG_L(1, 4); else if (d.nu == 2) OV_BG_L(2, 4); else OV_BG_L(3, 3);
#undef OV_BG_L
#undef OV_BG_L2
            OV_LAUNCH_CHECK();
        } else if (term_ok && (s.ldb >= 4096 || tp_mode == 2)) {
            const int tt = (d.C / d.nu + 31) / 32;
            const int4* tl = reinterpret_cast<const int4*>(c->tpanel_tiles);
#define OV_BG_T(N, W) hipLaunchKernelGGL((ov_bwd_gene_term_kernel<N, W>), dim3(ov_cb_grid(ncb, tt)), dim3(256), 0, s.st, c->csc_row, vals, s.ws.d, blk, d.C, \
                           s.ldb, ncb, tt, s.ws.act, s.gw_part, (int64_t)d.nnz, tl)
            if (d.nu == 1) OV_BG_T(1, 4); else if (d.nu == 2) OV_BG_T(2, 4); else OV_BG_T(3, 3);      // waves per SIMD the registers allow without spills
#undef OV_BG_T
            OV_LAUNCH_CHECK();
        } else {
#define OV_BG_CB(W) hipLaunchKernelGGL(ov_bwd_gene_cb_kernel<W>, dim3(ov_cb_grid(ncb, ntiles)), dim3(256), 0, s.st, c->cscptr, c->csc_split, c->csc_row, \
                           c->csc_k, vals, s.ws.d, blk, d.C, s.ldb, ncb, ntiles, s.ws.act, s.gw_part, (int64_t)d.nnz, \
                           reinterpret_cast<const int4*>(c->panel_tiles))
        if (wpe == 3) OV_BG_CB(3); else if (wpe == 4) OV_BG_CB(4); else OV_BG_CB(5);
#undef OV_BG_CB
        OV_LAUNCH_CHECK();
        }
    } else {
        OvProfScope prof(ONTOVAE_PHASE_DEC_BWD_GENE, s.st);
        dim3 grid((d.C + 3) / 4, s.nchunk);
#define OV_BG(V) hipLaunchKernelGGL((ov_dec_bwd_gene_kernel<V>), grid, dim3(256), 0, s.st, c->cscptr, c->csc_split, c->csc_row, c->csc_k, vals, s.ws.d, d.C, s.ldb)
        if (s.vec == 1) OV_BG(1); else if (s.vec == 2) OV_BG(2); else OV_BG(4);
#undef OV_BG
        OV_LAUNCH_CHECK();
    }
    if (per_level) {
        for (int lvl = d.K - 1; lvl >= 0; --lvl)
            if ((rc = ov_dec_bwd_level(s, c->level_begin[lvl], c->level_end[lvl], true))) return rc;
        return ONTOVAE_OK;
    }
    const ontovae_sell_t& sb = c->sell_bwd;
    OvSteps lv;
    ov_make_steps(sb, &lv);
    const size_t lds = (size_t)d.C * bt * sizeof(float);
    OvProfScope prof(ONTOVAE_PHASE_DEC_BWD, s.st);
    const int nrep = ov_chain_replicas(s.ldb / bt);
    const OvChainArgs ca{sb.node, sb.info, sb.chunk0, sb.idx, sb.w, nullptr, s.ws.d, blk, s.ldb, 0, d.C, 0, d.C,
                         (int)OV_DBG((c->flags >> 8) & 15), nrep};
    const int nchain = s.ldb / bt * nrep;
#define OV_TB2(BT, SG)                                                                                                \
    if (fin) {                                                                                                        \
        if ((rc = ov_set_lds(ov_chain_bwd_finish_kernel<BT, SG>, lds))) return rc;                                     \
        hipLaunchKernelGGL((ov_chain_bwd_finish_kernel<BT, SG>), dim3(nchain + fblocks), dim3(1024), lds, s.st, ca, lv,           \
                           nchain, *fin, fblocks);                                                                    \
    } else {                                                                                                          \
        if ((rc = ov_set_lds(ov_dec_chain_kernel<BT, true, SG>, lds))) return rc;                                      \
        hipLaunchKernelGGL((ov_dec_chain_kernel<BT, true, SG>), dim3(nchain), dim3(1024), lds, s.st, ca, lv);          \
    }
#define OV_TB(BT) do { if (sb.all_single_chunk) { OV_TB2(BT, true); } else { OV_TB2(BT, false); } } while (0)
    if (bt == 4) { OV_TB(4); } else if (bt == 2) { OV_TB(2); } else { OV_TB(1); }
#undef OV_TB
#undef OV_TB2
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

static int ov_loss_finalize(OvStep& s, hipStream_t st, int nrec, int nkl, float kc, float* loss_out) {
    OvProfScope prof(ONTOVAE_PHASE_LOSS, st);
    hipLaunchKernelGGL(ov_loss_finalize_kernel, dim3(1), dim3(1024), 0, st, s.ws.rec_part, nrec, s.ws.kl_part, nkl, kc, loss_out);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

static int ov_make_step(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int B, void* stream, OvStep* s) {
    s->c = ctx;
    ov_carve(ctx->dims, ctx->max_batch, ov_train_cap(ctx), ctx->workspace, &s->ws);
    s->B = B;
    s->ldb = ov_ldb(B);
    s->vec = ov_vec(B);
    s->nchunk = (s->ldb + 64 * s->vec - 1) / (64 * s->vec);
    s->st = reinterpret_cast<hipStream_t>(stream);
    s->x = x;
    s->xidx = xidx;
    s->xld = ctx->dims.G;
    s->heads_S = 0;
    s->side = nullptr;
    s->fm_out = nullptr;
    s->fm_ld = 0;
    s->want_xT = false;
    s->gw_part = nullptr; s->gw_bpart = nullptr; s->gw_rows = 0; s->gw_row0 = 0;
    s->tw_part = nullptr; s->tw_bpart = nullptr;
    s->gene_lds = false;
    s->tdrop = OvTermDrop{0.f, 0, 0};
    return ONTOVAE_OK;
}

// encoder (eval or fused-train) + heads + reparam: leaves mu/lv in ws.mulv, z in the panel
static int ov_encode(OvStep& s, const float* eps, int train, uint64_t rng_off, uint32_t eps_stream, int write_kl, int* nkl) {
    int S = 1, rc;
    if ((rc = ov_enc_gemm(s, &S))) return rc;
    if ((rc = ov_bn_fwd(s, S, train ? 1 : 0, s.B, nullptr, rng_off))) return rc;
    if ((rc = ov_heads_fwd(s))) return rc;
    return ov_reparam(s, eps, train, rng_off, eps_stream, write_kl, nkl, true);     // also refreshes the packed term-layer weights
}

static int ov_transpose_out(OvStep& s, const float* src, int F, float* dst, int64_t ld) {
    if (s.B >= 1024 && F >= 64) {
        hipLaunchKernelGGL(ov_transpose_out64_kernel, dim3((F + 63) / 64, (s.B + 63) / 64), dim3(256), 0, s.st, src, F, s.B, s.ldb, dst, ld);
        OV_LAUNCH_CHECK();
        return ONTOVAE_OK;
    }
    dim3 grid((F + 31) / 32, (s.B + 31) / 32);
    hipLaunchKernelGGL(ov_transpose_out_kernel, grid, dim3(256), 0, s.st, src, F, s.B, s.ldb, dst, ld);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// =================================================================================================
// C ABI: forward / eval / inference
// =================================================================================================
extern "C" int ontovae_forward(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, const float* eps,
                               uint64_t rng_offset, int32_t train, float* recon, float* mu, float* log_var, void* stream) {
    int rc = ov_check_ctx(ctx, B, false, train != 0);
    if (rc) return rc;
    if (!x) { ontovae_set_error("forward: x is NULL"); return ONTOVAE_EINVAL; }
    if (train && B < 2) { ontovae_set_error("forward: BatchNorm in train mode needs more than 1 sample per batch"); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, x, xidx, B, stream, &s);
    const ontovae_dims_t& d = ctx->dims;
    if ((rc = ov_encode(s, eps, train, rng_offset, OV_RNG_EPS, 0, nullptr))) return rc;
    if (mu && (rc = ov_transpose_out(s, s.ws.mulv, d.L, mu, d.L))) return rc;
    if (log_var && (rc = ov_transpose_out(s, s.ws.mulv + (int64_t)d.L * s.ldb, d.L, log_var, d.L))) return rc;
    if (recon) {
        if ((rc = ov_dec_forward_terms(s, true))) return rc;
        if ((rc = ov_dec_gene<2>(s, recon, d.G))) return rc;
    }
    return ONTOVAE_OK;
}

extern "C" int ontovae_eval_step(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, const float* eps,
                                 uint64_t rng_offset, float kl_coeff, float* loss_out, void* stream) {
    int rc = ov_check_ctx(ctx, B, false);
    if (rc) return rc;
    if (!x || !loss_out) { ontovae_set_error("eval_step: x or loss_out is NULL"); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, x, xidx, B, stream, &s);
    s.gene_lds = ov_gene_lds_on(s);
    int nkl = 0;
    if ((rc = ov_encode(s, eps, 0, rng_offset, OV_RNG_EPS, 1, &nkl))) return rc;
    if ((rc = ov_dec_forward_terms(s, true))) return rc;
    if ((rc = ov_dec_gene<1>(s, nullptr, 0))) return rc;
    return ov_loss_finalize(s, s.st, ov_nrec(s), nkl, kl_coeff, loss_out);
}

static int ov_act_out(OvStep& s, int lvl_from, int lvl_to, float* act_out, int64_t act_ld, bool feature_major = false) {
    const ontovae_dims_t& d = s.c->dims;
    if (feature_major && d.K > OV_MAX_LEVELS) { ontovae_set_error("feature-major activities need K <= %d levels", OV_MAX_LEVELS); return ONTOVAE_EINVAL; }
    if (d.K <= OV_MAX_LEVELS) {          // one launch for the whole range of levels
        OvActLevels lv;
        lv.n = d.K;
        int t0 = 0;
        for (int lvl = 0; lvl < d.K; ++lvl) {
            lv.tstart[lvl] = t0;
            lv.node0[lvl] = s.c->level_begin[lvl];
            t0 += (s.c->level_end[lvl] - s.c->level_begin[lvl]) / d.nu;
        }
        lv.tstart[d.K] = t0;
        const int c0 = lv.tstart[std::max(lvl_from, 0)], c1 = lv.tstart[std::min(lvl_to, (int)d.K)];
        if (c1 <= c0) return ONTOVAE_OK;
        if (feature_major) {
            hipLaunchKernelGGL(ov_act_fm_levels_kernel, dim3((s.B + 1023) / 1024, c1 - c0), dim3(256), 0, s.st, s.ws.act, lv, c0, c1, d.nu, s.B,
                               s.ldb, act_out, act_ld);
            OV_LAUNCH_CHECK();
            return ONTOVAE_OK;
        }
        dim3 grid((c1 - c0 + 31) / 32, (s.B + 31) / 32);
        hipLaunchKernelGGL(ov_act_out_levels_kernel, grid, dim3(256), 0, s.st, s.ws.act, lv, c0, c1, d.nu, s.B, s.ldb, act_out, act_ld);
        OV_LAUNCH_CHECK();
        return ONTOVAE_OK;
    }
    int t0 = 0;
    for (int lvl = 0; lvl < d.K; ++lvl) {
        const int n0 = s.c->level_begin[lvl];
        const int nterms = (s.c->level_end[lvl] - n0) / d.nu;
        if (lvl >= lvl_from && lvl < lvl_to && nterms > 0) {
            dim3 grid((nterms + 31) / 32, (s.B + 31) / 32);
            hipLaunchKernelGGL(ov_act_out_kernel, grid, dim3(256), 0, s.st, s.ws.act, n0, nterms, d.nu, s.B, s.ldb, act_out, act_ld, t0);
            OV_LAUNCH_CHECK();
        }
        t0 += nterms;
    }
    return ONTOVAE_OK;
}

static bool ov_fm_fused(const OvStep& s);
extern "C" int ontovae_pass_data(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, const float* eps1,
                                 const float* eps2, uint64_t rng_offset, float* act_out, int64_t act_ld, float* rec_out,
                                 int64_t rec_ld, void* stream) {
    int rc = ov_check_ctx(ctx, B, false);
    if (rc) return rc;
    if (!x) { ontovae_set_error("pass_data: x is NULL"); return ONTOVAE_EINVAL; }
    const ontovae_dims_t& d = ctx->dims;
    if ((act_out && act_ld < d.T) || (rec_out && rec_ld < d.G)) { ontovae_set_error("pass_data: output leading dimension too small"); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, x, xidx, B, stream, &s);
    // The reference runs the encoder twice (get_embedding, then forward); in eval mode both runs give
    // the same mu / log_var, so one encoder pass and two reparameterisations are identical.
    if ((rc = ov_encode(s, eps1, 0, rng_offset, OV_RNG_EPS, 0, nullptr))) return rc;
    // Inference batches: the level launches write the nu-means themselves, feature-major, into the (idle) D panel of the
    // workspace, and one transposition brings all T columns to the caller's sample-major matrix; otherwise the output
    // kernel reads the finished panel.
    const bool fused = act_out && ov_fm_fused(s);
    float* fm = s.ws.d;
    if (act_out && (rc = fused ? ov_act_out(s, 0, 1, fm, s.ldb, true) : ov_act_out(s, 0, 1, act_out, act_ld))) return rc;   // z part from draw #1
    if ((rc = ov_reparam(s, eps2, 0, rng_offset, OV_RNG_EPS2, 0, nullptr, false))) return rc;  // draw #2 feeds the decoder
    if (fused) { s.fm_out = fm; s.fm_ld = s.ldb; }
    if ((rc = ov_dec_forward_terms(s, true))) return rc;
    s.fm_out = nullptr;
    if (fused) {
        if ((rc = ov_transpose_out(s, fm, d.T, act_out, act_ld))) return rc;
    } else if (act_out && (rc = ov_act_out(s, 1, d.K, act_out, act_ld))) return rc;
    if (rec_out && (rc = ov_dec_gene<2>(s, rec_out, rec_ld))) return rc;
    return ONTOVAE_OK;
}

// ---- perturbation screens: the first encoder layer of a perturbed batch from a cached x W0^T ----------------------
// sum of the encoder GEMM's split-K slabs -> hpre [L][ldh] (no bias)
__global__ __launch_bounds__(256) void ov_slab_sum_kernel(const float* __restrict__ slab, int S, int64_t stride, int B, int ldb,
                                                         float* __restrict__ out, int64_t ldh) {
    const int l = blockIdx.y, b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float h = 0.f;
    for (int s = 0; s < S; ++s) h += slab[s * stride + (int64_t)l * ldb + b];
    out[(int64_t)l * ldh + b] = h;
}

#define OV_PERT_MAX ONTOVAE_MAX_PERTURBED_GENES
#define OV_PERT_LB 8
struct OvPert { int n; int32_t gene[OV_PERT_MAX]; float val[OV_PERT_MAX]; };
// slab[l][b] = hpre[l][b] + sum_p (val_p - x[b, gene_p]) W0[l, gene_p]: x' differs from x in the listed columns only, so
// x' W0^T is the cached product plus one rank-1 term per perturbed gene
__global__ __launch_bounds__(256) void ov_pert_slab_kernel(const float* __restrict__ hpre, int64_t ldh, const float* __restrict__ x,
                                                          const int32_t* __restrict__ xidx, int64_t xld, const float* __restrict__ W0,
                                                          int G, const OvPert pt, int L, int B, int ldb, float* __restrict__ slab) {
    const int l0 = blockIdx.y * OV_PERT_LB, b = blockIdx.x * 256 + threadIdx.x;
    if (b >= ldb) return;
    float h[OV_PERT_LB];
#pragma unroll
    for (int i = 0; i < OV_PERT_LB; ++i) h[i] = (b < B && l0 + i < L) ? hpre[(int64_t)(l0 + i) * ldh + b] : 0.f;
    if (b < B) {
        const float* xr = x + (int64_t)(xidx ? xidx[b] : b) * xld;
        for (int p = 0; p < pt.n; ++p) {
            const int g = pt.gene[p];
            const float d = pt.val[p] - xr[g];
#pragma unroll
            for (int i = 0; i < OV_PERT_LB; ++i)
                if (l0 + i < L) h[i] += d * W0[(int64_t)(l0 + i) * G + g];
        }
    }
#pragma unroll
    for (int i = 0; i < OV_PERT_LB; ++i)
        if (l0 + i < L) slab[(int64_t)(l0 + i) * ldb + b] = h[i];
}

// the term levels of this pass run as one launch per level (inference batch) and can write the activities themselves
static bool ov_fm_fused(const OvStep& s) {
    const ontovae_dims_t& d = s.c->dims;
    if (s.ldb < OV_CHAIN_MAX_LDB && ov_fused_bt(s.c) != 0) return false;          // the chain kernel takes the term levels
    if (s.vec != 4 || d.nu < 2 || d.nu > 4 || d.K > OV_MAX_LEVELS) return false;
    for (int lvl = 1; lvl < d.K; ++lvl)
        if ((s.c->level_end[lvl] - s.c->level_begin[lvl]) % d.nu != 0) return false;
    return true;
}

extern "C" int ontovae_encode_pre(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, float* hpre, int64_t ldh,
                                  void* stream) {
    int rc = ov_check_ctx(ctx, B, false);
    if (rc) return rc;
    if (!x || !hpre || ldh < B) { ontovae_set_error("encode_pre: x or hpre is NULL, or ldh < B"); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, x, xidx, B, stream, &s);
    int S = 1;
    if ((rc = ov_enc_gemm(s, &S))) return rc;
    const ontovae_dims_t& d = ctx->dims;
    hipLaunchKernelGGL(ov_slab_sum_kernel, dim3((B + 255) / 256, d.L), dim3(256), 0, s.st, s.ws.slab, S, (int64_t)d.L * s.ldb, B, s.ldb, hpre, ldh);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

extern "C" int ontovae_pass_data_perturbed(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, const float* hpre,
                                           int64_t ldh, int32_t n_pert, const int32_t* pert_gene, const float* pert_val,
                                           const float* eps1, const float* eps2, uint64_t rng_offset, float* act_out, int64_t act_ld,
                                           int32_t act_feature_major, float* rec_out, int64_t rec_ld, void* stream) {
    int rc = ov_check_ctx(ctx, B, false);
    if (rc) return rc;
    const ontovae_dims_t& d = ctx->dims;
    if (!x || !hpre || ldh < B) { ontovae_set_error("pass_data_perturbed: x or hpre is NULL, or ldh < B"); return ONTOVAE_EINVAL; }
    if (n_pert < 0 || n_pert > OV_PERT_MAX || (n_pert && (!pert_gene || !pert_val))) {
        ontovae_set_error("pass_data_perturbed: n_pert = %d outside [0, %d] or NULL lists", n_pert, OV_PERT_MAX);
        return ONTOVAE_EINVAL;
    }
    const bool fm = act_feature_major != 0;
    if ((act_out && act_ld < (fm ? B : d.T)) || (rec_out && rec_ld < d.G)) { ontovae_set_error("pass_data_perturbed: output leading dimension too small"); return ONTOVAE_EINVAL; }
    if (fm && act_out && ((reinterpret_cast<uintptr_t>(act_out) & 3) != 0)) { ontovae_set_error("pass_data_perturbed: act_out misaligned"); return ONTOVAE_EINVAL; }
    OvPert pt{};
    pt.n = n_pert;
    for (int p = 0; p < n_pert; ++p) {
        if (pert_gene[p] < 0 || pert_gene[p] >= d.G) { ontovae_set_error("pass_data_perturbed: gene index %d outside [0, %d)", pert_gene[p], d.G); return ONTOVAE_EINVAL; }
        for (int q = 0; q < p; ++q)
            if (pert_gene[q] == pert_gene[p]) { ontovae_set_error("pass_data_perturbed: gene index %d listed twice", pert_gene[p]); return ONTOVAE_EINVAL; }
        pt.gene[p] = pert_gene[p];
        pt.val[p] = pert_val[p];
    }
    OvStep s;
    ov_make_step(ctx, x, xidx, B, stream, &s);
    hipLaunchKernelGGL(ov_pert_slab_kernel, dim3((s.ldb + 255) / 256, (d.L + OV_PERT_LB - 1) / OV_PERT_LB), dim3(256), 0, s.st, hpre, ldh, s.x,
                       s.xidx, s.xld, ctx->P + ctx->lay.W0, d.G, pt, d.L, B, s.ldb, s.ws.slab);
    OV_LAUNCH_CHECK();
    if ((rc = ov_bn_fwd(s, 1, 0, s.B, nullptr, rng_offset))) return rc;
    if ((rc = ov_heads_fwd(s))) return rc;
    if ((rc = ov_reparam(s, eps1, 0, rng_offset, OV_RNG_EPS, 0, nullptr, true))) return rc;
    if (act_out && (rc = ov_act_out(s, 0, 1, act_out, act_ld, fm))) return rc;
    if ((rc = ov_reparam(s, eps2, 0, rng_offset, OV_RNG_EPS2, 0, nullptr, false))) return rc;
    const bool fused = fm && act_out && ov_fm_fused(s);
    if (fused) { s.fm_out = act_out; s.fm_ld = act_ld; }
    if ((rc = ov_dec_forward_terms(s, true))) return rc;
    s.fm_out = nullptr;
    if (act_out && !fused && (rc = ov_act_out(s, 1, d.K, act_out, act_ld, fm))) return rc;
    if (rec_out && (rc = ov_dec_gene<2>(s, rec_out, rec_ld))) return rc;
    return ONTOVAE_OK;
}

extern "C" int ontovae_decoder_forward(const ontovae_ctx_t* ctx, const float* z, int32_t B, float* act_out, int64_t act_ld,
                                       float* rec_out, int64_t rec_ld, void* stream) {
    int rc = ov_check_ctx(ctx, B, false);
    if (rc) return rc;
    if (!z) { ontovae_set_error("decoder_forward: z is NULL"); return ONTOVAE_EINVAL; }
    const ontovae_dims_t& d = ctx->dims;
    if ((act_out && act_ld < d.T) || (rec_out && rec_ld < d.G)) { ontovae_set_error("decoder_forward: output leading dimension too small"); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, nullptr, nullptr, B, stream, &s);
    dim3 grid((d.L + 31) / 32, (s.ldb + 31) / 32);
    hipLaunchKernelGGL(ov_transpose_in_kernel, grid, dim3(256), 0, s.st, z, d.L, B, s.ldb, s.ws.act + (int64_t)(d.C - d.L) * s.ldb);
    OV_LAUNCH_CHECK();
    if ((rc = ov_dec_forward_terms(s))) return rc;
    if (act_out && (rc = ov_act_out(s, 0, d.K, act_out, act_ld))) return rc;
    if (rec_out && (rc = ov_dec_gene<2>(s, rec_out, rec_ld))) return rc;
    return ONTOVAE_OK;
}

// Backward of ontovae_decoder_forward and the update of the masked operator: autograd of OntoDecoder.forward
// (modules.py:303-313) / OntoEncoder.forward (modules.py:226-238), the gradient mask (vae_model.py:168-169, 611-616: implicit,
// only mask positions exist), AdamW (vae_model.py:172, 619) and the clamp of the weights at zero (vae_model.py:175-176,
// 622-625; biases are not clamped).  Must follow ontovae_decoder_forward on the same ctx and batch: the panel of that pass is
// read from the workspace.  drec [B, G] sample-major, contiguous: dL/d(rec_out).  apply_update != 0: AdamW + clamp in place
// (opt, ctx->M, ctx->V required); else the gradients of [dec_vals | dec_bias] go to ctx->Gr.  dz_out [B, L] (may be NULL):
// dL/dz.  The general-purpose kernels (one wave per row for the weight gradient): this entry point is for models whose
// ontology sits in the encoder (OntoEncVAE), not for the OntoVAE step, which has its own fused path.
// ontovae_decoder_forward for a TRAINING pass of a model whose ontology sits in the encoder: Dropout(ctx->drop) on the output of
// every term layer (OntoEncoder's blocks, modules.py:219-223, 231-233), keyed by rng_offset; the last layer (the heads) carries
// none here (their Dropout(z_drop) belongs to ontovae_vae_decode_step).  With ctx->drop == 0 the same as ontovae_decoder_forward.
extern "C" int ontovae_decoder_forward_train(const ontovae_ctx_t* ctx, const float* z, int32_t B, uint64_t rng_offset, float* rec_out,
                                             int64_t rec_ld, void* stream) {
    int rc = ov_check_ctx(ctx, B, false, true);
    if (rc) return rc;
    if (!z || !rec_out) { ontovae_set_error("decoder_forward_train: z or rec_out is NULL"); return ONTOVAE_EINVAL; }
    const ontovae_dims_t& d = ctx->dims;
    if (rec_ld < d.G) { ontovae_set_error("decoder_forward_train: output leading dimension too small"); return ONTOVAE_EINVAL; }
    if (!(ctx->drop >= 0.f && ctx->drop < 1.f)) { ontovae_set_error("decoder_forward_train: drop = %g outside [0, 1)", (double)ctx->drop); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, nullptr, nullptr, B, stream, &s);
    s.tdrop = OvTermDrop{ctx->drop, ctx->seed, rng_offset};
    dim3 grid((d.L + 31) / 32, (s.ldb + 31) / 32);
    hipLaunchKernelGGL(ov_transpose_in_kernel, grid, dim3(256), 0, s.st, z, d.L, B, s.ldb, s.ws.act + (int64_t)(d.C - d.L) * s.ldb);
    OV_LAUNCH_CHECK();
    if ((rc = ov_dec_forward_terms(s))) return rc;
    return ov_dec_gene<2>(s, rec_out, rec_ld);
}

extern "C" int ontovae_decoder_backward(const ontovae_ctx_t* ctx, const float* drec, int32_t B, uint64_t rng_offset,
                                        const ontovae_adamw_t* opt, int32_t apply_update, float* dz_out, int64_t dz_ld, void* stream) {
    int rc = ov_check_ctx(ctx, B, apply_update != 0, true);
    if (rc) return rc;
    if (!drec) { ontovae_set_error("decoder_backward: drec is NULL"); return ONTOVAE_EINVAL; }
    if (apply_update && !opt) { ontovae_set_error("decoder_backward: opt is NULL"); return ONTOVAE_EINVAL; }
    if (!apply_update && !ctx->Gr) { ontovae_set_error("decoder_backward: gradient buffer Gr is NULL"); return ONTOVAE_EINVAL; }
    const ontovae_dims_t& d = ctx->dims;
    const ontovae_layout_t& lay = ctx->lay;
    if (dz_out && dz_ld < d.L) { ontovae_set_error("decoder_backward: dz leading dimension too small"); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, nullptr, nullptr, B, stream, &s);
    s.tdrop = OvTermDrop{ctx->drop, ctx->seed, rng_offset};      // the masks of ontovae_decoder_forward_train with the same offset
    // dL/d(rec) into the gene rows of D, feature-major, zeros in the padded columns
    dim3 grid((d.G + 31) / 32, (s.ldb + 31) / 32);
    hipLaunchKernelGGL(ov_transpose_in_kernel, grid, dim3(256), 0, s.st, drec, d.G, B, s.ldb, s.ws.d + (int64_t)d.C * s.ldb);
    OV_LAUNCH_CHECK();
    if ((rc = ov_dec_backward(s))) return rc;
    const OvAdam adam = ov_adam_from(opt);
    const int n_term_rows = d.C - d.L, wrows = n_term_rows + d.G;
    float* G_ = ctx->Gr;
    hipLaunchKernelGGL(ov_dec_wgrad_kernel, dim3((wrows + 3) / 4), dim3(256), 0, s.st, ctx->rowptr, ctx->colidx, ctx->row_k, s.ws.act, s.ws.d,
                       ctx->P + lay.dec_vals, ctx->M ? ctx->M + lay.dec_vals : nullptr, ctx->V ? ctx->V + lay.dec_vals : nullptr,
                       G_ ? G_ + lay.dec_vals : nullptr, ctx->P + lay.dec_bias, ctx->M ? ctx->M + lay.dec_bias : nullptr,
                       ctx->V ? ctx->V + lay.dec_bias : nullptr, G_ ? G_ + lay.dec_bias : nullptr, wrows, 0, n_term_rows, d.C, s.ldb,
                       apply_update ? 1 : 0, adam);
    OV_LAUNCH_CHECK();
    if (dz_out && (rc = ov_transpose_out(s, s.ws.d + (int64_t)(d.C - d.L) * s.ldb, d.L, dz_out, dz_ld))) return rc;
    return ONTOVAE_OK;
}

// =================================================================================================
// C ABI: training
// =================================================================================================
// Everything after the encoder forward: decoder, ELBO, full backward, weight gradients.
// bn_mode: 1 = fused BN backward (single GPU), 2 = stop after the DP phase-1 BN partial sums.
// refreshed: ov_sell_refresh was already enqueued on the side stream (joined here before the decoder).
// The side stream is still running when this returns: the caller ends the step with ov_side_join.
static int ov_train_tail_a(OvStep& s, const float* eps, uint64_t rng_off, float kc, const ontovae_adamw_t* opt, int apply,
                           float* loss_out, int bn_mode, float* bn_bwd_stats, int Bglobal, bool refreshed) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    const ontovae_layout_t& lay = c->lay;
    const OvAdam adam = ov_adam_from(opt);
    int rc, nkl = 0;
    if ((rc = ov_heads_fwd(s))) return rc;
    if ((rc = ov_reparam(s, eps, 1, rng_off, OV_RNG_EPS, 1, &nkl, !refreshed))) return rc;
    if (refreshed && (rc = ov_side_join(s))) return rc;          // packed decoder weights ready
    if ((rc = ov_dec_forward_terms(s, true))) return rc;
    // weight gradient of the decoder: rows = nodes [0, C - L) (term rows), then the genes; entries = packed range [0, nnz)
    const int n_term_rows = d.C - d.L;
    const int wrows = n_term_rows + d.G;
    const bool wcb = !(c->flags & ONTOVAE_FLAG_WIDE_GENE_TILES);
    // The gene layer's own weight gradient is formed inside its two column-block kernels (bias partials in the forward one,
    // link partials in the backward one, which has every D row it needs in registers anyway): no gathers of its own.
    // ONTOVAE_FLAG_NO_RIDERS keeps the stand-alone weight-gradient kernels for all rows (tests compare the two).
    const bool fused_gw = wcb && ov_gene_cb_on(s) && ov_fused_bt(c) != 0 && !s.side && !(c->flags & ONTOVAE_FLAG_NO_RIDERS);
    OvWgradArgs wa{};
    int ride_rows = 0;
    if (wcb) {
        const int ncb = s.ldb / 32;
        ride_rows = fused_gw ? d.G : 0;          // gene rows done elsewhere
        wa = OvWgradArgs{c->rowptr, c->colidx, c->row_k, s.ws.act, s.ws.d, s.ws.wpart, s.ws.wpart + (int64_t)ncb * d.nnz, (int64_t)d.nnz,
                         wrows - ride_rows, ride_rows, wrows, 0, n_term_rows, d.C, s.ldb, ncb, (ride_rows + 31) / 32};
        if (fused_gw) { s.gw_part = wa.part; s.gw_bpart = wa.bpart; s.gw_rows = wrows; s.gw_row0 = n_term_rows; }
    }
    if ((rc = ov_dec_gene<1>(s, nullptr, 0))) return rc;
    // decoder weight gradients (+ AdamW + clamp): partials of the rows the gene-layer kernels do not cover, then the finish
    // pass.  On the single-stream path the partials ride in ov_latent_bwd_kernel and the finish pass in two launches: the
    // gene layer's share in the backward chain's, the term layers' in ov_da_gemm_kernel's.
    float* G_ = c->Gr;
    OvWgradArgs wt = wa;
    OvFinishArgs fa{}, fa_gene{};
    int wteams = 0, fblocks = 0, gblocks = 0;
    bool ride_fin = false, ride_lat = false;
    const bool riders2 = wcb && !s.side && !(c->flags & ONTOVAE_FLAG_NO_RIDERS);
    if (wcb) {
        wt.row0 = 0;
        wt.nrows_part = fused_gw ? wrows - ride_rows : wrows;
        wt.ntiles = (wt.nrows_part + 31) / 32;
        wt.tiles = (wt.nrows_part == n_term_rows && wt.first_term_node == 0) ? reinterpret_cast<const int4*>(c->term_tiles) : nullptr;
        wteams = ov_cb_grid(wt.ncb, wt.ntiles);
        fblocks = (int)std::min<int64_t>(((int64_t)d.nnz + 255) / 256, 2048);
        fa = OvFinishArgs{wa.part, wa.bpart, wa.ncb, fused_gw && ov_bwd_gene_acc(s) ? std::min(wa.ncb, 8) : wa.ncb, wa.ncb, (int64_t)d.nnz, wrows, 0, n_term_rows, d.C, c->P + lay.dec_vals,
                          c->M ? c->M + lay.dec_vals : nullptr, c->V ? c->V + lay.dec_vals : nullptr,
                          G_ ? G_ + lay.dec_vals : nullptr, c->P + lay.dec_bias, c->M ? c->M + lay.dec_bias : nullptr,
                          c->V ? c->V + lay.dec_bias : nullptr, G_ ? G_ + lay.dec_bias : nullptr, apply, adam,
                          d.nnz_term, 0, nullptr, nullptr, 0, 0};
        if (s.gene_lds && fused_gw) { fa.gb4 = s.ws.gb4; fa.lane_gene = c->gsell.gene; fa.nl = c->gsell.n_slices * 64; fa.ngrp = (s.B + 3) / 4; }
        // only while the chain leaves most of the chip idle (<= 128 workgroups): at larger batches the chain's workgroups
        // fill every CU themselves and 1024-thread riders next to their LDS reservation only stretch the launch
        ride_fin = fused_gw && s.ldb / std::max(ov_fused_bt(c), 1) <= 128;
        // large batches: the gene layer's finish pass (a stream over ldb / 32 copies of the gene layer's links) rides with the
        // latent backward and the term rows' partials instead of stretching the da launch
        ride_lat = fused_gw && !ride_fin && riders2 && s.ldb >= 512;
        if (ride_fin || ride_lat) {
            fa_gene = fa;
            fa_gene.sel = 1;
            fa.sel = 2;
            gblocks = ride_fin ? (int)std::min<int64_t>(((int64_t)d.nnz + 1023) / 1024, 448)       // 1024-thread workgroups beside the chain
                               : (int)std::min<int64_t>(((int64_t)d.nnz + 255) / 256, 256);        // 256-thread ones, one per CU (350 us for the launch at 8192 rows; 367 with 512, 391 with 2048)
            fblocks = (int)std::min<int64_t>(((int64_t)d.nnz_term + d.C + 255) / 256, 2048);
        }
    }
    // from 4096 columns on the term part of the backward is one launch per level (ov_dec_backward: per_level), and each of those
    // waves holds exactly the rows the term layers' weight gradient needs: formed there, one partial copy per chunk of 256
    // columns, instead of by riders of the latent backward that gather every row again (one copy per 32 columns)
    if (ride_lat && s.ldb >= OV_CHAIN_FWD_MAX_LDB && s.tdrop.rate <= 0.f && ov_fused_bt(c) != 0 && !(c->flags & ONTOVAE_FLAG_PER_BLOCK_WGRAD)) {
        s.tw_part = wa.part; s.tw_bpart = wa.bpart;
        fa.ncb_term = s.nchunk;
        wteams = 0;
    }
    if ((rc = ov_dec_backward(s, ride_fin ? &fa_gene : nullptr, gblocks))) return rc;
    // fork: the decoder's weight gradients need nothing from the encoder backward, and vice versa
    if ((rc = ov_side_fork(s))) return rc;
    const hipStream_t sst = ov_side_stream(s);
    if (!riders2) {
        OvProfScope prof(ONTOVAE_PHASE_DEC_WGRAD, sst);
        if (wcb) {
            hipLaunchKernelGGL(ov_wgrad_cb_kernel, dim3(wteams), dim3(256), 0, sst, wt);
            OV_LAUNCH_CHECK();
            hipLaunchKernelGGL(ov_wgrad_finish_kernel, dim3(fblocks), dim3(256), 0, sst, fa);
            OV_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(ov_dec_wgrad_kernel, dim3((wrows + 3) / 4), dim3(256), 0, sst, c->rowptr, c->colidx, c->row_k, s.ws.act, s.ws.d,
                               c->P + lay.dec_vals, c->M ? c->M + lay.dec_vals : nullptr, c->V ? c->V + lay.dec_vals : nullptr,
                               G_ ? G_ + lay.dec_vals : nullptr, c->P + lay.dec_bias, c->M ? c->M + lay.dec_bias : nullptr,
                               c->V ? c->V + lay.dec_bias : nullptr, G_ ? G_ + lay.dec_bias : nullptr, wrows, 0, n_term_rows, d.C, s.ldb,
                               apply, adam);
            OV_LAUNCH_CHECK();
        }
    }
    // latent backward (+ head bias gradient / AdamW, + final reduction of the loss partials) [+ weight-gradient partials]
    {
        OvLatentArgs la{s.ws.d + (int64_t)(d.C - d.L) * s.ldb, s.ws.mulv, s.ws.epsw, s.ws.dmulv, d.L, s.B, s.ldb, kc, c->z_drop,
                        c->seed, rng_off, c->P + lay.bh, c->M ? c->M + lay.bh : nullptr, c->V ? c->V + lay.bh : nullptr,
                        c->Gr ? c->Gr + lay.bh : nullptr, apply, adam,
                        OvLossArgs{s.ws.rec_part, ov_nrec(s), s.ws.kl_part, nkl, kc, loss_out}, d.L + 1};
        OvProfScope prof(ONTOVAE_PHASE_LATENT_BWD, s.st);
        const int lfin = ride_lat ? gblocks : 0;
        hipLaunchKernelGGL(ov_latent_bwd_kernel, dim3(la.nblocks + lfin + (riders2 ? wteams : 0)), dim3(256), 0, s.st, la, wt,
                           riders2 ? wteams : 0, fa_gene, lfin);
        OV_LAUNCH_CHECK();
    }
    // da = Wh^T dmulv  (before Wh is updated) [+ finish pass of the decoder weight update]
    int da_S = 1;
    {
        OvGemm g{};
        g.M = d.L; g.N = s.B; g.K = 2 * d.L;
        g.A = c->P + lay.Wh; g.lda = d.L;          // A(m=l, k=j) = Wh[j*L + l]
        g.B = s.ws.dmulv; g.ldb = s.ldb;
        // split-K partials into the (free) encoder slab buffer; the BatchNorm backward adds them
        const int64_t one = (int64_t)d.L * s.ldb;
        // from 2048 columns on: 96 x 128 tiles staged through LDS (the 32 x 32 split-K tiles reach 21 TFLOP/s at 8192 columns, this
        // kernel 70; at 1024 columns the two are equal once the copy's launch is counted)
        const bool da_lds = s.ldb >= 2048 && (d.L % 4 == 0) && ov_al16(g.B) && ov_al16(s.ws.whT);
        const int tiles_lds = ((d.L + 95) / 96) * ((s.B + 127) / 128);
        const int S = (int)std::max<int64_t>(1, std::min<int64_t>(da_lds ? std::min(4, 256 / std::max(tiles_lds, 1)) : 4, s.ws.slab_floats / one));
        const int kchunk = ((2 * d.L + S - 1) / S + 31) / 32 * 32;
        da_S = (2 * d.L + kchunk - 1) / kchunk;
        g.C = da_S > 1 ? s.ws.slab : s.ws.da; g.ldc = s.ldb; g.slab_stride = one; g.kchunk = kchunk;
        OvProfScope prof(ONTOVAE_PHASE_DA_GEMM, s.st);
        if (da_lds) {
            const int ntx = (d.L + 31) / 32, nty = (2 * d.L + 31) / 32;
            hipLaunchKernelGGL(ov_whT_finish_kernel, dim3(ntx * nty + (riders2 ? fblocks : 0)), dim3(256), 0, s.st, c->P + lay.Wh, 2 * d.L, d.L,
                               s.ws.whT, ntx, nty, fa, riders2 ? fblocks : 0);
            OV_LAUNCH_CHECK();
            g.A = s.ws.whT; g.lda = 2 * d.L;          // A(m=l, k=j) = WhT[l * 2L + j]
            if ((rc = ov_launch_gemm_lds<1, 4, false, OV_EPI_STORE>(g, da_S, s.st))) return rc;
        } else {
        const int gx = (g.N + 31) / 32, gy = (g.M + 31) / 32;
        const size_t lds = (size_t)4 * 16 * 64 * sizeof(float);        // KW == 4 reduction area of a 1 x 1 tile
        hipLaunchKernelGGL((ov_da_gemm_kernel<false>), dim3(gx * gy * da_S + (riders2 ? fblocks : 0)), dim3(256), lds, s.st, g, gx, gy,
                           da_S, fa, riders2 ? fblocks : 0);
        OV_LAUNCH_CHECK();
        }
    }
    // dWh = dmulv a^T (+ AdamW) and the BatchNorm backward (dh; AdamW of gamma, beta, b0): independent, one launch on
    // the single-stream path.  (dbh is done by the latent backward launch.)
    OvGemm gw{};
    gw.M = 2 * d.L; gw.N = d.L; gw.K = s.B;
    gw.A = s.ws.dmulv; gw.lda = s.ldb;
    gw.B = s.ws.a; gw.ldb = s.ldb;
    gw.ldc = d.L; gw.kchunk = (s.B + 31) / 32 * 32;
    gw.adam = adam;
    if (apply) { gw.C = c->P + lay.Wh; gw.Pm = c->M + lay.Wh; gw.Pv = c->V + lay.Wh; }
    else gw.C = c->Gr + lay.Wh;
    float* Gr = c->Gr;
    const OvBnBwdArgs bn{da_S > 1 ? s.ws.slab : s.ws.da, s.ws.a, s.ws.xhat, s.ws.bn_invstd, c->P + lay.gamma, c->P + lay.beta,
                         c->P + lay.b0, c->M ? c->M + lay.gamma : nullptr, c->M ? c->M + lay.beta : nullptr,
                         c->M ? c->M + lay.b0 : nullptr, c->V ? c->V + lay.gamma : nullptr, c->V ? c->V + lay.beta : nullptr,
                         c->V ? c->V + lay.b0 : nullptr, Gr ? Gr + lay.gamma : nullptr, Gr ? Gr + lay.beta : nullptr,
                         Gr ? Gr + lay.b0 : nullptr, s.ws.dh, s.ws.bn_bwd_local, bn_bwd_stats, d.L, s.B, Bglobal, s.ldb, bn_mode,
                         apply, adam, c->drop, c->seed, rng_off, da_S, (int64_t)d.L * s.ldb, nullptr, 0};
    const int bn_blocks = d.L;
    if (!s.side && !(c->flags & ONTOVAE_FLAG_NO_RIDERS)) {
        ov_prof_pair_begin(s.st);
        const int64_t nwh = (int64_t)2 * d.L * d.L;
        // (with the combine launch: 23 -> 21 us at 1024 columns, 49 -> 28 at 2048, 169 -> 64 at 8192, BatchNorm part vectorised in both)
        if (s.ldb >= 1024 && (d.L % 4 == 0) && ov_al16(gw.A) && ov_al16(gw.B) && ov_al16(gw.C) && (!apply || (ov_al16(gw.Pm) && ov_al16(gw.Pv)))) {
            OvProfScope prof(ONTOVAE_PHASE_DWH, s.st);
            // 96 x 128 tiles of [2L, L], the batch in split-K slabs: about one workgroup per CU beside the L BatchNorm workgroups
            const int gx = (d.L + 127) / 128, gy = (2 * d.L + 95) / 96;
            const int S = std::max(1, std::min(std::min(OV_DWH_SMAX, 256 / std::max(gx * gy, 1)), s.ldb / 64));
            OvGemm gl = gw;
            gl.kchunk = ((s.B + S - 1) / S + 31) / 32 * 32;
            const int gz = (s.B + gl.kchunk - 1) / gl.kchunk;
            gl.C = s.ws.whslab; gl.ldc = d.L; gl.slab_stride = nwh;
            const size_t lds = (size_t)(96 + 128) * 36 * sizeof(float);
            hipLaunchKernelGGL(ov_dwh_lds_bn_kernel, dim3(gx * gy * gz + bn_blocks), dim3(256), lds, s.st, gl, gx, gy, gz, bn);
            OV_LAUNCH_CHECK();
            hipLaunchKernelGGL(ov_wh_combine_kernel, dim3((unsigned)((nwh / 4 + 255) / 256)), dim3(256), 0, s.st, s.ws.whslab, gz, nwh,
                               c->P + lay.Wh, c->M ? c->M + lay.Wh : nullptr, c->V ? c->V + lay.Wh : nullptr, Gr ? Gr + lay.Wh : nullptr,
                               apply ? 1 : 0, adam);
            OV_LAUNCH_CHECK();
        } else {
            OvProfScope prof(ONTOVAE_PHASE_DWH, s.st);
            const int gx = (gw.N + 31) / 32, gy = (gw.M + 31) / 32;
            const size_t lds = (size_t)4 * 16 * 64 * sizeof(float);
            if (apply) hipLaunchKernelGGL((ov_dwh_bn_kernel<OV_EPI_ADAMW>), dim3(gx * gy + bn_blocks), dim3(256), lds, s.st, gw, gx, gy, bn);
            else hipLaunchKernelGGL((ov_dwh_bn_kernel<OV_EPI_STORE>), dim3(gx * gy + bn_blocks), dim3(256), lds, s.st, gw, gx, gy, bn);
            OV_LAUNCH_CHECK();
        }
        ov_prof_pair_mid(s.st);
        return ONTOVAE_OK;
    }
    const bool dwh_main = OV_DBG((c->flags & 64u) != 0);
    if (!dwh_main && (rc = ov_side_sync_mid(s))) return rc;
    {
        const hipStream_t wst = dwh_main ? s.st : ov_side_stream(s);
        OvProfScope prof(ONTOVAE_PHASE_DWH, wst);
        if (apply) { if ((rc = ov_launch_gemm<1, 1, 4, 2, true, true, OV_EPI_ADAMW>(gw, 1, true, wst))) return rc; }
        else { if ((rc = ov_launch_gemm<1, 1, 4, 2, true, true, OV_EPI_STORE>(gw, 1, true, wst))) return rc; }
    }
    {
        OvProfScope prof(ONTOVAE_PHASE_BN_BWD, s.st);
        hipLaunchKernelGGL(ov_bn_bwd_kernel, dim3(bn_blocks), dim3(256), 0, s.st, bn);
        OV_LAUNCH_CHECK();
    }
    return ONTOVAE_OK;
}

// dW0 = dh x (+ AdamW)   (gradient of modules.py:46)
// rows [l0, l1) of W0 only (l1 < 0: all): the data-parallel all-reduce form walks W0 in row panels, each panel's exchange
// running under the next panel's product
static int ov_dw0(OvStep& s, const ontovae_adamw_t* opt, int apply, int l0 = 0, int l1 = -1) {
    const ontovae_ctx_t* c = s.c;
    const ontovae_dims_t& d = c->dims;
    if (l1 < 0) l1 = d.L;
    const int64_t roff = (int64_t)l0 * d.G;
    if (c->flags & ONTOVAE_FLAG_BF16_GEMM) {
        // bf16 variant: dh rows -> dhb, dW0 rows = dhb . xT^T with AdamW in the product's epilogue (fused-update callers) or into
        // the gradient buffer (data parallel: the exchange comes first)
        if (!apply && !c->Gr) { ontovae_set_error("dW0 (bf16 variant): the gradient buffer Gr is NULL"); return ONTOVAE_EINVAL; }
        OvBfWs bw;
        ov_bf16_carve(d, c->max_batch, c->bf16_ws, &bw);
        OvProfScope prof(ONTOVAE_PHASE_DW0, s.st);
        hipLaunchKernelGGL(ov_bf16_convert_rows_kernel, dim3((s.ldb + 1023) / 1024, l1 - l0), dim3(256), 0, s.st,
                           s.ws.dh + (int64_t)l0 * s.ldb, (int64_t)s.ldb, s.B, bw.dhb + (int64_t)l0 * s.ldb, (int64_t)s.ldb);
        OV_LAUNCH_CHECK();
        OvBfGemm bg{bw.dhb + (int64_t)l0 * s.ldb, s.ldb, bw.xT, s.ldb, apply ? c->P + c->lay.W0 + roff : c->Gr + c->lay.W0 + roff, d.G, 0,
                    l1 - l0, d.G, s.ldb, (s.ldb + OV_BF_BK - 1) / OV_BF_BK * OV_BF_BK};
        if (!apply) return ov_bf16_gemm(bg, 1, s.st, nullptr, nullptr, nullptr, true);
        const OvAdam adam = ov_adam_from(opt);
        return ov_bf16_gemm(bg, 1, s.st, &adam, c->M + c->lay.W0 + roff, c->V + c->lay.W0 + roff);
    }
    OvGemm g{};
    g.M = l1 - l0; g.N = d.G; g.K = s.B;
    g.A = s.ws.dh + (int64_t)l0 * s.ldb; g.lda = s.ldb;
    g.B = s.x; g.ldb = s.xld; g.bgather = s.xidx;     // B(k=b, n=g) = X[row(b) * G + g]
    g.ldc = d.G; g.kchunk = (s.B + 31) / 32 * 32;
    g.adam = ov_adam_from(opt);
    g.dbg = OV_DBG((c->flags >> 16) & 7);
    OvProfScope prof(ONTOVAE_PHASE_DW0, s.st);
    const bool resident = (s.ldb % 4 == 0) && ov_al16(g.A) && (int64_t)d.L * d.G * 4 < ((int64_t)1 << 31) &&
                          !OV_DBG((c->flags >> 22) & 1);
    if (apply) {
        g.C = c->P + c->lay.W0 + roff; g.Pm = c->M + c->lay.W0 + roff; g.Pv = c->V + c->lay.W0 + roff;
        if (resident) return ov_launch_dw0<OV_EPI_ADAMW>(g, s.st);
        const int var = OV_DBG((c->flags >> 12) & 3);
        const bool al = (d.G % 4 == 0) && ov_al16(g.C) && ov_al16(g.Pm) && ov_al16(g.Pv);
        if (al && ov_al16(g.B) && OV_DBG((c->flags >> 21) & 1)) return ov_launch_gemm_lds<4, 1, false, OV_EPI_ADAMW>(g, 1, s.st);
        if (!al || var == 0) return ov_launch_gemm<1, 1, 1, 4, true, false, OV_EPI_ADAMW>(g, 1, true, s.st);
        if (var == 1) return ov_launch_gemm<1, 2, 1, 2, true, false, OV_EPI_ADAMW>(g, 1, true, s.st);
        if (var == 2) return ov_launch_gemm<1, 2, 1, 2, true, false, OV_EPI_ADAMW_T>(g, 1, true, s.st);
        if (var == 3) return ov_launch_gemm<1, 4, 1, 1, true, false, OV_EPI_ADAMW_T>(g, 1, true, s.st);
        return ov_launch_gemm<1, 4, 1, 2, true, false, OV_EPI_ADAMW_T>(g, 1, true, s.st);
    }
    g.C = c->Gr + c->lay.W0 + roff;
    if (resident) return ov_launch_dw0<OV_EPI_STORE>(g, s.st);
    if ((d.G % 4 == 0) && ov_al16(g.B) && OV_DBG((c->flags >> 21) & 1)) return ov_launch_gemm_lds<4, 1, false, OV_EPI_STORE>(g, 1, s.st);
    return ov_launch_gemm<1, 4, 1, 2, true, false, OV_EPI_STORE>(g, 1, true, s.st);
}

extern "C" int ontovae_train_step(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, const float* eps,
                                  uint64_t rng_offset, float kl_coeff, const ontovae_adamw_t* opt, int32_t apply_update,
                                  float* loss_out, void* stream) {
    int rc = ov_check_ctx(ctx, B, apply_update != 0, true);
    if (rc) return rc;
    if (!x || !loss_out) { ontovae_set_error("train_step: x or loss_out is NULL"); return ONTOVAE_EINVAL; }
    if (apply_update && !opt) { ontovae_set_error("train_step: opt is NULL"); return ONTOVAE_EINVAL; }
    if (!apply_update && !ctx->Gr) { ontovae_set_error("train_step: gradient buffer Gr is NULL"); return ONTOVAE_EINVAL; }
    if ((ctx->flags & ONTOVAE_FLAG_BF16_GEMM) && !ctx->Gr) { ontovae_set_error("train_step: the bf16 variant stages dW0 in the gradient buffer, Gr is NULL"); return ONTOVAE_EINVAL; }
    if (B < 2) { ontovae_set_error("train_step: BatchNorm in train mode needs more than 1 sample per batch"); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, x, xidx, B, stream, &s);
    s.want_xT = true;
    s.gene_lds = ov_gene_lds_on(s);
    s.side = ov_side_get(ctx);
    bool refreshed = false;
    if (s.side && ov_fused_bt(ctx) != 0 && OV_DBG((ctx->flags & 32u) != 0)) {      // packed decoder weight copies || encoder (measurement only)
        if ((rc = ov_side_fork(s))) return rc;
        if ((rc = ov_sell_refresh(s, s.side->st))) return rc;
        refreshed = true;
    }
    int S = 1;
    if ((rc = ov_enc_gemm(s, &S))) return rc;
    if ((rc = ov_bn_fwd(s, S, 1, B, nullptr, rng_offset))) return rc;
    if ((rc = ov_train_tail_a(s, eps, rng_offset, kl_coeff, opt, apply_update, loss_out, 1, nullptr, B, refreshed))) return rc;
    if ((rc = ov_dw0(s, opt, apply_update))) return rc;
    ov_prof_pair_end(s.st);
    return ov_side_join(s);
}

extern "C" int ontovae_train_step_dp(const ontovae_ctx_t* ctx, int32_t phase, const float* x, const int32_t* xidx, int32_t B,
                                     int32_t B_global, const float* eps, uint64_t rng_offset, float kl_coeff, float* bn_stats,
                                     float* bn_bwd_stats, float* loss_out, void* stream) {
    int rc = ov_check_ctx(ctx, B, false, true);
    if (rc) return rc;
    if (!x || !bn_stats || !bn_bwd_stats || !loss_out || !ctx->Gr) { ontovae_set_error("train_step_dp: a required pointer is NULL"); return ONTOVAE_EINVAL; }
    if (B_global < 2 || B_global < B) { ontovae_set_error("train_step_dp: bad global batch %d (local %d)", B_global, B); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, x, xidx, B, stream, &s);
    s.want_xT = true;
    s.gene_lds = ov_gene_lds_on(s);
    if (phase == 0) {
        int S = 1;
        if ((rc = ov_enc_gemm(s, &S))) return rc;
        return ov_bn_fwd(s, S, 2, B_global, bn_stats, rng_offset);
    } else if (phase == 1 || phase >= 16) {
        // phase 16 + R: as phase 1, but bn_stats holds the R ranks' all-gathered blocks [mean | M2 | rows] (2 L + 1 floats
        // each) and the combination to the global mean / variance happens inside the BatchNorm kernel
        if (phase >= 16) { if ((rc = ov_bn_fwd(s, phase - 16, 5, B_global, bn_stats, rng_offset))) return rc; }
        else if ((rc = ov_bn_fwd(s, 1, 3, B_global, bn_stats, rng_offset))) return rc;
        s.side = ov_side_get(ctx);
        if ((rc = ov_train_tail_a(s, eps, rng_offset, kl_coeff, nullptr, 0, loss_out, 2, bn_bwd_stats, B_global, false))) return rc;
        return ov_side_join(s);
    } else if (phase == 2 || phase == 3) {
        const ontovae_dims_t& d = ctx->dims;
        const ontovae_layout_t& lay = ctx->lay;
        OvAdam adam{};
        const OvBnBwdArgs bn{s.ws.da, s.ws.a, s.ws.xhat, s.ws.bn_invstd, ctx->P + lay.gamma, ctx->P + lay.beta, ctx->P + lay.b0,
                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ctx->Gr + lay.gamma, ctx->Gr + lay.beta,
                             ctx->Gr + lay.b0, s.ws.dh, s.ws.bn_bwd_local, bn_bwd_stats, d.L, B, B_global, s.ldb, 3, 0, adam,
                             ctx->drop, ctx->seed, rng_offset, 1, (int64_t)0, nullptr, 0};
        hipLaunchKernelGGL(ov_bn_bwd_kernel, dim3(d.L), dim3(256), 0, s.st, bn);
        OV_LAUNCH_CHECK();
        if (phase == 3) return ONTOVAE_OK;          // dW0 follows in row panels (ontovae_dw0_rows)
        return ov_dw0(s, nullptr, 0);
    }
    ontovae_set_error("train_step_dp: phase %d not in {0,1,2,3}", phase);
    return ONTOVAE_EINVAL;
}

extern "C" int ontovae_dw0_rows(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, int32_t l0, int32_t l1,
                                void* stream) {
    int rc = ov_check_ctx(ctx, B, false, true);
    if (rc) return rc;
    if (!x || !ctx->Gr) { ontovae_set_error("dw0_rows: x or the gradient buffer Gr is NULL"); return ONTOVAE_EINVAL; }
    if (l0 < 0 || l1 > ctx->dims.L || l0 >= l1) { ontovae_set_error("dw0_rows: rows [%d, %d) outside [0, %d)", l0, l1, ctx->dims.L); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, x, xidx, B, stream, &s);
    return ov_dw0(s, nullptr, 0, l0, l1);
}

extern "C" int ontovae_adamw_apply(const ontovae_ctx_t* ctx, const ontovae_adamw_t* opt, void* stream) {
    if (!ctx || !opt || !ctx->P || !ctx->M || !ctx->V || !ctx->Gr) { ontovae_set_error("adamw_apply: a required pointer is NULL"); return ONTOVAE_EINVAL; }
    const ontovae_layout_t& lay = ctx->lay;
    const int64_t n = lay.total;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(ov_adamw_flat_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ctx->P, ctx->M, ctx->V,
                       ctx->Gr, n, lay.dec_vals, lay.dec_vals + ctx->dims.nnz, ov_adam_from(opt));
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

extern "C" int ontovae_adamw_apply_range(const ontovae_ctx_t* ctx, const ontovae_adamw_t* opt, int64_t begin, int64_t end,
                                         void* stream) {
    if (!ctx || !opt || !ctx->P || !ctx->M || !ctx->V || !ctx->Gr) { ontovae_set_error("adamw_apply_range: a required pointer is NULL"); return ONTOVAE_EINVAL; }
    const ontovae_layout_t& lay = ctx->lay;
    if (begin < 0 || end > lay.total || begin > end) { ontovae_set_error("adamw_apply_range: bad range"); return ONTOVAE_EINVAL; }
    const int64_t n = end - begin;
    if (n == 0) return ONTOVAE_OK;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    // clamp range relative to the sub-buffer
    hipLaunchKernelGGL(ov_adamw_flat_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ctx->P + begin,
                       ctx->M + begin, ctx->V + begin, ctx->Gr + begin, n, lay.dec_vals - begin, lay.dec_vals + ctx->dims.nnz - begin,
                       ov_adam_from(opt));
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

extern "C" int ontovae_dp_bn_bwd(const ontovae_ctx_t* ctx, int32_t B, int32_t B_global, const float* bn_bwd_stats,
                                 uint64_t rng_offset, float* send, int32_t blm, void* stream) {
    int rc = ov_check_ctx(ctx, B, false, true);
    if (rc) return rc;
    if (!bn_bwd_stats || !send || !ctx->Gr) { ontovae_set_error("dp_bn_bwd: a required pointer is NULL"); return ONTOVAE_EINVAL; }
    if (blm < B || (blm & 3) || B_global < B) { ontovae_set_error("dp_bn_bwd: blm %d must be a multiple of 4 and >= B = %d", blm, B); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, nullptr, nullptr, B, stream, &s);
    if (blm > s.ldb) { ontovae_set_error("dp_bn_bwd: blm %d exceeds the padded batch %d", blm, s.ldb); return ONTOVAE_EINVAL; }
    const ontovae_dims_t& d = ctx->dims;
    const ontovae_layout_t& lay = ctx->lay;
    OvAdam adam{};
    const OvBnBwdArgs bn{s.ws.da, s.ws.a, s.ws.xhat, s.ws.bn_invstd, ctx->P + lay.gamma, ctx->P + lay.beta, ctx->P + lay.b0,
                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ctx->Gr + lay.gamma, ctx->Gr + lay.beta,
                         ctx->Gr + lay.b0, s.ws.dh, s.ws.bn_bwd_local, const_cast<float*>(bn_bwd_stats), d.L, B, B_global, s.ldb, 4, 0,
                         adam, ctx->drop, ctx->seed, rng_offset, 1, (int64_t)0, send, blm};
    OvProfScope prof(ONTOVAE_PHASE_BN_BWD, s.st);
    hipLaunchKernelGGL(ov_bn_bwd_kernel, dim3(d.L), dim3(256), 0, s.st, bn);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// ---- one exchange instead of two between the backward pass and dW0 (data parallel) --------------------------------
// After phase 1 a rank holds dy = dL/d(BatchNorm output) of its columns, xhat, and its local sums (sum dy, sum dy xhat).
// Instead of all-reducing the sums and then all-gathering dh, the ranks all-gather [dy | . | xhat | sums] once; every
// rank then forms the global sums (fixed rank order: bitwise the same everywhere) and dh of ALL columns itself.
// Block of one rank, `rstride` floats: [L][blm] dy -> dh in place | [L] b0 gradient of the block | [L][blm] xhat | [2 L] sums.
__global__ __launch_bounds__(256) void ov_dp_pack_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                        const float* __restrict__ sums, int L, int B, int ldb, int blm,
                                                        float* __restrict__ send) {
    const int l = blockIdx.x;
    float* d0 = send + (int64_t)l * blm;
    float* x0 = send + (int64_t)L * blm + L + (int64_t)l * blm;
    for (int b = threadIdx.x; b < blm; b += 256) {
        d0[b] = b < B ? dy[(int64_t)l * ldb + b] : 0.f;
        x0[b] = b < B ? xhat[(int64_t)l * ldb + b] : 0.f;
    }
    if (threadIdx.x == 0) {
        send[(int64_t)L * blm + l] = 0.f;
        send[(int64_t)2 * L * blm + L + l] = sums[l];
        send[(int64_t)2 * L * blm + 2 * L + l] = sums[L + l];
    }
}

struct OvDpSizes { int n[ONTOVAE_MAX_DP_RANKS]; };
__global__ __launch_bounds__(256) void ov_dp_dh_kernel(float* __restrict__ gathered, int64_t rstride, int R, const OvDpSizes sz, int L,
                                                      int blm, int Bglobal, const float* __restrict__ gamma,
                                                      const float* __restrict__ invstd, float* __restrict__ G_gamma,
                                                      float* __restrict__ G_beta) {
    // one wave per feature, as ov_bn_bwd_body
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (l >= L) return;
    float gdy = 0.f, gdyx = 0.f;
    for (int r = 0; r < R; ++r) {
        const float* sums = gathered + r * rstride + (int64_t)2 * L * blm + L;
        gdy += sums[l];
        gdyx += sums[L + l];
    }
    const float nb = (float)Bglobal;
    const float scale = gamma[l] * invstd[l] / nb;
    for (int r = 0; r < R; ++r) {
        float* blk = gathered + r * rstride;
        float* dyr = blk + (int64_t)l * blm;
        const float* xr = blk + (int64_t)L * blm + L + (int64_t)l * blm;
        float sdh = 0.f;
        for (int b = lane; b < blm; b += 64) {
            float v = 0.f;
            if (b < sz.n[r]) {
                v = scale * (nb * dyr[b] - gdy - xr[b] * gdyx);
                sdh += v;
            }
            dyr[b] = v;
        }
        sdh = ov_wave_sum(sdh);
        if (lane == 0) blk[(int64_t)L * blm + l] = sdh;
    }
    if (lane == 0) { G_gamma[l] = gdyx; G_beta[l] = gdy; }
}

extern "C" int ontovae_dp_pack_dy(const ontovae_ctx_t* ctx, int32_t B, float* send, int32_t blm, void* stream) {
    int rc = ov_check_ctx(ctx, B, false, true);
    if (rc) return rc;
    if (!send) { ontovae_set_error("dp_pack_dy: send is NULL"); return ONTOVAE_EINVAL; }
    if (blm < B || (blm & 3)) { ontovae_set_error("dp_pack_dy: blm %d must be a multiple of 4 and >= B = %d", blm, B); return ONTOVAE_EINVAL; }
    OvStep s;
    ov_make_step(ctx, nullptr, nullptr, B, stream, &s);
    const ontovae_dims_t& d = ctx->dims;
    hipLaunchKernelGGL(ov_dp_pack_kernel, dim3(d.L), dim3(256), 0, s.st, s.ws.dh, s.ws.xhat, s.ws.bn_bwd_local, d.L, B, s.ldb, blm, send);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

extern "C" int ontovae_dp_dh_global(const ontovae_ctx_t* ctx, int32_t B_global, int32_t n_ranks, const int32_t* sizes, int32_t blm,
                                    float* gathered, int64_t rstride, void* stream) {
    if (!ctx || !gathered || !sizes || !ctx->Gr || !ctx->workspace) { ontovae_set_error("dp_dh_global: a required pointer is NULL"); return ONTOVAE_EINVAL; }
    const ontovae_dims_t& d = ctx->dims;
    if (n_ranks < 1 || n_ranks > ONTOVAE_MAX_DP_RANKS || blm < 4 || (blm & 3) || rstride < (int64_t)2 * d.L * blm + 3 * d.L) {
        ontovae_set_error("dp_dh_global: n_ranks = %d (at most %d), blm = %d, rstride = %lld", n_ranks, ONTOVAE_MAX_DP_RANKS, blm, (long long)rstride);
        return ONTOVAE_EINVAL;
    }
    OvDpSizes sz{};
    int total = 0;
    for (int r = 0; r < n_ranks; ++r) {
        if (sizes[r] < 0 || sizes[r] > blm) { ontovae_set_error("dp_dh_global: rank %d holds %d rows, blm = %d", r, sizes[r], blm); return ONTOVAE_EINVAL; }
        sz.n[r] = sizes[r];
        total += sizes[r];
    }
    if (total != B_global) { ontovae_set_error("dp_dh_global: the ranks hold %d rows, B_global = %d", total, B_global); return ONTOVAE_EINVAL; }
    OvWs ws;
    ov_carve(d, ctx->max_batch, ov_train_cap(ctx), ctx->workspace, &ws);
    const ontovae_layout_t& lay = ctx->lay;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    OvProfScope prof(ONTOVAE_PHASE_BN_BWD, st);
    hipLaunchKernelGGL(ov_dp_dh_kernel, dim3((d.L + 3) / 4), dim3(256), 0, st, gathered, rstride, n_ranks, sz, d.L, blm, B_global,
                       ctx->P + lay.gamma, ws.bn_invstd, ctx->Gr + lay.gamma, ctx->Gr + lay.beta);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

extern "C" int ontovae_dw0_global(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, const int32_t* colmap,
                                  int32_t n_ranks, int32_t blm, const float* gathered, int64_t rstride,
                                  const ontovae_adamw_t* opt, int32_t apply_update, void* stream) {
    if (!ctx || !x || (!xidx && !colmap) || !gathered || n_ranks < 1 || blm < 4 || (blm & 3) || (rstride & 3) ||
        rstride < (int64_t)ctx->dims.L * (blm + 1)) {
        ontovae_set_error("dw0_global: bad arguments");
        return ONTOVAE_EINVAL;
    }
    if (apply_update && (!opt || !ctx->M || !ctx->V)) { ontovae_set_error("dw0_global: optimizer state missing"); return ONTOVAE_EINVAL; }
    if (!apply_update && !ctx->Gr) { ontovae_set_error("dw0_global: gradient buffer Gr is NULL"); return ONTOVAE_EINVAL; }
    const ontovae_dims_t& d = ctx->dims;
    OvGemm g{};
    g.M = d.L; g.N = d.G; g.K = n_ranks * blm;
    g.A = gathered; g.lda = blm; g.a_blm = blm; g.a_rstride = rstride;
    g.B = x; g.ldb = d.G; g.bgather = xidx; g.colmap = colmap;
    g.ldc = d.G;
    g.adam = ov_adam_from(opt);
    g.b0sum = ctx->Gr ? ctx->Gr + ctx->lay.b0 : nullptr;       // gradient of b0: sum of the ranks' local sums
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    OvProfScope prof(ONTOVAE_PHASE_DW0, st);
    if (apply_update) {
        g.C = ctx->P + ctx->lay.W0; g.Pm = ctx->M + ctx->lay.W0; g.Pv = ctx->V + ctx->lay.W0;
        return ov_launch_dw0<OV_EPI_ADAMW>(g, st);
    }
    g.C = ctx->Gr + ctx->lay.W0;
    return ov_launch_dw0<OV_EPI_STORE>(g, st);
}

// =================================================================================================
// C ABI: batcher + state_dict helpers
// =================================================================================================
extern "C" int ontovae_compose_order(const int32_t* order_in, const int64_t* perm, int32_t* order_out, int64_t n, void* stream) {
    if (!order_in || !perm || !order_out || n < 0 || order_in == order_out) { ontovae_set_error("compose_order: bad arguments"); return ONTOVAE_EINVAL; }
    if (n == 0) return ONTOVAE_OK;
    hipLaunchKernelGGL(ov_compose_order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       order_in, perm, order_out, n);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

extern "C" int ontovae_gather_rows(const float* X, int64_t ld, const int32_t* idx, int32_t B, int32_t G, float* out, void* stream) {
    if (!X || !idx || !out || B < 0 || G < 1 || ld < G) { ontovae_set_error("gather_rows: bad arguments"); return ONTOVAE_EINVAL; }
    if (B == 0) return ONTOVAE_OK;
    hipLaunchKernelGGL(ov_gather_rows_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, ld, idx, G, out);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

extern "C" int ontovae_vals_to_dense(const float* vals, const int32_t* entry_row, const int32_t* entry_col, int64_t k0, int64_t k1,
                                     float* dense, int64_t ld, void* stream) {
    if (!vals || !entry_row || !entry_col || !dense || k1 < k0) { ontovae_set_error("vals_to_dense: bad arguments"); return ONTOVAE_EINVAL; }
    if (k1 == k0) return ONTOVAE_OK;
    hipLaunchKernelGGL(ov_vals_to_dense_kernel, dim3((unsigned)((k1 - k0 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       vals, entry_row, entry_col, k0, k1, dense, ld);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

extern "C" int ontovae_dense_to_vals(const float* dense, int64_t ld, const int32_t* entry_row, const int32_t* entry_col, int64_t k0,
                                     int64_t k1, float* vals, void* stream) {
    if (!vals || !entry_row || !entry_col || !dense || k1 < k0) { ontovae_set_error("dense_to_vals: bad arguments"); return ONTOVAE_EINVAL; }
    if (k1 == k0) return ONTOVAE_OK;
    hipLaunchKernelGGL(ov_dense_to_vals_kernel, dim3((unsigned)((k1 - k0 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       dense, ld, entry_row, entry_col, k0, k1, vals);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

#include "ontovae_dense.h"
