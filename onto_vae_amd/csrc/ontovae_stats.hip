// Statistics that follow a perturbation screen, on the device: the paired Wilcoxon signed-rank test per ontology term
// (or gene) between perturbed and control activities, and the Benjamini-Hochberg correction of its p-values.
//
// Reference: onto_vae/ontobj.py:651-701 (Ontobj.wilcox_test), which calls, per column,
//   scipy.stats.wilcoxon(perturbed[:, i], control[:, i], zero_method='zsplit', alternative='greater' | 'less')
// and statsmodels.stats.multitest.fdrcorrection(pvals) on the result.  Neither library is part of the reference tree;
// the kernels restate scipy 1.15's _wilcoxon_nd (method='auto', correction=False) and the 'indep' form of fdrcorrection:
//   d = x - y in float32, ranks of |d| with average ties, zeros ranked and split evenly between r_plus and r_minus;
//   n > 50, or 13 < n <= 50 with ties or zeros: normal approximation with tie correction;
//   n <= 50 without ties and zeros: exact null distribution;  n <= 13 with ties or zeros: all 2^n sign assignments.
//
// Layout: both matrices are feature-major, column c of the test is the contiguous run a[c * ld .. c * ld + n).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "ontovae_internal.h"

#define OV_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            ontovae_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ONTOVAE_EHIP;                                                                 \
        }                                                                                        \
    } while (0)
#define OV_LAUNCH_CHECK() OV_HIP_CHECK(hipGetLastError())

typedef unsigned long long u64;

// sort code of one difference: |d| bits in 31..1, bit 0 = (d > 0); zeros (either sign) are code 0
__device__ __forceinline__ uint32_t ov_wx_code(float d, bool* nan) {
    const uint32_t u = __float_as_uint(d), key = u & 0x7FFFFFFFu;
    if (key > 0x7F800000u) *nan = true;
    return (key << 1) | ((key != 0u && !(u >> 31)) ? 1u : 0u);
}

// scipy.special.ndtr (cephes): Phi(x) through erf near 0 and erfc in the tails
__device__ __forceinline__ double ov_ndtr(double x) {
    const double SQRTH = 0.70710678118654752440;
    const double t = x * SQRTH, z = fabs(t);
    if (z < SQRTH) return 0.5 + 0.5 * erf(t);
    const double y = 0.5 * erfc(z);
    return t > 0 ? 1.0 - y : y;
}

// normal approximation of _wilcoxon_statistic / _wilcoxon_nd: plus4 = 4 r_plus (zero half included), tie = sum t^3 - t
__device__ __forceinline__ void ov_wx_asymptotic(int n, u64 plus4, u64 tie, int alt, double* stat, double* p) {
    const double count = (double)n;
    const double r_plus = (double)plus4 * 0.25;
    const double mn = count * (count + 1.) * 0.25;
    double se = count * (count + 1.) * (2. * count + 1.);
    se -= (double)tie / 2;
    se = sqrt(se / 24);
    const double z = (r_plus - mn) / se;
    const double r_minus = count * (count + 1.) * 0.5 - r_plus;
    if (alt == ONTOVAE_ALT_GREATER) { *stat = r_plus; *p = ov_ndtr(-z); }
    else if (alt == ONTOVAE_ALT_LESS) { *stat = r_plus; *p = ov_ndtr(z); }
    else { *stat = fmin(r_plus, r_minus); *p = 2 * ov_ndtr(-fabs(z)); }
}

__device__ __forceinline__ u64 ov_wave_sum(u64 v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// n > 50: one workgroup of NT threads per column, E codes per thread in registers (n2 = NT * E >= n, padded with the
// largest code).  Bitonic network over the index bits [wave | lane | register]:
//   partner distance < E        compare-exchange between two registers of one thread;
//   E <= distance < 64 E        between lanes of one wave (cross-lane read, no barrier);
//   distance >= 64 E            between waves, through LDS (only log2(NT / 64) (log2(NT / 64) + 1) / 2 stages).
// The ranks are then read off the sorted positions in the registers (see the end of the kernel).
// ---------------------------------------------------------------------------------------------------------------------
// median of three: med3(x, y, 0) = min(x, y), med3(x, y, ~0) = max(x, y) -- one instruction for a compare-exchange half
// whose direction is only known at run time
__device__ __forceinline__ uint32_t ov_med3(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}

template <int E, int J>
__device__ __forceinline__ void ov_bitonic_regs(uint32_t (&v)[E], uint32_t lo_sel, int k) {
    // stages J, J/2, .. 1 inside the thread; direction from bit k of the element index t * E + r: a compile-time
    // property of r while k < E, the thread's lo_sel (0: ascending, ~0: descending) from then on
    if constexpr (J >= 1) {
#pragma unroll
        for (int r = 0; r < E; ++r)
            if ((r & J) == 0) {
                const uint32_t x = v[r], y = v[r | J];
                if (k >= E) {
                    v[r] = ov_med3(x, y, lo_sel);
                    v[r | J] = ov_med3(x, y, ~lo_sel);
                } else {
                    const bool asc = (r & k) == 0;
                    v[r] = asc ? min(x, y) : max(x, y);
                    v[r | J] = asc ? max(x, y) : min(x, y);
                }
            }
        ov_bitonic_regs<E, J / 2>(v, lo_sel, k);
    }
}
// merges of length 2, 4, .. E inside the thread
template <int E, int K>
__device__ __forceinline__ void ov_bitonic_thread_sort(uint32_t (&v)[E], int t) {
    if constexpr (K <= E) {
        ov_bitonic_regs<E, K / 2>(v, (t & 1) ? 0xFFFFFFFFu : 0u, K);
        ov_bitonic_thread_sort<E, K * 2>(v, t);
    }
}

template <int NT, int E>
__global__ __launch_bounds__(NT) void ov_wilcoxon_sort_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t ld,
                                                              int n, int alt, double* __restrict__ stat, double* __restrict__ pval) {
    constexpr int N2 = NT * E;
    extern __shared__ uint32_t codes[];
    __shared__ u64 red[3][NT / 64];
    __shared__ int s_nan;
    const int tid = threadIdx.x, col = blockIdx.x;
    const float* pa = a + (int64_t)col * ld;
    const float* pb = b + (int64_t)col * ld;
    if (tid == 0) s_nan = 0;
    bool nan = false;
    uint32_t v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {            // the input order is irrelevant: coalesced loads
        const int i = r * NT + tid;
        v[r] = i < n ? ov_wx_code(pa[i] - pb[i], &nan) : 0xFFFFFFFFu;
    }
    ov_bitonic_thread_sort<E, 2>(v, tid);
#pragma unroll 1
    for (int k = 2 * E; k <= N2; k <<= 1) {
        const bool asc = ((tid * E) & k) == 0;
        int j = k >> 1;
#pragma unroll 1
        for (; j >= 4 * E; j >>= 1) {
            const int m = j / E;                                  // partner thread = tid ^ m
            const uint32_t sel = (((tid & m) == 0) == asc) ? 0u : 0xFFFFFFFFu;     // 0: keep the smaller one
            if (m == 8) {                 // lane ^ 8 inside a row of 16: two DPP row shifts, each writing its half of the banks
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    int o = __builtin_amdgcn_update_dpp(0, (int)v[r], 0x108, 0xF, 0x3, false);       // row_shl:8 -> lanes 0..7 of the row
                    o = __builtin_amdgcn_update_dpp(o, (int)v[r], 0x118, 0xF, 0xC, false);           // row_shr:8 -> lanes 8..15
                    v[r] = ov_med3(v[r], (uint32_t)o, sel);
                }
            } else if (m == 4) {
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    int o = __builtin_amdgcn_update_dpp(0, (int)v[r], 0x104, 0xF, 0x5, false);       // row_shl:4 -> banks 0 and 2
                    o = __builtin_amdgcn_update_dpp(o, (int)v[r], 0x114, 0xF, 0xA, false);           // row_shr:4 -> banks 1 and 3
                    v[r] = ov_med3(v[r], (uint32_t)o, sel);
                }
            } else if (m >= 64) {                                        // other wave: through LDS
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; ++r) codes[r * NT + tid] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; ++r) v[r] = ov_med3(v[r], codes[r * NT + (tid ^ m)], sel);
            } else {
#pragma unroll
                for (int r = 0; r < E; ++r) v[r] = ov_med3(v[r], (uint32_t)__shfl_xor((int)v[r], m, 64), sel);
            }
        }
        if (j == 2 * E) {                                         // partner lane ^ 2: DPP quad_perm [2, 3, 0, 1]
            const uint32_t sel = (((tid & 2) == 0) == asc) ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] = ov_med3(v[r], (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[r], 0x4E, 0xF, 0xF, true), sel);
        }
        {                                                         // partner lane ^ 1: DPP quad_perm [1, 0, 3, 2]
            const uint32_t sel = (((tid & 1) == 0) == asc) ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] = ov_med3(v[r], (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[r], 0xB1, 0xF, 0xF, true), sel);
        }
        ov_bitonic_regs<E, E / 2>(v, asc ? 0u : 0xFFFFFFFFu, k);
    }
    // Ranks straight from the registers: thread t holds the sorted positions t E .. t E + E - 1.  A run of equal keys
    // [first, last) gives each member the average rank (first + 1 + last) / 2; `first` is the position of the latest
    // run head at or before the element (a max-scan over the positions), `last` that of the next head after it (a
    // min-scan from the right), both as a pass over the thread's registers plus one scan across the threads.  The
    // cost does not depend on how many ties the data holds.
    constexpr int NW = NT / 64;
    int* xch = reinterpret_cast<int*>(codes);          // [0, NW): last key of each wave; [NW, 2 NW): scan carries; [2 NW, 3 NW)
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();                                   // the sort's last LDS stage is read
    if (nan) s_nan = 1;
    if (lane == 63) xch[wave] = (int)(v[E - 1] >> 1);
    __syncthreads();
    uint32_t left = (uint32_t)__shfl_up((int)(v[E - 1] >> 1), 1, 64);
    if (lane == 0 && wave > 0) left = (uint32_t)xch[wave - 1];
    bool head[E];
    head[0] = tid == 0 || (v[0] >> 1) != left;
#pragma unroll
    for (int r = 1; r < E; ++r) head[r] = (v[r] >> 1) != (v[r - 1] >> 1);
    // forward: latest head at or before each position
    int first[E], lh = -1;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (head[r]) lh = tid * E + r;
        first[r] = lh;
    }
    int incl = lh;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl = max(incl, t);
    }
    if (lane == 63) xch[NW + wave] = incl;
    // backward: next head after each position (N2 if none)
    int fh = N2;
#pragma unroll
    for (int r = E - 1; r >= 0; --r)
        if (head[r]) fh = tid * E + r;
    int rincl = fh;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_down(rincl, o, 64);
        if (lane + o < 64) rincl = min(rincl, t);
    }
    if (lane == 0) xch[2 * NW + wave] = rincl;
    __syncthreads();
    int carry = __shfl_up(incl, 1, 64);
    if (lane == 0) carry = -1;
    for (int w = 0; w < wave; ++w) carry = max(carry, xch[NW + w]);
    int rcarry = __shfl_down(rincl, 1, 64);
    if (lane == 63) rcarry = N2;
    for (int w = wave + 1; w < NW; ++w) rcarry = min(rcarry, xch[2 * NW + w]);
    u64 plus4 = 0, zero4 = 0, tie = 0;
    uint32_t plus32 = 0, zero32 = 0;
    int nxt = rcarry;
#pragma unroll
    for (int r = E - 1; r >= 0; --r) {
        const int i = tid * E + r;
        const int f = max(first[r], carry), l = nxt;          // the run of position i is [f, l)
        if (head[r]) nxt = i;
        if (i < n) {
            const uint32_t r4 = 2u * (uint32_t)(f + l + 1);
            const uint32_t key = v[r] >> 1;
            if (key == 0u) zero32 += r4;
            else if (v[r] & 1u) plus32 += r4;
            if (head[r]) { const u64 t = (u64)(l - f); tie += t * t * t - t; }
        }
    }
    plus4 = plus32; zero4 = zero32;
    __syncthreads();                                   // xch is free again (red[] below is separate static LDS)
    plus4 = ov_wave_sum(plus4); zero4 = ov_wave_sum(zero4); tie = ov_wave_sum(tie);
    if ((tid & 63) == 0) { red[0][tid >> 6] = plus4; red[1][tid >> 6] = zero4; red[2][tid >> 6] = tie; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NT / 64; ++w) { plus4 += red[0][w]; zero4 += red[1][w]; tie += red[2][w]; }
        double s, p;
        ov_wx_asymptotic(n, plus4 + zero4 / 2, tie, alt, &s, &p);
        if (s_nan) s = p = __longlong_as_double(0x7FF8000000000000ll);
        stat[col] = s;
        pval[col] = p;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// n <= 50: one wave per column, lane i holds d_i; ranks by counting
// ---------------------------------------------------------------------------------------------------------------------
#define OV_WX_SMALL 50
#define OV_WX_PERM 13
__device__ __forceinline__ double ov_wave_sum_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return __shfl(v, 0, 64);
}

__global__ __launch_bounds__(64) void ov_wilcoxon_small_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t ld, int n,
                                                               int alt, double* __restrict__ stat, double* __restrict__ pval) {
    __shared__ double dist[2][OV_WX_SMALL * (OV_WX_SMALL + 1) / 2 + 1];
    __shared__ uint32_t r4s[64];
    const int lane = threadIdx.x, col = blockIdx.x;
    bool nan = false;
    uint32_t code = 0xFFFFFFFFu;
    if (lane < n) code = ov_wx_code(a[(int64_t)col * ld + lane] - b[(int64_t)col * ld + lane], &nan);
    const uint32_t key = code >> 1;
    int lo = 0, hi = 0;
    for (int k = 0; k < n; ++k) {
        const uint32_t other = __shfl(key, k, 64);
        lo += other < key;
        hi += other <= key;
    }
    const bool live = lane < n, zero = live && key == 0u, pos = live && !zero && (code & 1u);
    const uint32_t r4 = live ? 2u * (uint32_t)(lo + hi + 1) : 0u;
    const int t = hi - lo;
    const u64 zero4 = ov_wave_sum(zero ? r4 : 0u), plus4r = ov_wave_sum(pos ? r4 : 0u), tie = ov_wave_sum(live ? (u64)(t * t - 1) : 0ull);
    const u64 plus4 = __shfl(plus4r + zero4 / 2, 0, 64);
    const bool any_nan = __any(nan), ties_or_zeros = __any(live && (t > 1 || zero));
    double s = 0, p = 0;
    if (!ties_or_zeros) {
        // exact null distribution of r_plus (scipy _get_wilcoxon_distr): c_k[j] = (c_{k-1}[j] + c_{k-1}[j - k]) / 2
        int cur = 0, len = 1;
        if (lane == 0) dist[0][0] = 1.0;
        for (int k = 1; k <= n; ++k) {
            __syncthreads();
            const int nlen = k * (k + 1) / 2 + 1, shift = nlen - len;
            for (int j = lane; j < nlen; j += 64) {
                double v = j < len ? dist[cur][j] * 0.5 : 0.0;
                if (j >= shift) v += dist[cur][j - shift] * 0.5;
                dist[cur ^ 1][j] = v;
            }
            cur ^= 1; len = nlen;
        }
        __syncthreads();
        const int k = (int)(plus4 / 4);                       // integral: no ties, no zeros
        const double mn = (double)n * ((double)n + 1) / 4;
        // sf(k) = P(R >= k), cdf(k) = P(R <= k), each summed on its short side (WilcoxonDistribution.sf / .cdf)
        double head = 0, tail = 0, headk = 0, tailk = 0;      // sum [0, k), sum (k, len), sum [0, k], sum [k, len)
        for (int j = lane; j < len; j += 64) {
            const double v = dist[cur][j];
            if (j < k) head += v;
            if (j > k) tail += v;
            if (j <= k) headk += v;
            if (j >= k) tailk += v;
        }
        head = ov_wave_sum_f64(head); tail = ov_wave_sum_f64(tail); headk = ov_wave_sum_f64(headk); tailk = ov_wave_sum_f64(tailk);
        const double sf = (double)k <= mn ? tailk : 1 - head, cdf = (double)k <= mn ? headk : 1 - tail;
        const double r_plus = (double)plus4 * 0.25, r_minus = (double)n * ((double)n + 1) * 0.5 - r_plus;
        if (alt == ONTOVAE_ALT_GREATER) { s = r_plus; p = sf; }
        else if (alt == ONTOVAE_ALT_LESS) { s = r_plus; p = cdf; }
        else { s = fmin(r_plus, r_minus); p = fmin(fmax(2 * fmin(sf, cdf), 0.0), 1.0); }
    } else if (n <= OV_WX_PERM) {
        // scipy.stats.permutation_test(permutation_type='samples') enumerates the 2^n sign assignments when n <= 13
        r4s[lane] = (live && !zero) ? r4 : 0u;
        __syncthreads();
        const uint32_t half_zero = (uint32_t)__shfl(zero4 / 2, 0, 64);
        u64 ge = 0, le = 0;
        for (uint32_t m = lane; m < (1u << n); m += 64) {
            uint32_t v = half_zero;
            for (int i = 0; i < n; ++i) v += ((m >> i) & 1u) ? r4s[i] : 0u;
            ge += v >= (uint32_t)plus4;
            le += v <= (uint32_t)plus4;
        }
        ge = __shfl(ov_wave_sum(ge), 0, 64); le = __shfl(ov_wave_sum(le), 0, 64);
        const double total = (double)(1u << n), pg = (double)ge / total, pl = (double)le / total;
        const double r_plus = (double)plus4 * 0.25, r_minus = (double)n * ((double)n + 1) * 0.5 - r_plus;
        if (alt == ONTOVAE_ALT_GREATER) { s = r_plus; p = pg; }
        else if (alt == ONTOVAE_ALT_LESS) { s = r_plus; p = pl; }
        else { s = fmin(r_plus, r_minus); p = fmin(fmax(2 * fmin(pg, pl), 0.0), 1.0); }
    } else {
        ov_wx_asymptotic(n, plus4, __shfl(tie, 0, 64), alt, &s, &p);
    }
    if (any_nan) s = p = __longlong_as_double(0x7FF8000000000000ll);
    if (lane == 0) { stat[col] = s; pval[col] = p; }
}

// ---------------------------------------------------------------------------------------------------------------------
// n beyond what one workgroup's LDS holds: codes to the workspace, sorted there per column by a workgroup that merges
// LDS-sorted runs (each pass doubles the run length), then ranks from the sorted column
// ---------------------------------------------------------------------------------------------------------------------
#define OV_WX_RUN 16384          // codes per LDS-sorted run
__global__ __launch_bounds__(1024) void ov_wilcoxon_runs_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t ld, int n,
                                                                int64_t nruns, uint32_t* __restrict__ out, int64_t out_ld,
                                                                int* __restrict__ nanflag) {
    extern __shared__ uint32_t codes[];
    const int tid = threadIdx.x;
    const int64_t col = blockIdx.x / nruns, run = blockIdx.x % nruns;
    const float* pa = a + col * ld;
    const float* pb = b + col * ld;
    const int64_t i0 = run * OV_WX_RUN;
    bool nan = false;
    for (int i = tid; i < OV_WX_RUN; i += 1024) codes[i] = i0 + i < n ? ov_wx_code(pa[i0 + i] - pb[i0 + i], &nan) : 0xFFFFFFFFu;
    for (int k = 2; k <= OV_WX_RUN; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = tid; t < OV_WX_RUN / 2; t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                const uint32_t x = codes[i], y = codes[p];
                if ((x > y) == ((i & k) == 0)) { codes[i] = y; codes[p] = x; }
            }
        }
    __syncthreads();
    if (nan) nanflag[col] = 1;
    for (int i = tid; i < OV_WX_RUN; i += 1024) out[col * out_ld + i0 + i] = codes[i];
}

// one merge pass: runs of `len` codes in src -> runs of 2 len in dst; each thread places one element by binary search in
// the sibling run (stable: elements of the left run go before equal elements of the right run)
__global__ __launch_bounds__(256) void ov_wilcoxon_merge_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t ld,
                                                                int64_t len, int64_t total) {
    const int64_t col = blockIdx.y;
    const uint32_t* s = src + col * ld;
    uint32_t* d = dst + col * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pair = i / (2 * len) * (2 * len), off = i - pair;
        const bool left = off < len;
        const int64_t self0 = left ? pair : pair + len, sib0 = left ? pair + len : pair;
        const int64_t sib_n = std::max<int64_t>(0, std::min<int64_t>(len, total - sib0));
        const uint32_t v = s[i];
        int64_t lo = 0, hi = sib_n;
        while (lo < hi) {
            const int64_t m = (lo + hi) >> 1;
            const uint32_t w = s[sib0 + m];
            if (left ? (w < v) : (w <= v)) lo = m + 1; else hi = m;
        }
        d[pair + (i - self0) + lo] = v;
    }
}

__global__ __launch_bounds__(1024) void ov_wilcoxon_ranks_kernel(const uint32_t* __restrict__ sorted, int64_t ld, int n, int alt,
                                                                 const int* __restrict__ nanflag, double* __restrict__ stat,
                                                                 double* __restrict__ pval) {
    __shared__ u64 red[3][16];
    const int tid = threadIdx.x, col = blockIdx.x;
    const uint32_t* codes = sorted + (int64_t)col * ld;
    u64 plus4 = 0, zero4 = 0, tie = 0;
    for (int i = tid; i < n; i += 1024) {
        const uint32_t c = codes[i], key = c >> 1;
        const bool prev_same = i > 0 && (codes[i - 1] >> 1) == key, next_same = i + 1 < n && (codes[i + 1] >> 1) == key;
        u64 r4 = 4ull * (u64)(i + 1);
        if (prev_same || next_same) {
            int lo = 0, hi = i;
            while (lo < hi) { const int m = (lo + hi) >> 1; if ((codes[m] >> 1) < key) lo = m + 1; else hi = m; }
            const int first = lo;
            lo = i + 1; hi = n;
            while (lo < hi) { const int m = (lo + hi) >> 1; if ((codes[m] >> 1) <= key) lo = m + 1; else hi = m; }
            r4 = 2ull * (u64)(first + lo + 1);
            if (!prev_same) { const u64 t = (u64)(lo - first); tie += t * t * t - t; }
        }
        if (key == 0u) zero4 += r4;
        else if (c & 1u) plus4 += r4;
    }
    plus4 = ov_wave_sum(plus4); zero4 = ov_wave_sum(zero4); tie = ov_wave_sum(tie);
    if ((tid & 63) == 0) { red[0][tid >> 6] = plus4; red[1][tid >> 6] = zero4; red[2][tid >> 6] = tie; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) { plus4 += red[0][w]; zero4 += red[1][w]; tie += red[2][w]; }
        double s, p;
        ov_wx_asymptotic(n, plus4 + zero4 / 2, tie, alt, &s, &p);
        if (nanflag[col]) s = p = __longlong_as_double(0x7FF8000000000000ll);
        stat[col] = s;
        pval[col] = p;
    }
}

static int ov_stats_allow_lds(const void* kern, size_t bytes) {
    if (bytes <= 64 * 1024) return ONTOVAE_OK;
    OV_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return ONTOVAE_OK;
}

#define OV_WX_LDS_MAX 32768      // columns up to this length are sorted inside one workgroup's LDS

// per column of the large-n path: two code buffers of ld = n rounded up to whole runs, and one flag
static int64_t ov_wx_large_ld(int64_t n) { return (n + OV_WX_RUN - 1) / OV_WX_RUN * OV_WX_RUN; }

extern "C" int64_t ontovae_wilcoxon_workspace_bytes(int32_t n, int32_t ncols) {
    if (n <= OV_WX_LDS_MAX || ncols <= 0) return 0;
    // at least one column; as many as fit in 1 GiB otherwise (the call walks the columns in groups)
    const int64_t per_col = 2 * ov_wx_large_ld(n) * 4 + 256;
    const int64_t cols = std::max<int64_t>(1, std::min<int64_t>(ncols, (int64_t(1) << 30) / per_col));
    return cols * per_col;
}

extern "C" int ontovae_wilcoxon_zsplit(const float* perturbed, const float* control, int64_t ld, int32_t ncols, int32_t n,
                                       int32_t alternative, double* stat, double* pval, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
    if (ncols < 0 || n < 0 || ld < n) { ontovae_set_error("wilcoxon_zsplit: ncols = %d, n = %d, ld = %lld", ncols, n, (long long)ld); return ONTOVAE_EINVAL; }
    if (alternative != ONTOVAE_ALT_GREATER && alternative != ONTOVAE_ALT_LESS && alternative != ONTOVAE_ALT_TWO_SIDED) {
        ontovae_set_error("wilcoxon_zsplit: alternative %d is not one of ONTOVAE_ALT_*", alternative);
        return ONTOVAE_EINVAL;
    }
    if (ncols == 0) return ONTOVAE_OK;
    if (!perturbed || !control || !stat || !pval) { ontovae_set_error("wilcoxon_zsplit: NULL argument"); return ONTOVAE_EINVAL; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n == 0) {                  // scipy: empty input -> NaN, NaN
        OV_HIP_CHECK(hipMemsetAsync(stat, 0xFF, sizeof(double) * ncols, st));
        OV_HIP_CHECK(hipMemsetAsync(pval, 0xFF, sizeof(double) * ncols, st));
        return ONTOVAE_OK;
    }
    if (n <= OV_WX_SMALL) {
        hipLaunchKernelGGL(ov_wilcoxon_small_kernel, dim3(ncols), dim3(64), 0, st, perturbed, control, ld, n, alternative, stat, pval);
        OV_LAUNCH_CHECK();
        return ONTOVAE_OK;
    }
    if (n <= OV_WX_LDS_MAX) {
        int rc;
#define OV_WX_LAUNCH(NT, E)                                                                                                   \
        do {                                                                                                                  \
            if ((rc = ov_stats_allow_lds(reinterpret_cast<const void*>(ov_wilcoxon_sort_kernel<NT, E>), (size_t)NT * E * 4))) return rc; \
            hipLaunchKernelGGL((ov_wilcoxon_sort_kernel<NT, E>), dim3(ncols), dim3(NT), (size_t)NT * E * 4, st, perturbed, control, ld, n, \
                               alternative, stat, pval);                                                                      \
        } while (0)
        if (n <= 256) OV_WX_LAUNCH(256, 1);
        else if (n <= 512) OV_WX_LAUNCH(256, 2);
        else if (n <= 1024) OV_WX_LAUNCH(256, 4);
        else if (n <= 2048) OV_WX_LAUNCH(256, 8);
        else if (n <= 4096) OV_WX_LAUNCH(256, 16);          // (128, 32) and (512, 8) measured slower: 106 and 95 us against 82
        else if (n <= 8192) OV_WX_LAUNCH(1024, 8);
        else if (n <= 16384) OV_WX_LAUNCH(1024, 16);
        else OV_WX_LAUNCH(1024, 32);
#undef OV_WX_LAUNCH
        OV_LAUNCH_CHECK();
        return ONTOVAE_OK;
    }
    // large n: groups of columns through the workspace
    const int64_t cld = ov_wx_large_ld(n), per_col = 2 * cld * 4 + 256;
    const int64_t group = workspace ? std::min<int64_t>(workspace_bytes / per_col, 65535) : 0;      // grid rows of the merge passes
    if (group < 1) {
        ontovae_set_error("wilcoxon_zsplit: n = %d needs a workspace of at least %lld bytes (ontovae_wilcoxon_workspace_bytes)", n, (long long)per_col);
        return ONTOVAE_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) { ontovae_set_error("wilcoxon_zsplit: workspace must be 256-byte aligned"); return ONTOVAE_EINVAL; }
    int rc;
    if ((rc = ov_stats_allow_lds(reinterpret_cast<const void*>(ov_wilcoxon_runs_kernel), OV_WX_RUN * 4))) return rc;
    const int64_t nruns = cld / OV_WX_RUN;
    for (int64_t c0 = 0; c0 < ncols; c0 += group) {
        const int64_t nc = std::min<int64_t>(group, ncols - c0);
        uint32_t* buf0 = reinterpret_cast<uint32_t*>(workspace);
        uint32_t* buf1 = buf0 + nc * cld;
        int* nanflag = reinterpret_cast<int*>(buf1 + nc * cld);
        OV_HIP_CHECK(hipMemsetAsync(nanflag, 0, sizeof(int) * nc, st));
        hipLaunchKernelGGL(ov_wilcoxon_runs_kernel, dim3((unsigned)(nc * nruns)), dim3(1024), OV_WX_RUN * 4, st, perturbed + c0 * ld,
                           control + c0 * ld, ld, n, nruns, buf0, cld, nanflag);
        OV_LAUNCH_CHECK();
        uint32_t *src = buf0, *dst = buf1;
        for (int64_t len = OV_WX_RUN; len < cld; len <<= 1) {
            const unsigned bx = (unsigned)std::min<int64_t>((cld + 255) / 256, 4096);
            hipLaunchKernelGGL(ov_wilcoxon_merge_kernel, dim3(bx, (unsigned)nc), dim3(256), 0, st, src, dst, cld, len, cld);
            OV_LAUNCH_CHECK();
            std::swap(src, dst);
        }
        hipLaunchKernelGGL(ov_wilcoxon_ranks_kernel, dim3((unsigned)nc), dim3(1024), 0, st, src, cld, n, alternative, nanflag, stat + c0, pval + c0);
        OV_LAUNCH_CHECK();
    }
    return ONTOVAE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Benjamini-Hochberg ('indep' fdrcorrection): q_(i) = min_{j >= i} p_(j) n / j, capped at 1, returned in input order.
// Ranks by counting (rank_i = #{k : p_k <= p_i}; tied p-values end up with the same q either way), so no sort and no
// scratch beyond `raw`; nsets independent sets of n p-values each, one grid row per set.
// ---------------------------------------------------------------------------------------------------------------------
#define OV_FDR_TILE 1024
__global__ __launch_bounds__(256) void ov_fdr_raw_kernel(const double* __restrict__ p, int n, double* __restrict__ raw, int* __restrict__ nanflag) {
    __shared__ double tile[OV_FDR_TILE];
    const int set = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const double* ps = p + (int64_t)set * n;
    const double pi = i < n ? ps[i] : 0.0;
    int rank = 0;
    for (int t0 = 0; t0 < n; t0 += OV_FDR_TILE) {
        __syncthreads();
        for (int k = threadIdx.x; k < OV_FDR_TILE; k += 256) tile[k] = t0 + k < n ? ps[t0 + k] : __longlong_as_double(0x7FF0000000000000ll);
        __syncthreads();
        const int m = min(OV_FDR_TILE, n - t0);
        for (int k = 0; k < m; ++k) rank += tile[k] <= pi;
    }
    if (i < n) {
        raw[(int64_t)set * n + i] = pi / ((double)rank / (double)n);
        if (pi != pi) nanflag[set] = 1;
    }
}

__global__ __launch_bounds__(256) void ov_fdr_min_kernel(const double* __restrict__ p, const double* __restrict__ raw, int n,
                                                         const int* __restrict__ nanflag, double* __restrict__ q) {
    __shared__ double tp[OV_FDR_TILE], tr[OV_FDR_TILE];
    const int set = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const double* ps = p + (int64_t)set * n;
    const double* rs = raw + (int64_t)set * n;
    const double pi = i < n ? ps[i] : 0.0;
    double best = 1.0;
    for (int t0 = 0; t0 < n; t0 += OV_FDR_TILE) {
        __syncthreads();
        for (int k = threadIdx.x; k < OV_FDR_TILE; k += 256) {
            tp[k] = t0 + k < n ? ps[t0 + k] : -1.0;
            tr[k] = t0 + k < n ? rs[t0 + k] : 1.0;
        }
        __syncthreads();
        const int m = min(OV_FDR_TILE, n - t0);
        for (int k = 0; k < m; ++k) if (tp[k] >= pi) best = fmin(best, tr[k]);
    }
    // a NaN among the p-values makes every q NaN (numpy's minimum.accumulate over the sorted tail propagates it)
    if (i < n) q[(int64_t)set * n + i] = nanflag[set] ? __longlong_as_double(0x7FF8000000000000ll) : best;
}

extern "C" int64_t ontovae_fdr_workspace_bytes(int32_t n, int32_t nsets) {
    if (n <= 0 || nsets <= 0) return 0;
    return ((int64_t)n * nsets * 8 + 255) / 256 * 256 + ((int64_t)nsets * 4 + 255) / 256 * 256;
}

extern "C" int ontovae_fdr_bh(const double* pvals, int32_t n, int32_t nsets, double* qvals, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    if (n < 0 || nsets < 0) { ontovae_set_error("fdr_bh: n = %d, nsets = %d", n, nsets); return ONTOVAE_EINVAL; }
    if (n == 0 || nsets == 0) return ONTOVAE_OK;
    if (!pvals || !qvals) { ontovae_set_error("fdr_bh: NULL argument"); return ONTOVAE_EINVAL; }
    if (pvals == qvals) { ontovae_set_error("fdr_bh: qvals must not alias pvals (the second kernel reads p-values that other blocks' q-values would overwrite)"); return ONTOVAE_EINVAL; }
    if (!workspace || workspace_bytes < ontovae_fdr_workspace_bytes(n, nsets) || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
        ontovae_set_error("fdr_bh: workspace of %lld bytes, 256-byte aligned, required (ontovae_fdr_workspace_bytes)",
                          (long long)ontovae_fdr_workspace_bytes(n, nsets));
        return ONTOVAE_EINVAL;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    double* raw = reinterpret_cast<double*>(workspace);
    int* nanflag = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + ((int64_t)n * nsets * 8 + 255) / 256 * 256);
    OV_HIP_CHECK(hipMemsetAsync(nanflag, 0, sizeof(int) * nsets, st));
    const dim3 grid((n + 255) / 256, nsets);
    hipLaunchKernelGGL(ov_fdr_raw_kernel, grid, dim3(256), 0, st, pvals, n, raw, nanflag);
    OV_LAUNCH_CHECK();
    hipLaunchKernelGGL(ov_fdr_min_kernel, grid, dim3(256), 0, st, pvals, raw, n, nanflag, qvals);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ontovae_fdr_bh for callers that cannot allocate (HIP graph capture, serving loops): the `raw` values and the NaN flags
// live in a file-scope device array instead of a caller workspace.  Every GPU the library runs on holds its own copy of
// the array (2 MiB + 16 KiB) until the process ends; its device address is looked up once per device, with that device
// current, and only that pointer is ever handed to kernels or memsets (the array's name is a host-side stand-in).
// Calls on one GPU share the array: two of them in flight at once (different streams) race.
// ---------------------------------------------------------------------------------------------------------------------
#define OV_FDR_STATIC_PVALS (1 << 18)          // n * nsets at most
#define OV_FDR_STATIC_SETS 4096                // nsets at most
#define OV_FDR_STATIC_RAW_BYTES ((size_t)OV_FDR_STATIC_PVALS * sizeof(double))
#define OV_FDR_STATIC_BYTES (OV_FDR_STATIC_RAW_BYTES + (size_t)OV_FDR_STATIC_SETS * sizeof(int))
static __device__ __align__(256) unsigned char ov_fdr_scratch[OV_FDR_STATIC_BYTES];

#define OV_MAX_DEVICES 64
static std::mutex g_fdr_scratch_mutex;
static void* g_fdr_scratch_ptr[OV_MAX_DEVICES] = {nullptr};

// device address of this device's copy of ov_fdr_scratch (the current device), looked up on the first call per device
static int ov_fdr_scratch_get(void** out) {
    int dev = -1;
    OV_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= OV_MAX_DEVICES) { ontovae_set_error("fdr_bh_static: device index %d out of range", dev); return ONTOVAE_EINVAL; }
    std::lock_guard<std::mutex> lock(g_fdr_scratch_mutex);
    if (!g_fdr_scratch_ptr[dev]) {
        void* ptr = nullptr;
        OV_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(ov_fdr_scratch)));
        if (!ptr || (reinterpret_cast<uintptr_t>(ptr) & 255)) { ontovae_set_error("fdr_bh_static: bad scratch address %p", ptr); return ONTOVAE_EHIP; }
        g_fdr_scratch_ptr[dev] = ptr;
    }
    *out = g_fdr_scratch_ptr[dev];
    return ONTOVAE_OK;
}

extern "C" int ontovae_fdr_bh_static(const double* pvals, int32_t n, int32_t nsets, double* qvals, void* stream) {
    if (n < 0 || nsets < 0) { ontovae_set_error("fdr_bh_static: n = %d, nsets = %d", n, nsets); return ONTOVAE_EINVAL; }
    if (n == 0 || nsets == 0) return ONTOVAE_OK;
    if (!pvals || !qvals) { ontovae_set_error("fdr_bh_static: NULL argument"); return ONTOVAE_EINVAL; }
    if (pvals == qvals) { ontovae_set_error("fdr_bh_static: qvals must not alias pvals"); return ONTOVAE_EINVAL; }
    if ((int64_t)n * nsets > OV_FDR_STATIC_PVALS || nsets > OV_FDR_STATIC_SETS) {
        ontovae_set_error("fdr_bh_static: %d sets of %d p-values exceed the static scratch (n * nsets <= %d, nsets <= %d); "
                          "use ontovae_fdr_bh with a workspace", nsets, n, OV_FDR_STATIC_PVALS, OV_FDR_STATIC_SETS);
        return ONTOVAE_EINVAL;
    }
    void* base = nullptr;
    { const int rc = ov_fdr_scratch_get(&base); if (rc) return rc; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    double* raw = reinterpret_cast<double*>(base);
    int* nanflag = reinterpret_cast<int*>(reinterpret_cast<char*>(base) + OV_FDR_STATIC_RAW_BYTES);
    // the array keeps the previous call's contents: the flags must start at zero, raw is fully rewritten for [0, n * nsets)
    OV_HIP_CHECK(hipMemsetAsync(nanflag, 0, sizeof(int) * nsets, st));
    const dim3 grid((n + 255) / 256, nsets);
    hipLaunchKernelGGL(ov_fdr_raw_kernel, grid, dim3(256), 0, st, pvals, n, raw, nanflag);
    OV_LAUNCH_CHECK();
    hipLaunchKernelGGL(ov_fdr_min_kernel, grid, dim3(256), 0, st, pvals, raw, n, nanflag, qvals);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// sample-major [B, F] block -> columns [0, B) of a feature-major [F, dst_ld] matrix (the layout the test reads)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ov_to_feature_major_kernel(const float* __restrict__ src, int64_t src_ld, int B, int F,
                                                                  float* __restrict__ dst, int64_t dst_ld) {
    __shared__ float tile[32][33];
    const int f0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int b = b0 + j, f = f0 + tx;
        tile[j][tx] = (f < F && b < B) ? src[(int64_t)b * src_ld + f] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int f = f0 + j, b = b0 + tx;
        if (f < F && b < B) dst[(int64_t)f * dst_ld + b] = tile[tx][j];
    }
}

extern "C" int ontovae_to_feature_major(const float* src, int64_t src_ld, int32_t B, int32_t F, float* dst, int64_t dst_ld, void* stream) {
    if (B < 0 || F < 0 || src_ld < F || dst_ld < B) { ontovae_set_error("to_feature_major: B = %d, F = %d, src_ld = %lld, dst_ld = %lld", B, F, (long long)src_ld, (long long)dst_ld); return ONTOVAE_EINVAL; }
    if (B == 0 || F == 0) return ONTOVAE_OK;
    if (!src || !dst) { ontovae_set_error("to_feature_major: NULL argument"); return ONTOVAE_EINVAL; }
    hipLaunchKernelGGL(ov_to_feature_major_kernel, dim3((F + 31) / 32, (B + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       src, src_ld, B, F, dst, dst_ld);
    OV_LAUNCH_CHECK();
    return ONTOVAE_OK;
}
