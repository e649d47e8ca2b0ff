// fp32 MFMA GEMM for the dense encoder (gfx950, v_mfma_f32_32x32x2_f32: exact fp32 fma chains).
//
//   C[m, n] = sum_k A(m, k) * B(k, n)          C row-major [M, ldc]
//
// Operand storage is chosen per call so that no operand is ever transposed in memory:
//   A_KC : A(m, k) = A[m * lda + k]   (k contiguous)        else A(m, k) = A[k * lda + m]
//   B_KC : B(k, n) = B[row(n) * ldb + k]                    else B(k, n) = B[row(k) * ldb + n]
// row(i) = bgather ? bgather[i] : i is the batcher's row indirection into the HBM-resident dataset.
//
// Fragments go straight from global memory (L2) to registers: fp32 MFMA runs at 64 cycles per
// 32x32x2 per SIMD, so the arithmetic is slow enough that no LDS stage is needed as long as the
// loads are issued far enough ahead.  Two things provide that distance:
//   * UK k-steps (of 8) are loaded as one batch, and the next batch is issued before the current
//     one's MFMAs (double buffering in registers);
//   * a wave tile of MT x NT 32x32 blocks gives 4*MT*NT MFMAs per k-step (3072 cycles for the
//     96x128 tile of the encoder forward), more than an L2 round trip.
// The lane -> (row, k) map of one k-step of 8: lane (i = lane & 31, h = lane >> 5) supplies
// k = k0 + 4h + t to MFMA t (t = 0..3), for A and B alike, which is all the instruction needs.
//
// Work decomposition (256 threads = 4 waves):
//   KW == 1: the waves own a 2x2 arrangement of wave tiles, block tile (64 MT) x (64 NT);
//   KW == 4: the waves share ONE wave tile and split the block's K range four ways; the partial
//            tiles are combined through LDS (tree: 2,3 -> 0,1; 1 -> 0) and wave 0 runs the epilogue.
//            This is how skinny products (K >> M*N/chip) fill the chip without a slab per wave.
// blockIdx.z is a further split of K across workgroups; its partial results go to slabs
// (EPI_STORE) that the consumer sums.
#ifndef ONTOVAE_GEMM_H
#define ONTOVAE_GEMM_H

#include "ontovae_device.h"

// OV_EPI_ADAMW_T: AdamW with the tile transposed through LDS first, so that the parameter / moment traffic
// is 16 B per lane along n (512-B runs per row for a 128-wide wave tile) instead of 128-B pieces
enum { OV_EPI_STORE = 0, OV_EPI_BIAS = 1, OV_EPI_ADAMW = 2, OV_EPI_ADAMW_T = 3 };

struct OvGemm {
    int M, N, K;
    const float* A;
    int64_t lda;
    const float* B;
    int64_t ldb;
    const int32_t* bgather;
    float* C;             // EPI_STORE/BIAS: output (+ z * slab_stride); EPI_ADAMW: the parameter
    int64_t ldc;
    int64_t slab_stride;
    int kchunk;           // K range of one workgroup (blockIdx.z), multiple of 32
    const float* bias;    // EPI_BIAS: per-m bias
    float* Pm;            // EPI_ADAMW: moments, same indexing as C
    float* Pv;
    OvAdam adam;
    int dbg;              // measurement only: 1 = skip the main loop, 2 = skip the epilogue
    // ov_dw0_multi_kernel only: A given as rank blocks (data parallel, all-gathered dh): k = r * a_blm + j lives at
    // A[r * a_rstride + m * lda + j]; a_blm == 0: plain A[m * lda + k]
    int a_blm;
    int64_t a_rstride;
    const int32_t* colmap;     // optional: gathered column k is column colmap[k] of the global batch (row = bgather[colmap[k]])
    float* b0sum;              // optional: [M] <- sum over the rank blocks of their trailing [M] floats (local b0 gradients)
};

template <bool FULL, bool VEC>
__device__ __forceinline__ f32x4 ov_ld_kc(const float* __restrict__ rowbase, int k, int kend) {
    f32x4 v;
    if (FULL && VEC) {
        v = *reinterpret_cast<const f32x4*>(rowbase + k);
    } else if (FULL) {
        v.x = rowbase[k]; v.y = rowbase[k + 1]; v.z = rowbase[k + 2]; v.w = rowbase[k + 3];
    } else {
        v.x = k < kend ? rowbase[k] : 0.f;
        v.y = k + 1 < kend ? rowbase[k + 1] : 0.f;
        v.z = k + 2 < kend ? rowbase[k + 2] : 0.f;
        v.w = k + 3 < kend ? rowbase[k + 3] : 0.f;
    }
    return v;
}

// operand with the OTHER index contiguous: four separate rows k .. k+3; `rows` (LDS or NULL) holds the
// gathered row numbers of the whole K range, relative to krow0
template <bool FULL>
__device__ __forceinline__ f32x4 ov_ld_mc(const float* __restrict__ base, int64_t ld, int col, int k, int kend,
                                          const int32_t* rows, int krow0) {
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = k + j;
        if (FULL || kk < kend) {
            const int64_t r = rows ? (int64_t)rows[kk - krow0] : (int64_t)kk;
            t[j] = base[r * ld + col];
        } else {
            t[j] = 0.f;
        }
    }
    f32x4 v;
    v.x = t[0]; v.y = t[1]; v.z = t[2]; v.w = t[3];
    return v;
}

// (bx, by, bz) = the workgroup's tile coordinates: blockIdx for a plain launch, remapped when the GEMM shares its
// launch with riders (see ov_gemm_riders_kernel in ontovae_kernels.hip)
template <int MT, int NT, int KW, int UK, bool A_KC, bool B_KC, bool VEC, int EPI>
__device__ __forceinline__ void ov_gemm_body(const OvGemm& p, const int bx, const int by, const int bz) {
    extern __shared__ __attribute__((aligned(16))) float ov_gemm_lds[];
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int m0 = KW == 1 ? by * (64 * MT) + (wave >> 1) * (32 * MT) : by * (32 * MT);
    const int n0 = KW == 1 ? bx * (64 * NT) + (wave & 1) * (32 * NT) : bx * (32 * NT);
    const int kb0 = bz * p.kchunk;
    const int kb1 = min(p.K, kb0 + p.kchunk);
    int kbeg = kb0, kend = kb1;
    if (KW == 4) {   // contiguous quarters of the block's range, multiples of 8
        const int q = ((kb1 - kb0 + 4 * 8 - 1) / (4 * 8)) * 8;
        kbeg = min(kb1, kb0 + wave * q);
        kend = min(kb1, kbeg + q);
    }
    const bool tile_on = m0 < p.M && n0 < p.N;       // wave-uniform (block-uniform when KW == 4)

    // gathered row numbers of the K range (B with n contiguous, k = batch row): one LDS copy per block
    int32_t* rows = nullptr;
    if (!B_KC && p.bgather) {
        rows = reinterpret_cast<int32_t*>(ov_gemm_lds);
        for (int t = threadIdx.x; t < kb1 - kb0; t += 256) rows[t] = p.bgather[kb0 + t];
        __syncthreads();
    }
    float* red = ov_gemm_lds;     // KW == 4 reduction area (never used together with `rows`)

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // Fused AdamW: the tile's parameters and moments are requested BEFORE the K loop, so the optimizer's read traffic runs
    // under the MFMAs instead of after them (KW == 4: every wave for the quarter of the tile it ends up with).
    constexpr bool PREF = (EPI == OV_EPI_ADAMW) && (KW == 1 || KW == 4);
    float pw[PREF ? MT : 1][PREF ? NT : 1][16], pm[PREF ? MT : 1][PREF ? NT : 1][16], pv[PREF ? MT : 1][PREF ? NT : 1][16];
    if (PREF && tile_on) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + 32 * nt + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const bool ok = m < p.M && n < p.N && (KW != 4 || (r >> 2) == wave);
                    const int64_t idx = (int64_t)m * p.ldc + n;
                    pw[mt][nt][r] = ok ? p.C[idx] : 0.f;
                    pm[mt][nt][r] = ok ? p.Pm[idx] : 0.f;
                    pv[mt][nt][r] = ok ? p.Pv[idx] : 0.f;
                }
            }
    }

    if (tile_on && kbeg < kend && !(OV_DBG(p.dbg) & 1)) {
        // per-lane operand bases (rows clamped: out-of-range rows are computed and never stored)
        const float* abase[MT];
        const float* bbase[NT];
        int acol[MT], bcol[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(m0 + 32 * mt + i, p.M - 1);
            acol[mt] = m;
            abase[mt] = A_KC ? p.A + (int64_t)m * p.lda : p.A;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = min(n0 + 32 * nt + i, p.N - 1);
            bcol[nt] = n;
            if (B_KC) {
                const int64_t r = p.bgather ? (int64_t)p.bgather[n] : (int64_t)n;
                bbase[nt] = p.B + r * p.ldb;
            } else {
                bbase[nt] = p.B;
            }
        }
        f32x4 a[2][UK][MT], b[2][UK][NT];
        auto load_batch = [&](int buf, int k0) {
            if (OV_DBG(p.dbg) & 4) {      // measurement only: no global loads, MFMA pace alone
#pragma unroll
                for (int u = 0; u < UK; ++u) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[buf][u][mt] = f32x4{(float)k0, 1.f, (float)lane, 2.f};
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b[buf][u][nt] = f32x4{1.f, (float)k0, 3.f, (float)lane};
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                const int k = k0 + 8 * u + 4 * h;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    a[buf][u][mt] = A_KC ? ov_ld_kc<true, VEC>(abase[mt], k, kend)
                                         : ov_ld_mc<true>(abase[mt], p.lda, acol[mt], k, kend, nullptr, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[buf][u][nt] = B_KC ? ov_ld_kc<true, VEC>(bbase[nt], k, kend)
                                         : ov_ld_mc<true>(bbase[nt], p.ldb, bcol[nt], k, kend, rows, kb0);
            }
        };
        auto mfma_batch = [&](int buf) {
#pragma unroll
            for (int u = 0; u < UK; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][u][mt][t], b[buf][u][nt][t], acc[mt][nt], 0, 0, 0);
        };
        const int nbatch = (kend - kbeg) / (8 * UK);
        int k0 = kbeg;
        if (nbatch > 0) {
            load_batch(0, k0);
            int it = 0;
            for (; it + 2 <= nbatch; it += 2) {      // two batches per trip so that the buffers have static names
                load_batch(1, k0 + 8 * UK);
                mfma_batch(0);
                if (it + 2 < nbatch) load_batch(0, k0 + 16 * UK);
                mfma_batch(1);
                k0 += 16 * UK;
            }
            if (it < nbatch) {
                mfma_batch(0);
                k0 += 8 * UK;
            }
        }
        for (; k0 < kend; k0 += 8) {                  // remainder: single guarded k-steps
            const int k = k0 + 4 * h;
            f32x4 ta[MT], tb[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                ta[mt] = A_KC ? ov_ld_kc<false, VEC>(abase[mt], k, kend) : ov_ld_mc<false>(abase[mt], p.lda, acol[mt], k, kend, nullptr, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                tb[nt] = B_KC ? ov_ld_kc<false, VEC>(bbase[nt], k, kend) : ov_ld_mc<false>(bbase[nt], p.ldb, bcol[nt], k, kend, rows, kb0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[mt][t], tb[nt][t], acc[mt][nt], 0, 0, 0);
        }
    }

    if (KW == 4) {
        // combine the four K-quarters: every wave leaves its partial tile in LDS and picks up one quarter of the tile
        // (accumulator registers 4 w .. 4 w + 3: rows 8 w + {0..3} + 4 h), summed over the waves in wave order.  One
        // barrier, and the epilogue (stores, or 3 loads + AdamW + 3 stores per element) is shared by the four waves
        // instead of being wave 0's alone.
        constexpr int TT = MT * NT;
#pragma unroll
        for (int t = 0; t < TT; ++t) {              // one 32 x 32 tile at a time: 16 KB of LDS whatever the workgroup tile
            if (t > 0) __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t / NT][t % NT][r];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * wave + q;
                float v = red[(0 * 16 + r) * 64 + lane];
                v += red[(1 * 16 + r) * 64 + lane];
                v += red[(2 * 16 + r) * 64 + lane];
                v += red[(3 * 16 + r) * 64 + lane];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) if (rr == r) acc[t / NT][t % NT][rr] = v;      // r is wave-uniform
            }
        }
    }
    if (!tile_on) return;

    if (OV_DBG(p.dbg) & 2) {
        float sink = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink += acc[mt][nt][r];
        if (sink == 1.2345e-30f) p.C[0] = sink;
        return;
    }
    if (EPI == OV_EPI_ADAMW_T) {
        constexpr int TW = 32 * NT, LDW = TW + 4, TR = 32 * MT, C4 = TW / 4, RPI = 64 / C4;
        // wave-private LDS tile behind the (optional) gathered-row table
        const int rows_floats = (!B_KC && p.bgather) ? ((kb1 - kb0 + 3) / 4 * 4) : 0;
        float* tile = ov_gemm_lds + rows_floats + wave * (TR * LDW);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h) * LDW + 32 * nt + i] = acc[mt][nt][r];
        const int c4 = lane % C4, rr = lane / C4;
        const int n = n0 + 4 * c4;
#pragma unroll 1
        for (int row0 = 0; row0 < TR; row0 += 4 * RPI) {
            f32x4 w[4], mm[4], vv[4], g[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = row0 + u * RPI + rr;
                const int m = m0 + row;
                ok[u] = row < TR && m < p.M && n < p.N;
                const int64_t idx = (int64_t)m * p.ldc + n;
                if (ok[u]) {
                    w[u] = *reinterpret_cast<const f32x4*>(p.C + idx);
                    mm[u] = *reinterpret_cast<const f32x4*>(p.Pm + idx);
                    vv[u] = *reinterpret_cast<const f32x4*>(p.Pv + idx);
                    g[u] = *reinterpret_cast<const f32x4*>(tile + row * LDW + 4 * c4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ok[u]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float ww = w[u][e], m1 = mm[u][e], v1 = vv[u][e];
                        ov_adamw(ww, m1, v1, g[u][e], p.adam, false);
                        w[u][e] = ww; mm[u][e] = m1; vv[u][e] = v1;
                    }
                    const int64_t idx = (int64_t)(m0 + row0 + u * RPI + rr) * p.ldc + n;
                    *reinterpret_cast<f32x4*>(p.C + idx) = w[u];
                    *reinterpret_cast<f32x4*>(p.Pm + idx) = mm[u];
                    *reinterpret_cast<f32x4*>(p.Pv + idx) = vv[u];
                }
            }
        }
        return;
    }

    // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + 32 * nt + i;
            if (EPI == OV_EPI_ADAMW) {
                // all parameter / moment loads of the tile first, then the arithmetic, then the stores
                float w[16], mm[16], vv[16];
                bool ok[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    ok[r] = m < p.M && n < p.N && (KW != 4 || (r >> 2) == wave);
                    const int64_t idx = (int64_t)m * p.ldc + n;
                    if (PREF) {
                        w[r] = pw[mt][nt][r]; mm[r] = pm[mt][nt][r]; vv[r] = pv[mt][nt][r];
                    } else {
                        w[r] = ok[r] ? p.C[idx] : 0.f;
                        mm[r] = ok[r] ? p.Pm[idx] : 0.f;
                        vv[r] = ok[r] ? p.Pv[idx] : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (KW != 4 || (r >> 2) == wave) ov_adamw(w[r], mm[r], vv[r], acc[mt][nt][r], p.adam, false);      // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int64_t idx = (int64_t)m * p.ldc + n;
                    if (ok[r]) { p.C[idx] = w[r]; p.Pm[idx] = mm[r]; p.Pv[idx] = vv[r]; }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < p.M && n < p.N && (KW != 4 || (r >> 2) == wave)) {
                        const int64_t idx = (int64_t)m * p.ldc + n;
                        if (EPI == OV_EPI_STORE) p.C[(int64_t)bz * p.slab_stride + idx] = acc[mt][nt][r];
                        else p.C[idx] = acc[mt][nt][r] + p.bias[m];
                    }
                }
            }
        }
}

template <int MT, int NT, int KW, int UK, bool A_KC, bool B_KC, bool VEC, int EPI>
__global__ __launch_bounds__(256) void ov_gemm_kernel(const OvGemm p) {
    ov_gemm_body<MT, NT, KW, UK, A_KC, B_KC, VEC, EPI>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}


// -------------------------------------------------------------------------------------------------
// LDS-staged variant for the two large products (encoder forward, dW0).
//
// Fragment-shaped global loads touch 32 cache lines per wave instruction and every line four times;
// at the sizes of the first encoder layer that load-path work, not the MFMA pipe, set the pace.  Here
// a workgroup stages a (96 WM) x 32 tile of A and a 32 x (32 WN) tile of B per K-step of 32 through
// LDS with whole-line 16-B-per-lane loads (each line fetched once per workgroup, shared by its four
// waves), prefetches the next stage into registers under the current stage's MFMAs, and reads MFMA
// fragments with ds_read_b128 (k-contiguous operands, row stride 36 floats: conflict-free) or
// ds_read_b32 (n-contiguous B).  Wave grid WM x WN (WM * WN = 4), wave tile 96 x 32.
//   encoder forward: WM = 1, WN = 4, B_KC (x rows, gathered), split-K slabs over blockIdx.z
//   dW0            : WM = 4, WN = 1, B n-contiguous (x[row(b), g]), K = batch, AdamW epilogue
// Requires 16-B aligned rows (lda, ldb multiples of 4); the host falls back to ov_gemm_kernel otherwise.
// -------------------------------------------------------------------------------------------------
// (a device function, so that other work can share its launch: bid / gdim stand in for the block index and the grid size)
template <int WM, int WN, bool B_KC, int EPI>
__device__ __forceinline__ void ov_gemm_lds_body(const OvGemm& p, const dim3 bid, const dim3 gdim) {
    constexpr int MT = 3, BK = 32, LDK = BK + 4;
    constexpr int RA = 32 * MT * WM;              // A rows per workgroup
    constexpr int RB = 32 * WN;                   // B columns (n) per workgroup
    constexpr int UA = RA * 8 / 256;              // float4 units per thread, A tile
    constexpr int UB = B_KC ? RB * 8 / 256 : (BK * RB / 4 + 255) / 256;
    constexpr int LDN = RB + 4;
    extern __shared__ __attribute__((aligned(16))) float ov_gemm_lds[];
    float* As = ov_gemm_lds;                      // [RA][LDK]
    float* Bs = ov_gemm_lds + RA * LDK;           // B_KC: [RB][LDK]   else [BK][LDN]
    int32_t* rows = reinterpret_cast<int32_t*>(Bs + (B_KC ? RB * LDK : BK * LDN));   // gathered rows of the K range (!B_KC)
    const int tid = threadIdx.x;
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int wm = WM == 4 ? wave : 0, wn = WN == 4 ? wave : 0;
    // Workgroups are dealt to the 8 XCDs round-robin by linear id.  The row tiles (y) of one K slab (z) read the same
    // B tile, so they are renumbered to share an XCD (and its L2): linear id = xcd + 8 j  ->  z = xcd + 8 (j / ny), y = j % ny.
    int by = bid.y, bz = bid.z;
    if (gdim.x == 1 && gdim.z % 8 == 0) {
        const int id = bid.y + gdim.y * bid.z, ny = gdim.y;
        const int xcd = id & 7, j = id >> 3;
        bz = xcd + 8 * (j / ny);
        by = j % ny;
    }
    const int m0 = by * RA, n0 = bid.x * RB;
    const int kb0 = bz * p.kchunk;
    const int kb1 = min(p.K, kb0 + p.kchunk);
    if (!B_KC) {
        for (int t = tid; t < kb1 - kb0; t += 256) rows[t] = p.bgather ? p.bgather[kb0 + t] : kb0 + t;
        __syncthreads();
    }
    // staging assignments (fixed per thread over the whole K loop)
    const float* asrc[UA];
    int adst[UA];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int unit = tid + 256 * u, r = unit >> 3, kq = unit & 7;
        asrc[u] = p.A + (int64_t)min(m0 + r, p.M - 1) * p.lda + 4 * kq;
        adst[u] = r * LDK + 4 * kq;
    }
    const float* bsrc[UB];
    int bdst[UB], bk[UB];
    bool bok[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int unit = tid + 256 * u;
        if (B_KC) {
            const int r = unit >> 3, kq = unit & 7;
            const int n = min(n0 + r, p.N - 1);
            const int64_t row = p.bgather ? (int64_t)p.bgather[n] : (int64_t)n;
            bsrc[u] = p.B + row * p.ldb + 4 * kq;
            bdst[u] = r * LDK + 4 * kq;
            bk[u] = 0;
            bok[u] = true;
        } else {
            constexpr int Q = RB / 4;                         // float4 per k row
            const int k = unit / Q, q = unit % Q;
            bsrc[u] = p.B + (n0 + 4 * q < p.N ? n0 + 4 * q : 0);  // beyond N: a valid address, the piece is zeroed (bok); a piece that
                                                                  // starts below N may end in the row's padding (ldb >= N rounded up to 4)
            bdst[u] = k * LDN + 4 * q;
            bk[u] = k;
            bok[u] = k < BK && n0 + 4 * q < p.N;
        }
    }
    // Stage loads without lane-dependent control flow: every lane issues every load at an address clamped into its row, and the
    // elements beyond the K range are zeroed by selects where the stage goes to LDS.  With the tail handled by branches (one
    // 16-B load or up to three 4-B loads plus zero fills of the same registers) the compiler put a wait for ALL outstanding
    // loads behind every load of a stage -- a later write to a register that an earlier path still had a load pending on --
    // so the seven loads of a stage went out one memory latency after the other, in front of the stage's MFMAs.
    // Rows are 16-B aligned and lda, ldb are multiples of 4 that cover the range (host check), so a piece stays inside its row;
    // a clamped piece lies wholly beyond the K range.
    f32x4 ra[UA], rb[UB];
    const int kclamp = (p.K - 1) & ~3;             // the last 16-B piece of a row that holds an element of the K range
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int k = min(k0 + 4 * ((tid + 256 * u) & 7), kclamp);
            ra[u] = *reinterpret_cast<const f32x4*>(asrc[u] + (k - 4 * ((tid + 256 * u) & 7)));
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (B_KC) {
                const int k = min(k0 + 4 * ((tid + 256 * u) & 7), kclamp);
                rb[u] = *reinterpret_cast<const f32x4*>(bsrc[u] + (k - 4 * ((tid + 256 * u) & 7)));
            } else {
                const int k = min(k0 + bk[u], kb1 - 1);
                rb[u] = *reinterpret_cast<const f32x4*>(bsrc[u] + (int64_t)rows[k - kb0] * p.ldb);
            }
        }
    };
    // (a piece whose address was clamped starts at or beyond kb1 in unclamped terms: all four of its elements are zeroed)
    auto lstore = [&](int k0) {
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int k = k0 + 4 * ((tid + 256 * u) & 7);
            f32x4 v;
            v.x = k < kb1 ? ra[u].x : 0.f;
            v.y = k + 1 < kb1 ? ra[u].y : 0.f;
            v.z = k + 2 < kb1 ? ra[u].z : 0.f;
            v.w = k + 3 < kb1 ? ra[u].w : 0.f;
            *reinterpret_cast<f32x4*>(As + adst[u]) = v;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (B_KC) {
                const int k = k0 + 4 * ((tid + 256 * u) & 7);
                f32x4 v;
                v.x = k < kb1 ? rb[u].x : 0.f;
                v.y = k + 1 < kb1 ? rb[u].y : 0.f;
                v.z = k + 2 < kb1 ? rb[u].z : 0.f;
                v.w = k + 3 < kb1 ? rb[u].w : 0.f;
                *reinterpret_cast<f32x4*>(Bs + bdst[u]) = v;
            } else if (bk[u] < BK) {                 // (compile-time per u except for the last unit)
                const bool ok = bok[u] && k0 + bk[u] < kb1;
                f32x4 v;
                v.x = ok ? rb[u].x : 0.f; v.y = ok ? rb[u].y : 0.f; v.z = ok ? rb[u].z : 0.f; v.w = ok ? rb[u].w : 0.f;
                *reinterpret_cast<f32x4*>(Bs + bdst[u]) = v;
            }
        }
    };

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    gload(kb0);
    for (int k0 = kb0; k0 < kb1; k0 += BK) {
        __syncthreads();                  // everybody is done reading the previous stage
        lstore(k0);
        __syncthreads();
        gload(k0 + BK);                         // next stage in flight under this stage's MFMAs (beyond the range: clamped, unused)
        __builtin_amdgcn_sched_barrier(0);      // (in one block with the MFMAs now: keep the scheduler from sinking the loads below them)
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            f32x4 fa[MT], fb;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                fa[mt] = *reinterpret_cast<const f32x4*>(As + (wm * 32 * MT + 32 * mt + i) * LDK + 8 * s + 4 * h);
            if (B_KC) {
                fb = *reinterpret_cast<const f32x4*>(Bs + (wn * 32 + i) * LDK + 8 * s + 4 * h);
            } else {
                fb.x = Bs[(8 * s + 4 * h + 0) * LDN + wn * 32 + i];
                fb.y = Bs[(8 * s + 4 * h + 1) * LDN + wn * 32 + i];
                fb.z = Bs[(8 * s + 4 * h + 2) * LDN + wn * 32 + i];
                fb.w = Bs[(8 * s + 4 * h + 3) * LDN + wn * 32 + i];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][t], fb[t], acc[mt], 0, 0, 0);
        }
    }

    // epilogue; C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int n = n0 + wn * 32 + i;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mb = m0 + wm * 32 * MT + 32 * mt;
        if (EPI == OV_EPI_ADAMW) {
            float w[16], mm[16], vv[16];
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * h;
                ok[r] = m < p.M && n < p.N;
                const int64_t idx = (int64_t)m * p.ldc + n;
                w[r] = ok[r] ? p.C[idx] : 0.f;
                mm[r] = ok[r] ? p.Pm[idx] : 0.f;
                vv[r] = ok[r] ? p.Pv[idx] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) ov_adamw(w[r], mm[r], vv[r], acc[mt][r], p.adam, false);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int64_t idx = (int64_t)m * p.ldc + n;
                if (ok[r]) { p.C[idx] = w[r]; p.Pm[idx] = mm[r]; p.Pv[idx] = vv[r]; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M && n < p.N) p.C[(int64_t)bz * p.slab_stride + (int64_t)m * p.ldc + n] = acc[mt][r];
            }
        }
    }
}

template <int WM, int WN, bool B_KC, int EPI>
__global__ __launch_bounds__(256) void ov_gemm_lds_kernel(const OvGemm p) {
    ov_gemm_lds_body<WM, WN, B_KC, EPI>(p, blockIdx, gridDim);
}

// -------------------------------------------------------------------------------------------------
// The encoder-forward form of ov_gemm_lds_kernel (WM = 1, WN = 4, both operands K-contiguous) for large batches, where the
// K loop is long and the matrix pipe sets the pace: two LDS stages and ONE barrier per K-step instead of two, the MFMA
// fragments of the next group of 8 k read from LDS while the current group's 12 MFMAs run (two register sets), global loads
// two stages ahead (a whole step to land), and the next stage written to the other LDS buffer under the step's first MFMAs, so
// that at the barrier a wave has nothing left to wait for but the other waves.  Same tiles, same operand order and same summation order as ov_gemm_lds_kernel: bit-identical
// results.
// -------------------------------------------------------------------------------------------------
// WM: row tiles of 96 per workgroup, 4 waves each (WM = 2: 192 x 128, 8 waves that share one B stage -- 5 instead of 7 staging
// loads, selects and LDS stores per thread and step for the same MFMAs per wave).
template <int EPI, int WM>
__global__ __launch_bounds__(256 * WM) void ov_gemm_lds2_kernel(const OvGemm p) {
    constexpr int MT = 3, BK = 32, LDK = BK + 4, NT = 256 * WM;
    constexpr int RA = 32 * MT * WM, RB = 128;
    constexpr int UA = RA * 8 / NT, UB = RB * 8 / NT;
    constexpr int RS = NT / 8;                    // rows one pass of the workgroup's threads stages
    constexpr int STAGE = (RA + RB) * LDK;        // floats per LDS stage
    extern __shared__ __attribute__((aligned(16))) float ov_gemm_lds[];
    const int tid = threadIdx.x;
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    const int i = lane & 31, h = lane >> 5;
    int by = blockIdx.y, bz = blockIdx.z;
    if (gridDim.x == 1 && gridDim.z % 8 == 0) {          // (as ov_gemm_lds_kernel: row tiles of one K slab on one XCD)
        const int id = blockIdx.y + gridDim.y * blockIdx.z, ny = gridDim.y;
        const int xcd = id & 7, j = id >> 3;
        bz = xcd + 8 * (j / ny);
        by = j % ny;
    }
    const int m0 = by * RA, n0 = blockIdx.x * RB;
    const int kb0 = bz * p.kchunk;
    const int kb1 = min(p.K, kb0 + p.kchunk);
    const int kq4 = 4 * (tid & 7);
    const float* asrc[UA];
    const float* bsrc[UB];
#pragma unroll
    for (int u = 0; u < UA; ++u) asrc[u] = p.A + (int64_t)min(m0 + (tid >> 3) + RS * u, p.M - 1) * p.lda;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int n = min(n0 + (tid >> 3) + RS * u, p.N - 1);
        bsrc[u] = p.B + (p.bgather ? (int64_t)p.bgather[n] : (int64_t)n) * p.ldb;
    }
    const int dst0 = (tid >> 3) * LDK + kq4;      // + RS u rows; B behind the RA rows of A
    f32x4 ra[UA], rb[UB];
    const int kclamp = (p.K - 1) & ~3;
    auto gload = [&](int k0) {                    // every lane, every load; clamped into the row (see ov_gemm_lds_kernel)
        const int k = min(k0 + kq4, kclamp);
#pragma unroll
        for (int u = 0; u < UA; ++u) ra[u] = *reinterpret_cast<const f32x4*>(asrc[u] + k);
#pragma unroll
        for (int u = 0; u < UB; ++u) rb[u] = *reinterpret_cast<const f32x4*>(bsrc[u] + k);
    };
    // (a form without the K-tail selects -- walking pointers, plain LDS stores, 13 instead of ~50 vector-ALU instructions per
    // step -- measured 1-3 % SLOWER on the same box at 8192 and 16384 columns: the selects ride in the MFMAs' shadow)
    auto lstore = [&](int k0, float* st) {
        const int k = k0 + kq4;
        const bool o0 = k < kb1, o1 = k + 1 < kb1, o2 = k + 2 < kb1, o3 = k + 3 < kb1;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            f32x4 v;
            v.x = o0 ? ra[u].x : 0.f; v.y = o1 ? ra[u].y : 0.f; v.z = o2 ? ra[u].z : 0.f; v.w = o3 ? ra[u].w : 0.f;
            *reinterpret_cast<f32x4*>(st + dst0 + RS * u * LDK) = v;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            f32x4 v;
            v.x = o0 ? rb[u].x : 0.f; v.y = o1 ? rb[u].y : 0.f; v.z = o2 ? rb[u].z : 0.f; v.w = o3 ? rb[u].w : 0.f;
            *reinterpret_cast<f32x4*>(st + RA * LDK + dst0 + RS * u * LDK) = v;
        }
    };
    // fragments of 4 k: lane (i, h) holds k = 8 g + 4 h .. + 3 of group g
    const int wm = wave >> 2, wn = wave & 3;
    const int fa_off = (wm * 32 * MT + i) * LDK + 4 * h, fb_off = (RA + wn * 32 + i) * LDK + 4 * h;
    auto fread = [&](const float* st, int g, f32x4 (&fa)[MT], f32x4& fb) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const f32x4*>(st + fa_off + 32 * mt * LDK + 8 * g);
        fb = *reinterpret_cast<const f32x4*>(st + fb_off + 8 * g);
    };
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    auto mfmas = [&](const f32x4 (&fa)[MT], const f32x4& fb) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][t], fb[t], acc[mt], 0, 0, 0);
    };

    // LDS holds stage k (being multiplied) and stage k + 1 (written at the start of step k from registers whose loads were
    // requested a whole step earlier); the registers then take the loads of stage k + 2.
    gload(kb0);
    lstore(kb0, ov_gemm_lds);
    gload(kb0 + BK);
    __syncthreads();
    int cur = 0;
    f32x4 fa0[MT], fb0, fa1[MT], fb1;
    fread(ov_gemm_lds, 0, fa0, fb0);
    for (int k0 = kb0; k0 < kb1; k0 += BK) {
        const float* st = ov_gemm_lds + cur * STAGE;
        float* nx = ov_gemm_lds + (cur ^ 1) * STAGE;
        fread(st, 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        lstore(k0 + BK, nx);                     // the other buffer: nobody reads it before the barrier below; the scheduler
        gload(k0 + 2 * BK);                      // spreads these between the first group's MFMAs
        mfmas(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        fread(st, 2, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        fread(st, 3, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                         // next stage visible; this wave's reads of the current one have all returned
        fread(nx, 0, fa0, fb0);                  // the next step's first fragments land under this step's last 12 MFMAs
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fa1, fb1);
        cur ^= 1;
    }

    const int n = n0 + wn * 32 + i;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mb = m0 + wm * 32 * MT + 32 * mt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < p.M && n < p.N) {
                if (EPI == OV_EPI_STORE) p.C[(int64_t)bz * p.slab_stride + (int64_t)m * p.ldc + n] = acc[mt][r];
                else p.C[(int64_t)m * p.ldc + n] = acc[mt][r] + p.bias[m];
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// dW0 = dh x (+ AdamW) for batches of at most 128 rows: the weight-gradient of the first encoder layer
// (gradient of modules.py:46, optimizer step of vae_model.py:172).
//
// The product is a stream over W0 / exp_avg / exp_avg_sq (3 x 4 LG bytes in, 3 x 4 LG out) with only
// K = batch <= 128 of arithmetic depth per element, so the kernel is organised around that stream:
//   * a wave owns 32 genes (columns of W0) and keeps its whole x fragment, x[row(b), g0 .. g0+31] for all
//     b, in 64 registers: x is read once per wave, not once per k-step;
//   * a workgroup owns up to 128 rows of W0 (4 tiles of 32) and holds those rows of dh in LDS
//     (row stride 132 floats: conflict-free ds_read_b128 fragments);
//   * per 32x32 tile the parameter / moment loads are issued first, the 64 MFMAs run under them, then
//     AdamW and the stores; the second wave of the SIMD is in a different phase, which is what overlaps
//     the HBM stream with the matrix pipe.
// Workgroups that share a gene range are given ids congruent mod 8, i.e. the same XCD, so the x tile is
// fetched into one L2 only.
// -------------------------------------------------------------------------------------------------
// Bounds-checked buffer access for the parameter stream: a row beyond M or a column beyond N becomes an
// offset beyond num_records, which the hardware reads as 0 and does not store, so partial tiles need no
// branches.  (Vector memory path; offsets are 32-bit, the host checks M * ldc * 4 < 2^31.)
typedef int32_t ov_i32x4 __attribute__((ext_vector_type(4)));
// The dW0 kernels keep the batch's x rows in LDS as 64-bit ELEMENT OFFSETS (row * ldb), not row indices: a gathered x load then
// costs one 64-bit shift-add; with indices every load paid a sign extension, two 32-bit multiplies, a 64-bit multiply-add and
// two adds (quarter-rate integer ops, ~60 cycles; 32 loads per 64-row chunk and wave = a fifth of the chunk's MFMA time, and
// both waves of a SIMD did it at the same moment, before either had an MFMA to issue).
typedef int64_t ov_i64x2 __attribute__((ext_vector_type(2)));
__device__ float ov_buf_load(ov_i32x4 rsrc, int32_t voffset, int32_t soffset, int32_t aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void ov_buf_store(float v, ov_i32x4 rsrc, int32_t voffset, int32_t soffset, int32_t aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ __forceinline__ ov_i32x4 ov_buf_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    ov_i32x4 r;
    r.x = (int32_t)(uint32_t)a;
    r.y = (int32_t)((uint32_t)(a >> 32) & 0xffffu);      // stride 0: raw buffer
    r.z = (int32_t)bytes;
    r.w = 0x00020000;                                     // 32-bit untyped data format (gfx90a / gfx94x / gfx950)
    return r;
}

#ifdef OV_DW0_PROBE      // probe builds only: per-wave timestamps into p.bias (16 x uint64 per wave)
#define OV_PROBE(slot) do { probe_t[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define OV_PROBE(slot) do { } while (0)
#endif

template <int NW, int EPI>
__global__ __launch_bounds__(64 * NW) void ov_dw0_kernel(const OvGemm p, int njobs) {
#ifdef OV_DW0_PROBE
    uint64_t probe_t[16] = {};
#endif
    OV_PROBE(0);
    constexpr int LT = 4, RA = 32 * LT, KMAX = 128, LDA = KMAX + 4;
    extern __shared__ __attribute__((aligned(16))) float ov_gemm_lds[];
    float* As = ov_gemm_lds;                                             // [RA][LDA]
    int64_t* rows = reinterpret_cast<int64_t*>(ov_gemm_lds + RA * LDA);  // [KMAX] element offsets of the x rows; beyond K: the last row (dh is 0 there)
    const int tid = threadIdx.x;
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int job = (b >> 3) % njobs;
    const int gx = (b / (8 * njobs)) * 8 + (b & 7);
    const int m0 = job * RA;
    const int n0 = (gx * NW + wave) * 32;
    const bool wave_on = n0 < p.N;
    const uint32_t row_bytes = (uint32_t)p.ldc * 4u;
    // element (r, lane) of the tile at row mb: a per-row descriptor (scalar registers: base + row offset, the
    // bytes that remain) and ONE lane offset for all rows
    const int32_t lane_off = n0 + i < p.N ? (int32_t)(((uint32_t)(4 * h) * (uint32_t)p.ldc + (uint32_t)(n0 + i)) * 4u) : (int32_t)0x80000000u;
    auto row_rsrc = [&](const float* base, int mb, int r) {
        const int row = mb + (r & 3) + 8 * (r >> 2);
        const uint32_t ro = (uint32_t)row * row_bytes;
        return ov_buf_rsrc(reinterpret_cast<const char*>(base) + ro, (uint32_t)max(p.M - row, 0) * row_bytes);
    };

    // parameter / moment registers of two tiles: tile lt + 1 is requested before the MFMAs of tile lt
    float w[2][16], mm[2][16], vv[2][16];
    auto request = [&](int buf, int lt) {
        if (EPI != OV_EPI_ADAMW) return;
        const int mb = m0 + 32 * lt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#ifdef OV_DW0_NOLOAD
            w[buf][r] = mm[buf][r] = vv[buf][r] = (float)(lane + r);
#else
            w[buf][r] = ov_buf_load(row_rsrc(p.C, mb, r), lane_off, 0, 0);
            mm[buf][r] = ov_buf_load(row_rsrc(p.Pm, mb, r), lane_off, 0, 0);
            vv[buf][r] = ov_buf_load(row_rsrc(p.Pv, mb, r), lane_off, 0, 0);
#endif
        }
    };
    // Prologue, ordered by what the first MFMA needs (memory returns in issue order): the dh slab and the
    // x fragment first, then the first tile's parameters; the slab goes to LDS once it has landed.
    constexpr int SU = RA * (KMAX / 4) / (64 * NW);
    // slab loads without lane-dependent branches (clamped addresses, out-of-range elements zeroed by selects where the slab goes
    // to LDS): with branches around them the compiler made the slab's LDS stores wait for the 64 x loads issued after them as
    // well, and the first tile's parameter request went out only once all of x had landed.  lda % 4 == 0 and lda >= K (host check).
    f32x4 sv[SU];
    const int klast = (p.K - 1) & ~3;
#pragma unroll
    for (int q = 0; q < SU; ++q) {
        const int u = tid + q * 64 * NW;
        const int r = u / (KMAX / 4), k = min(4 * (u % (KMAX / 4)), klast);
        sv[q] = *reinterpret_cast<const f32x4*>(p.A + (int64_t)min(m0 + r, p.M - 1) * p.lda + k);
    }
    for (int t = tid; t < KMAX; t += 64 * NW) {
        const int tt = min(t, p.K - 1);
        rows[t] = (int64_t)(p.bgather ? p.bgather[tt] : tt) * p.ldb;
    }
    __syncthreads();
    OV_PROBE(1);
    float xb[KMAX / 2];
    {                                            // on every path (waves beyond N: clamped column, result unused)
        const float* xcol = p.B + min(n0 + i, p.N - 1);
#pragma unroll
        for (int s = 0; s < KMAX / 8; ++s) {
            const ov_i64x2 r01 = *reinterpret_cast<const ov_i64x2*>(rows + 8 * s + 4 * h);
            const ov_i64x2 r23 = *reinterpret_cast<const ov_i64x2*>(rows + 8 * s + 4 * h + 2);
            xb[4 * s + 0] = xcol[r01.x];
            xb[4 * s + 1] = xcol[r01.y];
            xb[4 * s + 2] = xcol[r23.x];
            xb[4 * s + 3] = xcol[r23.y];
        }
    }
    OV_PROBE(2);
#pragma unroll
    for (int q = 0; q < SU; ++q) {
        const int u = tid + q * 64 * NW;
        const int r = u / (KMAX / 4), k = 4 * (u % (KMAX / 4));
        const bool ok = m0 + r < p.M;
        f32x4 v;
        v.x = ok && k < p.K ? sv[q].x : 0.f;
        v.y = ok && k + 1 < p.K ? sv[q].y : 0.f;
        v.z = ok && k + 2 < p.K ? sv[q].z : 0.f;
        v.w = ok && k + 3 < p.K ? sv[q].w : 0.f;
        *reinterpret_cast<f32x4*>(As + r * LDA + k) = v;
    }
    __syncthreads();
    if (!wave_on) return;                       // wave-uniform; no barrier below
    request(0, 0);

    // One tile: MFMAs, AdamW arithmetic, THEN the next tile's request, then this tile's stores.  The order
    // keeps every wait exact: the hardware counts at most 63 outstanding vector memory operations, so a
    // wait for "all but the youngest n" with n > 63 would degrade to waiting for much more.
    auto tile = [&](int buf, int lt, bool more) {
        const int mb = m0 + 32 * lt;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* arow = As + (32 * lt + i) * LDA + 4 * h;
#pragma unroll
        for (int s = 0; s < KMAX / 8; ++s) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(arow + 8 * s);
#ifdef OV_DW0_NOMFMA
            if (s < 1)
#endif
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], xb[4 * s + t], acc, 0, 0, 0);
        }
        OV_PROBE(4 + 2 * (lt & 3));
        if (EPI == OV_EPI_ADAMW) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#if defined(OV_DW0_NOADAM)
                w[buf][r] += acc[r]; mm[buf][r] += acc[r]; vv[buf][r] += acc[r];
#elif defined(OV_DW0_IEEEADAM)
                ov_adamw(w[buf][r], mm[buf][r], vv[buf][r], acc[r], p.adam, false);
#else
                ov_adamw_fast(w[buf][r], mm[buf][r], vv[buf][r], acc[r], p.adam, false);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) request(buf ^ 1, lt + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ov_buf_store(w[buf][r], row_rsrc(p.C, mb, r), lane_off, 0, 0);
                ov_buf_store(mm[buf][r], row_rsrc(p.Pm, mb, r), lane_off, 0, 0);
                ov_buf_store(vv[buf][r], row_rsrc(p.Pv, mb, r), lane_off, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ov_buf_store(acc[r], row_rsrc(p.C, mb, r), lane_off, 0, 0);
        }
        OV_PROBE(5 + 2 * (lt & 3));
    };
    OV_PROBE(3);
#pragma unroll 1
    for (int lt = 0; lt < LT && m0 + 32 * lt < p.M; lt += 2) {
        const bool more1 = m0 + 32 * (lt + 1) < p.M;
        tile(0, lt, more1);
        if (!more1) break;
        tile(1, lt + 1, lt + 2 < LT && m0 + 32 * (lt + 2) < p.M);
    }
    OV_PROBE(15);
#ifdef OV_DW0_PROBE
    if (lane == 0 && p.bias) {
        uint64_t* out = reinterpret_cast<uint64_t*>(const_cast<float*>(p.bias)) + ((int64_t)blockIdx.x * NW + wave) * 16;
        for (int q = 0; q < 16; ++q) out[q] = probe_t[q];
    }
#endif
}

// The same product for any batch size (K > 128: larger single-GPU batches, and the data-parallel step where every
// rank forms dW0 of the GLOBAL batch from the all-gathered dh): K is walked in chunks of 128 with the four tile
// accumulators of the wave held across chunks, then the AdamW pass over the tiles with the same request ring.
// MFMA-bound from K of a few hundred on (2 L G K flops), so the chunk prologues are not pipelined.
template <int NW, int EPI>
__global__ __launch_bounds__(64 * NW) void ov_dw0_multi_kernel(const OvGemm p, int njobs) {
    constexpr int LT = 4, RA = 32 * LT, KC = 128, LDA = KC + 4;
    extern __shared__ __attribute__((aligned(16))) float ov_gemm_lds[];
    float* As = ov_gemm_lds;                                             // [RA][LDA]
    int64_t* rows = reinterpret_cast<int64_t*>(ov_gemm_lds + RA * LDA);  // [KC] element offsets of the chunk's x rows
    const int tid = threadIdx.x;
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int job = (b >> 3) % njobs;
    const int gx = (b / (8 * njobs)) * 8 + (b & 7);
    const int m0 = job * RA;
    const int n0 = (gx * NW + wave) * 32;
    const bool wave_on = n0 < p.N;
    const uint32_t row_bytes = (uint32_t)p.ldc * 4u;
    const int32_t lane_off = n0 + i < p.N ? (int32_t)(((uint32_t)(4 * h) * (uint32_t)p.ldc + (uint32_t)(n0 + i)) * 4u) : (int32_t)0x80000000u;
    auto row_rsrc = [&](const float* base, int mb, int r) {
        const int row = mb + (r & 3) + 8 * (r >> 2);
        const uint32_t ro = (uint32_t)row * row_bytes;
        return ov_buf_rsrc(reinterpret_cast<const char*>(base) + ro, (uint32_t)max(p.M - row, 0) * row_bytes);
    };
    const int blm = p.a_blm > 0 ? p.a_blm : 0x40000000;
    const float* xcol = p.B + min(n0 + i, p.N - 1);
    if (p.b0sum && blockIdx.x == 0 && p.a_blm > 0) {
        const int nr = p.K / p.a_blm;
        for (int l = tid; l < p.M; l += 64 * NW) {
            float sum = 0.f;
            for (int r = 0; r < nr; ++r) sum += p.A[(int64_t)r * p.a_rstride + (int64_t)p.M * p.a_blm + l];
            p.b0sum[l] = sum;
        }
    }

    f32x16 acc[LT];
#pragma unroll
    for (int lt = 0; lt < LT; ++lt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[lt][r] = 0.f;

    constexpr int SU = RA * (KC / 4) / (64 * NW);
#pragma unroll 1
    for (int kc = 0; kc < p.K; kc += KC) {
        f32x4 sv[SU];
#pragma unroll
        for (int q = 0; q < SU; ++q) {
            const int u = tid + q * 64 * NW;
            const int r = u / (KC / 4), k = kc + 4 * (u % (KC / 4));
            const int m = m0 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < p.M && k < p.K) {
                const int rb = k / blm, j = k - rb * blm;
                const float* src = p.A + (int64_t)rb * p.a_rstride + (int64_t)m * p.lda + j;
                if (k + 3 < p.K && j + 3 < blm) v = *reinterpret_cast<const f32x4*>(src);
                else {                          // the last columns of the batch (never across a rank block: a_blm % 4 == 0)
                    v.x = src[0];
                    v.y = k + 1 < p.K ? src[1] : 0.f;
                    v.z = k + 2 < p.K ? src[2] : 0.f;
                }
            }
            sv[q] = v;
        }
        for (int t = tid; t < KC; t += 64 * NW) {
            int tt = min(kc + t, p.K - 1);
            if (p.colmap) tt = p.colmap[tt];
            rows[t] = (int64_t)(p.bgather ? p.bgather[tt] : tt) * p.ldb;
        }
        __syncthreads();
        float xb[KC / 2];
        if (wave_on) {
#pragma unroll
            for (int s = 0; s < KC / 8; ++s) {
                const ov_i64x2 r01 = *reinterpret_cast<const ov_i64x2*>(rows + 8 * s + 4 * h);
                const ov_i64x2 r23 = *reinterpret_cast<const ov_i64x2*>(rows + 8 * s + 4 * h + 2);
                xb[4 * s + 0] = xcol[r01.x];
                xb[4 * s + 1] = xcol[r01.y];
                xb[4 * s + 2] = xcol[r23.x];
                xb[4 * s + 3] = xcol[r23.y];
            }
        }
#pragma unroll
        for (int q = 0; q < SU; ++q) {
            const int u = tid + q * 64 * NW;
            *reinterpret_cast<f32x4*>(As + (u / (KC / 4)) * LDA + 4 * (u % (KC / 4))) = sv[q];
        }
        __syncthreads();
        if (wave_on) {
#pragma unroll
            for (int lt = 0; lt < LT; ++lt) {
                if (m0 + 32 * lt < p.M) {            // uniform
                    const float* arow = As + (32 * lt + i) * LDA + 4 * h;
#pragma unroll
                    for (int s = 0; s < KC / 8; ++s) {
                        const f32x4 fa = *reinterpret_cast<const f32x4*>(arow + 8 * s);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[lt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], xb[4 * s + t], acc[lt], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                             // everybody is done with this chunk's LDS
    }
    if (!wave_on) return;

    if (EPI != OV_EPI_ADAMW) {
#pragma unroll
        for (int lt = 0; lt < LT; ++lt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ov_buf_store(acc[lt][r], row_rsrc(p.C, m0 + 32 * lt, r), lane_off, 0, 0);
        return;
    }
    float w[2][16], mm[2][16], vv[2][16];
    auto request = [&](int buf, int lt) {
        const int mb = m0 + 32 * lt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            w[buf][r] = ov_buf_load(row_rsrc(p.C, mb, r), lane_off, 0, 0);
            mm[buf][r] = ov_buf_load(row_rsrc(p.Pm, mb, r), lane_off, 0, 0);
            vv[buf][r] = ov_buf_load(row_rsrc(p.Pv, mb, r), lane_off, 0, 0);
        }
    };
    request(0, 0);
#pragma unroll
    for (int lt = 0; lt < LT; ++lt) {
        const int buf = lt & 1;
        const int mb = m0 + 32 * lt;
        if (mb < p.M) {                              // uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) ov_adamw_fast(w[buf][r], mm[buf][r], vv[buf][r], acc[lt][r], p.adam, false);
            __builtin_amdgcn_sched_barrier(0);
            if (lt + 1 < LT && mb + 32 < p.M) request(buf ^ 1, lt + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ov_buf_store(w[buf][r], row_rsrc(p.C, mb, r), lane_off, 0, 0);
                ov_buf_store(mm[buf][r], row_rsrc(p.Pm, mb, r), lane_off, 0, 0);
                ov_buf_store(vv[buf][r], row_rsrc(p.Pv, mb, r), lane_off, 0, 0);
            }
        }
    }
}

// ov_dw0_multi_kernel with the chunk prologue taken off the critical path.  There a workgroup loads a chunk's dh slab and
// its 131 KB of x, waits, and only then starts that chunk's MFMAs.  Here the chunks are 64 rows, the x
// fragment (32 registers per chunk) and the dh slab (LDS) are double-buffered, and chunk c + 1 is requested before the MFMAs
// of chunk c; the row indices of the whole batch are put in LDS once, so that the requests of a chunk depend on nothing
// that is still in flight.  One barrier per chunk.  The order of the sum over the batch is the same as in
// ov_dw0_multi_kernel (k ascending inside every accumulator), so the two kernels give the same bits.  Measured: 190 against
// 197 us at 1024 rows, 1.21 against 1.23 ms at 8192 (both MFMA-paced: 0.62 of the fp32 peak on the algorithmic flops, 0.72
// on the issued ones, with 237 workgroups on 256 CUs and 384 rows of tiles for 360 rows); the multi kernel stays for batches
// whose row indices do not fit LDS.
// Dynamic LDS: 2 * 128 * 68 floats + round_up(K, 64) 64-bit offsets.
template <int NW, int EPI>
__global__ __launch_bounds__(64 * NW) void ov_dw0_pipe_kernel(const OvGemm p, int njobs) {
    constexpr int LT = 4, RA = 32 * LT, KC = 64, LDA = KC + 4;
    extern __shared__ __attribute__((aligned(16))) float ov_gemm_lds[];
    float* As0 = ov_gemm_lds;                                            // [RA][LDA] x 2
    float* As1 = As0 + RA * LDA;
    int64_t* rows = reinterpret_cast<int64_t*>(As1 + RA * LDA);          // [round_up(K, KC)] element offsets of the x rows; beyond K: the last row (dh is 0 there)
    const int tid = threadIdx.x;
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int job = (b >> 3) % njobs;
    const int gx = (b / (8 * njobs)) * 8 + (b & 7);
    const int m0 = job * RA;
    const int n0 = (gx * NW + wave) * 32;
    const bool wave_on = n0 < p.N;
    const uint32_t row_bytes = (uint32_t)p.ldc * 4u;
    const int32_t lane_off = n0 + i < p.N ? (int32_t)(((uint32_t)(4 * h) * (uint32_t)p.ldc + (uint32_t)(n0 + i)) * 4u) : (int32_t)0x80000000u;
    auto row_rsrc = [&](const float* base, int mb, int r) {
        const int row = mb + (r & 3) + 8 * (r >> 2);
        const uint32_t ro = (uint32_t)row * row_bytes;
        return ov_buf_rsrc(reinterpret_cast<const char*>(base) + ro, (uint32_t)max(p.M - row, 0) * row_bytes);
    };
    const int blm = p.a_blm > 0 ? p.a_blm : 0x40000000;
    const float* xcol = p.B + min(n0 + i, p.N - 1);
    if (p.b0sum && blockIdx.x == 0 && p.a_blm > 0) {
        const int nr = p.K / p.a_blm;
        for (int l = tid; l < p.M; l += 64 * NW) {
            float sum = 0.f;
            for (int r = 0; r < nr; ++r) sum += p.A[(int64_t)r * p.a_rstride + (int64_t)p.M * p.a_blm + l];
            p.b0sum[l] = sum;
        }
    }
    const int kpad = (p.K + KC - 1) / KC * KC;
    for (int t = tid; t < kpad; t += 64 * NW) {
        int tt = min(t, p.K - 1);
        if (p.colmap) tt = p.colmap[tt];
        rows[t] = (int64_t)(p.bgather ? p.bgather[tt] : tt) * p.ldb;
    }

    f32x16 acc[LT];
#pragma unroll
    for (int lt = 0; lt < LT; ++lt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[lt][r] = 0.f;

    constexpr int SU = RA * (KC / 4) / (64 * NW);
    // No lane-dependent control flow around the loads of the K loop: every lane issues the same loads at clamped addresses and
    // the out-of-range elements are zeroed by selects (store_a), so that the compiler's wait-count bookkeeping stays exact.  With
    // branches around them it fell back to "wait for everything" in front of every slab store and made the MFMAs of a chunk wait
    // for the NEXT chunk's x loads one by one (30 % of the kernel at s_waitcnt: profiles/r03/gemm_counters.txt).
    // A row's 16-B pieces never cross its end: lda % 4 == 0 and lda >= K (host check), rank blocks are multiples of 4 columns.
    const int klast = (p.K - 1) & ~3;
    auto load_a = [&](int kc, f32x4 (&sv)[SU]) {
#pragma unroll
        for (int q = 0; q < SU; ++q) {
            const int u = tid + q * 64 * NW;
            const int r = u / (KC / 4), k = min(kc + 4 * (u % (KC / 4)), klast);
            const int m = min(m0 + r, p.M - 1);
            int rb = 0, j = k;
            if (p.a_blm > 0) { rb = k / blm; j = k - rb * blm; }          // uniform
            sv[q] = *reinterpret_cast<const f32x4*>(p.A + (int64_t)rb * p.a_rstride + (int64_t)m * p.lda + j);
        }
    };
    auto store_a = [&](float* As, int kc, const f32x4 (&sv)[SU]) {
#pragma unroll
        for (int q = 0; q < SU; ++q) {
            const int u = tid + q * 64 * NW;
            const int r = u / (KC / 4), k = kc + 4 * (u % (KC / 4));
            const bool ok = m0 + r < p.M;
            f32x4 v;
            v.x = ok && k < p.K ? sv[q].x : 0.f;
            v.y = ok && k + 1 < p.K ? sv[q].y : 0.f;
            v.z = ok && k + 2 < p.K ? sv[q].z : 0.f;
            v.w = ok && k + 3 < p.K ? sv[q].w : 0.f;
            *reinterpret_cast<f32x4*>(As + r * LDA + 4 * (u % (KC / 4))) = v;
        }
    };
    auto load_x = [&](int kc, float (&xb)[KC / 2]) {
#pragma unroll
        for (int s = 0; s < KC / 8; ++s) {
            const ov_i64x2 r01 = *reinterpret_cast<const ov_i64x2*>(rows + kc + 8 * s + 4 * h);
            const ov_i64x2 r23 = *reinterpret_cast<const ov_i64x2*>(rows + kc + 8 * s + 4 * h + 2);
            xb[4 * s + 0] = xcol[r01.x];
            xb[4 * s + 1] = xcol[r01.y];
            xb[4 * s + 2] = xcol[r23.x];
            xb[4 * s + 3] = xcol[r23.y];
        }
    };
    // One chunk: its 128 MFMAs with everything the NEXT chunk needs issued in between them, in ONE basic block.  Both waves of a
    // SIMD leave the chunk barrier together; while the loads and their address arithmetic stood in front of the MFMAs, neither
    // wave had an MFMA to issue for that long and the matrix pipe idled a fifth of every chunk (register-only and LDS-fed
    // loops of the same MFMAs run at 155 TFLOP/s, scratch/mfma_probe).  Eight groups (one per 8 k): the four tiles' fragments
    // of group s + 1 and the x-row offsets of the next chunk's group s + 1 are requested from LDS, the next chunk's x loads of
    // group s are issued, then the 16 MFMAs of group s run, the four accumulators taking turns (k ascending inside each, as in
    // ov_dw0_multi_kernel: same bits).  All four tiles are computed even where a workgroup's last tiles lie beyond M (their
    // dh rows are zeros in LDS and they are never stored): no branch inside the block.
    auto chunk = [&](const float* As, const float (&xc)[KC / 2], float (&xn)[KC / 2], int kn, f32x4 (&sv)[SU]) {
        const float* asrc[SU];
#pragma unroll
        for (int q = 0; q < SU; ++q) {
            const int u = tid + q * 64 * NW;
            const int r = u / (KC / 4), k = min(kn + 4 * (u % (KC / 4)), klast);
            const int m = min(m0 + r, p.M - 1);
            int rb = 0, j = k;
            if (p.a_blm > 0) { rb = k / blm; j = k - rb * blm; }          // uniform
            asrc[q] = p.A + (int64_t)rb * p.a_rstride + (int64_t)m * p.lda + j;
        }
        const float* arow = As + i * LDA + 4 * h;
        const int64_t* rown = rows + kn + 4 * h;
        f32x4 fa[2][LT];
        ov_i64x2 ro[2][2];
#pragma unroll
        for (int lt = 0; lt < LT; ++lt) fa[0][lt] = *reinterpret_cast<const f32x4*>(arow + 32 * lt * LDA);
        ro[0][0] = *reinterpret_cast<const ov_i64x2*>(rown);
        ro[0][1] = *reinterpret_cast<const ov_i64x2*>(rown + 2);
        // the slab loads first, ahead of the chunk's x loads (measured in scratch/mfma_probe: 18.7 k against 19.8 k cycles per
        // chunk when they are spread over the first four groups)
#pragma unroll
        for (int q = 0; q < SU; ++q) sv[q] = *reinterpret_cast<const f32x4*>(asrc[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KC / 8; ++s) {
            const int c = s & 1, n = c ^ 1;
            if (s + 1 < KC / 8) {
#pragma unroll
                for (int lt = 0; lt < LT; ++lt) fa[n][lt] = *reinterpret_cast<const f32x4*>(arow + 32 * lt * LDA + 8 * (s + 1));
            }
            xn[4 * s + 0] = xcol[ro[c][0].x];
            xn[4 * s + 1] = xcol[ro[c][0].y];
            xn[4 * s + 2] = xcol[ro[c][1].x];
            xn[4 * s + 3] = xcol[ro[c][1].y];
            if (s + 1 < KC / 8) {
                ro[n][0] = *reinterpret_cast<const ov_i64x2*>(rown + 8 * (s + 1));
                ro[n][1] = *reinterpret_cast<const ov_i64x2*>(rown + 8 * (s + 1) + 2);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int lt = 0; lt < LT; ++lt) acc[lt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][lt][t], xc[4 * s + t], acc[lt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x4 sv[SU];
    float xa[KC / 2], xb[KC / 2];
    // The loads of the next chunk are issued on EVERY path (waves beyond N and the last chunk included: clamped addresses, results
    // unused), never inside a branch: where a path without them joins one with them the compiler must place waits that are right
    // for both, and "the slab loads are the youngest four" of the path without x loads made every slab store wait for the x
    // prefetch as well (and the MFMAs of a chunk for the next chunk's loads).
    const int kend = kpad - KC;
    load_a(0, sv);
    __syncthreads();                                 // the row offsets are in LDS
    load_x(0, xa);
    store_a(As0, 0, sv);
    __syncthreads();
#pragma unroll 1
    for (int kc = 0; kc < p.K; kc += 2 * KC) {
        chunk(As0, xa, xb, min(kc + KC, kend), sv);
        if (kc + KC >= p.K) break;                   // workgroup-uniform
        store_a(As1, kc + KC, sv);
        __syncthreads();                             // slab kc + KC is in LDS, and everybody is done reading As0
        chunk(As1, xb, xa, min(kc + 2 * KC, kend), sv);
        if (kc + 2 * KC >= p.K) break;
        store_a(As0, kc + 2 * KC, sv);
        __syncthreads();
    }
    if (!wave_on) return;

    if (EPI != OV_EPI_ADAMW) {
#pragma unroll
        for (int lt = 0; lt < LT; ++lt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ov_buf_store(acc[lt][r], row_rsrc(p.C, m0 + 32 * lt, r), lane_off, 0, 0);
        return;
    }
    float w[2][16], mm[2][16], vv[2][16];
    auto request = [&](int buf, int lt) {
        const int mb = m0 + 32 * lt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            w[buf][r] = ov_buf_load(row_rsrc(p.C, mb, r), lane_off, 0, 0);
            mm[buf][r] = ov_buf_load(row_rsrc(p.Pm, mb, r), lane_off, 0, 0);
            vv[buf][r] = ov_buf_load(row_rsrc(p.Pv, mb, r), lane_off, 0, 0);
        }
    };
    request(0, 0);
#pragma unroll
    for (int lt = 0; lt < LT; ++lt) {
        const int buf = lt & 1;
        const int mb = m0 + 32 * lt;
        if (mb < p.M) {                              // uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) ov_adamw_fast(w[buf][r], mm[buf][r], vv[buf][r], acc[lt][r], p.adam, false);
            __builtin_amdgcn_sched_barrier(0);
            if (lt + 1 < LT && mb + 32 < p.M) request(buf ^ 1, lt + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ov_buf_store(w[buf][r], row_rsrc(p.C, mb, r), lane_off, 0, 0);
                ov_buf_store(mm[buf][r], row_rsrc(p.Pm, mb, r), lane_off, 0, 0);
                ov_buf_store(vv[buf][r], row_rsrc(p.Pv, mb, r), lane_off, 0, 0);
            }
        }
    }
}

#endif
