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
// 32x32x2 per SIMD, so per k-step of 8 a wave needs MT + NT 16-byte loads per 4*MT*NT MFMAs and the
// load path is nowhere near critical; there is no LDS stage and no barrier.  The lane -> (row, k)
// map of one k-step of 8: lane (i = lane & 31, h = lane >> 5) supplies k = k0 + 4h + t to MFMA t
// (t = 0..3), for A and B alike, which is all the instruction needs.
//
// grid = (ceil(N / (64 NT)), ceil(M / (64 MT)), S); blockIdx.z is the split-K slab.
#ifndef ONTOVAE_GEMM_H
#define ONTOVAE_GEMM_H

#include "ontovae_device.h"

enum { OV_EPI_STORE = 0, OV_EPI_BIAS = 1, OV_EPI_ADAMW = 2 };

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
    int kchunk;           // multiple of 8
    const float* bias;    // EPI_BIAS: per-m bias
    float* Pm;            // EPI_ADAMW: moments, same indexing as C
    float* Pv;
    OvAdam adam;
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

template <bool FULL>
__device__ __forceinline__ f32x4 ov_ld_mc(const float* __restrict__ base, int64_t ld, int col, int k, int kend,
                                          const int32_t* __restrict__ gather) {
    f32x4 v;
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = k + j;
        if (FULL || kk < kend) {
            const int64_t r = gather ? (int64_t)gather[kk] : (int64_t)kk;
            t[j] = base[r * ld + col];
        } else {
            t[j] = 0.f;
        }
    }
    v.x = t[0]; v.y = t[1]; v.z = t[2]; v.w = t[3];
    return v;
}

template <int MT, int NT, bool A_KC, bool B_KC, bool VEC, int EPI>
__global__ __launch_bounds__(256) void ov_gemm_kernel(const OvGemm p) {
    const int lane = ov_lane();
    const int wave = ov_wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * (64 * MT) + (wave >> 1) * (32 * MT);
    const int n0 = blockIdx.x * (64 * NT) + (wave & 1) * (32 * NT);
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    if (m0 >= p.M || n0 >= p.N) return;   // wave-uniform

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

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    f32x4 a[MT], b[NT];
    auto load_full = [&](int k0) {
        const int k = k0 + 4 * h;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a[mt] = A_KC ? ov_ld_kc<true, VEC>(abase[mt], k, kend) : ov_ld_mc<true>(abase[mt], p.lda, acol[mt], k, kend, nullptr);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            b[nt] = B_KC ? ov_ld_kc<true, VEC>(bbase[nt], k, kend) : ov_ld_mc<true>(bbase[nt], p.ldb, bcol[nt], k, kend, p.bgather);
    };
    auto load_tail = [&](int k0) {
        const int k = k0 + 4 * h;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a[mt] = A_KC ? ov_ld_kc<false, VEC>(abase[mt], k, kend) : ov_ld_mc<false>(abase[mt], p.lda, acol[mt], k, kend, nullptr);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            b[nt] = B_KC ? ov_ld_kc<false, VEC>(bbase[nt], k, kend) : ov_ld_mc<false>(bbase[nt], p.ldb, bcol[nt], k, kend, p.bgather);
    };
    auto mfma_step = [&](const f32x4 (&aa)[MT], const f32x4 (&bb)[NT]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[mt][t], bb[nt][t], acc[mt][nt], 0, 0, 0);
    };

    const int nfull = (kend - kbeg) / 8;
    int k0 = kbeg;
    if (nfull > 0) {
        load_full(k0);
        for (int it = 0; it < nfull; ++it) {
            f32x4 ca[MT], cb[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ca[mt] = a[mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) cb[nt] = b[nt];
            k0 += 8;
            if (it + 1 < nfull) load_full(k0);     // prefetch the next k-step under this one's MFMAs
            mfma_step(ca, cb);
        }
    }
    if (k0 < kend) {
        load_tail(k0);
        mfma_step(a, b);
    }

    // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + 32 * nt + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M && n < p.N) {
                    const int64_t idx = (int64_t)m * p.ldc + n;
                    float val = acc[mt][nt][r];
                    if (EPI == OV_EPI_STORE) {
                        p.C[(int64_t)blockIdx.z * p.slab_stride + idx] = val;
                    } else if (EPI == OV_EPI_BIAS) {
                        p.C[idx] = val + p.bias[m];
                    } else {
                        float w = p.C[idx], mm = p.Pm[idx], vv = p.Pv[idx];
                        ov_adamw(w, mm, vv, val, p.adam, false);
                        p.C[idx] = w; p.Pm[idx] = mm; p.Pv[idx] = vv;
                    }
                }
            }
        }
}

#endif
