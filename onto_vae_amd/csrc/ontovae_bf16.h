// bf16-MFMA variant of the two large dense products of the OntoVAE step (BASELINE.json configs[4]: "bf16 MFMA variant with
// stated tolerance"): inputs rounded to bf16 (round to nearest even), products accumulated in fp32 on
// v_mfma_f32_32x32x16_bf16, master weights, optimizer state and everything outside these two products in fp32.
//
//   encoder forward   h  [L, B] = W0 [L, G] . x^T        (reference onto_vae/modules.py:46)
//   first-layer grad  dW0[L, G] = dh [L, B] . x          (autograd of the same line)
//
// Both run through ONE kernel, C[M, N] = A[M, K] . B[N, K]^T with A and B K-contiguous bf16 matrices, so the operands are
// prepared once per step by conversion passes: x rows -> xb [B, G] (and, for training, the transposed copy xT [G, ldb]),
// W0 -> wb [L, G], dh -> dhb [L, ldb].  The conversions are HBM streams (a few % of the products' time at the batch sizes
// this variant is for: thousands of rows per GPU, where the fp32 products are MFMA-bound).
#ifndef ONTOVAE_BF16_H
#define ONTOVAE_BF16_H

#include "ontovae_device.h"

typedef __bf16 ov_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ov_f32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t ov_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint16_t ov_f2bf(float v) {
    const __bf16 b = (__bf16)v;                       // round to nearest even (v_cvt_pk_bf16_f32 on gfx950)
    return __builtin_bit_cast(uint16_t, b);
}

// flat fp32 -> bf16, n % 8 == 0 not required
__global__ __launch_bounds__(256) void ov_bf16_convert_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            const ov_f32x8 v = *reinterpret_cast<const ov_f32x8*>(src + i);
            *reinterpret_cast<ov_bf16x8*>(dst + i) = __builtin_convertvector(v, ov_bf16x8);
        } else {
            for (int64_t j = i; j < n; ++j) dst[j] = ov_f2bf(src[j]);
        }
    }
}

// [rows, ld_src] fp32 row block -> [rows, ld_dst] bf16 with columns [cols, ld_dst) zeroed (dh: columns beyond the batch)
__global__ __launch_bounds__(256) void ov_bf16_convert_rows_kernel(const float* __restrict__ src, int64_t ld_src, int cols,
                                                                  uint16_t* __restrict__ dst, int64_t ld_dst) {
    const int r = blockIdx.y;
    for (int c = (blockIdx.x * 256 + threadIdx.x) * 4; c < ld_dst; c += gridDim.x * 1024) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c + j < ld_dst) dst[(int64_t)r * ld_dst + c + j] = c + j < cols ? ov_f2bf(src[(int64_t)r * ld_src + c + j]) : (uint16_t)0;
    }
}

// batch rows of x (row b = X[xidx ? xidx[b] : b]) -> xb [B, G] bf16 and, if xT != NULL, xT [G, ldt] bf16 with columns
// [B, ldt) zeroed (the K padding of the dW0 product).  64 x 64 tiles through LDS; grid.y covers ldt / 64 tiles.
__global__ __launch_bounds__(256) void ov_bf16_x_kernel(const float* __restrict__ x, const int32_t* __restrict__ xidx, int64_t xld,
                                                       int B, int G, uint16_t* __restrict__ xb, uint16_t* __restrict__ xT, int64_t ldt) {
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    __shared__ uint16_t tile[64][68];                                // [b][g], row stride 136 B
    const int g0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 x 16 threads, 4 genes per thread and row
    const bool vec = (G % 4 == 0) && (xld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    for (int j = ty; j < 64; j += 16) {
        const int b = b0 + j, g = g0 + 4 * tx;
        u16x4 v = {0, 0, 0, 0};
        if (b < B && g < G) {
            const int64_t r = xidx ? (int64_t)xidx[b] : (int64_t)b;
            const float* src = x + r * xld + g;
            if (vec) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(src);
                v = u16x4{ov_f2bf(f.x), ov_f2bf(f.y), ov_f2bf(f.z), ov_f2bf(f.w)};
                *reinterpret_cast<u16x4*>(xb + (int64_t)b * G + g) = v;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (g + q < G) { v[q] = ov_f2bf(src[q]); xb[(int64_t)b * G + g + q] = v[q]; }
            }
        }
        *reinterpret_cast<u16x4*>(&tile[j][4 * tx]) = v;
    }
    if (!xT) return;
    __syncthreads();
    for (int j = ty; j < 64; j += 16) {
        const int g = g0 + j;
        if (g >= G) continue;
        const int b = b0 + 4 * tx;                                    // ldt % 64 == 0: the 4 columns are inside the row
        const u16x4 v = {tile[4 * tx][j], tile[4 * tx + 1][j], tile[4 * tx + 2][j], tile[4 * tx + 3][j]};
        *reinterpret_cast<u16x4*>(xT + (int64_t)g * ldt + b) = v;     // zeros beyond the batch: the K padding
    }
}

struct OvBfGemm {
    const uint16_t* A; int64_t lda;      // [M, K] bf16
    const uint16_t* B; int64_t ldb;      // [N, K] bf16
    float* C; int64_t ldc;               // [M, N] fp32, slab z at C + z * slab_stride
    int64_t slab_stride;
    int M, N, K, kchunk;                 // K % 8 == 0, kchunk % 64 == 0; lda, ldb % 8 == 0, A and B 16-B aligned
};

// 128 x 128 output tile per workgroup, 4 waves of 64 x 64 (2 x 2 MFMA tiles of 32 x 32), K in steps of 64 through LDS; the
// next step's global loads are in flight (registers) while the current one is multiplied.
#define OV_BF_BM 128
#define OV_BF_BN 128
#define OV_BF_BK 64
#define OV_BF_LD (OV_BF_BK + 8)          // row stride in LDS, bf16 elements: 144 B, conflict-free 16-B reads per 16 lanes
__global__ __launch_bounds__(256) void ov_bf16_gemm_tn_kernel(const OvBfGemm p) {
    __shared__ __attribute__((aligned(16))) uint16_t As[OV_BF_BM * OV_BF_LD];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[OV_BF_BN * OV_BF_LD];
    const int n0 = blockIdx.x * OV_BF_BN, m0 = blockIdx.y * OV_BF_BM, z = blockIdx.z;
    const int kbeg = z * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lc = tid & 7, lr = tid >> 3;                           // 16-B chunk of the K step, row (+ 32 i)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    ov_u32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
        const int k = k0 + lc * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lr + 32 * i, n = n0 + lr + 32 * i;
            ra[i] = ov_u32x4{0u, 0u, 0u, 0u};
            rb[i] = ov_u32x4{0u, 0u, 0u, 0u};
            if (m < p.M && k < kend) ra[i] = *reinterpret_cast<const ov_u32x4*>(p.A + (int64_t)m * p.lda + k);
            if (n < p.N && k < kend) rb[i] = *reinterpret_cast<const ov_u32x4*>(p.B + (int64_t)n * p.ldb + k);
        }
    };
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += OV_BF_BK) {
        __syncthreads();                                              // the previous step's LDS reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<ov_u32x4*>(&As[(lr + 32 * i) * OV_BF_LD + lc * 8]) = ra[i];
            *reinterpret_cast<ov_u32x4*>(&Bs[(lr + 32 * i) * OV_BF_LD + lc * 8]) = rb[i];
        }
        __syncthreads();
        if (k0 + OV_BF_BK < kend) gload(k0 + OV_BF_BK);
#pragma unroll
        for (int kk = 0; kk < OV_BF_BK / 16; ++kk) {
            ov_bf16x8 a[2], b[2];
            const int ko = kk * 16 + (lane >> 5) * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const ov_bf16x8*>(&As[(wm + 32 * i + (lane & 31)) * OV_BF_LD + ko]);
                b[i] = *reinterpret_cast<const ov_bf16x8*>(&Bs[(wn + 32 * i + (lane & 31)) * OV_BF_LD + ko]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float* C = p.C + (int64_t)z * p.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + 32 * i + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3);
                if (m < p.M && n < p.N) C[(int64_t)m * p.ldc + n] = acc[i][j][e];
            }
        }
}

#endif
