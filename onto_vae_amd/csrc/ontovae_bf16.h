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

// -------------------------------------------------------------------------------------------------
// Second version of the product kernel (round 3): the same 128 x 128 tile, operand order and summation order as
// ov_bf16_gemm_tn_kernel (bit-identical sums), built like ov_gemm_lds2_kernel: two LDS stages and ONE barrier per 64-deep
// K-step, the fragments of the next 16 k read while the current 4 MFMAs run (two register sets), global loads two steps ahead,
// the next step's first fragments read across the barrier.  EPI 0: fp32 tile to slab z.  EPI 1 (dW0): the tile is the
// gradient of a tile of W0 -- AdamW (the arithmetic of ov_adamw_flat_kernel) on w / exp_avg / exp_avg_sq in the epilogue,
// no gradient buffer, no second pass.
// -------------------------------------------------------------------------------------------------
struct OvBfGemm2 {
    OvBfGemm g;
    float* Pm; float* Pv;                // EPI 1: exp_avg, exp_avg_sq (g.C = the weights), same leading dimension g.ldc
    OvAdam adam;
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void ov_bf16_gemm2_kernel(const OvBfGemm2 q) {       // two workgroups per CU (LDS: 2 x 72 KB)
    const OvBfGemm& p = q.g;
    constexpr int ROWS = OV_BF_BM + OV_BF_BN, STAGE = ROWS * OV_BF_LD;      // bf16 elements per stage: A rows, then B rows
    extern __shared__ __attribute__((aligned(16))) uint16_t ov_bf_lds[];
    // workgroups are dealt to the 8 XCDs round-robin by linear id: the row tiles (y) of one column tile and slab read the same
    // B rows, so they are renumbered to be neighbours in time on one XCD
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nx = gridDim.x, ny = gridDim.y;
        const int id = blockIdx.x + nx * blockIdx.y;
        if ((nx * ny) % 8 == 0) {
            const int xcd = id & 7, j = id >> 3, per = nx * ny / 8;      // XCD xcd owns linear tiles [xcd * per, (xcd + 1) * per)
            const int t = xcd * per + j;
            by = t % ny; bx = t / ny;
        }
    }
    const int n0 = bx * OV_BF_BN, m0 = by * OV_BF_BM, z = blockIdx.z;
    const int kbeg = z * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lc8 = (tid & 7) * 8, lr = tid >> 3;                       // 16-B piece of the K step, row (+ 32 i)
    const uint16_t* asrc[4];
    const uint16_t* bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asrc[i] = p.A + (int64_t)min(m0 + lr + 32 * i, p.M - 1) * p.lda;
        bsrc[i] = p.B + (int64_t)min(n0 + lr + 32 * i, p.N - 1) * p.ldb;
    }
    const int kclamp = (p.K - 1) & ~7;                                   // the last 16-B piece of a row inside the K range
    ov_u32x4 ra[4], rb[4];
    auto gload = [&](int k0) {               // every lane, every load, addresses clamped into the row (exact wait counts)
        const int k = min(k0 + lc8, kclamp);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const ov_u32x4*>(asrc[i] + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const ov_u32x4*>(bsrc[i] + k);
    };
    const int dst0 = lr * OV_BF_LD + lc8;
    auto lstore = [&](int k0, uint16_t* st) {    // K % 8 == 0: a piece is wholly inside or wholly outside the range
        const bool ok = k0 + lc8 < kend;
        const ov_u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<ov_u32x4*>(st + dst0 + 32 * i * OV_BF_LD) = ok ? ra[i] : zero;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<ov_u32x4*>(st + (OV_BF_BM + 32 * i) * OV_BF_LD + dst0) = ok ? rb[i] : zero;
    };
    const int fa_off = (wm + (lane & 31)) * OV_BF_LD + (lane >> 5) * 8;
    const int fb_off = (OV_BF_BM + wn + (lane & 31)) * OV_BF_LD + (lane >> 5) * 8;
    auto fread = [&](const uint16_t* st, int g, ov_bf16x8 (&a)[2], ov_bf16x8 (&b)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = *reinterpret_cast<const ov_bf16x8*>(st + fa_off + 32 * i * OV_BF_LD + 16 * g);
            b[i] = *reinterpret_cast<const ov_bf16x8*>(st + fb_off + 32 * i * OV_BF_LD + 16 * g);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto mfmas = [&](const ov_bf16x8 (&a)[2], const ov_bf16x8 (&b)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };

    if (kbeg < kend) {
        gload(kbeg);
        lstore(kbeg, ov_bf_lds);
        gload(kbeg + OV_BF_BK);
        __syncthreads();
        int cur = 0;
        ov_bf16x8 a0[2], b0[2], a1[2], b1[2];
        fread(ov_bf_lds, 0, a0, b0);
        for (int k0 = kbeg; k0 < kend; k0 += OV_BF_BK) {
            const uint16_t* st = ov_bf_lds + cur * STAGE;
            uint16_t* nx = ov_bf_lds + (cur ^ 1) * STAGE;
            fread(st, 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            lstore(k0 + OV_BF_BK, nx);               // the other buffer: nobody reads it before the barrier below
            gload(k0 + 2 * OV_BF_BK);
            mfmas(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            fread(st, 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            fread(st, 3, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                         // next stage visible; this wave's reads of the current one have returned
            fread(nx, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a1, b1);
            cur ^= 1;
        }
    }

    // C/D map of a 32 x 32 tile: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + (lane & 31);
            const int mb = m0 + wm + 32 * i + (lane >> 5) * 4;
            if (EPI == 0) {
                float* C = p.C + (int64_t)z * p.slab_stride;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e >> 2) * 8 + (e & 3);
                    if (m < p.M && n < p.N) C[(int64_t)m * p.ldc + n] = acc[i][j][e];
                }
            } else {
                float w[16], mm[16], vv[16];
                bool ok[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e >> 2) * 8 + (e & 3);
                    ok[e] = m < p.M && n < p.N;
                    const int64_t idx = (int64_t)min(m, p.M - 1) * p.ldc + min(n, p.N - 1);
                    w[e] = p.C[idx]; mm[e] = q.Pm[idx]; vv[e] = q.Pv[idx];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) ov_adamw(w[e], mm[e], vv[e], acc[i][j][e], q.adam, false);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e >> 2) * 8 + (e & 3);
                    const int64_t idx = (int64_t)m * p.ldc + n;
                    if (ok[e]) { p.C[idx] = w[e]; q.Pm[idx] = mm[e]; q.Pv[idx] = vv[e]; }
                }
            }
        }
}

#endif
