// Microbenchmark: what the fp32 MFMA pipe gives for the dW0 / encoder inner loops' instruction patterns (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_probe/mfma_probe scratch/mfma_probe/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: operands in registers; NACC accumulators; ORDER 0: acc-major (32 dependent MFMAs on one accumulator, then the next),
// ORDER 1: interleaved (consecutive MFMAs on different accumulators).  MODE 1: A fragments read from LDS (ds_read_b128 per 4 MFMAs).
template <int NACC, int ORDER, int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float xb[32];
    for (int t = 0; t < 32; ++t) xb[t] = seed * (lane + t);
    for (int t = threadIdx.x; t < 128 * 68; t += blockDim.x) lds[t] = seed * t;
    __syncthreads();
    const int i = lane & 31, h = lane >> 5;
    float fa_reg[8];
    for (int t = 0; t < 8; ++t) fa_reg[t] = seed * (t + lane);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const float* arow = lds + (32 * a + i) * 68 + 4 * h;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    f32x4 fa;
                    if (MODE == 1) fa = *reinterpret_cast<const f32x4*>(arow + 8 * s);
                    else { fa.x = fa_reg[s]; fa.y = fa_reg[(s + 1) & 7]; fa.z = fa_reg[(s + 2) & 7]; fa.w = fa_reg[(s + 3) & 7]; }
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], xb[4 * s + t], acc[a], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                f32x4 fa[NACC];
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    if (MODE == 1) fa[a] = *reinterpret_cast<const f32x4*>(lds + (32 * a + i) * 68 + 4 * h + 8 * s);
                    else { fa[a].x = fa_reg[s]; fa[a].y = fa_reg[(s + 1) & 7]; fa[a].z = fa_reg[(s + 2) & 7]; fa[a].w = fa_reg[(s + 3) & 7]; }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][t], xb[4 * s + t], acc[a], 0, 0, 0);
            }
        }
    }
    float sum = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) sum += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int NACC, int ORDER, int MODE>
static void run(const char* name, int threads, int blocks, float* out) {
    const int iters = 2000;
    const size_t lds = 128 * 68 * sizeof(float);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, ORDER, MODE>), dim3(blocks), dim3(threads), lds, 0, out, 10, 1e-9f);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NACC, ORDER, MODE>), dim3(blocks), dim3(threads), lds, 0, out, iters, 1e-9f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * (threads / 64) * iters * NACC * 32 * 4096.0;
    printf("%-44s threads %3d blocks %4d  %8.3f ms  %7.1f TFLOP/s\n", name, threads, blocks, best, flops / best * 1e-9);
    fflush(stdout);
}

// The dW0 chunk loop, built up in steps.  STEP 1: A fragments from LDS, 16 MFMAs per group of 8 k, 8 groups per chunk.
// STEP 2: + a barrier per chunk.  STEP 3: + 4 ds_write_b128 per thread into the other slab before the barrier.
// STEP 4: + 32 gathered global loads per lane and chunk (next chunk's x), 4 per group.  STEP 5: + 4 16-B global loads (slab).
template <int STEP, bool TIMED = false>
__global__ __launch_bounds__(512) void chunkloop(float* out, const float* __restrict__ x, const float* __restrict__ a, int chunks, int ldx, int nrows, int lda) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LDA = 68;
    float* As0 = lds; float* As1 = lds + 128 * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    for (int t = tid; t < 2 * 128 * LDA; t += 512) lds[t] = 1e-9f * t;
    long long* rows = reinterpret_cast<long long*>(lds + 2 * 128 * LDA);          // element offsets of 8192 pseudo-random x rows
    for (int t = tid; t < 8192; t += 512) rows[t] = (long long)(((unsigned)t * 2654435761u) % (unsigned)nrows) * ldx;
    __syncthreads();
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float xa[32], xb[32];
    for (int t = 0; t < 32; ++t) { xa[t] = 1e-9f * (lane + t); xb[t] = xa[t]; }
    unsigned long long tw_vm = 0, tw_lds = 0, tw_bar = 0;
    float dummy = 0.f;
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
    f32x4 sv[4], sv2[4];
    for (int q = 0; q < 4; ++q) { sv[q] = f32x4{1e-9f, 2e-9f, 3e-9f, 4e-9f}; sv2[q] = sv[q]; }
    const float* xcol = x + ((blockIdx.x * 8 + wave) * 32 + i) % ldx;
    auto chunk = [&](const float* As, float* Asn, const float (&xc)[32], float (&xn)[32], int c, auto par) {
        constexpr bool PAR = decltype(par)::value;
        const float* arow = As + i * LDA + 4 * h;
        f32x4 fa[2][4];
#pragma unroll
        for (int lt = 0; lt < 4; ++lt) fa[0][lt] = *reinterpret_cast<const f32x4*>(arow + 32 * lt * LDA);
        if (STEP == 13) {     // all four slab loads ahead of the chunk's x loads
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] = *reinterpret_cast<const f32x4*>(a + ((int64_t)((tid + 512 * q) >> 4) * lda + (c * 64 + 4 * (tid & 15)) % 1024));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int cb = s & 1, nb = cb ^ 1;
            if (s + 1 < 8) {
#pragma unroll
                for (int lt = 0; lt < 4; ++lt) fa[nb][lt] = *reinterpret_cast<const f32x4*>(arow + 32 * lt * LDA + 8 * (s + 1));
            }
            if (STEP >= 4 && STEP != 11) {
                typedef long long i64x2 __attribute__((ext_vector_type(2)));
                const long long* rn = rows + ((c + 1) & 127) * 64 + 8 * s + 4 * h;
                const i64x2 r01 = *reinterpret_cast<const i64x2*>(rn), r23 = *reinterpret_cast<const i64x2*>(rn + 2);
                xn[4 * s + 0] = xcol[r01.x]; xn[4 * s + 1] = xcol[r01.y];
                if (STEP != 17) { xn[4 * s + 2] = xcol[r23.x]; xn[4 * s + 3] = xcol[r23.y]; }      // STEP 17: half of the x loads
            }
            // STEP 5: all workgroups read the same slab rows; 6: loaded but not stored (the stores write stale registers);
            // 7: every workgroup its own rows (no sharing); 8: slab loads in the LAST four groups instead of the first four
            if ((STEP == 5 || STEP == 7 || STEP == 11 || STEP == 14 || STEP == 15 || STEP == 17) && s < 4) sv[s] = *reinterpret_cast<const f32x4*>(a + ((int64_t)(((tid + 512 * s) >> 4) + (STEP == 7 ? (blockIdx.x % 3) * 128 : 0)) * lda + (c * 64 + 4 * (tid & 15)) % 1024));
            if (STEP == 9 && s < 4) sv[s] = *reinterpret_cast<const f32x4*>(a + ((int64_t)(((tid + 512 * s) >> 4) + blockIdx.x * 128) * lda + (c * 64 + 4 * (tid & 15)) % 1024));
            if (STEP == 10 && s < 4) sv[s] = *reinterpret_cast<const f32x4*>(a + ((int64_t)(((tid + 512 * s) >> 4) + (blockIdx.x & 7) * 128) * lda + (c * 64 + 4 * (tid & 15)) % 1024));
            if (STEP == 12 && s < 4) sv2[s] = *reinterpret_cast<const f32x4*>(a + ((int64_t)((tid + 512 * s) >> 4) * lda + (c * 64 + 4 * (tid & 15)) % 1024));
            if (STEP == 16 && s < 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(a + ((int64_t)((tid + 512 * s) >> 4) * lda + (c * 64 + 4 * (tid & 15)) % 1024));
                if (PAR) sv2[s] = v; else sv[s] = v;
            }
            if (STEP == 6 && s < 4) sv2[s] = *reinterpret_cast<const f32x4*>(a + ((int64_t)((tid + 512 * s) >> 4) * lda + (c * 64 + 4 * (tid & 15)) % 1024));
            if (STEP == 8 && s >= 4) sv[s - 4] = *reinterpret_cast<const f32x4*>(a + ((int64_t)((tid + 512 * (s - 4)) >> 4) * lda + (c * 64 + 4 * (tid & 15)) % 1024));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int lt = 0; lt < 4; ++lt) acc[lt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][lt][t], xc[4 * s + t], acc[lt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (TIMED) { t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); t1 = __builtin_amdgcn_s_memtime(); }
        if (STEP == 14) {       // consume the loaded slab in registers (forces the wait), store constants
#pragma unroll
            for (int q = 0; q < 4; ++q) { dummy += sv[q].x + sv[q].y + sv[q].z + sv[q].w; sv[q] = f32x4{1e-9f, 2e-9f, 3e-9f, 4e-9f}; }
        }
        if (STEP == 15) {       // store the loaded slab, but after an explicit wait for ALL loads
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (STEP >= 3) {   // (the stores of STEP 8 wait for loads issued at most four groups earlier)
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int u = tid + 512 * q; *reinterpret_cast<f32x4*>(Asn + (u >> 4) * LDA + 4 * (u & 15)) = (STEP == 16 && !PAR) ? sv2[q] : sv[q]; }
        }
        if (STEP == 12) { for (int q = 0; q < 4; ++q) sv[q] = sv2[q]; }      // what this chunk loaded is stored at the end of the next one
        if (TIMED) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t2 = __builtin_amdgcn_s_memtime(); }
        if (STEP >= 2) __syncthreads();
        if (TIMED) { t3 = __builtin_amdgcn_s_memtime(); tw_vm += t1 - t0; tw_lds += t2 - t1; tw_bar += t3 - t2; }
    };
#pragma unroll 1
    for (int c = 0; c < chunks; c += 2) {
        chunk(As0, As1, xa, xb, c, std::false_type{});
        chunk(As1, As0, xb, xa, c + 1, std::true_type{});
    }
    float sum = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    for (int q = 0; q < 4; ++q) sum += sv2[q].x + sv2[q].w;
    sum += dummy;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (!TIMED && tid == 0 && blockIdx.x == 0) {        // in-kernel clock: shader cycles per 100 MHz tick
        unsigned long long* o = reinterpret_cast<unsigned long long*>(out + 1024 * 256 + 65536);
        o[0] = __builtin_amdgcn_s_memtime() - tk0; o[1] = __builtin_amdgcn_s_memrealtime() - tr0;
    }
    if (TIMED && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(out + 1024 * 256) + (blockIdx.x * 8 + wave) * 4;
        o[0] = tw_vm; o[1] = tw_lds; o[2] = tw_bar; o[3] = __builtin_amdgcn_s_memtime() - tk0;
    }
}

template <int STEP>
static void run_chunk(const char* name, int blocks, float* out, const float* x, const float* a, int ldx, int nrows, int lda = 8192) {
    const int chunks = 128;
    const size_t lds = 2 * 128 * 68 * sizeof(float) + 8192 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(chunkloop<STEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chunkloop<STEP>), dim3(blocks), dim3(512), lds, 0, out, x, a, 4, ldx, nrows, lda);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((chunkloop<STEP>), dim3(blocks), dim3(512), lds, 0, out, x, a, chunks, ldx, nrows, lda);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 8 * chunks * 128 * 4096.0;
    unsigned long long clk[2];
    hipMemcpy(clk, out + 1024 * 256 + 65536, sizeof(clk), hipMemcpyDeviceToHost);
    printf("%-52s %8.3f ms %7.1f TFLOP/s (%.2f us, %.0f cycles per chunk, clock %.2f GHz)\n", name, best, flops / best * 1e-9, best * 1e3 / chunks,
           (double)clk[0] / chunks, (double)clk[0] / clk[1] * 0.1);
    fflush(stdout);
}

int main() {
    {
        float* out; hipMalloc(&out, 1024 * 512 * sizeof(float));
        const int ldx = 20000, nrows = 65536;
        float *x, *a;
        hipMalloc(&x, (size_t)ldx * nrows * sizeof(float));
        hipMemset(x, 0, (size_t)ldx * nrows * sizeof(float));
        hipMalloc(&a, (size_t)512 * 8448 * sizeof(float));
        hipMemset(a, 0, (size_t)512 * 8448 * sizeof(float));
        float* a2; hipMalloc(&a2, (size_t)256 * 128 * 1024 * sizeof(float)); hipMemset(a2, 0, (size_t)256 * 128 * 1024 * sizeof(float));
        run_chunk<1>("chunk loop: LDS A only", 256, out, x, a, ldx, nrows);
        run_chunk<2>("  + barrier per chunk", 256, out, x, a, ldx, nrows);
        run_chunk<3>("  + slab stores to LDS", 256, out, x, a, ldx, nrows);
        run_chunk<4>("  + 32 gathered x loads per chunk", 256, out, x, a, ldx, nrows);
        run_chunk<5>("  + 4 slab loads per chunk", 256, out, x, a, ldx, nrows);
        run_chunk<5>("  the same, 237 blocks", 237, out, x, a, ldx, nrows);
        run_chunk<5>("  x + slab loads, x rows: 64 (L2-resident)", 256, out, x, a, ldx, 64);
        run_chunk<5>("  x + slab loads, x rows: 4096", 256, out, x, a, ldx, 4096);
        run_chunk<5>("  slab row stride 8448 floats", 256, out, x, a, ldx, nrows, 8448);
        run_chunk<5>("  slab row stride 1024 floats", 256, out, x, a, ldx, nrows, 1024);
        run_chunk<9>("  slab loads, every workgroup its own rows", 256, out, x, a2, ldx, nrows, 1024);
        run_chunk<10>("  slab loads, one row block per XCD", 256, out, x, a2, ldx, nrows, 1024);
        for (int st = 0; st < 2; ++st) {
            const size_t lds = 2 * 128 * 68 * sizeof(float) + 8192 * 8;
            auto k = st ? chunkloop<5, true> : chunkloop<4, true>;
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, out, x, a, 128, ldx, nrows, 8192);
            hipDeviceSynchronize();
            static unsigned long long h[2048 * 4];
            hipMemcpy(h, out + 1024 * 256, sizeof(h), hipMemcpyDeviceToHost);
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int w = 0; w < 2048; ++w) { s0 += h[4 * w]; s1 += h[4 * w + 1]; s2 += h[4 * w + 2]; s3 += h[4 * w + 3]; }
            printf("  timed STEP %d: per wave and chunk, cycles: vm wait %.0f, lds store+wait %.0f, barrier %.0f; kernel %.0f per chunk\n",
                   st ? 5 : 4, s0 / 2048 / 128, s1 / 2048 / 128, s2 / 2048 / 128, s3 / 2048 / 128);
        }
        run_chunk<11>("  slab loads + stores, no x loads", 256, out, x, a, ldx, nrows);
        run_chunk<13>("  x + slab loads, slabs first in the chunk", 256, out, x, a, ldx, nrows);
        run_chunk<16>("  x + slab loads, slab stored a chunk after its loads", 256, out, x, a, ldx, nrows);
        run_chunk<17>("  slab loads + HALF of the x loads (16 per chunk)", 256, out, x, a, ldx, nrows);
        run_chunk<14>("  x + slab loads, slab consumed by VALU only", 256, out, x, a, ldx, nrows);
        run_chunk<15>("  x + slab loads, vmcnt(0) before the stores", 256, out, x, a, ldx, nrows);
        run_chunk<12>("  x + slab loads, slab stored one chunk later", 256, out, x, a, ldx, nrows);
        run_chunk<6>("  slab loads issued, result not stored", 256, out, x, a, ldx, nrows);
        run_chunk<7>("  slab loads, 3 different row blocks", 256, out, x, a, ldx, nrows);
        run_chunk<8>("  slab loads in groups 4-7", 256, out, x, a, ldx, nrows);
        run_chunk<3>("  (again) LDS A + barrier + slab stores", 256, out, x, a, ldx, nrows);
        hipFree(x); hipFree(a); hipFree(out);
    }
    float* out; hipMalloc(&out, 1024 * 512 * sizeof(float));
    run<4, 0, 0>("regs, 4 acc, acc-major", 512, 256, out);
    run<4, 1, 0>("regs, 4 acc, interleaved", 512, 256, out);
    run<4, 0, 0>("regs, 4 acc, acc-major, 1 wave/SIMD", 256, 256, out);
    run<4, 1, 0>("regs, 4 acc, interleaved, 1 wave/SIMD", 256, 256, out);
    run<1, 0, 0>("regs, 1 acc, 2 waves/SIMD", 512, 256, out);
    run<4, 0, 1>("LDS A, 4 acc, acc-major", 512, 256, out);
    run<4, 1, 1>("LDS A, 4 acc, interleaved", 512, 256, out);
    run<3, 1, 1>("LDS A, 3 acc, interleaved", 512, 256, out);
    run<4, 0, 1>("LDS A, 4 acc, acc-major, 1 wave/SIMD", 256, 256, out);
    run<4, 1, 1>("LDS A, 4 acc, interleaved, 1 wave/SIMD", 256, 256, out);
    run<4, 1, 1>("LDS A, 4 acc, interleaved, 237 blocks", 512, 237, out);
    hipFree(out);
    return 0;
}
