// Does the L2 keep a "column block" of a feature-major matrix?  Every XCD gathers 128-B lines  base + r * stride + xcd * 128  for random
// rows r of R rows, again and again (8 lanes per line, 16 B per lane, as the gene-layer kernels do), for several row strides.
// 20000 rows x 128 B = 2.56 MB per XCD: fits a 4 MB L2 if the lines spread over its sets and channels.
//   hipcc --offload-arch=gfx950 -O3 -o l2_stride_probe l2_stride_probe.hip && ./l2_stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gather(const float* __restrict__ base, int64_t stride_f, int R, int iters, float* out) {
    const int xcd = blockIdx.x & 7, lane8 = threadIdx.x & 7, grp = (blockIdx.x >> 3) * 32 + (threadIdx.x >> 3);
    unsigned s = 1234567u + 7919u * grp;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* col = base + xcd * 32 + lane8 * 4;
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s = s * 1664525u + 1013904223u;
            const int r = (int)((s >> 8) % (unsigned)R);
            v[i] = *reinterpret_cast<const f32x4*>(col + (int64_t)r * stride_f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
    }
    if (acc.x == 12345.f) out[0] = acc.y;
}
int main() {
    const int R = 20000, iters = 400;
    const int64_t strides[] = {1024, 1024 + 32, 1280, 2048, 2048 + 32, 8448, 8448 + 32, 8192, 8192 + 32};
    float *buf, *out;
    const size_t bytes = (size_t)R * (8448 + 64) * 4 + 4096;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int64_t st : strides) {
        const int grid = 8 * 94;                      // the backward gene kernel's grid: 94 workgroups per XCD, 3 per CU
        gather<<<grid, 256>>>(buf, st, R, iters, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        gather<<<grid, 256>>>(buf, st, R, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double lines = (double)grid * 32 * 8 * iters;
        printf("row stride %6lld floats (%6lld B): %8.1f us, %6.2f G lines/s, %6.2f TB/s of 128-B lines\n", (long long)st, (long long)st * 4,
               ms * 1e3, lines / ms / 1e6, lines * 128 / ms / 1e9);
    }
    if (hipGetLastError() != hipSuccess) { printf("error\n"); return 1; }
    return 0;
}
