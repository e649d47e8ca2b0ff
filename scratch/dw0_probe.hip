// Stand-alone probe of ov_dw0_kernel: event timing + per-wave phase timestamps.
//   hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -DOV_DW0_PROBE -Iinclude -Ionto_vae_amd/csrc scratch/dw0_probe.hip -o scratch/dw0_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#include "ontovae_gemm.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    const int L = 360, G = 20000, B = 128, NW = 8;
    float *dh, *x, *w, *m, *v; uint64_t* probe;
    CK(hipMalloc(&dh, (size_t)L * B * 4)); CK(hipMalloc(&x, (size_t)B * G * 4));
    CK(hipMalloc(&w, (size_t)L * G * 4)); CK(hipMalloc(&m, (size_t)L * G * 4)); CK(hipMalloc(&v, (size_t)L * G * 4));
    std::vector<float> h((size_t)L * G);
    for (auto& q : h) q = (rand() % 1000) * 1e-3f;
    CK(hipMemcpy(w, h.data(), (size_t)L * G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v, h.data(), (size_t)L * G * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(m, h.data(), (size_t)L * G * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, h.data(), (size_t)B * G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, h.data(), (size_t)L * B * 4, hipMemcpyHostToDevice));
    OvGemm g{};
    g.M = L; g.N = G; g.K = B; g.A = dh; g.lda = B; g.B = x; g.ldb = G; g.bgather = nullptr; g.C = w; g.Pm = m; g.Pv = v; g.ldc = G;
    g.adam = OvAdam{1.f - 1e-6f, 0.1f, 0.999f, 0.001f, 1e-4f, 0.5f, 1e-8f};
    const int njobs = (L + 127) / 128, gxn = (G + 32 * NW - 1) / (32 * NW), blocks = (gxn + 7) / 8 * 8 * njobs;
    const size_t lds = 128 * 132 * 4 + 512;
    CK(hipMalloc(&probe, (size_t)blocks * NW * 16 * 8)); CK(hipMemset(probe, 0, (size_t)blocks * NW * 16 * 8));
    auto kern = ov_dw0_kernel<NW, OV_EPI_ADAMW>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), lds, 0, g, njobs);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), lds, 0, g, njobs);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("blocks %d, avg %.2f us per launch (back-to-back)\n", blocks, ms * 1000 / 20);
    g.bias = reinterpret_cast<const float*>(probe);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), lds, 0, g, njobs);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> t((size_t)blocks * NW * 16);
    CK(hipMemcpy(t.data(), probe, t.size() * 8, hipMemcpyDeviceToHost));
    const char* names[16] = {"start", "lds_filled", "barrier", "x_frag", "mfma0", "adam0", "mfma1", "adam1", "mfma2", "adam2", "mfma3", "adam3", "", "", "", "end"};
    int prev = 0;
    for (int q = 1; q < 16; ++q) {
        double s = 0, mn = 1e30, mx = 0, tot = 0; int n = 0;
        for (size_t wv = 0; wv < t.size() / 16; ++wv) if (t[wv * 16] && t[wv * 16 + q] && t[wv * 16 + prev]) {
            double d = (double)(t[wv * 16 + q] - t[wv * 16 + prev]); s += d; mn = std::min(mn, d); mx = std::max(mx, d); ++n;
            tot += (double)(t[wv * 16 + q] - t[wv * 16]);
        }
        if (n) { printf("%-12s n=%5d dt mean %8.0f min %8.0f max %8.0f | since start %8.0f\n", names[q], n, s / n, mn, mx, tot / n); prev = q; }
    }
    return 0;
}
