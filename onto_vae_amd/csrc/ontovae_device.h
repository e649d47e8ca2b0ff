// Device-side helpers shared by the OntoVAE kernels (gfx950 only, wave64).
#ifndef ONTOVAE_DEVICE_H
#define ONTOVAE_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OV_WAVE 64

// Measurement switches (skip a kernel's main loop, select an experimental variant ...) are read from the upper bits of
// ctx->flags only in builds with -DOV_MEASUREMENT_SWITCHES; the shipped library compiles them out (every OV_DBG(...) is the
// constant 0, the branches it guards disappear) and ov_check_ctx refuses flag bits it does not know.
#ifdef OV_MEASUREMENT_SWITCHES
#define OV_DBG(expr) (expr)
#else
#define OV_DBG(expr) 0
#endif

__device__ __forceinline__ int ov_lane() { return threadIdx.x & 63; }
// wave index inside the block as a scalar (SGPR) value, so that everything indexed by it is loaded
// through the scalar path
__device__ __forceinline__ int ov_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Wave-wide sum on the DPP path (VALU cross-lane moves, no LDS traffic): quads, half rows, rows of
// 16, then row_bcast15 / row_bcast31 carry the row totals upwards; lane 63 holds the total, which
// v_readlane broadcasts.  Every lane of the wave must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ov_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float ov_wave_sum(float v) {
    v += ov_dpp<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v += ov_dpp<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v += ov_dpp<0x141, 0xF>(v);   // row_half_mirror
    v += ov_dpp<0x140, 0xF>(v);   // row_mirror: every lane holds the sum of its row of 16
    v += ov_dpp<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
    v += ov_dpp<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ double ov_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- AdamW, arithmetic order of torch.optim.AdamW's single-tensor path --------------------------
struct OvAdam {
    float decay_mul, one_m_beta1, beta2, one_m_beta2, step_size, bc2_sqrt, eps;
};

__device__ __forceinline__ void ov_adamw(float& w, float& m, float& v, float g, const OvAdam& h, bool clamp0) {
    w = w * h.decay_mul;                          // param.mul_(1 - lr * wd)
    m = m + (g - m) * h.one_m_beta1;              // exp_avg.lerp_(grad, 1 - beta1)
    v = v * h.beta2 + (h.one_m_beta2 * g) * g;    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    w = w + (-h.step_size * m) / denom;           // param.addcdiv_(exp_avg, denom, value=-step_size)
    if (clamp0) w = fmaxf(w, 0.0f);               // vae_model.py:175-176
}

// same update with the hardware reciprocal / square root (1 ulp each) in place of the IEEE sequences
__device__ __forceinline__ void ov_adamw_fast(float& w, float& m, float& v, float g, const OvAdam& h, bool clamp0) {
    w = w * h.decay_mul;
    m = m + (g - m) * h.one_m_beta1;
    v = v * h.beta2 + (h.one_m_beta2 * g) * g;
    const float denom = __builtin_amdgcn_sqrtf(v) * __builtin_amdgcn_rcpf(h.bc2_sqrt) + h.eps;
    w = w + (-h.step_size * m) * __builtin_amdgcn_rcpf(denom);
    if (clamp0) w = fmaxf(w, 0.0f);
}

// ---- Philox4x32-10 counter RNG for dropout masks and internally drawn eps ------------------------
struct OvPhilox {
    uint32_t c[4];
};

__device__ __forceinline__ OvPhilox ov_philox(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32) ^ (stream << 24), c2 = (uint32_t)offset,
             c3 = (uint32_t)(offset >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    OvPhilox o;
    o.c[0] = c0; o.c[1] = c1; o.c[2] = c2; o.c[3] = c3;
    return o;
}

__device__ __forceinline__ float ov_u01(uint32_t u) { return ((u >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

// stream ids
#define OV_RNG_DROP 1u
#define OV_RNG_ZDROP_MU 2u
#define OV_RNG_ZDROP_LV 3u
#define OV_RNG_EPS 4u
#define OV_RNG_EPS2 5u

// keep-scale of inverted dropout for element idx: 0 or 1/(1-p)
__device__ __forceinline__ float ov_dropout_scale(float p, uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx) {
    if (p <= 0.0f) return 1.0f;
    if (p >= 1.0f) return 0.0f;
    const OvPhilox r = ov_philox(seed, offset, stream, idx);
    return ov_u01(r.c[0]) >= p ? 1.0f / (1.0f - p) : 0.0f;
}

__device__ __forceinline__ float ov_randn(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx) {
    const OvPhilox r = ov_philox(seed, offset, stream, idx);
    const float u1 = ov_u01(r.c[0]), u2 = ov_u01(r.c[1]);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

#endif
