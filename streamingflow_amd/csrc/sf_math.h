// Small device helpers shared by the epilogues (activations, 16-byte loads / stores).  gfx950 only.
#pragma once
#include "sf_device.h"

namespace sf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// erf for the GELU epilogues: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 (+ fp32 rounding) on the hardware reciprocal and exp2 —
// 15 vector instructions without a branch where erff() is two divergent branches of ~35.  Vector instructions cost matrix time on this
// chip: the ConvNeXt 64 -> 256 layer (K = 64, 256 GELUs per pixel) ran at 0.31 of the MFMA peak bound by its epilogue (profiles/r05_m_*).
__device__ __forceinline__ float spm_erf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(fmaf(-p * t, e, 1.f), x);
#else
  return erff(x);
#endif
}
__device__ __forceinline__ float spm_act(float v, int act) {
  switch (act) {
    case ACT_LRELU:   return v > 0.f ? v : 0.1f * v;
    case ACT_RELU:    return v > 0.f ? v : 0.f;
    case ACT_TANH:    return tanhf(v);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case ACT_GELU:    return 0.5f * v * (1.f + spm_erf(v * 0.70710678118654752440f));
    default:          return v;
  }
}
__device__ __forceinline__ float4 spm_act4(float4 v, int act) {
  return make_float4(spm_act(v.x, act), spm_act(v.y, act), spm_act(v.z, act), spm_act(v.w, act));
}
__device__ __forceinline__ float spm_gelu(float v) { return 0.5f * v * (1.f + spm_erf(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float spm_softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }
__device__ __forceinline__ float4 spm_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void spm_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 spm_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), the result in every lane of the row: quad swaps, half-row
// mirror, row mirror — four VALU instructions, no LDS round trips (the ds_bpermute butterfly they replace adds the same
// operands in the same order: bitwise identical).
__device__ __forceinline__ float spm_row16_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1, 0, 3, 2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2, 3, 0, 1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
#endif
  return v;
}

// ---- in-kernel Gaussian noise (throughput mode of infer_state: no eps tensor) -------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter = (pixel, channel, draw, offset low), key = (seed low, seed high ^ offset
// high): the noise of an element depends only on (seed, offset, draw, pixel, channel) — not on the kernel, the tiling or the
// batch the pixel sits in — so one captured graph serves every call (the host bumps `offset` in a 16-byte device record).
__device__ __forceinline__ void spm_philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, const unsigned k0, const unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
  c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
}
// two independent N(0, 1) draws for channels ch, ch + 1 of pixel gp (Box-Muller on two of the four 32-bit words)
__device__ __forceinline__ float2 spm_philox_normal2(const unsigned long long* state, const int draw, const unsigned gp, const unsigned ch) {
  const unsigned long long seed = state[0], offset = state[1];
  unsigned c0 = gp, c1 = ch, c2 = (unsigned)draw, c3 = (unsigned)offset;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32) ^ (unsigned)(offset >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    spm_philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)c0 + 1.0f) * 2.3283064365386963e-10f;        // (0, 1]
  const float u2 = (float)c1 * 2.3283064365386963e-10f;                 // [0, 1]
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.28318530717958647692f * u2, &sn, &cs);
  return make_float2(r * cs, r * sn);
}

}  // namespace sf
