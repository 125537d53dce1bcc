// Small device helpers shared by the epilogues (activations, 16-byte loads / stores).  gfx950 only.
#pragma once
#include "sf_device.h"

namespace sf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float spm_act(float v, int act) {
  switch (act) {
    case ACT_LRELU:   return v > 0.f ? v : 0.1f * v;
    case ACT_RELU:    return v > 0.f ? v : 0.f;
    case ACT_TANH:    return tanhf(v);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case ACT_GELU:    return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    default:          return v;
  }
}
__device__ __forceinline__ float4 spm_act4(float4 v, int act) {
  return make_float4(spm_act(v.x, act), spm_act(v.y, act), spm_act(v.z, act), spm_act(v.w, act));
}
__device__ __forceinline__ float spm_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float spm_softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }
__device__ __forceinline__ float4 spm_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void spm_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 spm_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

}  // namespace sf
