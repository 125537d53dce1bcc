// The pointwise MLP of a ConvNeXt block in one launch: out = x + s2 * (W2 . gelu(s1 * (W1 . t) + b1)) + b2 at every pixel
// (streamingflow/layers/convolutions.py:338-345: pwconv1 -> GELU -> pwconv2 -> gamma -> residual), C = 64 channels, 256 hidden.
//
// As two launches the 256-channel hidden tensor makes a round trip through HBM (1 KB written and 1 KB read per pixel against the
// 768 B of t, x and out): on the 8.96 M pixels of the headline's head that is 18.4 GB per forward and the 64 -> 256 launch ran at
// 0.31 of the matrix peak behind its stores (profiles/r05_m_prof_dump.txt).  Here the hidden values never leave the registers:
//
//   * a wave owns 64 consecutive pixels (16 KB of t, contiguous: the layout is [pixel][channel]) and keeps them as the B operands of
//     v_mfma_f32_16x16x4_f32 for the whole kernel (64 registers), plus the 64 x 64 output accumulators (64 registers);
//   * per 16 hidden channels: 64 MFMAs give D1[16 hidden][64 px]; a lane then holds hidden rows 4g..4g+3 of pixel j — which is
//     exactly the B operand of K steps 0..3 of the second product if K step i is DEFINED as hidden row 4g + i (a sum may run over
//     its terms in any order as long as both operands agree), so the A operand of that step is W2[cout][16h + 4g + i]: one 16-byte
//     load per lane from the row-major packed weights.  No LDS, no cross-lane traffic, no barrier: waves are independent;
//   * W1 / W2 fragments (128 KB per wave and tile, L2-resident, the same for every wave) are requested one 16-row block ahead.
//
// GELU is vector work on the port the MFMAs issue from (sf_math.h): one reciprocal, one exp2 and about nine full-rate instructions per value
// (packed two values per instruction where the ISA has a packed form).
// gfx950 only.
#include <hip/hip_runtime.h>

#include "sf_device.h"
#include "sf_math.h"

namespace sf {

struct MlpLaunch {
  const float *t, *x;
  float* out;
  const float *w1, *s1, *b1, *w2, *s2, *b2;   // w1 [256][64], w2 [64][256] row-major (the packed 1x1 layout), scale / bias per row
  long P;                                      // pixels
};

constexpr int MLP_C = 64, MLP_HID = 256, MLP_NB = 4, MLP_WAVES = 4, MLP_PX = 16 * MLP_NB;

// gelu(v) = max(v, 0) - |v| * (0.5 * erfc(|v| / sqrt 2)) on a pair of values, erfc by Abramowitz & Stegun 7.1.26 (the polynomial of spm_erf with the
// factor 0.5 folded into its coefficients): the same approximation as the two-launch path, rounding aside
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 mlp_gelu2(const f32x2 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  // two values per instruction where the ISA has a packed form (v_pk_fma_f32 / v_pk_mul_f32); |v| rides as a source modifier
  const f32x2 t = {__builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(v.x), 1.f)),
                   __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(v.y), 1.f))};
  f32x2 p = __builtin_elementwise_fma(f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, t, f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
  const f32x2 u = v * 0.84932180028801904272f;     // sqrt(log2(e) / 2): exp(-v^2 / 2) = exp2(-u^2)
  const f32x2 w = -u * u;
  const f32x2 e = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const f32x2 r = (p * t) * e;
  return f32x2{fmaf(-fabsf(v.x), r.x, fmaxf(v.x, 0.f)), fmaf(-fabsf(v.y), r.y, fmaxf(v.y, 0.f))};
#else
  return v;
#endif
}

__global__ __launch_bounds__(64 * MLP_WAVES, 2) void convnext_mlp_kernel(const MlpLaunch L) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const long px0 = ((long)blockIdx.x * MLP_WAVES + wave) * MLP_PX;
  if (px0 >= L.P) return;
  const long left = L.P - px0;
  const int npx = left < MLP_PX ? (int)left : MLP_PX;
  auto make_rsrc = [](const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (int)bytes, 0x00020000);
  };
  // rows past the last pixel read zeros and drop their stores (buffer bounds)
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(L.t + px0 * MLP_C, (unsigned)npx * MLP_C * 4);
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(L.x + px0 * MLP_C, (unsigned)npx * MLP_C * 4);
  const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(L.out + px0 * MLP_C, (unsigned)npx * MLP_C * 4);
  const __amdgpu_buffer_rsrc_t rs_w1 = make_rsrc(L.w1, MLP_HID * MLP_C * 4);
  const __amdgpu_buffer_rsrc_t rs_w2 = make_rsrc(L.w2, MLP_C * MLP_HID * 4);
  // a layer packed without a scale (or bias) carries a null pointer: an empty buffer reads zeros, the scale is then set to one
  const bool one1 = L.s1 == nullptr, one2 = L.s2 == nullptr;
  const __amdgpu_buffer_rsrc_t rs_s1 = make_rsrc(one1 ? L.w1 : L.s1, one1 ? 0 : MLP_HID * 4);
  const __amdgpu_buffer_rsrc_t rs_b1 = make_rsrc(L.b1 ? L.b1 : L.w1, L.b1 ? MLP_HID * 4 : 0);
  const __amdgpu_buffer_rsrc_t rs_s2 = make_rsrc(one2 ? L.w2 : L.s2, one2 ? 0 : MLP_C * 4);
  const __amdgpu_buffer_rsrc_t rs_b2 = make_rsrc(L.b2 ? L.b2 : L.w2, L.b2 ? MLP_C * 4 : 0);
  const f32x4 ones = f32x4{1.f, 1.f, 1.f, 1.f};
  auto ld = [](const __amdgpu_buffer_rsrc_t r, const int voff, const int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  };

  // B operands of the first product: T[nb][q][i] = t[pixel 16 nb + j][channel 16 q + 4 g + i]
  const int v_px = j * (MLP_C * 4) + g * 16;             // row j of a 16-row block of 256-byte rows, 16-byte column g
  f32x4 T[MLP_NB][4];
#pragma unroll
  for (int nb = 0; nb < MLP_NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) T[nb][q] = ld(rs_t, v_px + q * 64, nb * 16 * MLP_C * 4);

  f32x4 O[4][MLP_NB];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nb = 0; nb < MLP_NB; ++nb) O[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int v_w2 = j * (MLP_HID * 4) + g * 16;            // W2 row j of a 16-row block, hidden columns 4 g .. 4 g + 3 of a 16-column block
  f32x4 A1[4], A2[4], S1, B1;
#pragma unroll
  for (int q = 0; q < 4; ++q) A1[q] = ld(rs_w1, v_px + q * 64, 0);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) A2[mt] = ld(rs_w2, v_w2, mt * 16 * MLP_HID * 4);
  S1 = ld(rs_s1, g * 16, 0);
  B1 = ld(rs_b1, g * 16, 0);

  for (int h = 0; h < MLP_HID / 16; ++h) {
    // ---- D1 = W1[16 h .. 16 h + 15][:] . T : K step (q, i) is channel 16 q + 4 g + i on both operands
    f32x4 D[MLP_NB];
#pragma unroll
    for (int nb = 0; nb < MLP_NB; ++nb) D[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int nb = 0; nb < MLP_NB; ++nb) D[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[q][i], T[nb][q][i], D[nb], 0, 0, 0);
    const f32x4 s1 = one1 ? ones : S1, b1 = B1;
    const int hn = h + 1 < MLP_HID / 16 ? h + 1 : h;      // the last block re-reads itself (cached, unused)
#pragma unroll
    for (int q = 0; q < 4; ++q) A1[q] = ld(rs_w1, v_px + q * 64, hn * 16 * MLP_C * 4);
    S1 = ld(rs_s1, g * 16, hn * 64);
    B1 = ld(rs_b1, g * 16, hn * 64);
    // ---- hidden activations: the lane's rows 16 h + 4 g + i of pixel 16 nb + j
#pragma unroll
    for (int nb = 0; nb < MLP_NB; ++nb)
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const f32x2 a = mlp_gelu2(__builtin_elementwise_fma(f32x2{D[nb][i], D[nb][i + 1]}, f32x2{s1[i], s1[i + 1]}, f32x2{b1[i], b1[i + 1]}));
        D[nb][i] = a.x;
        D[nb][i + 1] = a.y;
      }
    // ---- O += W2[:][16 h + 4 g + i] . H : K step i is hidden row 16 h + 4 g + i on both operands
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nb = 0; nb < MLP_NB; ++nb) O[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[mt][i], D[nb][i], O[mt][nb], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) A2[mt] = ld(rs_w2, v_w2, mt * 16 * MLP_HID * 4 + hn * 64);
  }

  // ---- out = x + s2 * O + b2: the lane holds channels 16 mt + 4 g + (0..3) of pixel 16 nb + j
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const f32x4 s2 = one2 ? ones : ld(rs_s2, g * 16, mt * 64);
    const f32x4 b2 = ld(rs_b2, g * 16, mt * 64);
#pragma unroll
    for (int nb = 0; nb < MLP_NB; ++nb) {
      const f32x4 xv = ld(rs_x, v_px + mt * 64, nb * 16 * MLP_C * 4);
      f32x4 y;
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = fmaf(O[mt][nb][i], s2[i], b2[i]) + xv[i];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, y), rs_o,
                                             v_px + mt * 64, nb * 16 * MLP_C * 4, 0);
    }
  }
#endif
}

// P pixels of [pixel][64] tensors; the weights in the packed 1x1 layout of sf_conv_w ([cout_pad][cin_pad] row-major)
hipError_t launch_convnext_mlp(const float* t, const float* x, float* out, const float* w1, const float* s1, const float* b1, const float* w2,
                               const float* s2, const float* b2, long P, hipStream_t stream) {
  MlpLaunch L{t, x, out, w1, s1, b1, w2, s2, b2, P};
  const long blocks = (P + MLP_PX * MLP_WAVES - 1) / (MLP_PX * MLP_WAVES);
  if (blocks <= 0 || blocks > 0x7fffffffL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(convnext_mlp_kernel, dim3((unsigned)blocks), dim3(64 * MLP_WAVES), 0, stream, L);
  return hipGetLastError();
}

}  // namespace sf
