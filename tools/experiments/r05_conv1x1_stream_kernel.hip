// Large 1x1 convolutions (out[p][:] = act(scale * (W . in[p][:]) + bias) (+ add[p][:])) without LDS: a wave owns 64 consecutive pixels
// and ALL 64 / 128 output channels, keeps the 64 x (64 | 128) accumulator tile in registers and streams K 16 channels at a time:
//
//   * B operands (pixels) are 16-byte loads straight from the [pixel][channel] tensor — lane (j, g) takes channels k0 + 4g .. + 3 of
//     pixel 16 nb + j, i.e. K step i of v_mfma_f32_16x16x4_f32 is channel k0 + 4g + i — requested one 16-channel block ahead;
//   * A operands (weights, [cout][cin_pad] row-major as packed) are 16-byte loads of the same columns of row 16 mt + j, L2-resident and
//     the same for every wave; the fragment of row block mt is refilled for the NEXT K block right after its last use in this one.
//
// No staging through LDS, no barrier, no workgroup-level pipeline to fill and drain: the two waves of a SIMD are independent and cover
// each other's load latency.  This is the form the fused ConvNeXt MLP (convnext_mlp.hip) showed to run at ~0.9 of the MFMA rate on its
// GEMM part; the LDS-DMA tiles of conv_igemm.hip reach 0.79 on the ASPP projection (K = 512) and are bound by HBM round trips on the
// thin layers.  api.hip sends a 1x1 layer here when it is plain (one input, AFFINE epilogue, no gate / SE scale / second output) and
// has at least 65536 pixels.  gfx950 only.
#include <hip/hip_runtime.h>

#include "sf_device.h"
#include "sf_math.h"

namespace sf {

struct PwLaunch {
  const float *in, *w, *scale, *bias, *add;
  float* out;
  long P;                       // pixels
  int HW;                       // pixels per image (bias_per_img)
  int cin, in_cs, w_pitch;      // channels read (multiple of 32), floats between pixels of `in`, floats between rows of `w`
  int add_cs, out_cs, out_co;
  int act, act_last, bias_per_img;
};

constexpr int PW_NB = 4, PW_WAVES = 4, PW_PX = 16 * PW_NB;

template <int MT>
__global__ __launch_bounds__(64 * PW_WAVES, 2) void conv1x1_stream_kernel(const PwLaunch L) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const long px0 = ((long)blockIdx.x * PW_WAVES + wave) * PW_PX;
  if (px0 >= L.P) return;
  const long left = L.P - px0;
  const int npx = left < PW_PX ? (int)left : PW_PX;
  auto make_rsrc = [](const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (int)bytes, 0x00020000);
  };
  const int cin = L.cin, in_cs = L.in_cs, wp = L.w_pitch;
  // rows past the last pixel read zeros and drop their stores (buffer bounds)
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(L.in + px0 * in_cs, (unsigned)npx * in_cs * 4);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(L.w, (unsigned)(16 * MT) * wp * 4);
  auto ld = [](const __amdgpu_buffer_rsrc_t r, const int voff, const int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  };
  const int v_in = j * in_cs * 4 + g * 16;      // pixel j of a 16-pixel block, channels 4g .. 4g + 3 of a 16-channel block
  const int v_w = j * wp * 4 + g * 16;          // row j of a 16-row block, the same columns
  const int s_nb = 16 * in_cs * 4, s_mt = 16 * wp * 4;

  f32x4 O[MT][PW_NB];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nb = 0; nb < PW_NB; ++nb) O[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 A[MT], B0[PW_NB], B1[PW_NB];
#pragma unroll
  for (int nb = 0; nb < PW_NB; ++nb) B0[nb] = ld(rs_in, v_in, nb * s_nb);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) A[mt] = ld(rs_w, v_w, mt * s_mt);

  // one 16-channel block: pixels of the block after it requested first, then the products row block by row block, each row block's
  // weights refilled for the next K block as soon as its last product has been issued (the last block re-reads itself: cached, unused)
  auto block = [&](f32x4 (&Bc)[PW_NB], f32x4 (&Bn)[PW_NB], const int k0) {
    const int kn = k0 + 16 < cin ? k0 + 16 : k0;
#pragma unroll
    for (int nb = 0; nb < PW_NB; ++nb) Bn[nb] = ld(rs_in, v_in, nb * s_nb + kn * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int nb = 0; nb < PW_NB; ++nb) O[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt][i], Bc[nb][i], O[mt][nb], 0, 0, 0);
      A[mt] = ld(rs_w, v_w, mt * s_mt + kn * 4);
    }
  };
  for (int k0 = 0; k0 < cin; k0 += 32) {
    block(B0, B1, k0);
    block(B1, B0, k0 + 16);
  }

  // ---- epilogue: the lane holds channels 16 mt + 4g + (0..3) of pixel 16 nb + j.  Branch-free per tile: absent operands are empty
  // buffers (they read zeros), the one wave-uniform switch on the activation encloses the whole tile loop
  const int cout = 16 * MT;
  const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(L.out + px0 * L.out_cs + L.out_co, (unsigned)npx * L.out_cs * 4 - (unsigned)L.out_co * 4);
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(L.add ? L.add + px0 * L.add_cs : L.w, L.add ? (unsigned)npx * L.add_cs * 4 : 0);
  const int nimg = L.bias_per_img ? (int)((L.P + L.HW - 1) / L.HW) : 1;
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(L.bias ? L.bias : L.w, L.bias ? (unsigned)nimg * cout * 4 : 0);
  const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(L.scale ? L.scale : L.w, L.scale ? (unsigned)cout * 4 : 0);
  const bool one = L.scale == nullptr, act_last = L.act_last != 0;
  int boff[PW_NB];          // byte offset of the pixel's bias row
#pragma unroll
  for (int nb = 0; nb < PW_NB; ++nb) {
    const long p = px0 + nb * 16 + j;
    boff[nb] = L.bias_per_img ? (int)((p < L.P ? p : L.P - 1) / L.HW) * cout * 4 : 0;
  }
  auto finish = [&](auto act_fn) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int c4 = (mt * 16 + 4 * g) * 4;
      const f32x4 sc = one ? f32x4{1.f, 1.f, 1.f, 1.f} : ld(rs_s, c4, 0);
#pragma unroll
      for (int nb = 0; nb < PW_NB; ++nb) {
        const f32x4 bi = ld(rs_b, boff[nb] + c4, 0);
        const f32x4 ad = ld(rs_a, (nb * 16 + j) * L.add_cs * 4 + c4, 0);
        f32x4 y;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pre = act_last ? ad[i] : 0.f;         // ResNet BasicBlock: the activation after the residual
          y[i] = act_fn(O[mt][nb][i] * sc[i] + bi[i] + pre) + (ad[i] - pre);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, y), rs_o,
                                               (nb * 16 + j) * L.out_cs * 4 + c4, 0, 0);
      }
    }
  };
  switch (L.act) {
    case ACT_LRELU: finish([](float v) { return v > 0.f ? v : 0.1f * v; }); break;
    case ACT_RELU:  finish([](float v) { return fmaxf(v, 0.f); }); break;
    case ACT_GELU:  finish([](float v) { return spm_gelu(v); }); break;
    default:        finish([](float v) { return v; }); break;      // ACT_NONE (api.hip keeps tanh / sigmoid layers on the LDS-DMA tiles)
  }
#endif
}

// cout = 64 or 128 (== cout_pad); cin a multiple of 32
hipError_t launch_conv1x1_stream(const PwLaunch& L, int cout, hipStream_t stream) {
  const long blocks = (L.P + PW_PX * PW_WAVES - 1) / (PW_PX * PW_WAVES);
  if (blocks <= 0 || blocks > 0x7fffffffL || (L.cin & 31)) return hipErrorInvalidValue;
  if (cout == 128) hipLaunchKernelGGL(conv1x1_stream_kernel<8>, dim3((unsigned)blocks), dim3(64 * PW_WAVES), 0, stream, L);
  else if (cout == 64) hipLaunchKernelGGL(conv1x1_stream_kernel<4>, dim3((unsigned)blocks), dim3(64 * PW_WAVES), 0, stream, L);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace sf
