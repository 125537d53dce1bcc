// Weight packing on the device (SURVEY.md §8b "sf_pack_*"): reference-format parameters (Conv2d / ConvTranspose2d OIHW fp32,
// BatchNorm buffers) -> the packed layout of include/sfnative.h.  Load-time work: one thread per packed element.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "../../include/sfnative.h"

namespace sf {

__host__ __device__ inline int pack_round_up(int v, int m) { return (v + m - 1) / m * m; }

// packed row -> source output channel (or -1: zero row).  Interleaved (the last p_model conv): row 16T + 4g + r holds
// loc channel 8T + 2g + r for r < 2 and raw-scale channel 8T + 2g + r - 2 for r >= 2
__device__ __forceinline__ int pack_src_row(int row, int cout, int interleave) {
  if (!interleave) return row < cout ? row : -1;
  const int Ch = cout >> 1;
  const int T = row >> 4, g = (row & 15) >> 2, r = row & 3;
  const int c = 8 * T + 2 * g + (r & 1);
  if (c >= Ch) return -1;
  return r < 2 ? c : Ch + c;
}

__global__ void pack_weights_kernel(const float* __restrict__ w, int cout, int cin, int kh, int kw, int cout_pad, int cin_pad, int flags,
                                    float* __restrict__ out) {
  const long total = (long)cout_pad * kh * kw * cin_pad;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % cin_pad);
  long r = idx / cin_pad;
  const int kx = (int)(r % kw);
  r /= kw;
  const int ky = (int)(r % kh);
  const int row = (int)(r / kh);
  const int co = pack_src_row(row, cout, flags & SF_PACK_INTERLEAVE);
  float v = 0.f;
  if (co >= 0 && c < cin) {
    if (flags & SF_PACK_TRANSPOSED) {          // ConvTranspose2d [cin][cout][kh][kw], k3 s1 p1 == conv with the kernel flipped
      v = w[(((size_t)c * cout + co) * kh + (kh - 1 - ky)) * kw + (kw - 1 - kx)];
    } else if (flags & SF_PACK_FOLD_DUP) {     // the layer reads cat[s, s]: W[:, :cin] + W[:, cin:] applied to s once
      const size_t a = (((size_t)co * 2 * cin + c) * kh + ky) * kw + kx;
      v = w[a] + w[a + (size_t)cin * kh * kw];
    } else {
      v = w[(((size_t)co * cin + c) * kh + ky) * kw + kx];
    }
  }
  out[idx] = v;
}

// opt-in math mode bf16x3: the packed fp32 weights split into bf16 pieces, a = hi + lo (round to nearest even both times).
// One thread per aligned group of 8 K values of a packed row: [8 x fp32] (32 bytes) -> [8 x bf16 hi][8 x bf16 lo] (32 bytes).
__device__ __forceinline__ unsigned pack_bf16_rne(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);      // finite weights only
  return u >> 16;
}
__global__ void pack_split_bf16_kernel(const float* __restrict__ packed, long groups, unsigned* __restrict__ out) {
  const long gi = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= groups) return;
  const float* src = packed + gi * 8;
  unsigned hi[8], lo[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a = src[i];
    hi[i] = pack_bf16_rne(a);
    lo[i] = pack_bf16_rne(a - __uint_as_float(hi[i] << 16));
  }
  unsigned* dst = out + gi * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dst[i] = hi[2 * i] | (hi[2 * i + 1] << 16);
    dst[4 + i] = lo[2 * i] | (lo[2 * i + 1] << 16);
  }
}

// per-output-channel affine: eval-mode BatchNorm folded onto the accumulator (+ conv bias), or the plain conv bias
__global__ void pack_affine_kernel(const float* __restrict__ conv_bias, const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                   const float* __restrict__ bn_mean, const float* __restrict__ bn_var, float eps,
                                   const float* __restrict__ scale_in, int cout, int cout_pad, int interleave,
                                   float* __restrict__ scale, float* __restrict__ bias) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= cout_pad) return;
  const int co = pack_src_row(row, cout, interleave);
  float sc = 0.f, bi = 0.f;
  if (co >= 0) {
    if (bn_w) {
      sc = bn_w[co] / sqrtf(bn_var[co] + eps);
      bi = bn_b[co] - bn_mean[co] * sc;
      if (conv_bias) bi = bi + conv_bias[co] * sc;
    } else {
      sc = scale_in ? scale_in[co] : 1.f;
      bi = conv_bias ? conv_bias[co] : 0.f;
    }
  }
  if (scale) scale[row] = sc;
  if (bias) bias[row] = bi;
}

}  // namespace sf

namespace sf {
hipError_t launch_wino_weights(const float* w, float* U, int cout_pad, int cin_pad, int ksz, hipStream_t stream);
}
using namespace sf;

extern "C" {

static inline size_t a64(size_t n) { return (n + 63) & ~size_t(63); }

static int packed_dims(int cout, int cin, int flags, int* cout_pad, int* cin_pad) {
  if (cout < 1 || cin < 1) return SF_ERR_INVALID;
  *cin_pad = pack_round_up(cin, 32);
  *cout_pad = (flags & SF_PACK_INTERLEAVE) ? pack_round_up(pack_round_up(cout / 2, 8) * 2, 16) : pack_round_up(cout, 16);
  if ((flags & SF_PACK_INTERLEAVE) && (cout & 1)) return SF_ERR_INVALID;
  return SF_OK;
}

// the Winograd copy exists for 3x3 layers whose input is whole 16-channel chunks and whose outputs fill 64-row tiles (conv_wino.hip)
static bool wino_packable(int cp, int ip, int cin, int kh, int kw, int flags) {
  // (interleaved rows — the sampling layer — are transformed in their packed order: only the small-P kernel's SAMPLE epilogue reads them;
  //  7x7: nine 3x3 sub-kernels, conv_wino.hip: wino_weights_kernel — the trusting gate's first layer on one latent)
  return (flags & SF_PACK_WINOGRAD) && ((kh == 3 && kw == 3) || (kh == 7 && kw == 7)) && cin == ip && (cp % 64) == 0;
}
static size_t wino_floats(int cp, int ip, int kh) { return (size_t)(kh == 7 ? 9 : 1) * 16 * cp * ip; }

size_t sf_pack_conv_bytes(int cout, int cin, int kh, int kw, int flags) {
  int cp = 0, ip = 0;
  if (packed_dims(cout, cin, flags, &cp, &ip) != SF_OK || kh < 1 || kw < 1) return 0;
  return (a64((size_t)cp * kh * kw * ip) * ((flags & SF_PACK_BF16X3) ? 2 : 1) + 2 * a64((size_t)cp) +
          (wino_packable(cp, ip, cin, kh, kw, flags) ? a64(wino_floats(cp, ip, kh)) : 0)) * sizeof(float);
}

int sf_pack_conv(const float* weight, const float* conv_bias, const float* scale, const float* bn_weight, const float* bn_bias,
                 const float* bn_mean, const float* bn_var, float bn_eps, int cout, int cin, int kh, int kw, int c0, int c1, int act,
                 int dil, int stride, int pad, int flags, void* blob, size_t blob_bytes, sf_conv_w* out, void* stream) {
  int cp = 0, ip = 0;
  if (!weight || !blob || !out || kh < 1 || kw < 1 || dil < 1 || stride < 1 || c0 < 0 || c1 < 0) return SF_ERR_INVALID;
  if (packed_dims(cout, cin, flags, &cp, &ip) != SF_OK) return SF_ERR_INVALID;
  if (c0 + c1 != cin || (c0 % 4) || (c1 % 4) || (cout % 4)) return SF_ERR_INVALID;      // the kernels move channels in fours
  if ((bn_weight != nullptr) != (bn_bias != nullptr) || (bn_weight != nullptr) != (bn_mean != nullptr) ||
      (bn_weight != nullptr) != (bn_var != nullptr) || (bn_weight && scale))
    return SF_ERR_INVALID;
  if ((flags & SF_PACK_TRANSPOSED) && (flags & SF_PACK_FOLD_DUP)) return SF_ERR_INVALID;
  if (blob_bytes < sf_pack_conv_bytes(cout, cin, kh, kw, flags)) return SF_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* wp = static_cast<float*>(blob);
  float* sp = wp + a64((size_t)cp * kh * kw * ip);
  float* bp = sp + a64((size_t)cp);
  const long total = (long)cp * kh * kw * ip;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight, cout, cin, kh, kw, cp, ip, flags, wp);
  const bool has_scale = bn_weight || scale, has_bias = bn_weight || conv_bias;
  hipLaunchKernelGGL(pack_affine_kernel, dim3((cp + 255) / 256), dim3(256), 0, st, conv_bias, bn_weight, bn_bias, bn_mean, bn_var, bn_eps, scale,
                     cout, cp, flags & SF_PACK_INTERLEAVE, has_scale ? sp : nullptr, has_bias ? bp : nullptr);
  float* w3 = nullptr;
  if (flags & SF_PACK_BF16X3) {      // ip is a multiple of 32: every row is a whole number of 8-value groups
    w3 = bp + a64((size_t)cp);
    const long groups = total / 8;
    hipLaunchKernelGGL(pack_split_bf16_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, wp, groups, reinterpret_cast<unsigned*>(w3));
  }
  float* wu = nullptr;
  if (wino_packable(cp, ip, cin, kh, kw, flags) && stride == 1 && (pad < 0 || pad == dil * (kh - 1) / 2) && (kh == 3 || dil == 1) && (c0 % 16) == 0 &&
      (c1 % 16) == 0) {
    wu = bp + a64((size_t)cp) + ((flags & SF_PACK_BF16X3) ? a64((size_t)total) : 0);
    if (launch_wino_weights(wp, wu, cp, ip, kh, st) != hipSuccess) return SF_ERR_LAUNCH;
  }
  if (hipGetLastError() != hipSuccess) return SF_ERR_LAUNCH;
  std::memset(out, 0, sizeof(*out));
  out->w = wp;
  out->w_bf16x3 = w3;
  out->w_wino = wu;
  out->scale = has_scale ? sp : nullptr;
  out->bias = has_bias ? bp : nullptr;
  out->cout = cout; out->cout_pad = cp; out->c0 = c0; out->c1 = c1; out->cin_pad = ip;
  out->kh = kh; out->kw = kw; out->dil = dil; out->stride = stride;
  out->pad = pad >= 0 ? pad : (dil * (kh - 1)) / 2;
  out->act = act;
  return SF_OK;
}

int sf_bn_fold(const float* conv_bias, const float* bn_weight, const float* bn_bias, const float* bn_mean, const float* bn_var,
               float bn_eps, int n, float* scale, float* bias, void* stream) {
  if (!bn_weight || !bn_bias || !bn_mean || !bn_var || !scale || !bias || n < 1) return SF_ERR_INVALID;
  hipLaunchKernelGGL(pack_affine_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), conv_bias, bn_weight, bn_bias,
                     bn_mean, bn_var, bn_eps, nullptr, n, n, 0, scale, bias);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

}  // extern "C"
