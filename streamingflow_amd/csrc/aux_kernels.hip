// Small HBM/LDS-bound helper kernels around the implicit-GEMM conv: layout transposes at the
// NCHW API boundary, 2x2 max-pool, nearest x2 upsample, SE gate, depthwise 7x7 + LayerNorm
// (ConvNeXt block head), ASPP global-pool branch.  gfx950 only, NHWC fp32 internally.
#include "sf_device.h"

namespace sf {

__device__ __forceinline__ float4 ld4a(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4a(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- [N][C][HW] <-> [N][HW][C] -------------------------------------------------------------
// 32x32 tiles through LDS (+1 pad): both the read and the write are 128-B contiguous per row.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int rows, int cols, size_t in_stride, size_t out_stride) {
  // in: image i at in + i*in_stride, [rows][cols] -> out: image i at out + i*out_stride, [cols][rows]
  __shared__ float tile[32][33];
  in += (size_t)blockIdx.z * in_stride;
  out += (size_t)blockIdx.z * out_stride;
  const size_t img = 0;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    int r = r0 + ty + i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + i][tx] = in[img + (size_t)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    int c = c0 + ty + i, r = r0 + tx;
    if (r < rows && c < cols) out[img + (size_t)c * rows + r] = tile[tx][ty + i];
  }
}

hipError_t launch_transpose_strided(const float* in, float* out, int n, int rows, int cols, size_t in_stride, size_t out_stride,
                                    hipStream_t s) {
  if (n <= 0 || rows <= 0 || cols <= 0) return hipSuccess;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, n);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, in, out, rows, cols, in_stride, out_stride);
  return hipGetLastError();
}
hipError_t launch_transpose(const float* in, float* out, int n, int rows, int cols, hipStream_t s) {
  return launch_transpose_strided(in, out, n, rows, cols, (size_t)rows * cols, (size_t)rows * cols, s);
}

// ---- 2x2 max-pool (stride 2, floor) and nearest x2 upsample, NHWC ----------------------------
// ceil_pad != 0: output is ceil(H/2) x ceil(W/2) and the missing row/column of an odd input counts
// as zeros (F.pad(value=0) + MaxPool2d(2,2), BEVerse Bottleneck skip path)
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       int n, int Hin, int Win, int C, int ceil_pad) {
  const int Ho = (Hin + ceil_pad) >> 1, Wo = (Win + ceil_pad) >> 1, C4 = C >> 2;
  const size_t total = (size_t)n * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = i % C4;
    size_t p = i / C4;
    int ox = p % Wo;
    size_t q = p / Wo;
    int oy = q % Ho;
    int img = q / Ho;
    const float* b = in + (((size_t)img * Hin + 2 * oy) * Win + 2 * ox) * C + c4 * 4;
    const bool xr = 2 * ox + 1 < Win, yb = 2 * oy + 1 < Hin;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a0 = ld4a(b), a1 = xr ? ld4a(b + C) : z, a2 = yb ? ld4a(b + (size_t)Win * C) : z,
           a3 = (xr && yb) ? ld4a(b + (size_t)Win * C + C) : z;
    float4 m;
    m.x = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
    m.y = fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y));
    m.z = fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z));
    m.w = fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w));
    st4a(out + p * C + c4 * 4, m);
  }
}

__global__ __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int n, int Hin, int Win, int C) {
  const int Ho = Hin * 2, Wo = Win * 2, C4 = C >> 2;
  const size_t total = (size_t)n * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c4 = i % C4;
    size_t p = i / C4;
    int ox = p % Wo;
    size_t q = p / Wo;
    int oy = q % Ho;
    int img = q / Ho;
    st4a(out + p * C + c4 * 4, ld4a(in + (((size_t)img * Hin + (oy >> 1)) * Win + (ox >> 1)) * C + c4 * 4));
  }
}

// UpsamplingAdd (convolutions.py:204-215): bilinear x2 (align_corners=False) of `in`, plus the skip tensor.
// source index as ATen's area_pixel_compute_source_index: src = 0.5*(dst + 0.5) - 0.5, clamped at 0.
__global__ __launch_bounds__(256) void upsample_bilinear2_add_kernel(const float* __restrict__ in, const float* __restrict__ skip,
                                                                     float* __restrict__ out, int n, int Hin, int Win, int C) {
  const int Ho = Hin * 2, Wo = Win * 2, C4 = C >> 2;
  const size_t total = (size_t)n * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = i % C4;
    const size_t p = i / C4;
    const int ox = p % Wo;
    const size_t q = p / Wo;
    const int oy = q % Ho;
    const int img = q / Ho;
    float sy = 0.5f * ((float)oy + 0.5f) - 0.5f, sx = 0.5f * ((float)ox + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* base = in + (size_t)img * Hin * Win * C + c4 * 4;
    const float4 a = ld4a(base + ((size_t)y0 * Win + x0) * C), b = ld4a(base + ((size_t)y0 * Win + x1) * C);
    const float4 c = ld4a(base + ((size_t)y1 * Win + x0) * C), d = ld4a(base + ((size_t)y1 * Win + x1) * C);
    float4 r = skip ? ld4a(skip + p * C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    r.x += hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x);
    r.y += hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y);
    r.z += hy * (hx * a.z + lx * b.z) + ly * (hx * c.z + lx * d.z);
    r.w += hy * (hx * a.w + lx * b.w) + ly * (hx * c.w + lx * d.w);
    st4a(out + p * C + c4 * 4, r);
  }
}

static int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

hipError_t launch_maxpool2(const float* in, float* out, int n, int Hin, int Win, int C, int ceil_pad, hipStream_t s) {
  size_t total = (size_t)n * ((Hin + ceil_pad) / 2) * ((Win + ceil_pad) / 2) * (C / 4);
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, n, Hin, Win, C, ceil_pad);
  return hipGetLastError();
}
hipError_t launch_upsample_bilinear2_add(const float* in, const float* skip, float* out, int n, int Hin, int Win, int C,
                                         hipStream_t s) {
  size_t total = (size_t)n * Hin * 2 * Win * 2 * (C / 4);
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(upsample_bilinear2_add_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, skip, out, n, Hin, Win, C);
  return hipGetLastError();
}
// LogSigmoid (the decoder of DistributionModule(method='BERNOULLI'), streamingflow/models/distributions.py:33): min(x, 0) - log1p(exp(-|x|))
__global__ __launch_bounds__(256) void logsigmoid_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = in[i];
    out[i] = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
  }
}
hipError_t launch_logsigmoid(const float* in, float* out, size_t n, hipStream_t s) {
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(logsigmoid_kernel, dim3((unsigned)(blocks < 65536 ? (blocks ? blocks : 1) : 65536)), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}
hipError_t launch_upsample2(const float* in, float* out, int n, int Hin, int Win, int C, hipStream_t s) {
  size_t total = (size_t)n * Hin * 2 * Win * 2 * (C / 4);
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(upsample2_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, n, Hin, Win, C);
  return hipGetLastError();
}

// ---- SE gate (res_models.py:161-165) ---------------------------------------------------------
// chansum: [ntile16][C] per-16-pixel channel sums written by the producing conv's epilogue.
// One workgroup: fixed-order sum => mean => fc0 (C/r x C) => ReLU => fc2 (C x C/r) => sigmoid.
__global__ __launch_bounds__(1024) void se_fc_kernel(const float* __restrict__ chansum, int ntile, int C, int Cr,
                                                     float inv_hw, const float* __restrict__ fc0,
                                                     const float* __restrict__ fc2, float* __restrict__ scale) {
  extern __shared__ float sm[];   // part[G][C] | mean[C] | hid[Cr]
  chansum += (size_t)blockIdx.x * ntile * C;   // one workgroup per image
  scale += (size_t)blockIdx.x * C;
  const int G = 1024 / C;         // tile groups summed in parallel (C <= 1024)
  float* part = sm;
  float* mean = sm + G * C;
  float* hid = mean + C;
  const int c = threadIdx.x % C, grp = threadIdx.x / C;
  if (grp < G) {
    // fixed assignment tile -> group and fixed order inside a group: bitwise reproducible
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = grp;
    for (; t + 3 * G < ntile; t += 4 * G) {
      s0 += chansum[(size_t)t * C + c];
      s1 += chansum[(size_t)(t + G) * C + c];
      s2 += chansum[(size_t)(t + 2 * G) * C + c];
      s3 += chansum[(size_t)(t + 3 * G) * C + c];
    }
    for (; t < ntile; t += G) s0 += chansum[(size_t)t * C + c];
    part[grp * C + c] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
    for (int q = 0; q < G; ++q) s += part[q * C + threadIdx.x];
    mean[threadIdx.x] = s * inv_hw;
  }
  __syncthreads();
  // hidden layer: one wave per output, lanes stride over the C inputs, shuffle reduce
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int h = wave; h < Cr; h += 16) {
    float s = 0.f;
    for (int k = lane; k < C; k += 64) s += fc0[h * C + k] * mean[k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) hid[h] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
    for (int h = 0; h < Cr; ++h) s += fc2[threadIdx.x * Cr + h] * hid[h];
    scale[threadIdx.x] = 1.f / (1.f + expf(-s));
  }
}

hipError_t launch_se_fc(const float* chansum, int ntile, int C, int Cr, int hw, const float* fc0,
                        const float* fc2, float* scale, int n_img, hipStream_t s) {
  if (C < 1 || C > 1024 || n_img < 1) return hipErrorInvalidValue;
  const int G = 1024 / C;
  hipLaunchKernelGGL(se_fc_kernel, dim3(n_img), dim3(1024), (G * C + C + Cr) * sizeof(float), s, chansum, ntile, C, Cr,
                     1.f / (float)hw, fc0, fc2, scale);
  return hipGetLastError();
}

// ---- depthwise 7x7 (+bias) + channels-last LayerNorm (convolutions.py:335-337) ---------------
// One workgroup per 8x8 pixel tile; the 14x14xC input patch and the [49][C] weights are staged in
// LDS once and shared by NW waves, each of which owns C/NW channels of every pixel (one pixel per
// lane).  LayerNorm statistics are combined across the waves through LDS in a fixed order.
template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void dwconv7_ln_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             const float* __restrict__ wdw /*[49][C]*/,
                                                             const float* __restrict__ bdw, const float* __restrict__ lnw,
                                                             const float* __restrict__ lnb, int H, int W, float eps) {
  constexpr int PS = C + 4;        // padded pixel stride (floats)
  constexpr int CW = C / NW;       // channels per wave
  constexpr int C4 = C / 4, CW4 = CW / 4;
  static_assert(CW % 4 == 0, "channels per wave must be a multiple of 4");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* patch = sm;               // [14*14][PS]
  float* wl = sm + 196 * PS;       // [49][C]
  float* red = wl + 49 * C;        // [2][NW][64] partial sums / sums of squares
  const int tiles_x = (W + 7) / 8;
  const int img = blockIdx.y;
  const int ty0 = (blockIdx.x / tiles_x) * 8, tx0 = (blockIdx.x % tiles_x) * 8;
  const float* src = in + (size_t)img * H * W * C;
  for (int i = threadIdx.x; i < 196 * C4; i += 64 * NW) {
    int c4 = i % C4, pp = i / C4;
    int py = pp / 14, px = pp % 14;
    int iy = ty0 + py - 3, ix = tx0 + px - 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld4a(src + ((size_t)iy * W + ix) * C + c4 * 4);
    st4a(patch + pp * PS + c4 * 4, v);
  }
  for (int i = threadIdx.x; i < 49 * C4; i += 64 * NW) st4a(wl + i * 4, ld4a(wdw + i * 4));
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ly = lane >> 3, lx = lane & 7;
  const int oy = ty0 + ly, ox = tx0 + lx;
  const int cb = wave * CW;
  float acc[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) acc[c] = bdw[cb + c];
  for (int ky = 0; ky < 7; ++ky)
    for (int kx = 0; kx < 7; ++kx) {
      const float* p = patch + ((ly + ky) * 14 + lx + kx) * PS + cb;
      const float* w = wl + (ky * 7 + kx) * C + cb;
#pragma unroll
      for (int c4 = 0; c4 < CW4; ++c4) {
        float4 x = ld4a(p + c4 * 4), ww = ld4a(w + c4 * 4);
        acc[c4 * 4 + 0] += x.x * ww.x; acc[c4 * 4 + 1] += x.y * ww.y;
        acc[c4 * 4 + 2] += x.z * ww.z; acc[c4 * 4 + 3] += x.w * ww.w;
      }
    }
  // mean over all C channels of the pixel (fixed wave order)
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CW; ++c) s += acc[c];
  red[wave * 64 + lane] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < NW; ++q) tot += red[q * 64 + lane];
  const float mean = tot / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < CW; ++c) { float d = acc[c] - mean; sq += d * d; }
  red[(NW + wave) * 64 + lane] = sq;
  __syncthreads();
  float tsq = 0.f;
#pragma unroll
  for (int q = 0; q < NW; ++q) tsq += red[(NW + q) * 64 + lane];
  const float rstd = 1.f / sqrtf(tsq / (float)C + eps);
  if (oy < H && ox < W) {
    float* dst = out + ((size_t)img * H * W + (size_t)oy * W + ox) * C + cb;
#pragma unroll
    for (int c4 = 0; c4 < CW4; ++c4) {
      float4 w = ld4a(lnw + cb + c4 * 4), b = ld4a(lnb + cb + c4 * 4), o;
      o.x = (acc[c4 * 4 + 0] - mean) * rstd * w.x + b.x;
      o.y = (acc[c4 * 4 + 1] - mean) * rstd * w.y + b.y;
      o.z = (acc[c4 * 4 + 2] - mean) * rstd * w.z + b.z;
      o.w = (acc[c4 * 4 + 3] - mean) * rstd * w.w + b.w;
      st4a(dst + c4 * 4, o);
    }
  }
}

template <int C, int NW>
static hipError_t launch_dw_t(const float* in, float* out, const float* wdw, const float* bdw, const float* lnw,
                              const float* lnb, int n, int H, int W, float eps, hipStream_t s) {
  int lds = (196 * (C + 4) + 49 * C + 2 * NW * 64) * sizeof(float);
  auto k = dwconv7_ln_kernel<C, NW>;
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    done = true;
  }
  dim3 grid(((H + 7) / 8) * ((W + 7) / 8), n);
  hipLaunchKernelGGL(k, grid, dim3(64 * NW), lds, s, in, out, wdw, bdw, lnw, lnb, H, W, eps);
  return hipGetLastError();
}

// C == 64 (the shipped width): lane = channel, so every load / store is one coalesced 256-B row, the 49 weights of
// the lane's channel sit in registers and nothing goes through LDS.  A wave produces 4 output rows x `seg` columns;
// its 10 x 7 input window (+ one prefetched column) lives in registers and slides along x: one new column (10 row
// loads, issued one step before they are used) per step, the column slots rotate by compile-time renaming (the x loop is unrolled by 8), 4 x 49 FMAs, and the channels-last LayerNorm
// is two butterfly reductions over the wave.  Taps are accumulated in the same (ky, kx) order as the generic kernel.
// Wave-wide sum without LDS traffic: four DPP steps inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror), then the four row sums are read into SGPRs.  (__shfl_xor lowers to ds_bpermute_b32: 12 LDS round trips per
// pixel for the two LayerNorm reductions.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum64(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  const int iv = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}
__global__ __launch_bounds__(256, 3) void dwconv7_ln_c64_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             const float* __restrict__ wdw, const float* __restrict__ bdw,
                                                             const float* __restrict__ lnw, const float* __restrict__ lnb, int n,
                                                             int H, int W, float eps, int seg) {
  constexpr int C = 64, R = 4;
  const int lane = threadIdx.x & 63;
  // everything but `lane` is wave-uniform: row / column addresses stay in SGPRs, loads are saddr + lane offset
  const long wv = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nseg = (W + seg - 1) / seg, nrb = (H + R - 1) / R;
  if (wv >= (long)n * nrb * nseg) return;
  const int sg = (int)(wv % nseg);
  const long t = wv / nseg;
  const int rb = (int)(t % nrb), img = (int)(t / nrb);
  const int y0 = rb * R, x0 = sg * seg, x1 = (x0 + seg < W) ? x0 + seg : W;
  float w[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) w[k] = wdw[k * C + lane];
  const float bias = bdw[lane], gam = lnw[lane], bet = lnb[lane];
  const float* src = in + (size_t)img * H * W * C;
  float* dst = out + (size_t)img * H * W * C;
  float win[R + 6][8];
  auto load_col = [&](int x, int slot) {
    const bool xok = (x >= 0) & (x < W);
#pragma unroll
    for (int r = 0; r < R + 6; ++r) {
      const int y = y0 - 3 + r;
      const bool ok = xok & (y >= 0) & (y < H);
      const float* rowp = src + (ok ? ((size_t)y * W + x) * C : (size_t)0);
      const float v = rowp[lane];
      win[r][slot] = ok ? v : 0.f;
    }
  };
#pragma unroll
  for (int kx = 0; kx < 7; ++kx) load_col(x0 - 3 + kx, kx);
  for (int xb = x0; xb < x1; xb += 8) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {        // window position kx lives in slot (p + kx) % 8; slot (p + 7) % 8 is filled one step ahead
      const int x = xb + p;
      if (x < x1) {                       // wave-uniform
        load_col(x + 4, (p + 7) % 8);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (y0 + r < H) {               // wave-uniform
            float acc = bias;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
              for (int kx = 0; kx < 7; ++kx) acc += win[r + ky][(p + kx) % 8] * w[ky * 7 + kx];
            const float mean = wave_sum64(acc) * (1.f / C);
            const float d = acc - mean;
            const float rstd = 1.f / sqrtf(wave_sum64(d * d) * (1.f / C) + eps);
            (dst + ((size_t)(y0 + r) * W + x) * C)[lane] = d * rstd * gam + bet;
          }
        }
      }
    }
  }
}

// The same with TWO channels per lane (v_pk_fma_f32: the 49 taps of both in one instruction each) and two strips per wave: lanes 0-31
// work on strip 2w, lanes 32-63 on strip 2w + 1 (a strip = 2 output rows x `seg` columns of one image; both strips of a wave lie in
// the same image, the host checks that the strips of an image come in pairs).  Window 8 rows x 7 columns (+ one prefetched) of
// channel pairs in registers, 8-byte loads through a buffer descriptor on the image — a row or column outside it gets an offset
// beyond the descriptor's range and reads zero, no masks — the LayerNorm sums over the 32 lanes of a half by DPP + four lane reads.
// Taps accumulate in the same (ky, kx) order as the one-channel kernel: the convolution is bitwise the same, the LayerNorm sums
// associate differently (last-bit differences).  Vector work per pixel 2.3x less: the one-channel kernel is bound by it.
typedef float dw_f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256, 2) void dwconv7_ln_c64_pk_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                const float* __restrict__ wdw, const float* __restrict__ bdw,
                                                                const float* __restrict__ lnw, const float* __restrict__ lnb, int n,
                                                                int H, int W, float eps, int seg, int nseg, unsigned m_per_img,
                                                                unsigned m_nseg) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = 64, R = 2;
  constexpr unsigned BAD = 0x40000000u;      // added to an offset: beyond any image (the host keeps images below 1 GB)
  const int lane = threadIdx.x & 63, half = lane >> 5, cp = lane & 31;
  const int wv = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nrb = (H + R - 1) / R;
  const int per_img = nrb * nseg;             // even (host)
  if (2 * (long)wv >= (long)n * per_img) return;
  // strip -> (image, row pair, segment): quotients by the host's ceil(2^32 / d) (0: d = 1) — a wave lives for 80 pixels, four
  // reciprocal sequences through the vector unit at its start were 5 % of that
  const int img = m_per_img ? (int)__umulhi((unsigned)(2 * wv), m_per_img) : 2 * wv;  // wave-uniform
  const int si = 2 * wv - img * per_img + half;
  const int rb = m_nseg ? (int)__umulhi((unsigned)si, m_nseg) : si, sg = si - rb * nseg;
  const int y0 = rb * R, x0 = sg * seg, x1 = (x0 + seg < W) ? x0 + seg : W;
  const size_t img_off = (size_t)img * H * W * C;
  const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + img_off), (short)0, (int)((unsigned)H * W * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out + img_off, (short)0, (int)((unsigned)H * W * C * 4), 0x00020000);
  dw_f2 w[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) w[k] = *reinterpret_cast<const dw_f2*>(wdw + k * C + 2 * cp);
  const dw_f2 bias = *reinterpret_cast<const dw_f2*>(bdw + 2 * cp), gam = *reinterpret_cast<const dw_f2*>(lnw + 2 * cp),
              bet = *reinterpret_cast<const dw_f2*>(lnb + 2 * cp);
  unsigned rowb[R + 6];                       // byte offset of (row, column 0, channel pair) or BAD
#pragma unroll
  for (int r = 0; r < R + 6; ++r) {
    const int y = y0 - 3 + r;
    rowb[r] = (y >= 0 && y < H) ? (unsigned)(y * W) * (C * 4) + (unsigned)cp * 8 : BAD;
  }
  dw_f2 win[R + 6][8];
  auto load_col = [&](const int x, const int slot) {
    const unsigned xo = (x >= 0 && x < W) ? (unsigned)x * (C * 4) : BAD;
#pragma unroll
    for (int r = 0; r < R + 6; ++r)
      win[r][slot] = __builtin_bit_cast(dw_f2, __builtin_amdgcn_raw_buffer_load_b64(rs_i, (int)(rowb[r] + xo), 0, 0));
  };
  // sum over the 32 lanes of the half (two values per lane): DPP inside each row of 16, then the two rows of the half
  auto half_sum = [&](const dw_f2 v) {
    float t = v.x + v.y;
    t += dpp_mov<0xB1>(t);
    t += dpp_mov<0x4E>(t);
    t += dpp_mov<0x141>(t);
    t += dpp_mov<0x140>(t);
    const int it = __float_as_int(t);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(it, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(it, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(it, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(it, 48));
    return half ? r2 + r3 : r0 + r1;
  };
#pragma unroll
  for (int kx = 0; kx < 7; ++kx) load_col(x0 - 3 + kx, kx);
  for (int xb = 0; xb < seg; xb += 8) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {             // window position kx lives in slot (p + kx) % 8; slot (p + 7) % 8 is filled one step ahead
      const int x = x0 + xb + p;
      if (xb + p >= seg) break;               // wave-uniform (seg not a multiple of 8); also keeps the scheduler from hoisting the next columns' loads over this one's products (132 spills without it)
      load_col(x + 4, (p + 7) % 8);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        dw_f2 acc = bias;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) acc = __builtin_elementwise_fma(win[r + ky][(p + kx) % 8], w[ky * 7 + kx], acc);
        const float mean = half_sum(acc) * (1.f / C);
        const dw_f2 d = acc - mean;
        const float rstd = 1.f / sqrtf(half_sum(d * d) * (1.f / C) + eps);
        const dw_f2 o = d * rstd * gam + bet;
        const bool ok = (y0 + r < H) && (x < x1);
        const unsigned off = ok ? (unsigned)((y0 + r) * W + x) * (C * 4) + (unsigned)cp * 8 : BAD;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, o), rs_o, (int)off, 0, 0);
      }
    }
  }
#endif
}

hipError_t launch_dwconv7_ln(const float* in, float* out, const float* wdw, const float* bdw, const float* lnw,
                             const float* lnb, int n, int H, int W, int C, float eps, hipStream_t s) {
  static const bool pk = [] { const char* v = std::getenv("SF_DWCONV_PK"); return v ? std::atoi(v) != 0 : true; }();
  if (pk && C == 64 && (long)n * H * W >= 65536 && (double)H * W * C * 4.0 < 1073741824.0) {
    // two channels per lane, two strips per wave: strips of an image must come in pairs
    int seg = 40;
    const int nrb = (H + 1) / 2;
    if ((nrb * ((W + seg - 1) / seg)) & 1) seg = W;      // one strip per row pair ...
    if (((nrb * ((W + seg - 1) / seg)) & 1) == 0) {      // ... (an odd number of row pairs of one strip each keeps the one-channel kernel)
      const int nseg = (W + seg - 1) / seg;
      const long per_img = (long)nrb * nseg, waves = ((long)n * per_img) / 2;
      auto magic = [](long d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); };
      if ((double)n * per_img * per_img < 4.0e9) {      // exact quotients (dividend x divisor < 2^32)
        hipLaunchKernelGGL(dwconv7_ln_c64_pk_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, in, out, wdw, bdw, lnw, lnb, n, H, W, eps,
                           seg, nseg, magic(per_img), magic(nseg));
        return hipGetLastError();
      }
    }
  }
  if (C == 64 && (long)n * H * W >= 65536) {   // large maps: register-window kernel (below that the LDS-tile kernel has more waves)
    const int seg = 40;
    const long waves = (long)n * ((H + 3) / 4) * ((W + seg - 1) / seg);
    hipLaunchKernelGGL(dwconv7_ln_c64_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, in, out, wdw, bdw, lnw, lnb, n, H, W, eps,
                       seg);
    return hipGetLastError();
  }
  switch (C) {
    case 8:  return launch_dw_t<8, 1>(in, out, wdw, bdw, lnw, lnb, n, H, W, eps, s);
    case 16: return launch_dw_t<16, 2>(in, out, wdw, bdw, lnw, lnb, n, H, W, eps, s);
    case 32: return launch_dw_t<32, 4>(in, out, wdw, bdw, lnw, lnb, n, H, W, eps, s);
    case 64: return launch_dw_t<64, 4>(in, out, wdw, bdw, lnw, lnb, n, H, W, eps, s);
  }
  return hipErrorInvalidValue;
}

// ---- ASPP image-pooling branch (convolutions.py:227-239) -------------------------------------
// stage 1: per-image, per-slab channel sums (fixed order inside a slab)
__global__ __launch_bounds__(256) void chan_partial_kernel(const float* __restrict__ in, float* __restrict__ part,
                                                           int HW, int C, int nslab) {
  extern __shared__ float sm[];   // [256/C4][C]
  const int C4 = C >> 2;
  const int lanes = 256 / C4;     // pixel lanes per block
  const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
  const int img = blockIdx.y, slab = blockIdx.x;
  const int per = (HW + nslab - 1) / nslab;
  const int p0 = slab * per, p1 = min(HW, p0 + per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pl < lanes)
    for (int p = p0 + pl; p < p1; p += lanes) {
      float4 v = ld4a(in + ((size_t)img * HW + p) * C + c4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  if (pl < lanes) st4a(sm + pl * C + c4 * 4, s);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += sm[l * C + c];
    part[((size_t)img * nslab + slab) * C + c] = t;
  }
}

// stage 2 (one workgroup per image): mean -> 1x1 conv (C->hid) -> BN -> ReLU -> its slice of the
// ASPP projection (hid->hid) -> folded into the projection's per-image bias:
//   bias_img[img][co] = proj_scale[co] * (Wp_pool[co][:] . g) + proj_bias[co]
// (bilinear upsampling of a 1x1 map, align_corners=False, is a constant broadcast.)
__global__ __launch_bounds__(256) void aspp_pool_kernel(const float* __restrict__ part, int nslab, int C, int hid,
                                                        float inv_hw, const float* __restrict__ w1 /*[hid][C]*/,
                                                        const float* __restrict__ s1, const float* __restrict__ b1,
                                                        const float* __restrict__ wp /*[hid][hid]*/,
                                                        const float* __restrict__ ps, const float* __restrict__ pb,
                                                        float* __restrict__ bias_img) {
  extern __shared__ float sm[];   // mean[C] | g[hid]
  float* mean = sm;
  float* gg = sm + C;
  const int img = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < nslab; ++t) s += part[((size_t)img * nslab + t) * C + c];
    mean[c] = s * inv_hw;
  }
  __syncthreads();
  for (int h = threadIdx.x; h < hid; h += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += w1[h * C + c] * mean[c];
    s = s * s1[h] + b1[h];
    gg[h] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int co = threadIdx.x; co < hid; co += blockDim.x) {
    float s = 0.f;
    for (int h = 0; h < hid; ++h) s += wp[co * hid + h] * gg[h];
    bias_img[(size_t)img * hid + co] = ps[co] * s + pb[co];
  }
}

// mean over pixels from per-slab sums: out[img][c] = inv_hw * sum_slab part[img][slab][c]
__global__ void mean_from_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int nslab, int C,
                                          float inv_hw) {
  const int img = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < nslab; ++t) s += part[((size_t)img * nslab + t) * C + c];
    out[(size_t)img * C + c] = s * inv_hw;
  }
}
hipError_t launch_mean_from_partials(const float* part, float* out, int n, int nslab, int C, int hw, hipStream_t s) {
  hipLaunchKernelGGL(mean_from_partials_kernel, dim3(n), dim3(256), 0, s, part, out, nslab, C, 1.f / (float)hw);
  return hipGetLastError();
}

// out[img][p][co + c] = vec[img][c]: a per-image channel vector broadcast over the pixels of a channel slice
__global__ __launch_bounds__(256) void broadcast_channels_kernel(const float* __restrict__ vec, float* __restrict__ out, int HW, int k,
                                                                 int out_cs, int out_co, size_t total) {
  const int k4 = k >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = i % k4;
    const size_t p = i / k4;                 // global pixel = img*HW + pixel
    const size_t img = p / HW;
    st4a(out + p * out_cs + out_co + c4 * 4, ld4a(vec + img * k + c4 * 4));
  }
}
hipError_t launch_broadcast_channels(const float* vec, float* out, int n, int HW, int k, int out_cs, int out_co, hipStream_t s) {
  size_t total = (size_t)n * HW * (k / 4);
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(broadcast_channels_kernel, dim3(grid_for(total)), dim3(256), 0, s, vec, out, HW, k, out_cs, out_co, total);
  return hipGetLastError();
}

hipError_t launch_chan_partial(const float* in, float* part, int n, int HW, int C, int nslab, hipStream_t s) {
  int C4 = C / 4;
  if (C4 < 1 || C4 > 256) return hipErrorInvalidValue;
  int lanes = 256 / C4;
  hipLaunchKernelGGL(chan_partial_kernel, dim3(nslab, n), dim3(256), lanes * C * sizeof(float), s, in, part, HW, C, nslab);
  return hipGetLastError();
}

hipError_t launch_aspp_pool(const float* in, float* part, float* bias_img, int n, int HW, int C, int hid,
                            const float* w1, const float* s1, const float* b1, const float* wp, const float* ps,
                            const float* pb, int nslab, hipStream_t s) {
  int C4 = C / 4;
  if (C4 < 1 || C4 > 256) return hipErrorInvalidValue;
  int lanes = 256 / C4;
  hipLaunchKernelGGL(chan_partial_kernel, dim3(nslab, n), dim3(256), lanes * C * sizeof(float), s, in, part, HW, C, nslab);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(aspp_pool_kernel, dim3(n), dim3(256), (C + hid) * sizeof(float), s, part, nslab, C, hid,
                     1.f / (float)HW, w1, s1, b1, wp, ps, pb, bias_img);
  return hipGetLastError();
}

}  // namespace sf
