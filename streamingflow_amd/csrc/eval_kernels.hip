// Evaluation-side kernels for gfx950 + C ABI: the per-pixel parts of
// streamingflow/metrics.py and streamingflow/utils/instance.py.  Integer / index work: results are exact.
//   sf_confusion_fwd          joint histogram of two label maps (IntersectionOverUnion's stat scores,
//                             PanopticMetric's bincount of prediction + K * target, metrics.py:37, :171-176)
//   sf_instance_centers_fwd   find_instance_centers (instance.py:80-92): threshold, 3x3 max-pool NMS, and the
//                             row-major list torch.nonzero returns (flags + exclusive scan + compaction)
//   sf_group_pixels_fwd       group_pixels + foreground mask (instance.py:95-116, :136-137): nearest centre of
//                             (pixel + offset), first minimum on ties
//   sf_instance_moments_fwd   per (frame, instance) pixel counts and position sums, plain and flow-warped — the masked means
//                             of make_instance_id_temporally_consistent (instance.py:213-236), all frames in one launch,
//                             integer atomics only (order-independent)
//   sf_confusion_frames_fwd   the joint histogram per frame of a whole label sequence in one launch
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "../../include/sfnative.h"

namespace sf {

__global__ void confusion_kernel(const long long* __restrict__ a, const long long* __restrict__ b, long n, int K,
                                 unsigned long long* __restrict__ out, int* __restrict__ bad) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long long x = a[i], y = b[i];
    if (x < 0 || x >= K || y < 0 || y >= K) { *bad = 1; continue; }
    atomicAdd(out + (size_t)y * K + x, 1ULL);
  }
}

// keep[i][j] = thresholded value is a strict-positive local maximum of its 3x3 neighbourhood
__global__ void center_flag_kernel(const float* __restrict__ c, int H, int W, float thr, int* __restrict__ flag) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const int i = idx / W, j = idx - i * W;
  auto val = [&](int y, int x) -> float {
    if (y < 0 || y >= H || x < 0 || x >= W) return -INFINITY;       // max_pool2d pads with -inf
    const float v = c[y * W + x];
    return v > thr ? v : -1.f;                                       // F.threshold(x, thr, -1)
  };
  const float v = val(i, j);
  float m = v;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) m = fmaxf(m, val(i + dy, j + dx));
  flag[idx] = (v == m && v > 0.f) ? 1 : 0;
}

__global__ void center_compact_kernel(const int* __restrict__ flag, const int* __restrict__ scan, int H, int W, int cap,
                                      int* __restrict__ centers, int* __restrict__ n_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = H * W;
  if (idx == 0) n_out[0] = scan[n - 1] + flag[n - 1];
  if (idx >= n || !flag[idx]) return;
  const int r = scan[idx];
  if (r >= cap) return;
  centers[2 * r] = idx / W;
  centers[2 * r + 1] = idx % W;
}

__global__ void group_pixels_kernel(const int* __restrict__ centers, int nc, const float* __restrict__ off, const unsigned char* __restrict__ fg,
                                    int H, int W, long long* __restrict__ inst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const int i = idx / W, j = idx - i * W;
  const float lx = __fadd_rn((float)i, off[idx]), ly = __fadd_rn((float)j, off[H * W + idx]);
  float best = INFINITY;
  int arg = 0;
  for (int k = 0; k < nc; ++k) {
    const float dx = __fsub_rn((float)centers[2 * k], lx), dy = __fsub_rn((float)centers[2 * k + 1], ly);
    const float d = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    if (d < best) { best = d; arg = k; }
  }
  inst[idx] = fg[idx] ? (long long)(arg + 1) : 0LL;
}

// Per (frame, instance id) pixel count, exact integer sums of (row, col) and sums of the flow-warped position
// (row + flow0, col + flow1) in 2^-20 fixed point.  Integer atomics only: the result does not depend on the order the
// pixels arrive in (bitwise reproducible), unlike a floating-point atomicAdd.
constexpr double MOMENT_FX = 1048576.0;
__global__ void instance_moments_kernel(const long long* __restrict__ inst, const float* __restrict__ flow, int F, int H, int W, int max_id,
                                        unsigned long long* __restrict__ pos, unsigned long long* __restrict__ warped, int* __restrict__ cnt) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long plane = (long)H * W;
  if (idx >= F * plane) return;
  const int f = (int)(idx / plane);
  const int pix = (int)(idx - f * plane);
  const long long id = inst[idx];
  if (id <= 0 || id > max_id) return;
  const int i = pix / W, j = pix - i * W;
  const size_t slot = (size_t)f * (max_id + 1) + (size_t)id;
  atomicAdd(pos + 2 * slot, (unsigned long long)i);
  atomicAdd(pos + 2 * slot + 1, (unsigned long long)j);
  atomicAdd(cnt + slot, 1);
  if (warped) {
    const float* fl = flow + (size_t)f * 2 * plane;
    const float x = __fadd_rn((float)i, fl[pix]), y = __fadd_rn((float)j, fl[plane + pix]);
    // two's complement: adding the unsigned image of a negative fixed-point value is the signed add
    atomicAdd(warped + 2 * slot, (unsigned long long)llrint((double)x * MOMENT_FX));
    atomicAdd(warped + 2 * slot + 1, (unsigned long long)llrint((double)y * MOMENT_FX));
  }
}

// joint histograms of F label-map pairs of n elements each: out[f][b][a] (labels outside [0, K) set *bad)
__global__ void confusion_frames_kernel(const long long* __restrict__ a, const long long* __restrict__ b, long n, int F, int K,
                                        unsigned long long* __restrict__ out, int* __restrict__ bad) {
  const long total = n * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long long x = a[i], y = b[i];
    if (x < 0 || x >= K || y < 0 || y >= K) { *bad = 1; continue; }
    const long f = i / n;
    // in BEV instance maps > 95 % of the pixels are (background, background): the lanes of a wave that hold that pair for
    // the same frame as the first of them add ONE ballot count instead of up to 64 atomics on one address (integer adds:
    // the result does not depend on who adds)
    const bool bg = (x == 0) & (y == 0);
    const unsigned long long m = __ballot(bg);
    bool done = false;
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      const long fl = __shfl(f, leader);
      const unsigned long long same = __ballot(bg && f == fl);
      if (bg && f == fl) {
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(out + (size_t)f * K * K, (unsigned long long)__popcll(same));
        done = true;
      }
    }
    if (!done) atomicAdd(out + ((size_t)f * K + (size_t)y) * K + x, 1ULL);
  }
}

// warp_features (utils/geometry.py:196-236): affine_grid(theta, align_corners=False) + grid_sample(mode, zeros
// padding, align_corners=False) on NCHW maps.  theta [b][6] row major (2 x 3).
//   base grid   x_j = (2j + 1)/W - 1,  y_i = (2i + 1)/H - 1
//   source      gx = t0*x + t1*y + t2, gy = t3*x + t4*y + t5;  ix = ((gx + 1)*W - 1)/2, iy likewise
//   nearest     index = nearbyint(ix) (ties to even), zero when outside; bilinear: 4 taps, zeros outside
__global__ void warp_affine_kernel(const float* __restrict__ x, const float* __restrict__ theta, int B, int C, int H, int W, int bilinear,
                                   float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  const int j = (int)(idx % W);
  const long r = idx / W;
  const int i = (int)(r % H), b = (int)(r / H);
  const float* t = theta + 6 * b;
  const float xs = (2.f * j + 1.f) / (float)W - 1.f, ys = (2.f * i + 1.f) / (float)H - 1.f;
  const float gx = __fadd_rn(__fadd_rn(__fmul_rn(t[0], xs), __fmul_rn(t[1], ys)), t[2]);
  const float gy = __fadd_rn(__fadd_rn(__fmul_rn(t[3], xs), __fmul_rn(t[4], ys)), t[5]);
  const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  const size_t plane = (size_t)H * W;
  const float* xb = x + (size_t)b * C * plane;
  float* ob = out + (size_t)b * C * plane + (size_t)i * W + j;
  if (!bilinear) {
    const float fx = nearbyintf(ix), fy = nearbyintf(iy);
    const bool ok = fx >= 0.f && fx < (float)W && fy >= 0.f && fy < (float)H;
    const size_t src = ok ? (size_t)((int)fy) * W + (int)fx : 0;
    for (int c = 0; c < C; ++c) ob[c * plane] = ok ? xb[c * plane + src] : 0.f;
    return;
  }
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  auto in = [&](int yy, int xx) { return yy >= 0 && yy < H && xx >= 0 && xx < W; };
  for (int c = 0; c < C; ++c) {
    const float* pc = xb + c * plane;
    float v = 0.f;
    if (in(y0, x0)) v += pc[(size_t)y0 * W + x0] * (wx0 * wy0);
    if (in(y0, x1)) v += pc[(size_t)y0 * W + x1] * (wx1 * wy0);
    if (in(y1, x0)) v += pc[(size_t)y1 * W + x0] * (wx0 * wy1);
    if (in(y1, x1)) v += pc[(size_t)y1 * W + x1] * (wx1 * wy1);
    ob[c * plane] = v;
  }
}

inline size_t a256e(size_t n) { return (n + 255) & ~size_t(255); }

}  // namespace sf

using namespace sf;

extern "C" {

int sf_confusion_fwd(const int64_t* a, const int64_t* b, long n, int K, int64_t* out, int32_t* bad, void* stream) {
  if (!out || !bad || K < 1 || n < 0) return SF_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out, 0, (size_t)K * K * sizeof(int64_t), st) != hipSuccess) return SF_ERR_LAUNCH;
  if (hipMemsetAsync(bad, 0, sizeof(int32_t), st) != hipSuccess) return SF_ERR_LAUNCH;
  if (n == 0) return SF_OK;
  if (!a || !b) return SF_ERR_INVALID;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const long long*>(a),
                     reinterpret_cast<const long long*>(b), n, K, reinterpret_cast<unsigned long long*>(out), bad);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

size_t sf_instance_centers_ws_bytes(int H, int W) {
  if (H < 1 || W < 1) return 0;
  size_t tb = 0;
  (void)rocprim::exclusive_scan(nullptr, tb, (const int*)nullptr, (int*)nullptr, 0, (size_t)H * W, rocprim::plus<int>(), nullptr);
  return 2 * a256e((size_t)H * W * 4) + a256e(tb) + 256;
}

int sf_instance_centers_fwd(const float* center, int H, int W, float conf_threshold, int32_t* centers, int cap, int32_t* n_centers,
                            void* ws, size_t ws_bytes, void* stream) {
  if (!center || !centers || !n_centers || H < 1 || W < 1 || cap < 1 || !ws) return SF_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t n = (size_t)H * W, a = a256e(n * 4);
  size_t tb = 0;
  (void)rocprim::exclusive_scan(nullptr, tb, (const int*)nullptr, (int*)nullptr, 0, n, rocprim::plus<int>(), st);
  if (ws_bytes < 2 * a + a256e(tb)) return SF_ERR_WORKSPACE;
  char* p = static_cast<char*>(ws);
  int* flag = reinterpret_cast<int*>(p);
  int* scan = reinterpret_cast<int*>(p + a);
  void* tmp = p + 2 * a;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipLaunchKernelGGL(center_flag_kernel, grid, block, 0, st, center, H, W, conf_threshold, flag);
  if (rocprim::exclusive_scan(tmp, tb, (const int*)flag, scan, 0, n, rocprim::plus<int>(), st) != hipSuccess) return SF_ERR_LAUNCH;
  hipLaunchKernelGGL(center_compact_kernel, grid, block, 0, st, flag, scan, H, W, cap, centers, n_centers);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

int sf_group_pixels_fwd(const int32_t* centers, int n_centers, const float* offsets, const uint8_t* foreground, int H, int W,
                        int64_t* instance, void* stream) {
  if (!centers || !offsets || !foreground || !instance || n_centers < 1 || H < 1 || W < 1) return SF_ERR_INVALID;
  hipLaunchKernelGGL(group_pixels_kernel, dim3((H * W + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), centers, n_centers,
                     offsets, foreground, H, W, reinterpret_cast<long long*>(instance));
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

int sf_instance_moments_fwd(const int64_t* instance, const float* flow, int F, int H, int W, int max_id, int64_t* pos_sums,
                            int64_t* warped_fx, int32_t* counts, void* stream) {
  if (!instance || !pos_sums || !counts || F < 1 || H < 1 || W < 1 || max_id < 0 || (warped_fx && !flow)) return SF_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t slots = (size_t)F * (max_id + 1);
  if (hipMemsetAsync(pos_sums, 0, slots * 2 * sizeof(int64_t), st) != hipSuccess) return SF_ERR_LAUNCH;
  if (warped_fx && hipMemsetAsync(warped_fx, 0, slots * 2 * sizeof(int64_t), st) != hipSuccess) return SF_ERR_LAUNCH;
  if (hipMemsetAsync(counts, 0, slots * sizeof(int32_t), st) != hipSuccess) return SF_ERR_LAUNCH;
  const long total = (long)F * H * W;
  hipLaunchKernelGGL(instance_moments_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const long long*>(instance), flow, F, H, W, max_id, reinterpret_cast<unsigned long long*>(pos_sums),
                     reinterpret_cast<unsigned long long*>(warped_fx), counts);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

int sf_confusion_frames_fwd(const int64_t* a, const int64_t* b, long n_per_frame, int F, int K, int64_t* out, int32_t* bad, void* stream) {
  if (!out || !bad || K < 1 || F < 1 || n_per_frame < 0) return SF_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out, 0, (size_t)F * K * K * sizeof(int64_t), st) != hipSuccess) return SF_ERR_LAUNCH;
  if (hipMemsetAsync(bad, 0, sizeof(int32_t), st) != hipSuccess) return SF_ERR_LAUNCH;
  if (n_per_frame == 0) return SF_OK;
  if (!a || !b) return SF_ERR_INVALID;
  long blocks = (n_per_frame * F + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(confusion_frames_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const long long*>(a),
                     reinterpret_cast<const long long*>(b), n_per_frame, F, K, reinterpret_cast<unsigned long long*>(out), bad);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

int sf_warp_affine_fwd(const float* x, const float* theta, int B, int C, int H, int W, int bilinear, float* out, void* stream) {
  if (!x || !theta || !out || B < 1 || C < 1 || H < 1 || W < 1) return SF_ERR_INVALID;
  const long total = (long)B * H * W;
  hipLaunchKernelGGL(warp_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, theta, B,
                     C, H, W, bilinear, out);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

}  // extern "C"
