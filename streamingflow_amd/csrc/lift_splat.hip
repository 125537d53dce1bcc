// Camera lift-splat voxel pooling for gfx950 + its C ABI.
//
// Reference: streamingflow/models/streamingflow.py:342-428 (bev_pool, projection_to_birds_eye_view),
// mmdet3d/ops/bev_pool/bev_pool.py:85-98 and src/bev_pool_cuda.cu:20-42.  The reference quantises
// the float geometry to cells on the host side of the op, filters, argsorts int64 ranks, gathers
// the [n, C] feature matrix into sorted order, builds interval tables with boolean indexing and
// only then runs a CUDA kernel (one thread per (interval, channel)).  Here:
//
//   lift_quantise_*   one thread per frustum point: cell id (or a sentinel for points outside the
//                     grid), either from a geometry tensor or straight from the camera rig
//                     (3x4 affine per camera/frame, no geometry tensor at all)
//   rocprim radix sort (stable) of (cell id, point id): points of a cell end up contiguous and in
//                     ascending point order -> the fp32 sum order is fixed and reproducible
//   lift_cell_start   CSR row starts by binary search in the sorted keys (covers empty cells, so
//                     the pooling kernel writes every output element: no memset, no atomics)
//   lift_pool_kernel  one lane per (cell, channel): sequential fp32 sum over the cell's points of
//                     either x[p][c] (materialised, drop-in) or depth_prob[p] * feat[ray(p)][c]
//                     (the depth (x) feature outer product of streamingflow.py:305-307 is never
//                     materialised: 124 MB per frame at the shipped size), optionally followed by
//                     the temporal blend  out = prev * discount + sum  (streamingflow.py:419)
//
// HBM-bound integer/gather work: coalesced 256-B feature rows (C = 64 lanes of a wave read one
// row), broadcast loads of the point list, 4 gathers in flight per lane.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/sfnative.h"

namespace sf {

struct LiftGrid {
  float lo[3], res[3];
  int X, Y, Z;
};

// streamingflow.py:353: ((g - (start - res/2)) / res).long()   [fp32 sub, IEEE fp32 divide, trunc]
// :358-366: keep 0 <= idx < dim.  trunc(q) >= 0  <=>  q > -1 ;  trunc(q) < X  <=>  q < X.
__device__ __forceinline__ unsigned cell_of(float gx, float gy, float gz, const LiftGrid& G, int b, unsigned sentinel,
                                            int* ix, int* iy, int* iz) {
  const float qx = __fdiv_rn(__fsub_rn(gx, G.lo[0]), G.res[0]);
  const float qy = __fdiv_rn(__fsub_rn(gy, G.lo[1]), G.res[1]);
  const float qz = __fdiv_rn(__fsub_rn(gz, G.lo[2]), G.res[2]);
  const bool ok = (qx > -1.f) && (qx < (float)G.X) && (qy > -1.f) && (qy < (float)G.Y) && (qz > -1.f) && (qz < (float)G.Z);
  if (!ok) { *ix = *iy = *iz = -1; return sentinel; }
  *ix = (int)qx; *iy = (int)qy; *iz = (int)qz;
  return (unsigned)(((b * G.Z + *iz) * G.X + *ix) * G.Y + *iy);
}

__global__ void lift_quantise_geom_kernel(const float* __restrict__ geom, int n, int pts_per_batch, LiftGrid G, unsigned sentinel,
                                          unsigned* __restrict__ key, unsigned* __restrict__ val, int* __restrict__ coords) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float gx = geom[3 * (size_t)p], gy = geom[3 * (size_t)p + 1], gz = geom[3 * (size_t)p + 2];
  const int b = p / pts_per_batch;
  int ix, iy, iz;
  key[p] = cell_of(gx, gy, gz, G, b, sentinel, &ix, &iy, &iz);
  val[p] = (unsigned)p;
  if (coords) {
    int4 c = make_int4(ix, iy, iz, ix < 0 ? -1 : b);
    *reinterpret_cast<int4*>(coords + 4 * (size_t)p) = c;
  }
}

// Geometry straight from the rig: point p = ((cam*D + d)*fH + h)*fW + w of batch element b;
// ego position = A[b*n_cam + cam] (3x4, row major) applied to (u*depth, v*depth, depth, 1) with
// u = us[w], v = vs[h], depth = ds[d] (streamingflow.py:149-168, :277-292 and the ego-motion
// warps of :386-396 composed on the host in float64).
__global__ void lift_quantise_rig_kernel(const float* __restrict__ A, const float* __restrict__ us, const float* __restrict__ vs,
                                         const float* __restrict__ ds, int n, int n_cam, int D, int fH, int fW, LiftGrid G,
                                         unsigned sentinel, unsigned* __restrict__ key, unsigned* __restrict__ val) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int w = p % fW;
  int t = p / fW;
  const int h = t % fH; t /= fH;
  const int d = t % D; t /= D;          // t = b*n_cam + cam
  const float dep = ds[d];
  const float px = __fmul_rn(us[w], dep), py = __fmul_rn(vs[h], dep);
  const float* a = A + 12 * (size_t)t;
  const float gx = fmaf(a[0], px, fmaf(a[1], py, fmaf(a[2], dep, a[3])));
  const float gy = fmaf(a[4], px, fmaf(a[5], py, fmaf(a[6], dep, a[7])));
  const float gz = fmaf(a[8], px, fmaf(a[9], py, fmaf(a[10], dep, a[11])));
  int ix, iy, iz;
  key[p] = cell_of(gx, gy, gz, G, t / n_cam, sentinel, &ix, &iy, &iz);
  val[p] = (unsigned)p;
}

// integer coordinates (x, y, z, b) -> cell id (bev_pool.py:88-93 builds an int64 rank for the same purpose)
__global__ void lift_key_coords_kernel(const int* __restrict__ coords, int n, int B, int Z, int X, int Y, unsigned sentinel,
                                       unsigned* __restrict__ key, unsigned* __restrict__ val) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (size_t)p);
  const bool ok = c.x >= 0 && c.x < X && c.y >= 0 && c.y < Y && c.z >= 0 && c.z < Z && c.w >= 0 && c.w < B;
  key[p] = ok ? (unsigned)(((c.w * Z + c.z) * X + c.x) * Y + c.y) : sentinel;
  val[p] = (unsigned)p;
}

// cell_start[c] = first position in the sorted key list whose key is >= c, c = 0..ncells
__global__ void lift_cell_start_kernel(const unsigned* __restrict__ keys, int n, int ncells, int* __restrict__ cell_start) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > ncells) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < (unsigned)c) lo = mid + 1; else hi = mid;
  }
  cell_start[c] = lo;
}

// One wave per (cell, 64-channel chunk); lane = channel.  The cell's point list is read 64 entries at a
// time, one entry per lane (coalesced), together with everything that depends on the point only
// (MODE 1: its depth probability and its ray index — two integer divisions done once per point
// instead of once per (point, channel)); the entries are then broadcast one by one with
// v_readlane and every lane gathers its channel of that point's feature row (one 256-B line per
// wave).  Adds are sequential in list order: the sum is reproducible and bit-identical to a scalar
// loop.  MODE 0: v = x[p][c].  MODE 1: v = depth[p] * feat[ray(p)][c], ray(p) = (p / (D*fHW))*fHW + p % fHW.
#ifndef LIFT_INFLIGHT
#define LIFT_INFLIGHT 16
#endif
template <int MODE>
__global__ __launch_bounds__(256) void lift_pool_kernel(const int* __restrict__ order, const int* __restrict__ cell_start, int ncells,
                                                        int C, const float* __restrict__ x, const float* __restrict__ depth,
                                                        const float* __restrict__ feat, int D, int fHW,
                                                        const float* __restrict__ prev, float discount, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int chunks = (C + 63) >> 6;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int cell = (int)(wave / chunks);
  if (cell >= ncells) return;                       // wave-uniform
  const int c = (int)(wave - (long)cell * chunks) * 64 + lane;
  const bool act = c < C;
  const int cc = act ? c : 0;
  const int s = cell_start[cell], e = cell_start[cell + 1];
  const int DF = D * fHW;
  const float* __restrict__ table = (MODE == 0) ? x : feat;
  float acc = 0.f;
  for (int base = s; base < e; base += 64) {
    const int m = (e - base) < 64 ? (e - base) : 64;
    int row = 0;
    float dv = 0.f;
    if (lane < m) {
      const int p = order[base + lane];
      if (MODE == 0) row = p;
      else {
        dv = depth[p];
        row = (p / DF) * fHW + (p % fHW);
      }
    }
    for (int j = 0; j < m; j += LIFT_INFLIGHT) {    // gathers in flight; padded slots contribute +0 (identity)
      float v[LIFT_INFLIGHT];
#pragma unroll
      for (int i = 0; i < LIFT_INFLIGHT; ++i) {
        const int jj = (j + i) < m ? (j + i) : j;
        const int r = __builtin_amdgcn_readlane(row, jj);
        const float t = table[(size_t)r * C + cc];
        if (MODE == 0) v[i] = t;
        else v[i] = __fmul_rn(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), jj)), t);
        if ((j + i) >= m) v[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < LIFT_INFLIGHT; ++i) acc = __fadd_rn(acc, v[i]);
    }
  }
  if (!act) return;
  const size_t o = (size_t)cell * C + c;
  if (prev) acc = __fadd_rn(__fmul_rn(prev[o], discount), acc);      // streamingflow.py:419
  out[o] = acc;
}

// mmdet3d/ops/bev_pool/src/bev_pool_cuda.cu:20-42, coalesced the same way (adjacent lanes = channels)
__global__ void bev_pool_intervals_kernel(int d, int h, int w, int c, int n_intervals, const float* __restrict__ x,
                                          const int* __restrict__ geom, const int* __restrict__ starts,
                                          const int* __restrict__ lengths, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(idx / c);
  const int cc = (int)(idx - (long)i * c);
  if (i >= n_intervals) return;
  const int s = starts[i], L = lengths[i];
  const int* g = geom + 4 * (size_t)s;
  const float* px = x + (size_t)s * c + cc;
  float acc = 0.f;
  int k = 0;
  for (; k + 4 <= L; k += 4) {
    const float v0 = px[(size_t)k * c], v1 = px[(size_t)(k + 1) * c], v2 = px[(size_t)(k + 2) * c], v3 = px[(size_t)(k + 3) * c];
    acc = __fadd_rn(acc, v0); acc = __fadd_rn(acc, v1); acc = __fadd_rn(acc, v2); acc = __fadd_rn(acc, v3);
  }
  for (; k < L; ++k) acc = __fadd_rn(acc, px[(size_t)k * c]);
  out[((((size_t)g[3] * d + g[2]) * h + g[0]) * w + g[1]) * c + cc] = acc;
}

// softmax over the depth axis of [rows][D][fHW] (streamingflow.py:304): one thread per (row, hw).  DT > 0: D == DT (a multiple of 16,
// the shipped 48 bins among them): the logits of a ray are read once, all loads in flight together, and kept in registers — no
// per-bin bounds test (64 of them as scalar conditions cost 34 SGPR spills).  DT == 0: any D, three passes over the ray.
template <int DT>
__global__ void depth_softmax_kernel(const float* __restrict__ logits, float* __restrict__ prob, int rows, int D, int fHW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * fHW) return;
  const int r = i / fHW, hw = i - r * fHW;
  const float* in = logits + (size_t)r * D * fHW + hw;
  float* o = prob + (size_t)r * D * fHW + hw;
  if constexpr (DT > 0) {
    float v[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) v[d] = in[(size_t)d * fHW];
    float m = -INFINITY;
#pragma unroll
    for (int d = 0; d < DT; ++d) m = fmaxf(m, v[d]);
    float sum = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      v[d] = expf(v[d] - m);
      sum += v[d];
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) o[(size_t)d * fHW] = v[d] / sum;
  } else {
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, in[(size_t)d * fHW]);
    float sum = 0.f;
    for (int d = 0; d < D; ++d) sum += expf(in[(size_t)d * fHW] - m);
    for (int d = 0; d < D; ++d) o[(size_t)d * fHW] = expf(in[(size_t)d * fHW] - m) / sum;
  }
}

inline int key_bits(unsigned sentinel) {
  int b = 1;
  while (b < 32 && (sentinel >> b)) ++b;
  return b;
}
inline size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

// workspace layout: key[n] | val[n] | key_sorted[n] | rocprim temp
struct IndexWs { unsigned *key, *val, *key_sorted; void* tmp; size_t tmp_bytes; };

inline size_t sort_temp_bytes(int n, int bits, hipStream_t st) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr,
                                  (unsigned*)nullptr, (size_t)n, 0u, (unsigned)bits, st);
  return bytes;
}

int finish_index(IndexWs& W, int n, int ncells, int* order, int* cell_start, hipStream_t st) {
  const int bits = key_bits((unsigned)ncells);
  if (rocprim::radix_sort_pairs(W.tmp, W.tmp_bytes, (const unsigned*)W.key, W.key_sorted, (const unsigned*)W.val,
                                reinterpret_cast<unsigned*>(order), (size_t)n, 0u, (unsigned)bits, st) != hipSuccess)
    return SF_ERR_LAUNCH;
  hipLaunchKernelGGL(lift_cell_start_kernel, dim3((ncells + 1 + 255) / 256), dim3(256), 0, st, W.key_sorted, n, ncells, cell_start);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

int carve(void* ws, size_t ws_bytes, int n, int ncells, hipStream_t st, IndexWs* W) {
  const size_t a = align256((size_t)n * 4);
  const int bits = key_bits((unsigned)ncells);
  const size_t tb = sort_temp_bytes(n, bits, st);
  if (!ws || ws_bytes < 3 * a + align256(tb)) return SF_ERR_WORKSPACE;
  char* p = static_cast<char*>(ws);
  W->key = reinterpret_cast<unsigned*>(p);
  W->val = reinterpret_cast<unsigned*>(p + a);
  W->key_sorted = reinterpret_cast<unsigned*>(p + 2 * a);
  W->tmp = p + 3 * a;
  W->tmp_bytes = tb;
  return SF_OK;
}

bool grid_ok(const float* lo, const float* res, const int32_t* dim, int nb, LiftGrid* G, long* ncells) {
  if (!lo || !res || !dim || nb < 1) return false;
  for (int i = 0; i < 3; ++i) { G->lo[i] = lo[i]; G->res[i] = res[i]; if (!(res[i] > 0.f) || dim[i] < 1) return false; }
  G->X = dim[0]; G->Y = dim[1]; G->Z = dim[2];
  *ncells = (long)nb * dim[0] * dim[1] * dim[2];
  return *ncells < (1L << 30);
}

}  // namespace sf

using namespace sf;

extern "C" {

int sf_bev_pool_fwd(const float* x, const int32_t* geom_feats, const int32_t* interval_lengths, const int32_t* interval_starts,
                    int n, int c, int n_intervals, int b, int d, int h, int w, float* out, void* stream) {
  if (!out || b < 1 || d < 1 || h < 1 || w < 1 || c < 1 || n < 0 || n_intervals < 0) return SF_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out, 0, (size_t)b * d * h * w * c * sizeof(float), st) != hipSuccess) return SF_ERR_LAUNCH;   // bev_pool.cpp:42
  if (n_intervals == 0) return SF_OK;
  if (!x || !geom_feats || !interval_lengths || !interval_starts) return SF_ERR_INVALID;
  const long total = (long)n_intervals * c;
  hipLaunchKernelGGL(bev_pool_intervals_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d, h, w, c, n_intervals, x,
                     geom_feats, interval_starts, interval_lengths, out);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

size_t sf_lift_index_ws_bytes(int n_points, int n_cells) {
  if (n_points < 1 || n_cells < 1) return 0;
  const size_t tb = sort_temp_bytes(n_points, key_bits((unsigned)n_cells), nullptr);
  return 3 * align256((size_t)n_points * 4) + align256(tb) + 256;
}

int sf_lift_index_fwd(const float* geom, int n_points, int n_batch, const float* lo, const float* res, const int32_t* dim,
                      int32_t* coords, int32_t* order, int32_t* cell_start, void* ws, size_t ws_bytes, void* stream) {
  LiftGrid G;
  long ncells;
  if (!geom || !order || !cell_start || n_points < 1 || !grid_ok(lo, res, dim, n_batch, &G, &ncells) || n_points % n_batch)
    return SF_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  IndexWs W;
  int rc = carve(ws, ws_bytes, n_points, (int)ncells, st, &W);
  if (rc != SF_OK) return rc;
  hipLaunchKernelGGL(lift_quantise_geom_kernel, dim3((n_points + 255) / 256), dim3(256), 0, st, geom, n_points, n_points / n_batch,
                     G, (unsigned)ncells, W.key, W.val, coords);
  if (hipGetLastError() != hipSuccess) return SF_ERR_LAUNCH;
  return finish_index(W, n_points, (int)ncells, order, cell_start, st);
}

int sf_lift_index_coords_fwd(const int32_t* coords, int n_points, int B, int Z, int X, int Y, int32_t* order, int32_t* cell_start,
                             void* ws, size_t ws_bytes, void* stream) {
  if (!coords || !order || !cell_start || n_points < 1 || B < 1 || Z < 1 || X < 1 || Y < 1) return SF_ERR_INVALID;
  const long ncells = (long)B * Z * X * Y;
  if (ncells >= (1L << 30)) return SF_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  IndexWs W;
  int rc = carve(ws, ws_bytes, n_points, (int)ncells, st, &W);
  if (rc != SF_OK) return rc;
  hipLaunchKernelGGL(lift_key_coords_kernel, dim3((n_points + 255) / 256), dim3(256), 0, st, coords, n_points, B, Z, X, Y,
                     (unsigned)ncells, W.key, W.val);
  if (hipGetLastError() != hipSuccess) return SF_ERR_LAUNCH;
  return finish_index(W, n_points, (int)ncells, order, cell_start, st);
}

int sf_lift_index_rig_fwd(const float* affine, const float* us, const float* vs, const float* ds, int n_batch, int n_cam, int D,
                          int fH, int fW, const float* lo, const float* res, const int32_t* dim, int32_t* order,
                          int32_t* cell_start, void* ws, size_t ws_bytes, void* stream) {
  LiftGrid G;
  long ncells;
  if (!affine || !us || !vs || !ds || !order || !cell_start || n_cam < 1 || D < 1 || fH < 1 || fW < 1 ||
      !grid_ok(lo, res, dim, n_batch, &G, &ncells))
    return SF_ERR_INVALID;
  const long np = (long)n_batch * n_cam * D * fH * fW;
  if (np >= (1L << 31)) return SF_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  IndexWs W;
  int rc = carve(ws, ws_bytes, (int)np, (int)ncells, st, &W);
  if (rc != SF_OK) return rc;
  hipLaunchKernelGGL(lift_quantise_rig_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, affine, us, vs, ds, (int)np, n_cam,
                     D, fH, fW, G, (unsigned)ncells, W.key, W.val);
  if (hipGetLastError() != hipSuccess) return SF_ERR_LAUNCH;
  return finish_index(W, (int)np, (int)ncells, order, cell_start, st);
}

int sf_lift_pool_fwd(const float* x, const int32_t* order, const int32_t* cell_start, int n_cells, int C, const float* prev,
                     float discount, float* out, void* stream) {
  if (!x || !order || !cell_start || !out || n_cells < 1 || C < 1) return SF_ERR_INVALID;
  const long waves = (long)n_cells * ((C + 63) / 64);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(lift_pool_kernel<0>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), order, cell_start, n_cells, C, x, nullptr, nullptr, 1, 1, prev, discount, out);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

int sf_lift_pool_fused_fwd(const float* feat, const float* depth_prob, int D, int fHW, const int32_t* order,
                           const int32_t* cell_start, int n_cells, int C, const float* prev, float discount, float* out,
                           void* stream) {
  if (!feat || !depth_prob || !order || !cell_start || !out || n_cells < 1 || C < 1 || D < 1 || fHW < 1) return SF_ERR_INVALID;
  const long waves = (long)n_cells * ((C + 63) / 64);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(lift_pool_kernel<1>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), order, cell_start, n_cells, C, nullptr, depth_prob, feat, D, fHW, prev, discount,
                     out);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

int sf_depth_softmax_fwd(const float* logits, float* prob, int rows, int D, int fHW, void* stream) {
  if (!logits || !prob || rows < 1 || D < 1 || fHW < 1) return SF_ERR_INVALID;
  const long total = (long)rows * fHW;
  const dim3 grid((unsigned)((total + 63) / 64));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (D) {      // same arithmetic and summation order in every instantiation
    case 16: hipLaunchKernelGGL(depth_softmax_kernel<16>, grid, dim3(64), 0, st, logits, prob, rows, D, fHW); break;
    case 32: hipLaunchKernelGGL(depth_softmax_kernel<32>, grid, dim3(64), 0, st, logits, prob, rows, D, fHW); break;
    case 48: hipLaunchKernelGGL(depth_softmax_kernel<48>, grid, dim3(64), 0, st, logits, prob, rows, D, fHW); break;
    case 64: hipLaunchKernelGGL(depth_softmax_kernel<64>, grid, dim3(64), 0, st, logits, prob, rows, D, fHW); break;
    default: hipLaunchKernelGGL(depth_softmax_kernel<0>, grid, dim3(64), 0, st, logits, prob, rows, D, fHW); break;
  }
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

}  // extern "C"
