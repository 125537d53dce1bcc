// Index building for sparse 3-D convolutions on gfx950 (the SparseEncoder) + C ABI.
//
// spconv 1.x (mmdet3d/ops/spconv) builds, per convolution, "indice pairs" (input row, output row) for each
// of the 27 kernel offsets with a hash table, then runs gather -> GEMM -> scatter-add per offset.  Here the
// convolution itself is the implicit-GEMM kernel of conv_igemm.hip reading a neighbour table
// nbr[output row][kernel tap] -> input row (or -1), so the index work is:
//   sp_key_kernel          (batch, x, y, z) -> linear key
//   rocprim radix sort     of (key, row): membership / lookup structure (binary search)
//   sp_table_kernel        one thread per (output row, tap): coordinates of the input site that tap reads,
//                          bounds check, binary search, row index or -1      (submanifold and strided alike)
//   strided convs only:    sp_candidates_kernel (every (input site, tap) names the output site it feeds, or a
//                          sentinel) -> radix sort -> head flags -> exclusive scan -> compaction = the sorted,
//                          unique output sites; their count goes back to the host (one sync per strided conv)
//   sp_to_dense_kernel     structure.py dense() + the permute/view of sparse_encoder.py:133-137, written
//                          directly as the NHWC BEV tensor [b][x][y][c*D + z]
// Integer work, HBM/latency-bound; deterministic (sorts + scans, no atomics).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/sfnative.h"

namespace sf {

struct SpGeom {
  int shape[3];      // input spatial shape (X, Y, Z)
  int oshape[3];     // output spatial shape
  int k[3], s[3], p[3];
  int subm;
};

__device__ __forceinline__ unsigned sp_key(int b, int x, int y, int z, const int* sh) {
  return (((unsigned)b * (unsigned)sh[0] + (unsigned)x) * (unsigned)sh[1] + (unsigned)y) * (unsigned)sh[2] + (unsigned)z;
}

__global__ void sp_key_kernel(const int* __restrict__ coords, int n, SpGeom G, unsigned* __restrict__ key, unsigned* __restrict__ val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (size_t)i);
  key[i] = sp_key(c.x, c.y, c.z, c.w, G.shape);
  val[i] = (unsigned)i;
}

__device__ __forceinline__ int sp_find(const unsigned* __restrict__ keys, int n, unsigned v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < v) lo = mid + 1; else hi = mid;
  }
  return (lo < n && keys[lo] == v) ? lo : -1;
}

// nbr[j][t]: input row read by output site j through kernel tap t = (kx*KY + ky)*KZ + kz
__global__ void sp_table_kernel(const int* __restrict__ out_coords, int n_out, const unsigned* __restrict__ in_keys,
                                const unsigned* __restrict__ in_rows, int n_in, SpGeom G, int* __restrict__ nbr) {
  const int ntaps = G.k[0] * G.k[1] * G.k[2];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_out * ntaps) return;
  const int j = (int)(idx / ntaps), t = (int)(idx - (long)j * ntaps);
  const int kz = t % G.k[2], ky = (t / G.k[2]) % G.k[1], kx = t / (G.k[2] * G.k[1]);
  const int4 c = *reinterpret_cast<const int4*>(out_coords + 4 * (size_t)j);
  int q[3];
  if (G.subm) {
    q[0] = c.y + kx - G.k[0] / 2; q[1] = c.z + ky - G.k[1] / 2; q[2] = c.w + kz - G.k[2] / 2;
  } else {
    q[0] = c.y * G.s[0] - G.p[0] + kx; q[1] = c.z * G.s[1] - G.p[1] + ky; q[2] = c.w * G.s[2] - G.p[2] + kz;
  }
  int r = -1;
  if (q[0] >= 0 && q[0] < G.shape[0] && q[1] >= 0 && q[1] < G.shape[1] && q[2] >= 0 && q[2] < G.shape[2]) {
    const int pos = sp_find(in_keys, n_in, sp_key(c.x, q[0], q[1], q[2], G.shape));
    if (pos >= 0) r = (int)in_rows[pos];
  }
  nbr[idx] = r;
}

// strided conv: the output site input i feeds through tap t (o = (p_in + pad - k) / stride when divisible)
__global__ void sp_candidates_kernel(const int* __restrict__ coords, int n, SpGeom G, unsigned sentinel, unsigned* __restrict__ cand) {
  const int ntaps = G.k[0] * G.k[1] * G.k[2];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * ntaps) return;
  const int i = (int)(idx / ntaps), t = (int)(idx - (long)i * ntaps);
  const int kk[3] = {t / (G.k[2] * G.k[1]), (t / G.k[2]) % G.k[1], t % G.k[2]};
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * (size_t)i);
  const int pin[3] = {c.y, c.z, c.w};
  int o[3];
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int num = pin[a] + G.p[a] - kk[a];
    ok = ok && num >= 0 && (num % G.s[a]) == 0;
    o[a] = num / G.s[a];
    ok = ok && o[a] < G.oshape[a];
  }
  cand[idx] = ok ? sp_key(c.x, o[0], o[1], o[2], G.oshape) : sentinel;
}

__global__ void sp_head_kernel(const unsigned* __restrict__ keys, long n, unsigned sentinel, int* __restrict__ flag) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned k = keys[j];
  flag[j] = (k != sentinel && (j == 0 || keys[j - 1] != k)) ? 1 : 0;
}

__global__ void sp_compact_kernel(const unsigned* __restrict__ keys, const int* __restrict__ flag, const int* __restrict__ scan, long n,
                                  SpGeom G, int cap, int* __restrict__ out_coords, int* __restrict__ n_out) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0 && n_out) n_out[0] = n > 0 ? scan[n - 1] + flag[n - 1] : 0;
  if (j >= n || !flag[j]) return;
  const int r = scan[j];
  if (r >= cap) return;
  unsigned k = keys[j];
  const int z = (int)(k % (unsigned)G.oshape[2]); k /= (unsigned)G.oshape[2];
  const int y = (int)(k % (unsigned)G.oshape[1]); k /= (unsigned)G.oshape[1];
  const int x = (int)(k % (unsigned)G.oshape[0]); k /= (unsigned)G.oshape[0];
  *reinterpret_cast<int4*>(out_coords + 4 * (size_t)r) = make_int4((int)k, x, y, z);
}

// out[b][x][y][c*D + z] = feats[row][c]   (zero elsewhere: the caller's memset)
__global__ void sp_to_dense_kernel(const float* __restrict__ feats, const int* __restrict__ coords, int n, int C, int X, int Y, int D,
                                   float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * C) return;
  const int i = (int)(idx / C), c = (int)(idx - (long)i * C);
  const int4 q = *reinterpret_cast<const int4*>(coords + 4 * (size_t)i);
  out[((((size_t)q.x * X + q.y) * Y + q.z) * C + c) * D + q.w] = feats[idx];
}

inline size_t a256s(size_t n) { return (n + 255) & ~size_t(255); }
inline int sp_bits(unsigned v) {
  int b = 1;
  while (b < 32 && (v >> b)) ++b;
  return b;
}
inline size_t sp_sort_pairs_tmp(size_t n, hipStream_t st) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr,
                                  (unsigned*)nullptr, n, 0u, 32u, st);
  return bytes;
}
inline size_t sp_sort_keys_tmp(size_t n, hipStream_t st) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, n, 0u, 32u, st);
  return bytes;
}
inline size_t sp_scan_tmp(size_t n, hipStream_t st) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const int*)nullptr, (int*)nullptr, 0, n, rocprim::plus<int>(), st);
  return bytes;
}

bool sp_geom(const int32_t* shape, const int32_t* k, const int32_t* s, const int32_t* p, int subm, int batch, SpGeom* G, double* cells_in,
             double* cells_out) {
  if (!shape || !k || batch < 1) return false;
  *cells_in = batch;
  *cells_out = batch;
  for (int a = 0; a < 3; ++a) {
    G->shape[a] = shape[a]; G->k[a] = k[a];
    G->s[a] = subm ? 1 : (s ? s[a] : 1);
    G->p[a] = subm ? 0 : (p ? p[a] : 0);
    if (shape[a] < 1 || k[a] < 1 || G->s[a] < 1 || G->p[a] < 0) return false;
    G->oshape[a] = subm ? shape[a] : (shape[a] + 2 * G->p[a] - (k[a] - 1) - 1) / G->s[a] + 1;      // spconv/ops.py:19-33
    if (G->oshape[a] < 1) return false;
    *cells_in *= shape[a];
    *cells_out *= G->oshape[a];
  }
  G->subm = subm;
  return *cells_in < 4294967295.0 && *cells_out < 4294967295.0;
}

}  // namespace sf

using namespace sf;

extern "C" {

size_t sf_sparse_index_ws_bytes(int n_in, int ntaps) {
  if (n_in < 1 || ntaps < 1) return 0;
  const size_t nc = (size_t)n_in * ntaps;
  const size_t a = a256s((size_t)n_in * 4), c = a256s(nc * 4);
  return 4 * a + a256s(sp_sort_pairs_tmp(n_in, nullptr)) + 4 * c + a256s(sp_sort_keys_tmp(nc, nullptr)) + a256s(sp_scan_tmp(nc, nullptr)) + 512;
}

// sorted (key, row) of the input sites into ws; returns pointers through *keys / *rows
static int sp_sort_inputs(const int32_t* coords, int n, const SpGeom& G, char*& p, size_t& left, hipStream_t st, unsigned** keys,
                          unsigned** rows) {
  const size_t a = a256s((size_t)n * 4);
  const size_t tb = sp_sort_pairs_tmp(n, st);
  if (left < 4 * a + a256s(tb)) return SF_ERR_WORKSPACE;
  unsigned* key = reinterpret_cast<unsigned*>(p);
  unsigned* val = reinterpret_cast<unsigned*>(p + a);
  *keys = reinterpret_cast<unsigned*>(p + 2 * a);
  *rows = reinterpret_cast<unsigned*>(p + 3 * a);
  void* tmp = p + 4 * a;
  p += 4 * a + a256s(tb);
  left -= 4 * a + a256s(tb);
  hipLaunchKernelGGL(sp_key_kernel, dim3((n + 255) / 256), dim3(256), 0, st, coords, n, G, key, val);
  size_t t2 = tb;
  double cells = 1.0;
  for (int a2 = 0; a2 < 3; ++a2) cells *= G.shape[a2];
  if (rocprim::radix_sort_pairs(tmp, t2, (const unsigned*)key, *keys, (const unsigned*)val, *rows, (size_t)n, 0u, 32u, st) != hipSuccess)
    return SF_ERR_LAUNCH;
  return SF_OK;
}

int sf_sparse_table_fwd(const int32_t* in_coords, int n_in, const int32_t* out_coords, int n_out, int batch, const int32_t* shape,
                        const int32_t* ksize, const int32_t* stride, const int32_t* padding, int subm, int32_t* nbr, void* ws,
                        size_t ws_bytes, void* stream) {
  SpGeom G;
  double ci, co;
  if (!in_coords || !out_coords || !nbr || n_in < 1 || n_out < 0 || !sp_geom(shape, ksize, stride, padding, subm, batch, &G, &ci, &co) || !ws)
    return SF_ERR_INVALID;
  if (n_out == 0) return SF_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* p = static_cast<char*>(ws);
  size_t left = ws_bytes;
  unsigned *keys, *rows;
  int rc = sp_sort_inputs(in_coords, n_in, G, p, left, st, &keys, &rows);
  if (rc != SF_OK) return rc;
  const long total = (long)n_out * G.k[0] * G.k[1] * G.k[2];
  hipLaunchKernelGGL(sp_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out_coords, n_out, keys, rows, n_in, G, nbr);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

int sf_sparse_out_sites_fwd(const int32_t* in_coords, int n_in, int batch, const int32_t* shape, const int32_t* ksize,
                            const int32_t* stride, const int32_t* padding, int32_t* out_coords, int cap, int32_t* n_out, void* ws,
                            size_t ws_bytes, void* stream) {
  SpGeom G;
  double ci, co;
  if (!in_coords || !out_coords || !n_out || n_in < 1 || cap < 1 || !sp_geom(shape, ksize, stride, padding, 0, batch, &G, &ci, &co) || !ws)
    return SF_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ntaps = G.k[0] * G.k[1] * G.k[2];
  const size_t nc = (size_t)n_in * ntaps;
  const size_t c = a256s(nc * 4);
  const size_t tb_sort = sp_sort_keys_tmp(nc, st), tb_scan = sp_scan_tmp(nc, st);
  if (ws_bytes < 4 * c + a256s(tb_sort) + a256s(tb_scan)) return SF_ERR_WORKSPACE;
  char* p = static_cast<char*>(ws);
  unsigned* cand = reinterpret_cast<unsigned*>(p);
  unsigned* sorted = reinterpret_cast<unsigned*>(p + c);
  int* flag = reinterpret_cast<int*>(p + 2 * c);
  int* scan = reinterpret_cast<int*>(p + 3 * c);
  void* tmp_sort = p + 4 * c;
  void* tmp_scan = p + 4 * c + a256s(tb_sort);
  const unsigned sentinel = 0xFFFFFFFFu;
  const dim3 grid((unsigned)((nc + 255) / 256)), block(256);
  hipLaunchKernelGGL(sp_candidates_kernel, grid, block, 0, st, in_coords, n_in, G, sentinel, cand);
  size_t t2 = tb_sort;
  if (rocprim::radix_sort_keys(tmp_sort, t2, (const unsigned*)cand, sorted, nc, 0u, 32u, st) != hipSuccess) return SF_ERR_LAUNCH;
  hipLaunchKernelGGL(sp_head_kernel, grid, block, 0, st, sorted, (long)nc, sentinel, flag);
  t2 = tb_scan;
  if (rocprim::exclusive_scan(tmp_scan, t2, (const int*)flag, scan, 0, nc, rocprim::plus<int>(), st) != hipSuccess) return SF_ERR_LAUNCH;
  hipLaunchKernelGGL(sp_compact_kernel, grid, block, 0, st, sorted, flag, scan, (long)nc, G, cap, out_coords, n_out);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

int sf_sparse_to_dense_fwd(const float* feats, const int32_t* coords, int n, int C, int batch, int X, int Y, int D, float* out,
                           void* stream) {
  if (!out || batch < 1 || X < 1 || Y < 1 || D < 1 || C < 1 || n < 0) return SF_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out, 0, (size_t)batch * X * Y * C * D * sizeof(float), st) != hipSuccess) return SF_ERR_LAUNCH;
  if (n == 0) return SF_OK;
  if (!feats || !coords) return SF_ERR_INVALID;
  const long total = (long)n * C;
  hipLaunchKernelGGL(sp_to_dense_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, feats, coords, n, C, X, Y, D, out);
  return hipGetLastError() == hipSuccess ? SF_OK : SF_ERR_LAUNCH;
}

}  // extern "C"
