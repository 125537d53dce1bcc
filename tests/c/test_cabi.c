/* Plain C host of libsfnative.so (no Python, no torch): packs a reference-format convolution with eval-mode BatchNorm on
 * the device (sf_pack_conv), runs it through sf_conv2d_fwd and checks the result against a CPU loop.
 *   gcc -std=c99 -O1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/test_cabi.c \
 *       -Lstreamingflow_amd -lsfnative -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/streamingflow_amd -o build_r02/test_cabi
 * Exit code 0 = pass.  Needs the MI355X (tests/test_gpu_cabi_c.py builds and runs it). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sfnative.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define SFCK(x) do { int s_ = (x); if (s_ != SF_OK) { printf("libsfnative: %s at line %d\n", sf_status_string(s_), __LINE__); return 3; } } while (0)

static unsigned long long rng = 88172645463325252ULL;
static float frand(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 11) % 20001) / 10000.0f - 1.0f; }

int main(void) {
  enum { N = 2, H = 7, W = 5, CIN = 12, COUT = 20, K = 3 };
  static float w[COUT][CIN][K][K], cb[COUT], g[COUT], b[COUT], mu[COUT], var[COUT], x[N][H][W][CIN], y[N][H][W][COUT], ref[N][H][W][COUT];
  const float eps = 1e-3f;
  for (int o = 0; o < COUT; ++o) {
    cb[o] = frand(); g[o] = 1.0f + 0.5f * frand(); b[o] = frand(); mu[o] = frand(); var[o] = 0.5f + 0.4f * frand();
    for (int c = 0; c < CIN; ++c) for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) w[o][c][i][j] = 0.2f * frand();
  }
  for (int n = 0; n < N; ++n) for (int i = 0; i < H; ++i) for (int j = 0; j < W; ++j) for (int c = 0; c < CIN; ++c) x[n][i][j][c] = frand();
  /* CPU: conv 3x3 pad 1 + bias -> BatchNorm (eval) -> ReLU */
  for (int n = 0; n < N; ++n) for (int i = 0; i < H; ++i) for (int j = 0; j < W; ++j) for (int o = 0; o < COUT; ++o) {
    double acc = 0.0;
    for (int di = 0; di < K; ++di) for (int dj = 0; dj < K; ++dj) {
      const int ii = i + di - 1, jj = j + dj - 1;
      if (ii < 0 || ii >= H || jj < 0 || jj >= W) continue;
      for (int c = 0; c < CIN; ++c) acc += (double)w[o][c][di][dj] * x[n][ii][jj][c];
    }
    const double v = (acc + cb[o] - mu[o]) / sqrt((double)var[o] + eps) * g[o] + b[o];
    ref[n][i][j][o] = (float)(v > 0.0 ? v : 0.0);
  }
  if (sf_version() < 110) { printf("unexpected library version\n"); return 1; }
  /* ABI guard: the struct sizes THIS compiler saw against the library's; a stale size must be refused */
  if (sf_abi_check_header() != SF_OK) { printf("ABI mismatch: header %d, library %d\n", SF_ABI_VERSION, sf_abi_version()); return 1; }
  { size_t stale[SF_STRUCT_COUNT];
    for (int i = 0; i < SF_STRUCT_COUNT; ++i) stale[i] = sf_abi_sizeof(i);
    stale[SF_STRUCT_CONV_W] = 72;   /* the round-2 sf_conv_w */
    if (sf_abi_check(SF_ABI_VERSION, stale, SF_STRUCT_COUNT) != SF_ERR_INVALID || sf_abi_check(SF_ABI_VERSION - 1, NULL, 0) != SF_ERR_INVALID) return 1; }
  float *dw, *dcb, *dg, *db, *dmu, *dvar, *dx, *dy;
  void* blob;
  CK(hipMalloc((void**)&dw, sizeof(w))); CK(hipMalloc((void**)&dcb, sizeof(cb))); CK(hipMalloc((void**)&dg, sizeof(g)));
  CK(hipMalloc((void**)&db, sizeof(b))); CK(hipMalloc((void**)&dmu, sizeof(mu))); CK(hipMalloc((void**)&dvar, sizeof(var)));
  CK(hipMalloc((void**)&dx, sizeof(x))); CK(hipMalloc((void**)&dy, sizeof(y)));
  CK(hipMemcpy(dw, w, sizeof(w), hipMemcpyHostToDevice)); CK(hipMemcpy(dcb, cb, sizeof(cb), hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, g, sizeof(g), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b, sizeof(b), hipMemcpyHostToDevice));
  CK(hipMemcpy(dmu, mu, sizeof(mu), hipMemcpyHostToDevice)); CK(hipMemcpy(dvar, var, sizeof(var), hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, x, sizeof(x), hipMemcpyHostToDevice));
  const size_t nb = sf_pack_conv_bytes(COUT, CIN, K, K, 0);
  if (nb == 0) { printf("sf_pack_conv_bytes returned 0\n"); return 1; }
  CK(hipMalloc(&blob, nb));
  sf_conv_w cw;
  SFCK(sf_pack_conv(dw, dcb, NULL, dg, db, dmu, dvar, eps, COUT, CIN, K, K, CIN, 0, SF_ACT_RELU, 1, 1, -1, 0, blob, nb, &cw, NULL));
  if (cw.cout != COUT || cw.cout_pad != 32 || cw.cin_pad != 32 || cw.pad != 1 || !cw.scale || !cw.bias) { printf("unexpected packed descriptor\n"); return 1; }
  /* too small a blob and inconsistent BatchNorm arguments are rejected, not written through */
  if (sf_pack_conv(dw, dcb, NULL, dg, db, dmu, dvar, eps, COUT, CIN, K, K, CIN, 0, SF_ACT_RELU, 1, 1, -1, 0, blob, nb - 4, &cw, NULL) != SF_ERR_WORKSPACE) return 1;
  if (sf_pack_conv(dw, dcb, NULL, dg, NULL, dmu, dvar, eps, COUT, CIN, K, K, CIN, 0, SF_ACT_RELU, 1, 1, -1, 0, blob, nb, &cw, NULL) != SF_ERR_INVALID) return 1;
  SFCK(sf_pack_conv(dw, dcb, NULL, dg, db, dmu, dvar, eps, COUT, CIN, K, K, CIN, 0, SF_ACT_RELU, 1, 1, -1, 0, blob, nb, &cw, NULL));
  SFCK(sf_conv2d_fwd(&cw, dx, NULL, NULL, dy, N, H, W, 0, NULL));
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(y, dy, sizeof(y), hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (size_t i = 0; i < sizeof(y) / sizeof(float); ++i) {
    const double d = fabs((double)((float*)y)[i] - ((float*)ref)[i]);
    if (d > worst) worst = d;
  }
  printf("sf_pack_conv + sf_conv2d_fwd from C: max-abs error vs the CPU loop = %.3e\n", worst);
  return worst <= 1e-4 ? 0 : 1;
}
