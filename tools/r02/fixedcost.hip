// Calibration of the fixed costs of a short dependent kernel chain on MI355X (round 2 design input; not product code).
//   hipcc --offload-arch=gfx950 -O3 -o fixedcost tools/r02/fixedcost.hip && ./fixedcost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Big { float* p[4]; int a[320]; };   // ~1.3 KB kernarg like sf::ConvLaunch

__global__ void k_trivial(float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1.f;
}
__global__ void k_bigarg(const Big b) {
  const int i = b.a[blockIdx.y * 80 + 3];
  if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0][0] += (float)i;
}

// stamps: [launch][wg][8]
__global__ void k_stamp_copy(const float* __restrict__ in, float* __restrict__ out, unsigned long long* st, int n4, int erf_reps) {
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float4 v = make_float4(0, 0, 0, 0);
  if (tid < n4) v = reinterpret_cast<const float4*>(in)[tid];
  // force the wait
  float s = v.x + v.y + v.z + v.w;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < erf_reps; ++r) {
    v.x = 0.5f * v.x * (1.f + erff(v.x * 0.70710678f));
    v.y = 0.5f * v.y * (1.f + erff(v.y * 0.70710678f));
    v.z = 0.5f * v.z * (1.f + erff(v.z * 0.70710678f));
    v.w = 0.5f * v.w * (1.f + erff(v.w * 0.70710678f));
  }
  unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  if (tid < n4) reinterpret_cast<float4*>(out)[tid] = v;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    unsigned long long* q = st + (size_t)blockIdx.x * 8;
    q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3;
    q[4] = (unsigned long long)(s != 123.f);
  }
}

static double time_chain(void (*launch)(hipStream_t, int), hipStream_t st, int n, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < n; ++i) launch(st, i);
  CK(hipStreamSynchronize(st));
  std::vector<float> ts;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, st));
    for (int i = 0; i < n; ++i) launch(st, i);
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ts.push_back(ms * 1e3f / n);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

static float* g_buf; static float* g_buf2; static Big g_big; static unsigned long long* g_st;
static int g_wgs = 157, g_thr = 256, g_erf = 1;
static void l_trivial(hipStream_t s, int) { hipLaunchKernelGGL(k_trivial, dim3(g_wgs), dim3(g_thr), 0, s, g_buf); }
static void l_bigarg(hipStream_t s, int) { hipLaunchKernelGGL(k_bigarg, dim3(g_wgs, 2), dim3(g_thr), 0, s, g_big); }
static void l_copy(hipStream_t s, int i) {
  const int n4 = 2500 * 64 / 4;
  float* in = (i & 1) ? g_buf2 : g_buf; float* out = (i & 1) ? g_buf : g_buf2;
  hipLaunchKernelGGL(k_stamp_copy, dim3((n4 + g_thr - 1) / g_thr), dim3(g_thr), 0, s, in, out, g_st + (size_t)i * 4096 * 8, n4, g_erf);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  CK(hipMalloc(&g_buf, 4 << 20)); CK(hipMalloc(&g_buf2, 4 << 20)); CK(hipMalloc(&g_st, (size_t)64 * 4096 * 8 * 8));
  CK(hipMemset(g_buf, 0, 4 << 20)); CK(hipMemset(g_buf2, 0, 4 << 20));
  g_big.p[0] = g_buf;
  for (int i = 0; i < 320; ++i) g_big.a[i] = i;
  for (int wgs : {157, 256, 628, 1024}) {
    g_wgs = wgs;
    printf("trivial chain %4d WGs x256 : %.2f us/kernel\n", wgs, time_chain(l_trivial, st, 26, 20));
  }
  g_wgs = 157;
  printf("big-kernarg (1.3 KB) chain    : %.2f us/kernel\n", time_chain(l_bigarg, st, 26, 20));
  // graph replay of the trivial chain
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 26; ++i) l_trivial(st, i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    std::vector<float> ts;
    for (int r = 0; r < 20; ++r) {
      CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f / 26);
    }
    std::sort(ts.begin(), ts.end());
    printf("trivial chain as hipGraph      : %.2f us/kernel\n", ts[10]);
  }
  // dependent copy chain with stamps: thread counts x erf reps
  for (int thr : {64, 256}) for (int erf : {0, 1, 4}) {
    g_thr = thr; g_erf = erf;
    double us = time_chain(l_copy, st, 16, 10);
    CK(hipStreamSynchronize(st));
    const int n4 = 2500 * 64 / 4, nwg = (n4 + thr - 1) / thr;
    std::vector<unsigned long long> h((size_t)16 * 4096 * 8);
    CK(hipMemcpy(h.data(), g_st, h.size() * 8, hipMemcpyDeviceToHost));
    // launch 8 vs launch 7: boundary = min start of 8 - max end of 7
    auto mm = [&](int L, int k, bool mx) { unsigned long long r = mx ? 0 : ~0ull; for (int w = 0; w < nwg; ++w) { auto v = h[((size_t)L * 4096 + w) * 8 + k]; r = mx ? std::max(r, v) : std::min(r, v); } return r; };
    double med_load = 0, med_erf = 0, med_store = 0;
    { std::vector<double> a, b, c; for (int w = 0; w < nwg; ++w) { auto* q = &h[((size_t)8 * 4096 + w) * 8]; a.push_back((q[1] - q[0]) * 0.01); b.push_back((q[2] - q[1]) * 0.01); c.push_back((q[3] - q[2]) * 0.01); }
      std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end()); med_load = a[a.size() / 2]; med_erf = b[b.size() / 2]; med_store = c[c.size() / 2]; }
    printf("copy chain thr %3d erf x%d (%4d WGs): %.2f us/kernel | first-start skew %.2f us, kernel span %.2f us, boundary (end(7)->start(8)) %.2f us | per-WG median: load %.2f erf %.2f store %.2f us\n",
           thr, erf, nwg, us, (mm(8, 0, true) - mm(8, 0, false)) * 0.01, (mm(8, 3, true) - mm(8, 0, false)) * 0.01,
           ((double)mm(8, 0, false) - (double)mm(7, 3, true)) * 0.01, med_load, med_erf, med_store);
  }
  return 0;
}
