// Cost model input for the round-5 Winograd kernel: what one global->VGPR load, one LDS-DMA piece, one ds_read_b128, one ds_write_b128
// and one VALU instruction cost in fp32-MFMA time (v_mfma_f32_16x16x4_f32 shares the vector issue port) at 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_cost tools/r05/mfma_cost.hip && /tmp/mfma_cost
// A "step" = 16 MFMAs per wave (16 independent accumulators) + the fillers named by the template arguments; every CU busy.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int VLOAD, int LDMA, int DSR, int DSW, int VALU, int PKV>
__global__ __launch_bounds__(512) void k_step(const float* __restrict__ src, float* __restrict__ out, unsigned long long* cyc, int iters, int src_bytes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, src_bytes, 0x00020000);
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 a[2][4], b[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[0][i] = a[1][i] = (f32x4){1.f, 2.f, 3.f, 4.f}; b[0][i] = b[1][i] = (f32x4){0.5f, 0.25f, 0.125f, 1.f}; }
  float vv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) vv[i] = (float)(tid + i);
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pk[i] = (f32x2){(float)tid, (float)i};
  for (int i = tid; i < 8192; i += blockDim.x) smem[i] = (float)i;
  __syncthreads();
  const int voff = (wave * 64 + lane) * 16;
  const int mask = src_bytes - 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  auto step = [&](const int it, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value, nxt = cur ^ 1;
    const int so = __builtin_amdgcn_readfirstlane(((it * 8 + (int)blockIdx.x * 64) * 1024) & mask & ~8191);
#pragma unroll
    for (int v = 0; v < VLOAD; ++v) a[nxt][v & 3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + v * 8192, so, 0));
#pragma unroll
    for (int v = 0; v < LDMA; ++v)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + 8192 + ((it & 3) * 8 + v) * 8 * 256 + wave * 256), 16, voff + v * 8192, so, 0, 0);
#pragma unroll
    for (int v = 0; v < DSR; ++v) {
      typedef const __attribute__((address_space(3))) f32x4 lds_f4;
      b[nxt][v & 3] = *(lds_f4*)(smem + ((v * 512 + (it & 7) * 64 + lane) & 2047) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[q * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][q >> 1][e], b[cur][q & 1][e], acc[q * 4 + e], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < DSW; ++v) {
      typedef __attribute__((address_space(3))) f32x4 lds_f4w;
      *(lds_f4w*)(smem + 2048 * 4 + ((v * 512 + tid) & 2047) * 4) = acc[v & 15];      // timing only
    }
#pragma unroll
    for (int v = 0; v < VALU; ++v) vv[v & 7] = vv[v & 7] + vv[(v + 1) & 7];
#pragma unroll
    for (int v = 0; v < PKV; ++v) pk[v & 3] = pk[v & 3] + pk[(v + 1) & 3];
    if (LDMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LDMA * 2 < 60 ? LDMA * 2 : 60) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int it = 0; it < iters; it += 2) {
    step(it, std::integral_constant<int, 0>{});
    step(it + 1, std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += a[0][i] + a[1][i] + b[0][i] + b[1][i];
  float r = s[0] + s[1] + s[2] + s[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += vv[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) r += pk[i][0] + pk[i][1];
  out[(size_t)blockIdx.x * blockDim.x + tid] = r;
  if (lane == 0) cyc[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

static float* g_src; static float* g_out; static unsigned long long* g_cyc;
template <int VLOAD, int LDMA, int DSR, int DSW, int VALU, int PKV>
static void run(const char* name, int threads, int wg_per_cu, int iters) {
  auto kern = k_step<VLOAD, LDMA, DSR, DSW, VALU, PKV>;
  const int lds = 160 * 1024 / wg_per_cu;      // occupancy through LDS
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = 256 * wg_per_cu;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int src_bytes = 1 << 20;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, g_src, g_out, g_cyc, iters, src_bytes);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, g_src, g_out, g_cyc, iters, src_bytes);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = grid * (threads / 64);
  std::vector<unsigned long long> c((size_t)grid * 8);
  CK(hipMemcpy(c.data(), g_cyc, c.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> per;
  for (int g = 0; g < grid; ++g) for (int w = 0; w < threads / 64; ++w) per.push_back((double)c[(size_t)g * 8 + w] / iters);
  std::sort(per.begin(), per.end());
  const int wps = threads / 64 * wg_per_cu / 4;      // waves per SIMD
  const double ideal = 16.0 * 32 * wps;
  printf("%-44s thr %4d wg/cu %d waves/simd %d: cycles/step median %8.1f (ideal %6.0f, x%.3f)  wall %.3f ms  -> extra per step & wave %7.1f cyc\n", name, threads,
         wg_per_cu, wps, per[per.size() / 2], ideal, per[per.size() / 2] / ideal, ms, (per[per.size() / 2] - ideal) / wps);
  (void)nw;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  CK(hipMalloc(&g_src, 1 << 20)); CK(hipMemset(g_src, 0x3c, 1 << 20));
  CK(hipMalloc(&g_out, 256 * 8 * 1024 * 4)); CK(hipMalloc(&g_cyc, 256 * 8 * 8 * 8));
  const int cfg[][2] = {{256, 1}, {512, 1}, {512, 2}};
  for (auto& c : cfg) {
    const int t = c[0], w = c[1];
    run<0, 0, 0, 0, 0, 0>("bare 16 MFMA", t, w, iters);
    run<2, 0, 0, 0, 0, 0>("+2 buffer_load_dwordx4 -> VGPR", t, w, iters);
    run<4, 0, 0, 0, 0, 0>("+4 buffer_load_dwordx4 -> VGPR", t, w, iters);
    run<0, 2, 0, 0, 0, 0>("+2 LDS-DMA pieces (1 KB)", t, w, iters);
    run<0, 4, 0, 0, 0, 0>("+4 LDS-DMA pieces (1 KB)", t, w, iters);
    run<0, 0, 2, 0, 0, 0>("+2 ds_read_b128", t, w, iters);
    run<0, 0, 4, 0, 0, 0>("+4 ds_read_b128", t, w, iters);
    run<0, 0, 8, 0, 0, 0>("+8 ds_read_b128", t, w, iters);
    run<0, 0, 0, 2, 0, 0>("+2 ds_write_b128", t, w, iters);
    run<0, 0, 0, 4, 0, 0>("+4 ds_write_b128", t, w, iters);
    run<0, 0, 0, 0, 8, 0>("+8 v_add_f32", t, w, iters);
    run<0, 0, 0, 0, 32, 0>("+32 v_add_f32", t, w, iters);
    run<0, 0, 0, 0, 0, 8>("+8 v_pk_add_f32", t, w, iters);
    run<0, 0, 0, 0, 0, 16>("+16 v_pk_add_f32", t, w, iters);
    run<2, 0, 2, 0, 0, 0>("+2 VGPR loads +2 ds_read (planned step)", t, w, iters);
    run<0, 2, 4, 0, 0, 0>("+2 LDS-DMA +4 ds_read (old-style step x2)", t, w, iters);
  }
  return 0;
}
