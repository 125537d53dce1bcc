// Probe (round 4): does the HIP runtime of this box accept kernel arguments beyond 4 KB, eagerly and from a captured graph?
// hipcc --offload-arch=gfx950 -O2 tools/r04/kernarg_probe.hip -o /tmp/kp && /tmp/kp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N> struct Big { unsigned v[N]; };
template <int N> __global__ void k(const Big<N> b, unsigned* out) {
  unsigned s = 0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) s += b.v[i] * (unsigned)(i + 1);
  atomicAdd(out, s);
}
template <int N> int probe() {
  Big<N> b; unsigned want = 0;
  for (int i = 0; i < N; ++i) { b.v[i] = (unsigned)(i * 2654435761u + 7u); want += b.v[i] * (unsigned)(i + 1); }
  unsigned* d; if (hipMalloc(&d, 4) != hipSuccess) return 1;
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(256), 0, 0, b, d);
  hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
  unsigned got = 0; hipMemcpy(&got, d, 4, hipMemcpyDeviceToHost);
  printf("kernarg %6zu bytes eager: launch %s sync %s result %s\n", sizeof(b), hipGetErrorName(e), hipGetErrorName(e2), got == want ? "ok" : "WRONG");
  hipStream_t st; hipStreamCreate(&st); hipGraph_t g; hipGraphExec_t ge;
  hipMemset(d, 0, 4); hipDeviceSynchronize();
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(256), 0, st, b, d);
  e = hipStreamEndCapture(st, &g);
  for (int i = 0; i < N; ++i) b.v[i] = 0;      // the host copy must not matter after capture
  hipError_t e3 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipError_t e4 = hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipMemcpy(&got, d, 4, hipMemcpyDeviceToHost);
  printf("kernarg %6zu bytes graph: capture %s instantiate %s launch %s result %s\n", sizeof(Big<N>), hipGetErrorName(e), hipGetErrorName(e3), hipGetErrorName(e4), got == want ? "ok" : "WRONG");
  return 0;
}
int main() { probe<960>(); probe<2048>(); probe<4096>(); probe<16384>(); return 0; }
