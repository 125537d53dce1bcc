#include <hip/hip_runtime.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out, int n) {
  extern __shared__ float sm[];
  for (int i = 0; i < 4; ++i) sm[threadIdx.x * 4 + i] = -7.f;
  __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
  int voff = threadIdx.x < 32 ? threadIdx.x * 16 : -1;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, (short)0, n * 4, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)sm, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __syncthreads();
  out[threadIdx.x] = sm[threadIdx.x * 4];
}
int main() {
  float *in, *out; (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 64 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i + 1; (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, in, out, 4096);
  float o[64]; (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 8) printf("%g ", o[i]); printf("\n");
  for (int i = 33; i < 36; ++i) printf("%g ", o[i]); printf("\n");
  return 0;
}
