// Probe: does the immediate offset of global_load_lds_dwordx4 move the LDS destination, the
// global source, or both?  (decides whether consecutive DMA pieces can share one M0 setting)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* src, float* out) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
  __syncthreads();
  unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
  unsigned voff = threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(base), "s"(src) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<float> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, 8192); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o);
  std::vector<float> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  printf("lds[0]=%g lds[255]=%g lds[256]=%g lds[511]=%g lds[512]=%g\n", r[0], r[255], r[256], r[511], r[512]);
  return 0;
}
