// Probe: sustained v_mfma_f32_16x16x4_f32 rate vs. number of independent accumulators,
// waves per SIMD, and accumulator-reuse distance.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WPS>
__global__ void __launch_bounds__(256, WPS) k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x * 1e-3f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = MFMA16(a, b, acc[i]);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int NACC, int WPS>
__global__ void __launch_bounds__(256, WPS) k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = a0 + threadIdx.x * 1e-3f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = MFMA32(a, b, acc[i]);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int blocks, int iters, double flop_per_iter_per_wave, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kern<<<blocks, 256>>>(out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, 256>>>(out, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double fl = flop_per_iter_per_wave * iters * blocks * 4.0;
  printf("%-34s blocks=%4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  const int it = 4000;
#define R16(N, W, B) run("16x16x4 nacc=" #N " wps=" #W, k16<N, W>, B, it, 16.0 * N * 2048.0, out)
#define R32(N, W, B) run("32x32x2 nacc=" #N " wps=" #W, k32<N, W>, B, it, 8.0 * N * 4096.0, out)
  R16(1, 1, 256); R16(2, 1, 256); R16(4, 1, 256); R16(8, 1, 256);
  R16(1, 2, 512); R16(2, 2, 512); R16(4, 2, 512); R16(8, 2, 512);
  R16(4, 4, 1024); R16(2, 4, 1024);
  R32(1, 1, 256); R32(2, 1, 256); R32(4, 1, 256);
  R32(1, 2, 512); R32(2, 2, 512); R32(4, 2, 512);
  return 0;
}
