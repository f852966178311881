// Probe: sustained v_mfma_f32_16x16x32_bf16 rate vs. number of independent accumulator chains and waves per SIMD
// (does a lone wave with 2 interleaved chains -- the packed chain kernels' gemm_q -- reach the pipe's rate?).  Not product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
template <int NACC, int WPS>
__global__ void __launch_bounds__(256, WPS) k(float* out, int iters, unsigned seed) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  u32x4 a = {seed + threadIdx.x, seed * 3u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, seed, 0x3f803f80u, seed ^ 7u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = MFMA(a, b, acc[i]);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <typename K>
void run(const char* name, K kern, int blocks, int iters, int nacc, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 256>>>(out, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, 256>>>(out, iters, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = 12.0 * nacc * iters;                   // MFMAs per wave
  double fl = nm * 16384.0 * blocks * 4.0;
  // cycles per MFMA per SIMD assuming 2.4 GHz: waves per SIMD = blocks*4/(256*4)
  double wps = blocks * 4.0 / 1024.0;
  printf("%-28s blocks=%4d  %8.3f ms  %7.1f TFLOP/s   %.1f ns per MFMA per wave\n", name, blocks, ms, fl / ms / 1e9, ms * 1e6 / nm);
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  const int it = 20000;
  run("1 chain, 1 wave/SIMD", k<1, 1>, 256, it, 1, out);
  run("2 chains, 1 wave/SIMD", k<2, 1>, 256, it, 2, out);
  run("4 chains, 1 wave/SIMD", k<4, 1>, 256, it, 4, out);
  run("8 chains, 1 wave/SIMD", k<8, 1>, 256, it, 8, out);
  run("1 chain, 2 waves/SIMD", k<1, 2>, 512, it, 1, out);
  run("2 chains, 2 waves/SIMD", k<2, 2>, 512, it, 2, out);
  run("4 chains, 2 waves/SIMD", k<4, 2>, 512, it, 4, out);
  run("8 chains, 2 waves/SIMD", k<8, 2>, 512, it, 8, out);
  return 0;
}
