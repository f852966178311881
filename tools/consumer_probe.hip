// Probe [r5]: what does the CONSUMER pattern of k_wgrad_pc cost on its own?  One tile = 96 v_mfma_f32_16x16x32_bf16 over 16 accumulators
// (or 48 v_mfma_f32_32x32x16_bf16 over 4) with 24 distinct 16-byte operand vectors, optionally re-loaded from LDS by 24 ds_read_b128
// spread through the tile as the kernel spreads them (every piece re-loaded right after its last use).  WAVES = 4 or 8 per workgroup
// (one or two such waves per SIMD), one workgroup per CU.  s_memtime around ITERS tiles of wave 0.  Not product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/consumer_probe tools/consumer_probe.hip && tools/consumer_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
#define MF32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)

// SHAPE 16 / 32; READS: 0 none, 1 = 24 ds_read_b128 per tile spread as in the kernel, 2 = the same reads in ONE burst at the tile's start
template <int SHAPE, int READS, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) k(unsigned long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __attribute__((address_space(3))) char* sm = (__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 49152 / 16; i += 64 * WAVES) ((__attribute__((address_space(3))) u32x4*)sm)[i] = u32x4{0x3f803f80u, 0x3c003c00u, 0x3f803f80u, (unsigned)i};
  __syncthreads();
  __attribute__((address_space(3))) char* ra = sm + 16 * lane;
  u32x4 ap[4][3], bp[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int p = 0; p < 3; ++p) ap[j][p] = *(lds_cu32x4*)(ra + (j * 3 + p) * 1024), bp[j][p] = *(lds_cu32x4*)(ra + (12 + j * 3 + p) * 1024);
  auto rdA = [&](int p) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ap[j][p] = *(lds_cu32x4*)(ra + (j * 3 + p) * 1024);
  };
  auto rdB = [&](int p) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bp[j][p] = *(lds_cu32x4*)(ra + (12 + j * 3 + p) * 1024);
  };
  unsigned long long t0, t1;
  float res = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc[j][kk] = f32x4{0, 0, 0, 0};
    auto term = [&](int pa, int pb) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[j][kk] = MF16(ap[j][pa], bp[kk][pb], acc[j][kk]);
    };
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if (READS == 2) { rdB(2), rdB(1), rdA(0), rdA(1), rdA(2), rdB(0); PIN(); }
      term(0, 2); PIN();
      if (READS == 1) { rdB(2); PIN(); }
      term(0, 1); term(1, 1); PIN();
      if (READS == 1) { rdB(1); PIN(); }
      term(0, 0); PIN();
      if (READS == 1) { rdA(0); PIN(); }
      term(1, 0); PIN();
      if (READS == 1) { rdA(1); PIN(); }
      term(2, 0); PIN();
      if (READS == 1) { rdA(2); rdB(0); PIN(); }
    }
    t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) res += acc[j][kk][0];
  } else {
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][kk][r] = 0.f;
    // the same 24 vectors: ap[2 * blk + kstep][piece]
    auto term = [&](int pa, int pb) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) acc[j][kk] = MF32(ap[2 * j + ks][pa], bp[2 * kk + ks][pb], acc[j][kk]);
    };
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if (READS == 2) { rdB(2), rdB(1), rdA(0), rdA(1), rdA(2), rdB(0); PIN(); }
      term(0, 2); PIN();
      if (READS == 1) { rdB(2); PIN(); }
      term(0, 1); term(1, 1); PIN();
      if (READS == 1) { rdB(1); PIN(); }
      term(0, 0); PIN();
      if (READS == 1) { rdA(0); PIN(); }
      term(1, 0); PIN();
      if (READS == 1) { rdA(1); PIN(); }
      term(2, 0); PIN();
      if (READS == 1) { rdA(2); rdB(0); PIN(); }
    }
    t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) res += acc[j][kk][0];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (res == 12345.678f) sink[0] = res;
}

template <int SHAPE, int READS, int WAVES>
static void run(const char* name, unsigned long long* d_out, float* sink) {
  const int iters = 400;
  hipFuncSetAttribute((const void*)k<SHAPE, READS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, READS, WAVES>), dim3(256), dim3(64 * WAVES), 49152, 0, d_out, sink, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<SHAPE, READS, WAVES>), dim3(256), dim3(64 * WAVES), 49152, 0, d_out, sink, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c;
  hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
  const int nm = SHAPE == 16 ? 96 : 48;
  const double flop = 256.0 * WAVES * iters * 96.0 * 16384.0;   // 96 MFMA-equivalents of 16x16x32 per tile and wave
  printf("%-64s %7.0f cycles per tile = %5.1f per MFMA (pipe floor %d); launch %.3f ms = %.0f TFLOP/s of bf16 terms\n", name, c / (double)iters,
         c / (double)iters / nm, SHAPE == 16 ? 16 : 32, ms, flop / (ms * 1e-3) / 1e12);
}
int main() {
  unsigned long long* d_out;
  float* sink;
  hipMalloc(&d_out, 64), hipMalloc(&sink, 64);
  run<16, 0, 4>("16x16x32, 24 static operand vectors, 1 wave / SIMD", d_out, sink);
  run<16, 1, 4>("16x16x32, 24 ds_read_b128 spread through the tile, 1 wave / SIMD", d_out, sink);
  run<16, 2, 4>("16x16x32, 24 ds_read_b128 in one burst, 1 wave / SIMD", d_out, sink);
  run<16, 0, 8>("16x16x32, static operands, 2 waves / SIMD", d_out, sink);
  run<16, 1, 8>("16x16x32, spread reads, 2 waves / SIMD", d_out, sink);
  run<32, 0, 4>("32x32x16, 24 static operand vectors, 1 wave / SIMD", d_out, sink);
  run<32, 1, 4>("32x32x16, 24 ds_read_b128 spread through the tile, 1 wave / SIMD", d_out, sink);
  run<32, 2, 4>("32x32x16, 24 ds_read_b128 in one burst, 1 wave / SIMD", d_out, sink);
  run<32, 0, 8>("32x32x16, static operands, 2 waves / SIMD", d_out, sink);
  run<32, 1, 8>("32x32x16, spread reads, 2 waves / SIMD", d_out, sink);
  return 0;
}
