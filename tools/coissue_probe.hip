// Probe: how fast does a wave issue VALU instructions while its SIMD partner streams MFMAs?  (The ping-pong kernels put one
// wave group into a GEMM quarter and the other into its non-GEMM work on the same SIMDs.)  One 512-thread workgroup per CU;
// waves 0-3 run an MFMA stream, waves 4-7 a VALU stream, each measured with s_memtime.  Not product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe tools/coissue_probe.hip && tools/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE_M: 0 none, 1 = 16x16x32 bf16 (4 accumulators round robin), 2 = 32x32x16 bf16 (2 accumulators)
// MODE_V: 0 none, 1 = v_fma_f32 x8 independent, 2 = v_pk_add_f32, 3 = v_cvt_pk_bf16_f32, 4 = v_and_b32, 5 = the bf16x3 split chain,
//         6 = ds_read_b128 stream (LDS instead of VALU), 7 = v_max_f32
template <int MODE_M, int MODE_V>
__global__ void __launch_bounds__(512, 2) k(unsigned long long* out, float* sink, int iters_m, int iters_v, int dummy) {
  __shared__ f32x4 lds[1024];
  const int wv = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, (float)threadIdx.x};
  lds[threadIdx.x + 512] = f32x4{1.f, 2.f, 3.f, (float)threadIdx.x};
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  float res = 0.f;
  if (wv < 4) {
    if (MODE_M == 1) {
      f32x4 acc[4];
      for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
      u32x4 a = {0x3f803f80u + (unsigned)dummy, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < iters_m; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
      }
      t1 = __builtin_readcyclecounter();
      res = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if (MODE_M == 2) {
      f32x16 acc[2];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
      u32x4 a = {0x3f803f80u + (unsigned)dummy, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < iters_m; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
      }
      t1 = __builtin_readcyclecounter();
      res = acc[0][0] + acc[1][1];
    }
  } else {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i + dummy);
    t0 = __builtin_readcyclecounter();
    if (MODE_V == 1) {
      for (int it = 0; it < iters_v; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
      }
    } else if (MODE_V == 2) {
      f32x2 p[4];
      for (int i = 0; i < 4; ++i) p[i] = f32x2{v[2 * i], v[2 * i + 1]};
      for (int it = 0; it < iters_v; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
      }
      for (int i = 0; i < 4; ++i) v[i] = p[i][0] + p[i][1];
    } else if (MODE_V == 3) {
      for (int it = 0; it < iters_v; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
      }
    } else if (MODE_V == 4) {
      for (int it = 0; it < iters_v; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[i]));
      }
    } else if (MODE_V == 7) {
      for (int it = 0; it < iters_v; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %0" : "+v"(v[i]));
      }
    } else if (MODE_V == 5) {  // 4 pairs x (cvt, shl, and, pk_add) x 2 + cvt = 44 instructions ~ one K-slice split; x1 per iteration
      for (int it = 0; it < iters_v; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float a = v[2 * i], b = v[2 * i + 1];
          unsigned q1, q2, q3;
          f32x2 pr = {a, b};
          q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
          float ra = a - __builtin_bit_cast(float, q1 << 16), rb = b - __builtin_bit_cast(float, q1 & 0xffff0000u);
          f32x2 pr2 = {ra, rb};
          q2 = __builtin_bit_cast(unsigned, __builtin_convertvector(pr2, bf16x2));
          float sa = ra - __builtin_bit_cast(float, q2 << 16), sb = rb - __builtin_bit_cast(float, q2 & 0xffff0000u);
          f32x2 pr3 = {sa, sb};
          q3 = __builtin_bit_cast(unsigned, __builtin_convertvector(pr3, bf16x2));
          v[2 * i] = __builtin_bit_cast(float, q1 ^ q3) + 1.0f;
          v[2 * i + 1] = __builtin_bit_cast(float, q2) + 1.0f;
          asm volatile("" : "+v"(v[2 * i]), "+v"(v[2 * i + 1]));
        }
      }
    } else if (MODE_V == 6) {
      f32x4 s = {0, 0, 0, 0};
      for (int it = 0; it < iters_v; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
          f32x4 x = *(volatile f32x4*)&lds[(lane + 64 * u + it) & 1023];
          s += x;
        }
      }
      v[0] = s[0] + s[1] + s[2] + s[3];
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) res += v[i];
  }
  if (lane == 0) {
    out[(blockIdx.x * 8 + wv) * 2] = t1 - t0;
    out[(blockIdx.x * 8 + wv) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
  }
  if (res == 123.456f) sink[threadIdx.x] = res;
}

template <int MM, int MV>
void run(const char* name, int nm_per_iter, int nv_per_iter, unsigned long long* d_out, float* sink) {
  const int iters_m = MM ? (MV ? 1600 : 400) : 0;  // paired: the MFMA stream outlasts the other one
  // size the VALU stream so that both streams last about equally long when they do not disturb each other
  const int iters_v = MV ? 400 : 0;
  k<MM, MV><<<256, 512>>>(d_out, sink, iters_m, iters_v, 0);
  hipDeviceSynchronize();
  k<MM, MV><<<256, 512>>>(d_out, sink, iters_m, iters_v, 0);
  hipDeviceSynchronize();
  static unsigned long long h[256 * 8 * 2];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  double cm = 0, cv = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? cm : cv) += (double)h[(b * 8 + w) * 2];
  cm /= 1024, cv /= 1024;
  // SIMD ids of waves 0 and 4 of block 0 (HW_ID bits 5:4)
  const int s0 = (int)((h[1] >> 4) & 3), s4 = (int)((h[4 * 2 + 1] >> 4) & 3);
  printf("%-44s", name);
  if (MM) printf("  MFMA %7.1f cyc/instr", cm / (iters_m * (double)nm_per_iter));
  if (MV) printf("  other %7.1f cyc/instr (%d per iter)", cv / (iters_v * (double)nv_per_iter), nv_per_iter);
  printf("   [simd w0=%d w4=%d]\n", s0, s4);
}

int main() {
  unsigned long long* d_out;
  float* sink;
  hipMalloc(&d_out, 256 * 8 * 2 * 8);
  hipMalloc(&sink, 4096);
  run<1, 0>("16x16x32 alone", 48, 0, d_out, sink);
  run<2, 0>("32x32x16 alone", 24, 0, d_out, sink);
  run<0, 1>("v_fma_f32 alone", 0, 48, d_out, sink);
  run<0, 2>("v_pk_add_f32 alone", 0, 48, d_out, sink);
  run<0, 3>("v_cvt_pk_bf16_f32 alone", 0, 48, d_out, sink);
  run<0, 4>("v_and_b32 alone", 0, 48, d_out, sink);
  run<0, 7>("v_max_f32 alone", 0, 48, d_out, sink);
  run<0, 5>("split chain alone (44 instr)", 0, 44, d_out, sink);
  run<0, 6>("ds_read_b128 alone", 0, 12, d_out, sink);
  run<1, 1>("16x16x32 + v_fma_f32", 48, 48, d_out, sink);
  run<1, 2>("16x16x32 + v_pk_add_f32", 48, 48, d_out, sink);
  run<1, 3>("16x16x32 + v_cvt_pk_bf16_f32", 48, 48, d_out, sink);
  run<1, 4>("16x16x32 + v_and_b32", 48, 48, d_out, sink);
  run<1, 7>("16x16x32 + v_max_f32", 48, 48, d_out, sink);
  run<1, 5>("16x16x32 + split chain", 48, 44, d_out, sink);
  run<1, 6>("16x16x32 + ds_read_b128", 48, 12, d_out, sink);
  run<2, 1>("32x32x16 + v_fma_f32", 24, 48, d_out, sink);
  run<2, 2>("32x32x16 + v_pk_add_f32", 24, 48, d_out, sink);
  run<2, 3>("32x32x16 + v_cvt_pk_bf16_f32", 24, 48, d_out, sink);
  run<2, 4>("32x32x16 + v_and_b32", 24, 48, d_out, sink);
  run<2, 5>("32x32x16 + split chain", 24, 44, d_out, sink);
  run<2, 6>("32x32x16 + ds_read_b128", 24, 12, d_out, sink);
  return 0;
}
