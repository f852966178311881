// Probe [r6]: register-resident weights.  Feasibility of the "N-split" chain kernel: a 256-thread workgroup (ONE wave per SIMD, up
// to 512 registers), wave w keeps output blocks {2w, 2w+1} of all four 128x128 units as bf16x3 pieces in registers (96 x u32x4 =
// 384 registers), the B operands (row pieces) of a 16-row tile come from LDS (12 ds_read_b128 per tile and unit), 48 MFMAs per tile
// and unit, plus FILL vector instructions / 3 ds_write_b128 / 2 global stores of "post" work on the other group's accumulators
// between them.  One s_barrier per slot of R tiles.  Prints shader cycles per slot of wave 0.  Not product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/regw_probe tools/regw_probe.hip && tools/regw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) char lds_char;
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(u32, __builtin_convertvector(v, bf16x2));
}

// R tiles per slot; FILL: 0 = MFMAs + B reads only, 1 = with the post work of the other group (blocks between MFMA groups, compiler's order),
// 2 = the same work, one scheduling region per tile with sched_group_barrier patterns (1 MFMA : 1 VALU : ...), B operands prefetched
template <int R, int FILL>
__global__ void __launch_bounds__(256, 1) k(const u32x4* __restrict__ wsrc, unsigned long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* sm = (lds_char*)smem;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // pieces image: [grp 2][tile R][j 4][piece 3] x 1 KB
  for (int i = threadIdx.x; i < 2 * R * 12 * 64; i += 256) ((lds_u32x4*)sm)[i] = u32x4{0x3f803f80u, 0x3c003c00u, 0x3f803f80u, 0x3c003c00u};
  u32x4 w[4][3][2][4];  // unit, piece, ob, K-slice: 96 x 4 registers
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[u][p][q][j] = wsrc[((((u * 3 + p) * 8 + 2 * wv + q) * 4 + j) * 64) + lane];
  f32x4 acc[2][R][2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[g][r][0] = acc[g][r][1] = f32x4{0, 0, 0, 0};
  __syncthreads();
  lds_char* ra = sm + 16 * lane;
  float* gout = sink + (size_t)blockIdx.x * 65536 + threadIdx.x * 4;
  // selector of the matrix-pipe split (see mgn_x6.inc split_sel)
  u32x4 s0, s1;
  {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      u32 w0 = 0, w1 = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = 2 * d + h, f = 16 * (i >> 2) + 4 * g + (i & 3);
        if (f == c) w0 |= 0xbf80u << (16 * h);
        if (f == c + 16) w1 |= 0xbf80u << (16 * h);
      }
      s0[d] = w0, s1[d] = w1;
    }
  }
  unsigned long long t0 = __builtin_readcyclecounter();
  if (FILL == 2) {
    u32x4 xb[2][4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) xb[0][j][p] = *(lds_u32x4*)(ra + (j * 3 + p) * 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
          __syncthreads();
          lds_char* pin = ra + grp * (R * 12 * 1024);
          lds_char* pout = ra + (grp ^ 1) * (R * 12 * 1024);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int cb = r & 1, nb = cb ^ 1;
            // next tile's B operands (the next slot's first tile is read after the barrier in the real kernel; here: same image)
            lds_char* nx = (r + 1 < R) ? pin + (r + 1) * 12 * 1024 : pout;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int p = 0; p < 3; ++p) xb[(R & 1) ? ((r + grp + 2 * u) & 1) ^ 1 : nb][j][p] = *(lds_u32x4*)(nx + (j * 3 + p) * 1024);
            u32x4(&x)[4][3] = xb[(R & 1) ? ((r + grp + 2 * u) & 1) : cb];
            f32x4(&a)[2] = acc[grp][r];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int q = 0; q < 2; ++q) a[q] = MF16(w[u][2][q][j], x[j][0], a[q]);
#pragma unroll
              for (int q = 0; q < 2; ++q) a[q] = MF16(w[u][1][q][j], x[j][1], a[q]);
#pragma unroll
              for (int q = 0; q < 2; ++q) a[q] = MF16(w[u][1][q][j], x[j][0], a[q]);
#pragma unroll
              for (int q = 0; q < 2; ++q) a[q] = MF16(w[u][0][q][j], x[j][2], a[q]);
#pragma unroll
              for (int q = 0; q < 2; ++q) a[q] = MF16(w[u][0][q][j], x[j][1], a[q]);
#pragma unroll
              for (int q = 0; q < 2; ++q) a[q] = MF16(w[u][0][q][j], x[j][0], a[q]);
            }
            // post of tile r of the other group
            f32x4(&pa)[2] = acc[grp ^ 1][r];
            u32 bits = 0;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float v = pa[q][e];
                asm("v_max_f32 %0, 0, %0" : "+v"(v));
                pa[q][e] = v;
                int t = __builtin_bit_cast(int, v);
                t = (t < 0) ? 0 : ((t > 1) ? 1 : t);
                bits |= (u32)t << (4 * q + e);
              }
            __builtin_nontemporal_store(pa[0], (f32x4*)(gout + (it & 15) * 4096));
            __builtin_nontemporal_store(pa[1], (f32x4*)(gout + (it & 15) * 4096 + 1024));
            if (lane < 16) __builtin_nontemporal_store(bits, (u32*)(gout + (it & 15) * 4096 + 2048));
            u32x4 pc[3];
            pc[0] = u32x4{pk_bf16(pa[0][0], pa[0][1]), pk_bf16(pa[0][2], pa[0][3]), pk_bf16(pa[1][0], pa[1][1]), pk_bf16(pa[1][2], pa[1][3])};
            const f32x4 r0 = MF16(s0, pc[0], pa[0]), r1 = MF16(s1, pc[0], pa[1]);
            pc[1] = u32x4{pk_bf16(r0[0], r0[1]), pk_bf16(r0[2], r0[3]), pk_bf16(r1[0], r1[1]), pk_bf16(r1[2], r1[3])};
            const f32x4 t0 = MF16(s0, pc[1], r0), t1 = MF16(s1, pc[1], r1);
            pc[2] = u32x4{pk_bf16(t0[0], t0[1]), pk_bf16(t0[2], t0[3]), pk_bf16(t1[0], t1[1]), pk_bf16(t1[2], t1[3])};
#pragma unroll
            for (int p = 0; p < 3; ++p) *(lds_u32x4*)(pout + ((r * 4 + wv) * 3 + p) * 1024) = pc[p];
            pa[0] = pa[1] = f32x4{0, 0, 0, 0};
            // the order: one MFMA, then one or two other instructions
#pragma unroll
            for (int i = 0; i < 52; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
              if (i < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              else if (i % 8 == 0) __builtin_amdgcn_sched_group_barrier(0x200 | 0x040, 1, 0);
              else __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            PIN();
          }
        }
      }
    }
  } else
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int grp = 0; grp < 2; ++grp) {
        __syncthreads();
        lds_char* pin = ra + grp * (R * 12 * 1024);
        lds_char* pout = ra + (grp ^ 1) * (R * 12 * 1024);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          u32x4 xb[4][3];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) xb[j][p] = *(lds_u32x4*)(pin + ((r * 4 + j) * 3 + p) * 1024);
          PIN();
          // post of tile r of the other group (its previous unit): ReLU, mask, conversions, piece writes, row stores
          f32x4(&pa)[2] = acc[grp ^ 1][r];
          u32 bits = 0;
          u32x4 pc[3];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              acc[grp][r][q] = MF16(w[u][2][q][j], xb[j][0], acc[grp][r][q]);
              acc[grp][r][q] = MF16(w[u][1][q][j], xb[j][1], acc[grp][r][q]);
              acc[grp][r][q] = MF16(w[u][1][q][j], xb[j][0], acc[grp][r][q]);
            }
            PIN();
            if (FILL) {
              if (j == 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    float v = pa[q][e];
                    asm("v_max_f32 %0, 0, %0" : "+v"(v));
                    pa[q][e] = v;
                    int t = __builtin_bit_cast(int, v);
                    t = (t < 0) ? 0 : ((t > 1) ? 1 : t);
                    bits |= (u32)t << (4 * q + e);
                  }
              } else if (j == 1) {
                pc[0] = u32x4{pk_bf16(pa[0][0], pa[0][1]), pk_bf16(pa[0][2], pa[0][3]), pk_bf16(pa[1][0], pa[1][1]), pk_bf16(pa[1][2], pa[1][3])};
                __builtin_nontemporal_store(pa[0], (f32x4*)(gout + (it & 15) * 4096));
                __builtin_nontemporal_store(pa[1], (f32x4*)(gout + (it & 15) * 4096 + 1024));
              } else if (j == 2) {
                f32x4 r0, r1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  r0[e] = pa[0][e] - __builtin_bit_cast(float, (pc[0][e >> 1] << (16 * (1 - (e & 1)))) & 0xffff0000u);
                  r1[e] = pa[1][e] - __builtin_bit_cast(float, (pc[0][2 + (e >> 1)] << (16 * (1 - (e & 1)))) & 0xffff0000u);
                }
                pc[1] = u32x4{pk_bf16(r0[0], r0[1]), pk_bf16(r0[2], r0[3]), pk_bf16(r1[0], r1[1]), pk_bf16(r1[2], r1[3])};
                pc[2] = pc[1] ^ u32x4{bits, bits, bits, bits};
              } else {
#pragma unroll
                for (int p = 0; p < 3; ++p) *(lds_u32x4*)(pout + ((r * 4 + wv) * 3 + p) * 1024) = pc[p];
                pa[0] = pa[1] = f32x4{0, 0, 0, 0};
              }
              PIN();
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              acc[grp][r][q] = MF16(w[u][0][q][j], xb[j][2], acc[grp][r][q]);
              acc[grp][r][q] = MF16(w[u][0][q][j], xb[j][1], acc[grp][r][q]);
              acc[grp][r][q] = MF16(w[u][0][q][j], xb[j][0], acc[grp][r][q]);
            }
            PIN();
          }
        }
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < R; ++r) res += acc[g][r][0][0] + acc[g][r][1][1];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (res == 123.456f) sink[0] = res;
}

template <int R, int FILL>
void run(const char* name, const u32x4* w, unsigned long long* out, float* sink) {
  const int iters = 200, lds = 2 * R * 12 * 1024;
  hipFuncSetAttribute((const void*)k<R, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, FILL>), dim3(256), dim3(256), lds, 0, w, out, sink, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0;
  for (int i = 0; i < 256; ++i) cyc += (double)h[i];
  cyc /= 256;
  const double slots = iters * 8.0, mf = R * 48.0;
  printf("%-28s %8.1f us  %8.0f cycles/slot  %6.1f cycles/MFMA  (%.0f MFMAs per slot)  clock %.2f GHz  %.0f TF/s\n", name, ms * 1e3,
         cyc / slots, cyc / slots / mf, mf, cyc / (ms * 1e6), 256.0 * 4 * slots * mf * 16384 / (ms * 1e-3) / 1e12);
}

int main() {
  u32x4* w;
  unsigned long long* out;
  float* sink;
  hipMalloc(&w, 96 * 4 * 64 * 16);
  hipMemset(w, 0x3c, 96 * 4 * 64 * 16);
  hipMalloc(&out, 256 * 8);
  hipMalloc(&sink, (size_t)256 * 65536 * 4 + 65536);
  run<1, 0>("R=1 bare", w, out, sink);
  run<1, 1>("R=1 with post", w, out, sink);
  run<2, 0>("R=2 bare", w, out, sink);
  run<2, 1>("R=2 with post", w, out, sink);
  run<2, 2>("R=2 post, group pattern", w, out, sink);
  run<4, 0>("R=4 bare", w, out, sink);
  run<4, 1>("R=4 with post", w, out, sink);
  run<4, 2>("R=4 post, group pattern", w, out, sink);
  return 0;
}
