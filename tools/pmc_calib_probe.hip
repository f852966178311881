// Calibration of FETCH_SIZE / WRITE_SIZE in the engine's OWN access patterns (MI355X_MICROARCH.md, HBM section: "calibrate on a known
// byte count in your own access pattern before trusting an absolute").  Every kernel touches each byte of a 1 GiB buffer exactly once
// (far past the 256 MiB Infinity Cache), so  counter x 1024 / 2^30  is the factor of that pattern.
//   rd_contig16   : 16 B per lane, 1 KB contiguous per wave instruction (the calibration copy's pattern; expected to report 1/2)
//   rd_rows64     : the T-layout row tile of the chain kernels: one instruction reads 64 B of each of 16 consecutive 512-byte rows,
//                   eight such instructions back to back cover the rows (ld2u / ld3u in mgn_x6.inc)
//   rd_rows64_perm: the same with the 16 rows taken through a random permutation (the gathered Pd / Ps rows)
//   rd_ldsdma16   : global_load_lds_dwordx4, 1 KB contiguous per wave instruction (k_wgrad_x6's tiles)
//   wr_contig16 / wr_rows64: the two store patterns (st4: 64 B of each of 16 rows per instruction)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib_probe tools/pmc_calib_probe.hip
// run  : rocprofv3 --pmc FETCH_SIZE -d out/fetch --output-format csv -- tools/pmc_calib_probe   (and the same with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GRID 2048
__global__ void __launch_bounds__(256) rd_contig16(const f32x4* __restrict__ in, f32x4* __restrict__ sink, size_t n) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)GRID * 256) acc += in[i];
  if (acc[0] == 12345.678f) sink[0] = acc;
}
// rows of 512 B = 32 f32x4; a wave owns 16 rows per iteration
template <bool PERM>
__global__ void __launch_bounds__(256) rd_rows64(const f32x4* __restrict__ in, const int* __restrict__ perm, f32x4* __restrict__ sink, size_t nrows) {
  const int lane = threadIdx.x & 63, r = lane >> 2, g = lane & 3;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)GRID * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t t = wave; t * 16 < nrows; t += nw) {
    size_t row = t * 16 + r;
    if (PERM) row = (size_t)perm[row];
    const f32x4* p = in + row * 32 + g;
    f32x4 v[8];
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) v[ib] = p[4 * ib];
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) acc += v[ib];
  }
  if (acc[0] == 12345.678f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) rd_ldsdma16(const float* __restrict__ in, f32x4* __restrict__ sink, size_t nbytes) {
  __shared__ f32x4 buf[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t wave = (size_t)blockIdx.x * 4 + wv, nw = (size_t)GRID * 4;
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(&buf[wv][0]));
  const unsigned voff = lane * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t t = wave; t * 1024 < nbytes; t += nw) {
    const unsigned long long sv = (unsigned long long)((const char*)in + t * 1024);
    const unsigned slo = __builtin_amdgcn_readfirstlane((unsigned)sv), shi = __builtin_amdgcn_readfirstlane((unsigned)(sv >> 32));
    const float* src = (const float*)(((unsigned long long)shi << 32) | slo);
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&s"(keep)
        : "v"(voff), "s"(src), "s"(lds)
        : "memory");
    acc += buf[wv][lane];
  }
  if (acc[0] == 12345.678f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) wr_contig16(f32x4* __restrict__ out, size_t n) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)GRID * 256) out[i] = v;
}
__global__ void __launch_bounds__(256) wr_rows64(f32x4* __restrict__ out, size_t nrows) {
  const int lane = threadIdx.x & 63, r = lane >> 2, g = lane & 3;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)GRID * 4;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t t = wave; t * 16 < nrows; t += nw) {
    f32x4* p = out + (t * 16 + r) * 32 + g;
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) p[4 * ib] = v;
  }
}
int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 16, nrows = bytes / 512;
  f32x4 *in, *out, *sink;
  int* perm;
  hipMalloc(&in, bytes), hipMalloc(&out, bytes), hipMalloc(&sink, 64), hipMalloc(&perm, nrows * 4);
  hipMemset(in, 0, bytes), hipMemset(out, 0, bytes);
  int* h = (int*)malloc(nrows * 4);
  for (size_t i = 0; i < nrows; ++i) h[i] = (int)i;
  unsigned long long s = 88172645463325252ull;  // xorshift: a fixed permutation of the rows
  for (size_t i = nrows - 1; i > 0; --i) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    const size_t j = s % (i + 1);
    const int t = h[i];
    h[i] = h[j], h[j] = t;
  }
  hipMemcpy(perm, h, nrows * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(rd_contig16, dim3(GRID), dim3(256), 0, 0, in, sink, n);
    hipLaunchKernelGGL(rd_rows64<false>, dim3(GRID), dim3(256), 0, 0, in, (const int*)nullptr, sink, nrows);
    hipLaunchKernelGGL(rd_rows64<true>, dim3(GRID), dim3(256), 0, 0, in, (const int*)perm, sink, nrows);
    hipLaunchKernelGGL(rd_ldsdma16, dim3(GRID), dim3(256), 0, 0, (const float*)in, sink, bytes);
    hipLaunchKernelGGL(wr_contig16, dim3(GRID), dim3(256), 0, 0, out, n);
    hipLaunchKernelGGL(wr_rows64, dim3(GRID), dim3(256), 0, 0, out, nrows);
  }
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
