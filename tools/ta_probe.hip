// Probe: what a CU's vector-memory front end (address coalescer / L1) sustains for the T-layout row-tile access of the
// chain kernels (lane (c, g) -> row c, 16 B at 64 * blk + 16 * g: 64 separate 16-byte pieces per wave instruction) against
// a row-contiguous access (lane L -> 16 B at 16 * L: two whole 512-byte rows per instruction), on L2-resident and on
// HBM-resident data, loads and stores, 8 waves per CU.  Not product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ta_probe tools/ta_probe.hip && tools/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 g_f32x4;

// MODE 0: T-layout, 1: row-contiguous (2 rows per instruction), 2: half-row contiguous (4 rows x 256 B per instruction)
template <int MODE, bool STORE>
__global__ void __launch_bounds__(512) k(float* buf, long tiles_per_wave_span, int reps, unsigned long long* cyc, float* sink) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const long wave = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
  // each wave walks its own span of 16-row tiles (8 KB each), `reps` times
  float* base0 = buf + wave * tiles_per_wave_span * 16 * 128;
  f32x4 s = {0, 0, 0, 0}, v = {1.f * lane, 2.f, 3.f, 4.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r)
    for (long t = 0; t < tiles_per_wave_span; ++t) {
      float* base = base0 + t * 16 * 128;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        float* p;
        if (MODE == 0) p = base + c * 128 + 16 * kb + 4 * g;
        else if (MODE == 1) p = base + kb * 256 + lane * 4;
        else p = base + ((kb & 3) * 4 + (lane >> 4)) * 128 + (kb >> 2) * 64 + (lane & 15) * 4;
        if (STORE) *(g_f32x4*)p = v;
        else s += *(const g_f32x4*)p;
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[wave] = t1 - t0;
  if (s[0] == 12345.f) sink[0] = s[1];
}

template <int MODE, bool STORE>
void run(const char* name, float* buf, long span, int reps, unsigned long long* d_cyc, float* sink) {
  k<MODE, STORE><<<256, 512>>>(buf, span, reps, d_cyc, sink);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE, STORE><<<256, 512>>>(buf, span, reps, d_cyc, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[2048];
  hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0;
  for (int i = 0; i < 2048; ++i) c += (double)h[i];
  c /= 2048;
  const double instr_per_wave = (double)reps * span * 8;
  const double bytes = 2048.0 * instr_per_wave * 1024;
  printf("%-44s %8.3f ms %7.0f GB/s   %6.1f cycles per wave instruction, %5.1f per CU (8 waves), %5.1f B/cycle/CU\n", name, ms,
         bytes / ms / 1e6, c / instr_per_wave, c / instr_per_wave / 8, 8 * 1024.0 / (c / instr_per_wave));
}

int main() {
  float *buf, *sink;
  unsigned long long* d_cyc;
  const long big = 1L << 31;  // 2 GiB
  hipMalloc(&buf, big);
  hipMalloc(&sink, 4096);
  hipMalloc(&d_cyc, 2048 * 8);
  hipMemset(buf, 0, big);
  // L2-resident: 2048 waves x 2 tiles x 8 KB = 32 MB (fits the 8 x 4 MB L2 s); HBM: 2048 x 128 tiles x 8 KB = 2 GiB
  run<0, false>("load  T-layout        L2-resident (32 MB)", buf, 2, 400, d_cyc, sink);
  run<1, false>("load  row-contiguous  L2-resident", buf, 2, 400, d_cyc, sink);
  run<2, false>("load  half-row (256B) L2-resident", buf, 2, 400, d_cyc, sink);
  run<0, true>("store T-layout        L2-resident", buf, 2, 400, d_cyc, sink);
  run<1, true>("store row-contiguous  L2-resident", buf, 2, 400, d_cyc, sink);
  run<2, true>("store half-row (256B) L2-resident", buf, 2, 400, d_cyc, sink);
  run<0, false>("load  T-layout        MALL-resident (128 MB)", buf, 8, 100, d_cyc, sink);
  run<1, false>("load  row-contiguous  MALL-resident", buf, 8, 100, d_cyc, sink);
  run<0, true>("store T-layout        MALL-resident", buf, 8, 100, d_cyc, sink);
  run<1, true>("store row-contiguous  MALL-resident", buf, 8, 100, d_cyc, sink);
  run<0, false>("load  T-layout        HBM (2 GiB)", buf, 128, 4, d_cyc, sink);
  run<1, false>("load  row-contiguous  HBM", buf, 128, 4, d_cyc, sink);
  run<0, true>("store T-layout        HBM", buf, 128, 4, d_cyc, sink);
  run<1, true>("store row-contiguous  HBM", buf, 128, 4, d_cyc, sink);
  return 0;
}
