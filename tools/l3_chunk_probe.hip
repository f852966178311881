// Does a producer -> consumer pair run faster when the intermediate fits the 256 MiB Infinity Cache?  ("chunk the
// backward so the dZ pair is Infinity-Cache-resident".)  Producer: reads a cold stream A (S bytes), writes the intermediate D (S bytes).
// Consumer: reads D and a second cold stream B (S bytes each), writes nothing.  Cold streams walk through 2 GiB so they never hit.
// One "round" = TOTAL bytes of D split into TOTAL / S producer/consumer pairs that reuse the SAME D buffer.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/l3_chunk_probe tools/l3_chunk_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_prod(const f32x4* __restrict__ a, f32x4* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = a[i] * 2.f;
}
__global__ void __launch_bounds__(256) k_cons(const f32x4* __restrict__ d, const f32x4* __restrict__ b, f32x4* __restrict__ sink, size_t n) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += d[i] * b[i];
  if (acc[0] == 12345.678f) sink[0] = acc;
}
int main() {
  const size_t cold = (size_t)2 << 30, total = (size_t)384 << 20;  // bytes
  f32x4 *a, *b, *d, *sink;
  hipMalloc(&a, cold), hipMalloc(&b, cold), hipMalloc(&d, total), hipMalloc(&sink, 64);
  hipMemset(a, 0, cold), hipMemset(b, 0, cold), hipMemset(d, 0, total);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)  // 0: D reused per chunk (same addresses); 1: D walks through the 384 MiB buffer
    for (size_t mb : {384, 192, 128, 96, 64, 48, 32, 16}) {
      const size_t S = mb << 20, n = S / 16, chunks = total / S;
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        size_t off = 0;
        hipEventRecord(e0, 0);
        for (int round = 0; round < 4; ++round)
          for (size_t c = 0; c < chunks; ++c) {
            f32x4* dc = d + (mode ? c * n : 0);
            hipLaunchKernelGGL(k_prod, dim3(2048), dim3(256), 0, 0, a + off / 16, dc, n);
            hipLaunchKernelGGL(k_cons, dim3(2048), dim3(256), 0, 0, dc, b + off / 16, sink, n);
            off = (off + S) % cold;
          }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      // per round: producer moves 2 * total, consumer 2 * total
      printf("%s chunk %4zu MiB x %2zu : %8.1f us per 384 MiB round  (%.0f GB/s over the 4 x 384 MiB of nominal traffic)\n",
             mode ? "walking" : "reused ", mb, chunks, best / 4 * 1e3, 4.0 * total / 1e9 / (best / 4 * 1e-3));
    }
  return 0;
}
