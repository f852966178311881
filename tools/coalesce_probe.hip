// Probe: cost of T-layout (lane (c,g) -> row c, 16 B at 64*blk+16*g) vs lane-linear 16-byte
// global stores / loads, same bytes.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 g_f32x4;

template <int MODE>  // 0: T-layout, 1: lane-linear (2 rows per instruction)
__global__ void __launch_bounds__(256) k_store(float* out, long rows, int reps) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
  for (int r = 0; r < reps; ++r)
    for (long t = wave; t < rows / 16; t += nw) {
      float* base = out + t * 16 * 128;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        float* p = (MODE == 0) ? base + c * 128 + 16 * kb + 4 * g : base + kb * 256 + lane * 4;
        *(g_f32x4*)p = v;
      }
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) k_load(const float* in, float* out, long rows, int reps) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  f32x4 s = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r)
    for (long t = wave; t < rows / 16; t += nw) {
      const float* base = in + t * 16 * 128;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const float* p = (MODE == 0) ? base + c * 128 + 16 * kb + 4 * g : base + kb * 256 + lane * 4;
        s += *(const g_f32x4*)p;
      }
    }
  if (s[0] == 12345.f) out[0] = s[1];
}
int main() {
  const long rows = 1 << 21;  // 1 GiB
  float *a, *b;
  hipMalloc(&a, rows * 512);
  hipMalloc(&b, 4096);
  hipMemset(a, 0, rows * 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %7.0f GB/s\n", name, ms, rows * 512.0 / ms / 1e6);
  };
  for (int blocks : {512, 2048}) {
    printf("blocks=%d\n", blocks);
    timeit("store T-layout", [&] { k_store<0><<<blocks, 256>>>(a, rows, 1); });
    timeit("store lane-linear", [&] { k_store<1><<<blocks, 256>>>(a, rows, 1); });
    timeit("load  T-layout", [&] { k_load<0><<<blocks, 256>>>(a, b, rows, 1); });
    timeit("load  lane-linear", [&] { k_load<1><<<blocks, 256>>>(a, b, rows, 1); });
  }
  return 0;
}
