// Probe [r6]: what does the SHAPE of a wave's store instruction cost?  Every wave writes [16 rows x 128 bytes] pieces of fp32 row
// tensors of 512-byte rows (the 32 features a wave of the register-resident-weights kernels owns), 1 GiB in all, as
//   0: two instructions of 16 rows x 64 bytes (the MFMA T layout: lane (c, g) -> row c, bytes 64 q + 16 g): HALF lines
//   1: two instructions of 8 rows x 128 bytes (lanes c < 8: half 0, c >= 8: half 1 of row c & 7): whole 128-byte lines
//   2: one wave writes whole 512-byte rows (4 instructions of 2 rows): the x6 kernels' pattern is 8 instructions of 16 x 64
//   3: fully contiguous 1 KB per instruction (a streaming store)
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_pattern_probe tools/store_pattern_probe.hip && tools/store_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long rows, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
  const long tiles = rows / 16;
  for (int rp = 0; rp < reps; ++rp)
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
      char* base = (char*)out + t * 16 * 512;
      if (MODE == 0) {  // waves 0-3 / 4-7: two tensors' worth is not modelled; wave w & 3 owns bytes 128 (w & 3)
        if (wave < 4) {
          char* p = base + c * 512 + 128 * wave + 16 * g;
          *(f32x4*)p = v;
          *(f32x4*)(p + 64) = v;
        }
      } else if (MODE == 1) {
        if (wave < 4) {
          char* p = base + (c & 7) * 512 + 128 * wave + ((c >> 3) * 64) + 16 * g;
          *(f32x4*)p = v;
          *(f32x4*)(p + 8 * 512) = v;
        }
      } else if (MODE == 2) {
        if (wave < 4) {
          char* p = base + (4 * wave + (lane >> 5)) * 512 + 16 * (lane & 31);
          *(f32x4*)p = v;
          *(f32x4*)(p + 2 * 512) = v;
        }
      } else {
        if (wave < 4) {
          char* p = base + wave * 2048 + 16 * lane;
          *(f32x4*)p = v;
          *(f32x4*)(p + 1024) = v;
        }
      }
    }
}
template <int MODE>
void run(const char* name, float* out, long rows) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, rows, 2);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-44s %8.1f us  %7.1f GB/s\n", name, best * 1e3, 2.0 * rows * 512 / (best * 1e-3) / 1e9);
}
int main() {
  const long rows = 2L << 20;  // 1 GiB
  float* out;
  (void)hipMalloc(&out, rows * 512);
  run<3>("contiguous 1 KB per instruction", out, rows);
  run<0>("16 rows x 64 B per instruction (half lines)", out, rows);
  run<1>("8 rows x 128 B per instruction (whole lines)", out, rows);
  run<2>("2 rows x 512 B per instruction (whole rows)", out, rows);
  run<0>("16 rows x 64 B per instruction (half lines)", out, rows);
  run<1>("8 rows x 128 B per instruction (whole lines)", out, rows);
  return 0;
}
