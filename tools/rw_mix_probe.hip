// What HBM sustains for a given read : write mix (streaming, 16-byte lanes, buffers far past the 256 MiB Infinity Cache).
// The training-mode chain kernels write two thirds of their traffic; the 8 TB/s spec is a read figure.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/rw_mix_probe tools/rw_mix_probe.hip ;  run: tools/rw_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int R, int W, bool NT = false>
__global__ void __launch_bounds__(256) k_mix(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int r = 0; r < R; ++r) v += in[(size_t)r * n + i];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      if (NT) __builtin_nontemporal_store(v * (float)(w + 1), &out[(size_t)w * n + i]);
      else out[(size_t)w * n + i] = v * (float)(w + 1);
    }
    if (W == 0 && v[0] == 12345.678f) out[0] = v;
  }
}
template <int R, int W, bool NT = false>
static void run(const f32x4* in, f32x4* out, size_t n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int grid : {2048, 8192}) {
    hipLaunchKernelGGL((k_mix<R, W, NT>), dim3(grid), dim3(256), 0, 0, in, out, n);
    hipEventRecord(e0, 0);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((k_mix<R, W, NT>), dim3(grid), dim3(256), 0, 0, in, out, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double gb = 5.0 * (R + W) * n * 16 / 1e9;
    printf("read %d : write %d%s  grid %5d  %7.1f GB/s\n", R, W, NT ? " (nt stores)" : "", grid, gb / (ms * 1e-3));
  }
}
int main() {
  const size_t n = (size_t)1 << 26;  // 1 GiB per stream
  f32x4 *in, *out;
  hipMalloc(&in, 3 * n * 16), hipMalloc(&out, 3 * n * 16);
  hipMemset(in, 0, 3 * n * 16), hipMemset(out, 0, 3 * n * 16);
  run<1, 0>(in, out, n), run<0, 1>(in, out, n), run<1, 1>(in, out, n), run<1, 2>(in, out, n), run<2, 1>(in, out, n), run<3, 1>(in, out, n),
      run<1, 3>(in, out, n);
  run<0, 1, true>(in, out, n), run<1, 1, true>(in, out, n), run<1, 2, true>(in, out, n), run<1, 3, true>(in, out, n);
  return 0;
}
