// Probe: the engine's LDS-fed half-GEMM loop in isolation.  Not part of the product.
#include "../graph-physics_amd/csrc/mgn_kernels.hip"
#include <cstdio>

template <int MT, int WPS, int MODE>
__global__ void __launch_bounds__(256, WPS) k_probe(const float* W, const float* X, float* out, int reps) {
  constexpr int HB = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* wl = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  int off[4];
  for (int j = 0; j < 4; ++j) off[j] = c * 256 + (((4 * j + g) ^ c) & 15) * 16;
  dma_weights(W, 128, 0, wl, wv, lane);
  dma_weights(W, 128, 1, wl + WBUF_BYTES, wv, lane);
  f32x4 in[MT][HB], acc[MT][HB];
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  const float* nx[MT];
  for (int t = 0; t < MT; ++t) {
    nx[t] = X + ((row * MT + t) % 4096) * 128 + 4 * g;
    for (int kb = 0; kb < HB; ++kb) {
      in[t][kb] = ld4(nx[t] + 16 * kb);
      acc[t][kb] = f32x4{0, 0, 0, 0};
    }
  }
  __syncthreads();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 1) __syncthreads();
    gemm_lds_half<MT, 0>(acc, in, wl, off, nx);
    if (MODE == 1) __syncthreads();
    gemm_lds_half<MT, 1>(acc, in, wl + WBUF_BYTES, off, nx);
  }
  f32x4 s = {0, 0, 0, 0};
  for (int t = 0; t < MT; ++t)
    for (int kb = 0; kb < HB; ++kb) s += acc[t][kb];
  out[row] = s[0] + s[1] + s[2] + s[3];
}

template <typename K>
void run(const char* name, K kern, int blocks, int mt, const float* W, const float* X, float* out) {
  const int reps = 60;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kern<<<blocks, 256, 65536>>>(W, X, out, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, 256, 65536>>>(W, X, out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double fl = 2.0 * 128 * 128 * 16.0 * mt * reps * blocks * 4.0;
  printf("%-44s blocks=%4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
}

int main() {
  float *W, *X, *out;
  hipMalloc(&W, 128 * 128 * 4);
  hipMalloc(&X, 4096 * 128 * 4);
  hipMalloc(&out, 4096 * 256 * 4);
  hipMemset(W, 0, 128 * 128 * 4);
  hipMemset(X, 0, 4096 * 128 * 4);
#define R(MT, WPS, MODE, B) run("lds gemm MT=" #MT " wps=" #WPS " barriers=" #MODE, k_probe<MT, WPS, MODE>, B, MT, W, X, out)
  R(1, 2, 0, 512); R(1, 2, 1, 512); R(2, 2, 0, 512); R(2, 2, 1, 512);
  R(1, 1, 0, 256); R(2, 1, 0, 256);
  return 0;
}
