// Probe: the engine's gemm_full in isolation.  Not part of the product.
#include "../graph-physics_amd/csrc/mgn_kernels.hip"
#include <cstdio>

// MODE 0: gemm_full<NEXT=false>; 1: gemm_full<NEXT=true>; 2: no weight loads (constant operand)
template <int MT, int MODE, int WPS>
__global__ void __launch_bounds__(256, WPS) k_probe(const float* W, const float* X, float* out, int reps) {
  constexpr int HB = 8;
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
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
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) gemm_full<HB, MT, false>(acc, in, W, 128, c, g, nx);
    if (MODE == 1) gemm_full<HB, MT, true>(acc, in, W, 128, c, g, nx);
    if (MODE == 2) {
      f32x4 w = ld4(W + c * 128 + 4 * g);
#pragma unroll
      for (int kb = 0; kb < HB; ++kb)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][ib] = MFMA16(w[rr], in[t][kb][rr], acc[t][ib]);
    }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int t = 0; t < MT; ++t)
    for (int kb = 0; kb < HB; ++kb) s += acc[t][kb];
  out[row] = s[0] + s[1] + s[2] + s[3];
}

template <typename K>
void run(const char* name, K kern, int blocks, int mt, const float* W, const float* X, float* out) {
  const int reps = 60;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kern<<<blocks, 256>>>(W, X, out, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, 256>>>(W, X, out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double fl = 2.0 * 128 * 128 * 16.0 * mt * reps * blocks * 4.0;
  printf("%-40s blocks=%4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
}

int main() {
  float *W, *X, *out;
  hipMalloc(&W, 128 * 128 * 4);
  hipMalloc(&X, 4096 * 128 * 4);
  hipMalloc(&out, 4096 * 256 * 4);
  hipMemset(W, 0, 128 * 128 * 4);
  hipMemset(X, 0, 4096 * 128 * 4);
#define R(MT, MODE, WPS, B) run("MT=" #MT " mode=" #MODE " wps=" #WPS, k_probe<MT, MODE, WPS>, B, MT, W, X, out)
  R(2, 2, 2, 512); R(2, 0, 2, 512); R(2, 1, 2, 512);
  R(2, 2, 1, 256); R(2, 0, 1, 256); R(2, 1, 1, 256);
  R(1, 2, 2, 512); R(1, 0, 2, 512); R(1, 1, 2, 512);
  R(1, 0, 4, 1024); R(1, 1, 4, 1024);
  return 0;
}
