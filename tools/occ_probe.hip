// Probe: blocks/CU the runtime grants the engine's kernels at their real LDS sizes. Not product.
#include "../graph-physics_amd/csrc/mgn_kernels.hip"
#include <cstdio>
template <typename K>
void q(const char* name, K k, int threads, size_t smem) {
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  int nb = -1;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, threads, smem);
  hipFuncAttributes at;
  hipFuncGetAttributes(&at, (const void*)k);
  printf("%-24s smem=%6zu  blocks/CU=%d  regs=%d staticLDS=%zu err=%d\n", name, smem, nb, at.numRegs, at.sharedSizeBytes, (int)e);
}
int main() {
  q("k_wgrad_lds", k_wgrad_lds, 256, 4 * WG_TILE_BYTES);
  q("k_mlp_fwd_lds<1>", k_mlp_fwd_lds<1>, 256, FWD_LDS_BYTES);
  q("k_mlp_bwd_lds<1>", k_mlp_bwd_lds<1>, 256, 2 * WBUF_BYTES + 512 + 4 * 5 * 512);
  q("k_mlp_fwd_lds<1>@64K", k_mlp_fwd_lds<1>, 256, 65536);
  q("k_wgrad_lds@48K", k_wgrad_lds, 256, 49152);
  return 0;
}
