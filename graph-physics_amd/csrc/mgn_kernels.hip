// MI355X (gfx950 / CDNA4) kernels of the MeshGraphNet message-passing engine.
//
// Register layouts (one wave = 64 lanes; c = lane & 15, g = lane >> 4):
//   T-layout: a 16-row x H tile held as f32x4 v[H/16]; lane (c,g) owns row c and
//             features 16*kb + 4*g + {0,1,2,3} of block kb.  It is (a) a plain
//             16-byte load/store per block from a row-major [M,H] matrix and
//             (b) simultaneously the B operand of v_mfma_f32_16x16x4_f32 for
//             k = 4*g + r AND its C/D layout, so a chain of Linear layers
//             computed transposed (Z^T = W X^T) never moves data between lanes:
//             the accumulator of layer l is the B operand of layer l+1.
//   N-layout: lane (c,g) owns feature 16*b + c of rows 4*g + {0..3}; the operand
//             layout of the weight-gradient product dW = dZ^T X.
// fp32 MFMA (v_mfma_f32_16x16x4_f32) is an exact fp32 fma chain; it runs at the
// fp32 vector rate (157 TF/s), so every kernel here is fed straight from L2
// with 16-byte loads and needs no LDS staging for operands.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mgn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// --------------------------------------------------------------------------
// Z^T[16*ib.., tile] += W[16*ib + c, 16*kb + 4g + r] * in[kb][r]   (T-layout chain)
// W row-major with leading dimension ldw; nib / nkb = active output / input blocks.
template <int HB, int MT>
__device__ __forceinline__ void gemm_tl(f32x4 (&acc)[MT][HB], const f32x4 (&in)[MT][HB],
                                        const float* __restrict__ W, int ldw, int nib, int nkb,
                                        int c, int g) {
  const float* wl = W + (size_t)c * ldw + 4 * g;
  constexpr int IP = (HB >= 2) ? 2 : 1;  // interleave two output blocks: independent accumulators
#pragma unroll
  for (int ib = 0; ib < HB; ib += IP) {
    if (ib < nib) {
#pragma unroll
      for (int kb = 0; kb < HB; ++kb) {
        if (kb < nkb) {
          f32x4 w[IP];
#pragma unroll
          for (int q = 0; q < IP; ++q)
            w[q] = (ib + q < nib) ? *(const f32x4*)(wl + (size_t)(16 * (ib + q)) * ldw + 16 * kb)
                                  : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
              for (int q = 0; q < IP; ++q)
                acc[t][ib + q] = MFMA16(w[q][r], in[t][kb][r], acc[t][ib + q]);
        }
      }
    }
  }
}

template <int HB, int MT>
__device__ __forceinline__ void load_tl(f32x4 (&v)[MT][HB], const float* __restrict__ src,
                                        const int32_t* __restrict__ idx, int kw, const long (&mm)[MT],
                                        int g, int& nkb) {
  constexpr int H = 16 * HB;
  if (kw == H) {
    nkb = HB;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const long row = idx ? (long)idx[mm[t]] : mm[t];
      const float* p = src + row * H + 4 * g;
#pragma unroll
      for (int kb = 0; kb < HB; ++kb) v[t][kb] = *(const f32x4*)(p + 16 * kb);
    }
  } else {  // ragged width: scalar guarded loads, zero padded to a multiple of 16
    nkb = (kw + 15) >> 4;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const long row = idx ? (long)idx[mm[t]] : mm[t];
      const float* p = src + row * kw;
#pragma unroll
      for (int kb = 0; kb < HB; ++kb) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (kb < nkb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = 16 * kb + 4 * g + r;
            if (k < kw) x[r] = p[k];
          }
        }
        v[t][kb] = x;
      }
    }
  }
}

template <int HB, int MT>
__device__ __forceinline__ void store_tl(float* __restrict__ dst, const f32x4 (&v)[MT][HB], int w,
                                         const long (&mm)[MT], const bool (&valid)[MT], int g) {
  constexpr int H = 16 * HB;
  if (w == H) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (valid[t]) {
        float* p = dst + mm[t] * H + 4 * g;
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) *(f32x4*)(p + 16 * kb) = v[t][kb];
      }
  } else {
    const int nb = (w + 15) >> 4;
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (valid[t]) {
        float* p = dst + mm[t] * w;
#pragma unroll
        for (int kb = 0; kb < HB; ++kb)
          if (kb < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * kb + 4 * g + r;
              if (k < w) p[k] = v[t][kb][r];
            }
          }
      }
  }
}

template <int HB, int MT>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[MT][HB], const float* __restrict__ b, int nib, int g) {
#pragma unroll
  for (int ib = 0; ib < HB; ++ib) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (b != nullptr && ib < nib) bv = *(const f32x4*)(b + 16 * ib + 4 * g);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t][ib] = bv;
  }
}

// sum over the 4 lane groups that share a row (lanes c, c+16, c+32, c+48)
__device__ __forceinline__ float rowsum4(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
// sum over the 16 rows of a tile (lanes with equal g)
__device__ __forceinline__ float colsum16(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}

// ========================================================================= forward
template <int HB, int MT>
__global__ void __launch_bounds__(256, (MT >= 4) ? 1 : 2) k_mlp_fwd(const mgn_mlp_fwd_args a) {
  constexpr int H = 16 * HB;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const long row0 = ((long)blockIdx.x * 4 + wv) * (16 * MT);
  if (row0 >= a.M) return;
  long mm[MT];
  bool valid[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const long m = row0 + 16 * t + c;
    valid[t] = m < a.M;
    mm[t] = valid[t] ? m : a.M - 1;
  }
  const int nib_last = (a.out_w + 15) >> 4;

  f32x4 in[MT][HB], acc[MT][HB];
  // ---- layer 0: phases of the concatenated input, gathered straight into MFMA operands
  {
    const int nib0 = (a.NL == 1) ? nib_last : HB;
    int ktot = 0;
    for (int p = 0; p < a.nphase; ++p) ktot += (a.kw[p] + 15) & ~15;
    init_bias<HB, MT>(acc, a.b[0], nib0, g);
    int koff = 0;
    for (int p = 0; p < a.nphase; ++p) {
      int nkb;
      load_tl<HB, MT>(in, a.src[p], a.idx[p], a.kw[p], mm, g, nkb);
      gemm_tl<HB, MT>(acc, in, a.W[0] + koff, ktot, nib0, nkb, c, g);
      koff += 16 * nkb;
    }
  }
  // ---- layers 1..NL-1: the accumulator IS the next B operand
  for (int l = 1; l < a.NL; ++l) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int ib = 0; ib < HB; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) in[t][ib][r] = fmaxf(acc[t][ib][r], 0.f);
    if (a.saveH[l - 1] != nullptr) store_tl<HB, MT>(a.saveH[l - 1], in, H, mm, valid, g);
    const int nib = (l == a.NL - 1) ? nib_last : HB;
    init_bias<HB, MT>(acc, a.b[l], nib, g);
    gemm_tl<HB, MT>(acc, in, a.W[l], H, nib, HB, c, g);
  }
  // ---- epilogue: RMSNorm (reference epsilon placement), residual, stores
  if (a.scale != nullptr) {
    f32x4 sc[HB];
#pragma unroll
    for (int ib = 0; ib < HB; ++ib) sc[ib] = *(const f32x4*)(a.scale + 16 * ib + 4 * g);
    const float sqrt_d = sqrtf((float)H);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      float ss = 0.f;
#pragma unroll
      for (int ib = 0; ib < HB; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) ss = fmaf(acc[t][ib][r], acc[t][ib][r], ss);
      ss = rowsum4(ss);
      const float rms = sqrtf(ss) / sqrt_d;
      const float den = rms + a.eps;
#pragma unroll
      for (int ib = 0; ib < HB; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) in[t][ib][r] = acc[t][ib][r] / den;  // u = z / (rms + eps)
      if (a.saveR != nullptr && valid[t] && g == 0) a.saveR[mm[t]] = rms;
#pragma unroll
      for (int ib = 0; ib < HB; ++ib) acc[t][ib] = sc[ib] * in[t][ib];
    }
    if (a.saveU != nullptr) store_tl<HB, MT>(a.saveU, in, H, mm, valid, g);
  }
  if (a.y_out != nullptr) store_tl<HB, MT>(a.y_out, acc, a.out_w, mm, valid, g);
  if (a.resid != nullptr) {
    int nkb;
    load_tl<HB, MT>(in, a.resid, nullptr, a.out_w, mm, g, nkb);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int ib = 0; ib < HB; ++ib) acc[t][ib] = in[t][ib] + acc[t][ib];
  }
  store_tl<HB, MT>(a.out, acc, a.out_w, mm, valid, g);
}

// ======================================================================== backward
// Per-block partial column sums live in LDS as [wave][slot][H]; slot l < NL is db[l],
// slot NL is dscale.  They go to red_ws[block][slot][H] and are reduced by k_colred.
template <int HB, int MT>
__device__ __forceinline__ void colsum_to_lds(float* lds_w, const f32x4 (&v)[MT][HB], int c, int g) {
#pragma unroll
  for (int ib = 0; ib < HB; ++ib) {
    f32x4 s = v[0][ib];
#pragma unroll
    for (int t = 1; t < MT; ++t) s += v[t][ib];
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = colsum16(s[r]);
    if (c == 0) *(f32x4*)(lds_w + 16 * ib + 4 * g) = s;
  }
}

template <int HB, int MT>
__global__ void __launch_bounds__(256, (MT >= 4) ? 1 : 2) k_mlp_bwd(const mgn_mlp_bwd_args a) {
  constexpr int H = 16 * HB;
  __shared__ float lds[4 * (MGN_MAX_LAYERS + 1) * H];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const long row0 = ((long)blockIdx.x * 4 + wv) * (16 * MT);
  const int nslot = a.NL + 1;
  float* lds_w = lds + wv * nslot * H;
  // zero this wave's partial slots (waves past M contribute zeros)
  for (int i = lane; i < nslot * H; i += 64) lds_w[i] = 0.f;

  if (row0 < a.M) {
    long mm[MT];
    bool valid[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const long m = row0 + 16 * t + c;
      valid[t] = m < a.M;
      mm[t] = valid[t] ? m : a.M - 1;
    }
    const int nkb_last = (a.out_w + 15) >> 4;
    f32x4 dz[MT][HB], acc[MT][HB];
    int nkb;
    load_tl<HB, MT>(dz, a.dOut, nullptr, a.out_w, mm, g, nkb);
    if (a.dOut2 != nullptr) {
      load_tl<HB, MT>(acc, a.dOut2, a.idx2, H, mm, g, nkb);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) dz[t][ib] += acc[t][ib];
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (!valid[t]) {
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) dz[t][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    // ---- RMSNorm backward:  dz = g/(rms+eps) - u * <g,u> / (H*rms),  g = scale*dy
    if (a.scale != nullptr) {
      f32x4 sc[HB];
#pragma unroll
      for (int ib = 0; ib < HB; ++ib) sc[ib] = *(const f32x4*)(a.scale + 16 * ib + 4 * g);
      load_tl<HB, MT>(acc, a.U, nullptr, H, mm, g, nkb);  // acc <- u
      f32x4 du[MT][HB];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        float dot = 0.f;
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) {
          du[t][ib] = dz[t][ib] * acc[t][ib];  // dy*u -> dscale
          const f32x4 gg = sc[ib] * dz[t][ib];
#pragma unroll
          for (int r = 0; r < 4; ++r) dot = fmaf(gg[r], acc[t][ib][r], dot);
          dz[t][ib] = gg;
        }
        dot = rowsum4(dot);
        const float rms = a.R[mm[t]];
        const float inv = 1.0f / (rms + a.eps);
        const float k2 = (rms > 0.f) ? dot / ((float)H * rms) : 0.f;
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) dz[t][ib] = dz[t][ib] * inv - acc[t][ib] * k2;
      }
      if (a.dscale != nullptr) colsum_to_lds<HB, MT>(lds_w + a.NL * H, du, c, g);
    }
    if (a.dZ[a.NL - 1] != nullptr) store_tl<HB, MT>(a.dZ[a.NL - 1], dz, 16 * nkb_last, mm, valid, g);
    if (a.db[a.NL - 1] != nullptr) colsum_to_lds<HB, MT>(lds_w + (a.NL - 1) * H, dz, c, g);
    // ---- dgrad chain with ReLU masks from the saved activations
    for (int l = a.NL - 1; l >= 1; --l) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) acc[t][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nk = (l == a.NL - 1) ? nkb_last : HB;
      gemm_tl<HB, MT>(acc, dz, a.WT[l], 16 * nk, HB, nk, c, g);
      load_tl<HB, MT>(dz, a.Hs[l - 1], nullptr, H, mm, g, nkb);  // dz <- h_l (post-ReLU)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dz[t][ib][r] = (valid[t] && dz[t][ib][r] > 0.f) ? acc[t][ib][r] : 0.f;
      if (a.dZ[l - 1] != nullptr) store_tl<HB, MT>(a.dZ[l - 1], dz, H, mm, valid, g);
      if (a.db[l - 1] != nullptr) colsum_to_lds<HB, MT>(lds_w + (l - 1) * H, dz, c, g);
    }
    // ---- gradients wrt the requested first-layer input blocks
    for (int q = 0; q < a.n_din; ++q) {
      if (a.din_resid[q] != nullptr) {
        load_tl<HB, MT>(acc, a.din_resid[q], nullptr, H, mm, g, nkb);
      } else {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int ib = 0; ib < HB; ++ib) acc[t][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const int nk0 = (a.NL == 1) ? nkb_last : HB;
      gemm_tl<HB, MT>(acc, dz, a.WT0[q], 16 * nk0, HB, nk0, c, g);
      store_tl<HB, MT>(a.dIn[q], acc, H, mm, valid, g);
    }
  }
  __syncthreads();
  // fixed-order sum of the 4 waves' partials -> red_ws[block][slot][H]
  float* ws = (float*)a.red_ws + (size_t)blockIdx.x * nslot * H;
  for (int i = threadIdx.x; i < nslot * H; i += 256)
    ws[i] = ((lds[i] + lds[nslot * H + i]) + lds[2 * nslot * H + i]) + lds[3 * nslot * H + i];
}

// out[s][j] = sum_b ws[b][s][j];  one thread per (slot, j), serial over blocks in chunks
struct ColredOuts {
  float* o[MGN_MAX_LAYERS + 1];
};
__global__ void k_colred(const float* __restrict__ ws, int nblocks, int nslot, int H, const ColredOuts outs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nslot * H) return;
  float* o = outs.o[i / H];
  if (o == nullptr) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const size_t st = (size_t)nslot * H;
  int b = 0;
  for (; b + 4 <= nblocks; b += 4) {
    s0 += ws[(size_t)b * st + i];
    s1 += ws[(size_t)(b + 1) * st + i];
    s2 += ws[(size_t)(b + 2) * st + i];
    s3 += ws[(size_t)(b + 3) * st + i];
  }
  for (; b < nblocks; ++b) s0 += ws[(size_t)b * st + i];
  o[i % H] = (s0 + s1) + (s2 + s3);
}

// ===================================================================== weight grads
struct WgradLaunch {
  int njobs;
  mgn_wgrad_job job[MGN_MAX_WGRAD_JOBS];
  int wg0[MGN_MAX_WGRAD_JOBS + 1];  // first workgroup of each job
  float* partial;                   // [total_wg][H*H]
  int H;
};

template <int HB>
__global__ void __launch_bounds__(256, 2) k_wgrad(const WgradLaunch L) {
  constexpr int H = 16 * HB;
  constexpr int KPW = (HB + 3) / 4;  // k-blocks per wave
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  int j = 0;
  while (j + 1 < L.njobs && (int)blockIdx.x >= L.wg0[j + 1]) ++j;
  const mgn_wgrad_job J = L.job[j];
  const int nwg = L.wg0[j + 1] - L.wg0[j];
  const int wg = blockIdx.x - L.wg0[j];
  const long ntiles = (J.M + 15) >> 4;
  const long t0 = ntiles * wg / nwg, t1 = ntiles * (wg + 1) / nwg;

  f32x4 acc[KPW][HB];
#pragma unroll
  for (int kk = 0; kk < KPW; ++kk)
#pragma unroll
    for (int jb = 0; jb < HB; ++jb) acc[kk][jb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kb0 = wv * KPW;
  if (kb0 < J.nkb) {
    for (long tile = t0; tile < t1; ++tile) {
      float av[HB][4], bv[KPW][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = tile * 16 + 4 * g + r;
        const bool ok = row < J.M;
        const float* ap = J.A + (ok ? row : 0) * J.lda + c;
        const float* bp = J.B + (ok ? row : 0) * J.ldb + c;
#pragma unroll
        for (int jb = 0; jb < HB; ++jb) av[jb][r] = (ok && jb < J.nja) ? ap[16 * jb] : 0.f;
#pragma unroll
        for (int kk = 0; kk < KPW; ++kk) {
          const int col = 16 * (kb0 + kk) + c;
          bv[kk][r] = (ok && col < J.kw) ? bp[16 * (kb0 + kk)] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int kk = 0; kk < KPW; ++kk)
#pragma unroll
          for (int jb = 0; jb < HB; ++jb)
            if (jb < J.nja) acc[kk][jb] = MFMA16(av[jb][r], bv[kk][r], acc[kk][jb]);
    }
  }
  // D layout: lane (c,g), reg q -> dW[16*jb + 4g + q][16*kb + c]
  float* P = L.partial + (size_t)blockIdx.x * (H * H);
#pragma unroll
  for (int kk = 0; kk < KPW; ++kk) {
    const int kb = kb0 + kk;
    if (kb < J.nkb) {
#pragma unroll
      for (int jb = 0; jb < HB; ++jb)
        if (jb < J.nja) {
#pragma unroll
          for (int q = 0; q < 4; ++q) P[(16 * jb + 4 * g + q) * H + 16 * kb + c] = acc[kk][jb][q];
        }
    }
  }
}

__global__ void k_wgrad_red(const WgradLaunch L) {
  const int H = L.H;
  int j = blockIdx.y;
  const mgn_wgrad_job J = L.job[j];
  const int rows = 16 * J.nja, cols = 16 * J.nkb;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, k = i % cols;
  const float* P = L.partial + (size_t)L.wg0[j] * (H * H) + r * H + k;
  const int nwg = L.wg0[j + 1] - L.wg0[j];
  float s0 = 0.f, s1 = 0.f;
  int b = 0;
  for (; b + 2 <= nwg; b += 2) {
    s0 += P[(size_t)b * (H * H)];
    s1 += P[(size_t)(b + 1) * (H * H)];
  }
  if (b < nwg) s0 += P[(size_t)b * (H * H)];
  if (k < J.ldw) J.dW[(size_t)r * J.ldw + k] = s0 + s1;
}

// ===================================================================== segment sum
// One row = H floats = H/4 lanes of float4.  A group of LPR lanes walks one CSR
// segment in k order (the CPU index_add_ order) with 8 independent loads in flight.
template <int HB>
__global__ void __launch_bounds__(256) k_segsum(const float* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                const int32_t* __restrict__ perm, float* __restrict__ out, long N) {
  constexpr int H = 16 * HB;
  constexpr int LPR = H / 4;
  const long node = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  if (node >= N) return;
  const int beg = rowptr[node], end = rowptr[node + 1];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const float* base = src + 4 * l;
  for (int k = beg; k < end; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = k + u;
      if (kk < end) {
        const long row = perm ? (long)perm[kk] : (long)kk;
        v[u] = *(const f32x4*)(base + row * H);
      } else {
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < end) s += v[u];
  }
  *(f32x4*)(out + node * H + 4 * l) = s;
}

// ======================================================================= CSR build
__global__ void k_csr_hist(const int64_t* __restrict__ key, long E, long N, int* __restrict__ cnt, int* __restrict__ err) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t k = key[e];
  if (k < 0 || k >= N) {
    *err = 1;
    return;
  }
  atomicAdd(&cnt[k], 1);
}

// exclusive scan of cnt[0..N) into rowptr[0..N]; single block of 1024 threads
__global__ void __launch_bounds__(1024) k_csr_scan(const int* __restrict__ cnt, long N, int32_t* __restrict__ rowptr) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (long base = 0; base < N; base += 1024) {
    const long i = base + tid;
    const int v = (i < N) ? cnt[i] : 0;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; ++w) woff += wsum[w];
    const int carry = carry_s;
    if (i < N) rowptr[i] = carry + woff + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) rowptr[N] = carry_s;
}

__global__ void k_csr_fill(const int64_t* __restrict__ key, long E, long N, const int32_t* __restrict__ rowptr,
                           int* __restrict__ cursor, int32_t* __restrict__ perm) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t k = key[e];
  if (k < 0 || k >= N) return;
  const int pos = atomicAdd(&cursor[k], 1);
  perm[rowptr[k] + pos] = (int32_t)e;
}

// make every segment ascending in edge id (stable counting sort == CPU order).
// Small segments: insertion sort by one thread.  Large ones go to a worklist and
// are rank-sorted by a whole workgroup (k_csr_sortbig) through a scratch copy.
#define CSR_SMALL_SEG 48
__global__ void k_csr_sortseg(const int32_t* __restrict__ rowptr, long N, int32_t* __restrict__ perm,
                              int* __restrict__ worklist, int* __restrict__ nbig) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int beg = rowptr[i], end = rowptr[i + 1];
  if (end - beg > CSR_SMALL_SEG) {
    worklist[atomicAdd(nbig, 1)] = (int)i;
    return;
  }
  for (int a = beg + 1; a < end; ++a) {
    const int v = perm[a];
    int b = a - 1;
    while (b >= beg && perm[b] > v) {
      perm[b + 1] = perm[b];
      --b;
    }
    perm[b + 1] = v;
  }
}

__global__ void __launch_bounds__(256) k_csr_sortbig(const int32_t* __restrict__ rowptr, int32_t* __restrict__ perm,
                                                     const int* __restrict__ worklist, const int* __restrict__ nbig,
                                                     int32_t* __restrict__ tmp) {
  const int n = *nbig;
  for (int w = blockIdx.x; w < n; w += gridDim.x) {
    const int node = worklist[w];
    const int beg = rowptr[node], d = rowptr[node + 1] - beg;
    for (int i = threadIdx.x; i < d; i += 256) tmp[beg + i] = perm[beg + i];
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += 256) {
      const int v = tmp[beg + i];
      int rank = 0;
      for (int j = 0; j < d; ++j) rank += (tmp[beg + j] < v) ? 1 : 0;  // edge ids are distinct
      perm[beg + rank] = v;
    }
    __syncthreads();
  }
}

// ============================================================================ C ABI
static thread_local char g_err[256] = "";
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}
static int pick_mt(int64_t M) {
  // enough 16*MT-row wave tiles to give each of the 1024 SIMDs ~2 waves
  if (M >= (int64_t)64 * 2048) return 2;
  return 1;
}

template <int HB>
static void launch_fwd(const mgn_mlp_fwd_args& a, hipStream_t s) {
  const int mt = pick_mt(a.M);
  const unsigned grid = (unsigned)((a.M + 64 * mt - 1) / (64 * mt));
  if (mt == 2)
    hipLaunchKernelGGL((k_mlp_fwd<HB, 2>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_mlp_fwd<HB, 1>), dim3(grid), dim3(256), 0, s, a);
}

static unsigned bwd_grid(int64_t M) {
  const int mt = pick_mt(M);
  return (unsigned)((M + 64 * mt - 1) / (64 * mt));
}

template <int HB>
static void launch_bwd(const mgn_mlp_bwd_args& a, hipStream_t s) {
  const int mt = pick_mt(a.M);
  const unsigned grid = bwd_grid(a.M);
  if (mt == 2)
    hipLaunchKernelGGL((k_mlp_bwd<HB, 2>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_mlp_bwd<HB, 1>), dim3(grid), dim3(256), 0, s, a);
}

extern "C" {

int mgn_version(void) { return 100; }
const char* mgn_last_error(void) { return g_err; }

size_t mgn_csr_workspace_bytes(int64_t E, int64_t N) {
  // cnt/cursor[N] + err + nbig (16 ints) | worklist[N] | tmp[E]
  return (size_t)(2 * N + E + 32) * sizeof(int);
}

int mgn_csr_build(const int64_t* key, int64_t E, int64_t N, int32_t* rowptr, int32_t* perm, void* ws,
                  size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (N < 0 || E < 0 || E > 2147483647LL || N > 2147483646LL) return fail(1, "mgn_csr_build: size out of int32 range");
  if (ws_bytes < mgn_csr_workspace_bytes(E, N)) return fail(1, "mgn_csr_build: workspace too small");
  int* cnt = (int*)ws;
  int* err = cnt + N;
  if (hipMemsetAsync(ws, 0, (size_t)(N + 16) * sizeof(int), s) != hipSuccess) return fail(2, "mgn_csr_build: memset");
  if (E > 0) hipLaunchKernelGGL(k_csr_hist, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, key, (long)E, (long)N, cnt, err);
  hipLaunchKernelGGL(k_csr_scan, dim3(1), dim3(1024), 0, s, cnt, (long)N, rowptr);
  int herr = 0;
  if (hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return fail(2, "mgn_csr_build: memcpy");
  if (hipMemsetAsync(ws, 0, (size_t)N * sizeof(int), s) != hipSuccess) return fail(2, "mgn_csr_build: memset");
  if (E > 0) {
    hipLaunchKernelGGL(k_csr_fill, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, key, (long)E, (long)N, rowptr, cnt, perm);
    if (N > 0) {
      int* nbig = cnt + N + 1;
      int* worklist = cnt + N + 16;
      int32_t* tmp = worklist + N;
      hipLaunchKernelGGL(k_csr_sortseg, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, rowptr, (long)N, perm, worklist, nbig);
      hipLaunchKernelGGL(k_csr_sortbig, dim3(512), dim3(256), 0, s, rowptr, perm, worklist, nbig, tmp);
    }
  }
  if (int rc = check_launch("mgn_csr_build")) return rc;
  if (hipStreamSynchronize(s) != hipSuccess) return fail(2, "mgn_csr_build: sync failed");
  if (herr) return fail(3, "mgn_csr_build: edge index outside [0, N)");
  return 0;
}

int mgn_segsum(const float* src, const int32_t* rowptr, const int32_t* perm, float* out, int64_t N, int H, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) return 0;
  const int lpr = H / 4;
  const unsigned grid = (unsigned)((N * lpr + 255) / 256);
  switch (H) {
    case 128: hipLaunchKernelGGL(k_segsum<8>, dim3(grid), dim3(256), 0, s, src, rowptr, perm, out, (long)N); break;
    case 64: hipLaunchKernelGGL(k_segsum<4>, dim3(grid), dim3(256), 0, s, src, rowptr, perm, out, (long)N); break;
    case 32: hipLaunchKernelGGL(k_segsum<2>, dim3(grid), dim3(256), 0, s, src, rowptr, perm, out, (long)N); break;
    case 16: hipLaunchKernelGGL(k_segsum<1>, dim3(grid), dim3(256), 0, s, src, rowptr, perm, out, (long)N); break;
    default: return fail(1, "mgn_segsum: H must be 16, 32, 64 or 128");
  }
  return check_launch("mgn_segsum");
}

static int check_mlp_common(int H, int NL, int out_w, const char* who) {
  if (!(H == 16 || H == 32 || H == 64 || H == 128)) return fail(1, "H must be 16, 32, 64 or 128");
  if (NL < 1 || NL > MGN_MAX_LAYERS) return fail(1, "NL out of range");
  if (out_w < 1 || out_w > H) return fail(1, "out_w out of range");
  (void)who;
  return 0;
}


int mgn_mlp_fwd(const mgn_mlp_fwd_args* args, void* stream) {
  const mgn_mlp_fwd_args& a = *args;
  if (int rc = check_mlp_common(a.H, a.NL, a.out_w, "mgn_mlp_fwd")) return rc;
  if (a.nphase < 1 || a.nphase > MGN_MAX_PHASES) return fail(1, "mgn_mlp_fwd: nphase out of range");
  for (int p = 0; p < a.nphase; ++p)
    if (a.kw[p] < 1 || a.kw[p] > a.H) return fail(1, "mgn_mlp_fwd: phase width out of range");
  if (a.scale != nullptr && a.out_w != a.H) return fail(1, "mgn_mlp_fwd: RMSNorm needs out_w == H");
  if (a.M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  switch (a.H) {
    case 128: launch_fwd<8>(a, s); break;
    case 64: launch_fwd<4>(a, s); break;
    case 32: launch_fwd<2>(a, s); break;
    default: launch_fwd<1>(a, s); break;
  }
  return check_launch("mgn_mlp_fwd");
}


size_t mgn_mlp_bwd_workspace_bytes(int64_t M, int H, int NL) {
  return (size_t)(bwd_grid(M) + 1) * (NL + 1) * H * sizeof(float);
}


int mgn_mlp_bwd(const mgn_mlp_bwd_args* args, void* stream) {
  const mgn_mlp_bwd_args& a = *args;
  if (int rc = check_mlp_common(a.H, a.NL, a.out_w, "mgn_mlp_bwd")) return rc;
  if (a.n_din < 0 || a.n_din > MGN_MAX_PHASES) return fail(1, "mgn_mlp_bwd: n_din out of range");
  if (a.M == 0) return 0;
  if (a.red_ws_bytes < mgn_mlp_bwd_workspace_bytes(a.M, a.H, a.NL)) return fail(1, "mgn_mlp_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  switch (a.H) {
    case 128: launch_bwd<8>(a, s); break;
    case 64: launch_bwd<4>(a, s); break;
    case 32: launch_bwd<2>(a, s); break;
    default: launch_bwd<1>(a, s); break;
  }
  if (int rc = check_launch("mgn_mlp_bwd")) return rc;
  // column reductions: db[l], dscale
  const unsigned grid = bwd_grid(a.M);
  const int nslot = a.NL + 1;
  ColredOuts outs;
  bool any = false;
  for (int l = 0; l <= MGN_MAX_LAYERS; ++l) outs.o[l] = nullptr;
  for (int l = 0; l < a.NL; ++l) {
    outs.o[l] = a.db[l];
    any |= a.db[l] != nullptr;
  }
  outs.o[a.NL] = (a.scale != nullptr) ? a.dscale : nullptr;
  any |= outs.o[a.NL] != nullptr;
  if (!any) return 0;
  const int n = nslot * a.H;
  hipLaunchKernelGGL(k_colred, dim3((n + 127) / 128), dim3(128), 0, s, (const float*)a.red_ws, (int)grid, nslot, a.H, outs);
  return check_launch("mgn_mlp_bwd/colred");
}

static int wgrad_plan(int njobs, const mgn_wgrad_job* jobs, int* wg0) {
  // ~1024 workgroups in total, shared out in proportion to the rows of each job
  int64_t tot = 0;
  for (int j = 0; j < njobs; ++j) tot += (jobs[j].M + 15) / 16;
  int acc = 0;
  for (int j = 0; j < njobs; ++j) {
    const int64_t tiles = (jobs[j].M + 15) / 16;
    int64_t n = tot > 0 ? (tiles * 1024 + tot - 1) / tot : 1;
    const int64_t cap = (tiles + 7) / 8;  // at least 8 tiles (128 rows) per workgroup
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    wg0[j] = acc;
    acc += (int)n;
  }
  wg0[njobs] = acc;
  return acc;
}

size_t mgn_wgrad_workspace_bytes(int njobs, const mgn_wgrad_job* jobs) {
  if (njobs < 1 || njobs > MGN_MAX_WGRAD_JOBS) return 0;
  int wg0[MGN_MAX_WGRAD_JOBS + 1];
  const int total = wgrad_plan(njobs, jobs, wg0);
  return (size_t)total * 128 * 128 * sizeof(float);
}

int mgn_wgrad(int njobs, const mgn_wgrad_job* jobs, void* ws, size_t ws_bytes, void* stream) {
  if (njobs < 1 || njobs > MGN_MAX_WGRAD_JOBS) return fail(1, "mgn_wgrad: njobs out of range");
  WgradLaunch L;
  L.njobs = njobs;
  int maxb = 1;
  for (int j = 0; j < njobs; ++j) {
    L.job[j] = jobs[j];
    if (jobs[j].nja < 1 || jobs[j].nkb < 1 || jobs[j].nja > 8 || jobs[j].nkb > 8) return fail(1, "mgn_wgrad: block counts out of range");
    if (jobs[j].nja > maxb) maxb = jobs[j].nja;
    if (jobs[j].nkb > maxb) maxb = jobs[j].nkb;
  }
  const int HB = maxb <= 1 ? 1 : maxb <= 2 ? 2 : maxb <= 4 ? 4 : 8;
  L.H = 16 * HB;
  const int total = wgrad_plan(njobs, jobs, L.wg0);
  if (ws_bytes < (size_t)total * L.H * L.H * sizeof(float)) return fail(1, "mgn_wgrad: workspace too small");
  L.partial = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  switch (HB) {
    case 8: hipLaunchKernelGGL(k_wgrad<8>, dim3(total), dim3(256), 0, s, L); break;
    case 4: hipLaunchKernelGGL(k_wgrad<4>, dim3(total), dim3(256), 0, s, L); break;
    case 2: hipLaunchKernelGGL(k_wgrad<2>, dim3(total), dim3(256), 0, s, L); break;
    default: hipLaunchKernelGGL(k_wgrad<1>, dim3(total), dim3(256), 0, s, L); break;
  }
  if (int rc = check_launch("mgn_wgrad")) return rc;
  hipLaunchKernelGGL(k_wgrad_red, dim3((L.H * L.H + 255) / 256, njobs), dim3(256), 0, s, L);
  return check_launch("mgn_wgrad/reduce");
}

}  // extern "C"
