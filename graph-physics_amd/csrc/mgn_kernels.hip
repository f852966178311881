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
#include <stdlib.h>
#include <type_traits>
#include "mgn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Every hot access is forced into the GLOBAL address space: a pointer that went through
// a select loses its provenance and becomes a FLAT access, and one flat load among
// global loads makes hipcc fall back to s_waitcnt vmcnt(0) -- a full stall per step.
typedef __attribute__((address_space(1))) const f32x4 g_cf32x4;
typedef __attribute__((address_space(1))) f32x4 g_f32x4;
typedef __attribute__((address_space(1))) const float g_cfloat;
typedef __attribute__((address_space(1))) float g_float;
#ifdef MGN_EXP_NOLOAD   // timing experiment only: rows are not fetched
__device__ __forceinline__ f32x4 ld4(const float* p) { return f32x4{1.f, 2.f, (float)(size_t)p, 0.5f}; }
#else
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(g_cf32x4*)p; }
#endif
#ifdef MGN_EXP_NOSTORE  // timing experiment only: results are dropped unless they are NaN (keeps the math alive)
__device__ __forceinline__ void st4(float* p, f32x4 v) { if (v[0] != v[0]) *(g_f32x4*)p = v; }
#else
__device__ __forceinline__ void st4(float* p, f32x4 v) { *(g_f32x4*)p = v; }
#endif
// Write-once data that is only read again much later (saved activations for the backward pass).  A store
// instruction covers 64 B of each of 16 rows -- HALF a 128-byte line -- and the other half follows one
// instruction later: through plain stores the XCD's L2 merges the halves into whole-line write-backs;
// non-temporal stores (the round-1 choice, -2.7 % on the exact-fp32 generation) let them leave separately
// and cost the packed kernel 12 % more write traffic (545 vs 487 MB per launch, WRITE_SIZE) and 6 % of its
// time (239/232/237 vs 224/222/221 us, alternating A/B in one GPU call).  -DMGN_EXP_NT_SAVES rebuilds the
// non-temporal variant.
#ifdef MGN_EXP_NOSTORE
__device__ __forceinline__ void st4_stream(float* p, f32x4 v) { if (v[0] != v[0]) *(g_f32x4*)p = v; }
#elif defined(MGN_EXP_NT_SAVES)
__device__ __forceinline__ void st4_stream(float* p, f32x4 v) { __builtin_nontemporal_store(v, (g_f32x4*)p); }
#else
__device__ __forceinline__ void st4_stream(float* p, f32x4 v) { *(g_f32x4*)p = v; }
#endif
__device__ __forceinline__ float ld1(const float* p) { return *(g_cfloat*)p; }
__device__ __forceinline__ void st1(float* p, float v) { *(g_float*)p = v; }

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
            w[q] = (ib + q < nib) ? ld4(wl + (size_t)(16 * (ib + q)) * ldw + 16 * kb) : f32x4{0.f, 0.f, 0.f, 0.f};
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

// Guard-free variant for full H x H blocks -- the hot one.  Straight-line code with
//  * kb-outer order: input block kb is dead once every output block consumed it, so
//    (NEXT) the rows of the NEXT tensor (next gathered phase, or the saved activation
//    the backward chain masks with) are loaded into in[.][kb] while the MFMAs of the
//    remaining blocks run -- a prefetch that costs no extra registers;
//  * a 3-deep ring of weight fragments loaded PD=2 steps ahead (L2 latency ~ 2 steps).
// Consecutive MFMAs hit 2*MT different accumulators (dependent latency 40 cyc > issue 32).
template <int HB, int MT, bool NEXT>
__device__ __forceinline__ void gemm_full(f32x4 (&acc)[MT][HB], f32x4 (&in)[MT][HB],
                                          const float* __restrict__ W, int ldw, int c, int g,
                                          const float* const (&nxt)[MT]) {
  constexpr int IP = (HB >= 2) ? 2 : 1;
  constexpr int NI = HB / IP;
  constexpr int NS = HB * NI;
  constexpr int PD = 2;
  const float* wl = W + (size_t)c * ldw + 4 * g;
  f32x4 w[PD + 1][IP];
#pragma unroll
  for (int s = 0; s < PD && s < NS; ++s)
#pragma unroll
    for (int q = 0; q < IP; ++q)
      w[s % (PD + 1)][q] = ld4(wl + (size_t)(16 * ((s % NI) * IP + q)) * ldw + 16 * (s / NI));
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int kb = s / NI, ibp = s % NI;
    if (s + PD < NS) {
      const int s2 = s + PD;
#pragma unroll
      for (int q = 0; q < IP; ++q)
        w[s2 % (PD + 1)][q] = ld4(wl + (size_t)(16 * ((s2 % NI) * IP + q)) * ldw + 16 * (s2 / NI));
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the hand-placed pipeline: no hoisting of later loads
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < IP; ++q)
          acc[t][ibp * IP + q] = MFMA16(w[s % (PD + 1)][q][r], in[t][kb][r], acc[t][ibp * IP + q]);
    if (NEXT && ibp == NI - 1) {
#pragma unroll
      for (int t = 0; t < MT; ++t) in[t][kb] = ld4(nxt[t] + 16 * kb);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int HB, int MT, bool RAGGED = true>
__device__ __forceinline__ void load_tl(f32x4 (&v)[MT][HB], const float* __restrict__ src,
                                        const int32_t* __restrict__ idx, int kw, const long (&mm)[MT],
                                        int g, int& nkb) {
  constexpr int H = 16 * HB;
  if (!RAGGED || kw == H) {
    nkb = HB;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const long row = idx ? (long)idx[mm[t]] : mm[t];
      const float* p = src + row * H + 4 * g;
#pragma unroll
      for (int kb = 0; kb < HB; ++kb) v[t][kb] = ld4(p + 16 * kb);
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
            if (k < kw) x[r] = ld1(p + k);
          }
        }
        v[t][kb] = x;
      }
    }
  }
}

template <int HB, int MT, bool RAGGED = true>
__device__ __forceinline__ void store_tl(float* __restrict__ dst, const f32x4 (&v)[MT][HB], int w,
                                         const long (&mm)[MT], const bool (&valid)[MT], int g) {
  constexpr int H = 16 * HB;
  if (!RAGGED || w == H) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (valid[t]) {
        float* p = dst + mm[t] * H + 4 * g;
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) st4(p + 16 * kb, v[t][kb]);
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
              if (k < w) st1(p + k, v[t][kb][r]);
            }
          }
      }
  }
}

template <int HB, int MT>
__device__ __forceinline__ void store_tl_stream(float* __restrict__ dst, const f32x4 (&v)[MT][HB],
                                                const long (&mm)[MT], const bool (&valid)[MT], int g) {
  constexpr int H = 16 * HB;
#pragma unroll
  for (int t = 0; t < MT; ++t)
    if (valid[t]) {
      float* p = dst + mm[t] * H + 4 * g;
#pragma unroll
      for (int kb = 0; kb < HB; ++kb) st4_stream(p + 16 * kb, v[t][kb]);
    }
}

template <int HB, int MT>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[MT][HB], const float* __restrict__ b, int nib, int g) {
#pragma unroll
  for (int ib = 0; ib < HB; ++ib) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (b != nullptr && ib < nib) bv = ld4(b + 16 * ib + 4 * g);
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

// SiLU (layers.py:132-160, nn.SiLU): z * sigmoid(z); its derivative sigma * (1 + z * (1 - sigma)).
__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + expf(-z)); }
__device__ __forceinline__ float dsilu_f(float z) {
  const float sg = 1.0f / (1.0f + expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}
// GELU (nn.GELU(), exact erf form; build_mlp(act="gelu"), layers.py:150-160): z Phi(z); derivative Phi(z) + z phi(z).
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float z) {
  return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * expf(-0.5f * z * z);
}
__device__ __forceinline__ bool act_smooth(int act) { return act != MGN_ACT_RELU; }  // needs the saved PRE-activations
__device__ __forceinline__ float act_f(float z, int act) {
  return act == MGN_ACT_SILU ? silu_f(z) : (act == MGN_ACT_GELU ? gelu_f(z) : fmaxf(z, 0.f));
}
__device__ __forceinline__ float dact_f(float z, int act) { return act == MGN_ACT_GELU ? dgelu_f(z) : dsilu_f(z); }

// ========================================================================= forward
// bf16 matrix mode of the GENERIC kernels [r4] (any supported width; precision == 1 as in the packed kernels): every Linear
// takes operands rounded to bf16 and returns a bf16 value, fp32 accumulation in between -- on the exact-fp32 MFMA, whose products
// of bf16-representable operands are exact (the semantic of the reference under Lightning bf16-mixed, train.py:74-78,268-293).
// The caller hands over weights and biases already rounded; the kernels round the row operands and each layer's result.
__device__ __forceinline__ float bf16_round(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);   // round to nearest even (finite values)
  return __builtin_bit_cast(float, ((u & 0x7f800000u) == 0x7f800000u ? u : r) & 0xffff0000u);
}
template <int HB, int MT>
__device__ __forceinline__ void round_tl(f32x4 (&v)[MT][HB]) {
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int ib = 0; ib < HB; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[t][ib][r] = bf16_round(v[t][ib][r]);
}

template <int HB, int MT, bool RAGGED>
__global__ void __launch_bounds__(256, (MT >= 4) ? 1 : 2) k_mlp_fwd(const mgn_mlp_fwd_args a) {
  constexpr int H = 16 * HB;
  const bool bf = a.precision == 1;
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
  const float* dummy[MT];  // any readable address: target of the redundant "next" loads
#pragma unroll
  for (int t = 0; t < MT; ++t) dummy[t] = a.W[0] + 4 * g;
  const int nib0 = (a.NL == 1) ? nib_last : HB;
  int ktot = 0;
  bool full0 = (nib0 == HB);
  for (int p = 0; p < a.nphase; ++p) {
    ktot += (a.kw[p] + 15) & ~15;
    full0 = full0 && (a.kw[p] == H);
  }
  init_bias<HB, MT>(acc, a.b[0], nib0, g);
  int nkb;
  load_tl<HB, MT, RAGGED>(in, a.src[0], a.idx[0], a.kw[0], mm, g, nkb);  // first phase: blocking
  int l = 0, p = 0, koff = 0;
  bool layer_open = full0;  // acc is collecting layer 0's phases through the pipelined path
  if (RAGGED && !full0) {  // ragged layer 0 (encoders: 11 / 3 input features): guarded generic path
    for (int pp = 0; pp < a.nphase; ++pp) {
      if (pp > 0) load_tl<HB, MT, RAGGED>(in, a.src[pp], a.idx[pp], a.kw[pp], mm, g, nkb);
      if (bf) round_tl<HB, MT>(in);
      gemm_tl<HB, MT>(acc, in, a.W[0] + koff, ktot, nib0, nkb, c, g);
      koff += 16 * nkb;
    }
    l = 1;
  }
  // One pipelined GEMM call site serves every phase of layer 0 and every later layer.
  while (l < a.NL) {
    const float* Wp;
    int ldw;
    const float* nx[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) nx[t] = dummy[t];
    if (layer_open) {  // layer 0, phase p; prefetch the gathered rows of phase p+1
      if (bf) round_tl<HB, MT>(in);   // (the rows of this phase: loaded above or by the previous call's prefetch)
      Wp = a.W[0] + koff;
      ldw = ktot;
      if (p + 1 < a.nphase) {
        const float* sp = a.src[p + 1];
        const int32_t* ip = a.idx[p + 1];
#pragma unroll
        for (int t = 0; t < MT; ++t) nx[t] = sp + (ip ? (long)ip[mm[t]] : mm[t]) * H + 4 * g;
      }
    } else {  // layer l >= 1: the accumulator IS the next B operand
      if (bf) round_tl<HB, MT>(acc);   // the previous Linear returns bf16
      if (act_smooth(a.act) && a.saveZ[l - 1] != nullptr) store_tl<HB, MT, RAGGED>(a.saveZ[l - 1], acc, H, mm, valid, g);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib)
#pragma unroll
          for (int r = 0; r < 4; ++r) in[t][ib][r] = act_f(acc[t][ib][r], a.act);
      if (bf && act_smooth(a.act)) round_tl<HB, MT>(in);   // a smooth activation of a bf16 tensor is bf16 again (ReLU: exact)
      if (a.saveH[l - 1] != nullptr) store_tl<HB, MT, RAGGED>(a.saveH[l - 1], in, H, mm, valid, g);
      const int nib = (l == a.NL - 1) ? nib_last : HB;
      init_bias<HB, MT>(acc, a.b[l], nib, g);
      if (RAGGED && nib != HB) {  // ragged last layer (decoder: 2 / 3 outputs)
        gemm_tl<HB, MT>(acc, in, a.W[l], H, nib, HB, c, g);
        ++l;
        continue;
      }
      Wp = a.W[l];
      ldw = H;
    }
    gemm_full<HB, MT, true>(acc, in, Wp, ldw, c, g, nx);
    if (layer_open && p + 1 < a.nphase) {
      ++p;
      koff += H;
    } else {
      layer_open = false;
      ++l;
    }
  }
  // ---- epilogue: RMSNorm (reference epsilon placement), residual, stores
  if (bf) round_tl<HB, MT>(acc);   // the last Linear's bf16 result (norm, residual in fp32)
  if (a.scale != nullptr) {
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
      if (a.saveR != nullptr && valid[t] && g == 0) st1(a.saveR + mm[t], rms);
#pragma unroll
      for (int ib = 0; ib < HB; ++ib) acc[t][ib] = ld4(a.scale + 16 * ib + 4 * g) * in[t][ib];
    }
    if (a.saveU != nullptr) store_tl<HB, MT, RAGGED>(a.saveU, in, H, mm, valid, g);
  }
  if (a.y_out != nullptr) store_tl<HB, MT, RAGGED>(a.y_out, acc, a.out_w, mm, valid, g);
  if (a.resid != nullptr) {
    int nkb;
    load_tl<HB, MT, RAGGED>(in, a.resid, nullptr, a.out_w, mm, g, nkb);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int ib = 0; ib < HB; ++ib) acc[t][ib] = in[t][ib] + acc[t][ib];
  }
  if (a.out_relu) {  // stand-alone first layer of an encoder: the activation feeding the packed kernel
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int ib = 0; ib < HB; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][ib][r] = act_f(acc[t][ib][r], a.act);
    if (a.saveM[0] != nullptr) {  // ReLU mask bits in the packed kernels' layout (saveM of mgn_hip.h)
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        uint32_t bits = 0;
#pragma unroll
        for (int ib = 0; ib < HB; ++ib)
#pragma unroll
          for (int r = 0; r < 4; ++r) bits |= (acc[t][ib][r] > 0.f) ? (1u << (4 * ib + r)) : 0u;
        if (valid[t]) a.saveM[0][mm[t] * 4 + g] = bits;
      }
    }
  }
  store_tl<HB, MT, RAGGED>(a.out, acc, a.out_w, mm, valid, g);
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

// same, accumulating over the tiles a persistent wave walks (same-wave LDS ops are ordered)
template <int HB, int MT>
__device__ __forceinline__ void colsum_to_lds_add(float* lds_w, const f32x4 (&v)[MT][HB], int c, int g) {
#pragma unroll
  for (int ib = 0; ib < HB; ++ib) {
    f32x4 s = v[0][ib];
#pragma unroll
    for (int t = 1; t < MT; ++t) s += v[t][ib];
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = colsum16(s[r]);
    if (c == 0) *(f32x4*)(lds_w + 16 * ib + 4 * g) += s;
  }
}

template <int HB, int MT, bool RAGGED>
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
    load_tl<HB, MT, RAGGED>(dz, a.dOut, nullptr, a.out_w, mm, g, nkb);
    if (a.dOut2 != nullptr) {
      load_tl<HB, MT, RAGGED>(acc, a.dOut2, a.idx2, H, mm, g, nkb);
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
      for (int ib = 0; ib < HB; ++ib) sc[ib] = ld4(a.scale + 16 * ib + 4 * g);
      load_tl<HB, MT, RAGGED>(acc, a.U, nullptr, H, mm, g, nkb);  // acc <- u
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
        const float rms = ld1(a.R + mm[t]);
        const float inv = 1.0f / (rms + a.eps);
        const float k2 = (rms > 0.f) ? dot / ((float)H * rms) : 0.f;
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) dz[t][ib] = dz[t][ib] * inv - acc[t][ib] * k2;
      }
      if (a.dscale != nullptr) colsum_to_lds<HB, MT>(lds_w + a.NL * H, du, c, g);
    }
    if (a.precision == 1) round_tl<HB, MT>(dz);   // the gradient of a bf16 Linear output is bf16 (see k_mlp_fwd)
    if (a.dZ[a.NL - 1] != nullptr) store_tl<HB, MT, RAGGED>(a.dZ[a.NL - 1], dz, 16 * nkb_last, mm, valid, g);
    if (a.db[a.NL - 1] != nullptr) colsum_to_lds<HB, MT>(lds_w + (a.NL - 1) * H, dz, c, g);
    // ---- dgrad chain with ReLU masks from the saved activations, then the requested
    //      first-layer input gradients: ONE pipelined GEMM call site for all of them.
    int l = a.NL - 1;
    if (RAGGED && l >= 1 && nkb_last != HB) {  // ragged last layer (decoder): guarded generic path
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) acc[t][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
      gemm_tl<HB, MT>(acc, dz, a.WT[l], 16 * nkb_last, HB, nkb_last, c, g);
      load_tl<HB, MT, RAGGED>(dz, act_smooth(a.act) ? a.Zs[l - 1] : a.Hs[l - 1], nullptr, H, mm, g, nkb);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dz[t][ib][r] = !valid[t] ? 0.f : act_smooth(a.act) ? acc[t][ib][r] * dact_f(dz[t][ib][r], a.act)
                                                                       : (dz[t][ib][r] > 0.f ? acc[t][ib][r] : 0.f);
      if (a.precision == 1) round_tl<HB, MT>(dz);
      if (a.dZ[l - 1] != nullptr) store_tl<HB, MT, RAGGED>(a.dZ[l - 1], dz, H, mm, valid, g);
      if (a.db[l - 1] != nullptr) colsum_to_lds<HB, MT>(lds_w + (l - 1) * H, dz, c, g);
      --l;
    }
    const bool din_full = (a.NL > 1) || (nkb_last == HB);
    int q = 0;
    for (;;) {
      const bool chain = (l >= 1);
      if (!chain && q >= a.n_din) break;
      const float* Wp;
      const float* nx[MT];
      if (chain) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int ib = 0; ib < HB; ++ib) acc[t][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
        Wp = a.WT[l];
        const float* hz = act_smooth(a.act) ? a.Zs[l - 1] : a.Hs[l - 1];
#pragma unroll
        for (int t = 0; t < MT; ++t) nx[t] = hz + mm[t] * H + 4 * g;  // dz <- h_l (ReLU) / z_l (SiLU) on the way out
      } else {
        if (a.din_resid[q] != nullptr) {
          load_tl<HB, MT, RAGGED>(acc, a.din_resid[q], nullptr, H, mm, g, nkb);
        } else {
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int ib = 0; ib < HB; ++ib) acc[t][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (RAGGED && !din_full) {  // NL == 1 with a ragged output: generic
          gemm_tl<HB, MT>(acc, dz, a.WT0[q], 16 * nkb_last, HB, nkb_last, c, g);
          store_tl<HB, MT, RAGGED>(a.dIn[q], acc, H, mm, valid, g);
          ++q;
          continue;
        }
        Wp = a.WT0[q];
        // the "next" loads re-read dZ[0] (just stored by this wave): dz keeps its value
        const float* base = (a.dZ[0] != nullptr) ? a.dZ[0] : a.WT0[q];
#pragma unroll
        for (int t = 0; t < MT; ++t) nx[t] = base + ((a.dZ[0] != nullptr) ? mm[t] * H : 0) + 4 * g;
      }
      gemm_full<HB, MT, true>(acc, dz, Wp, H, c, g, nx);
      if (chain) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int ib = 0; ib < HB; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              dz[t][ib][r] = !valid[t] ? 0.f : act_smooth(a.act) ? acc[t][ib][r] * dact_f(dz[t][ib][r], a.act)
                                                                         : (dz[t][ib][r] > 0.f ? acc[t][ib][r] : 0.f);
        if (a.precision == 1) round_tl<HB, MT>(dz);
        if (a.dZ[l - 1] != nullptr) store_tl<HB, MT, RAGGED>(a.dZ[l - 1], dz, H, mm, valid, g);
        if (a.db[l - 1] != nullptr) colsum_to_lds<HB, MT>(lds_w + (l - 1) * H, dz, c, g);
        --l;
      } else {
        store_tl<HB, MT, RAGGED>(a.dIn[q], acc, H, mm, valid, g);
        ++q;
      }
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
__global__ void __launch_bounds__(1024) k_colred(const float* __restrict__ ws, int nblocks, int nslot, int H, const ColredOuts outs) {
  __shared__ float red[32][33];
  const int col = blockIdx.x * 32 + (threadIdx.x & 31);
  const int rl = threadIdx.x >> 5;  // 0..31
  const size_t st = (size_t)nslot * H;
  float s = 0.f;
  if (col < nslot * H)
    for (int b = rl; b < nblocks; b += 32) s += ws[(size_t)b * st + col];
  red[rl][threadIdx.x & 31] = s;
  __syncthreads();
  if (rl == 0 && col < nslot * H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
    float* o = outs.o[col / H];
    if (o != nullptr) o[col % H] = t;
  }
}

// The same for up to CR_MAX deferred launches at once (grid.y = job): the per-launch reduce is a
// 7 us latency-bound launch, 30 of them per training step.
#define CR_MAX 32
struct ColredJob {
  const float* ws;
  int nblocks, nslot, H;
  ColredOuts outs;
};
struct ColredBatch {
  ColredJob j[CR_MAX];
};
__global__ void __launch_bounds__(1024) k_colred_batch(const ColredBatch B) {
  __shared__ float red[32][33];
  const ColredJob J = B.j[blockIdx.y];
  const int ncol = J.nslot * J.H;
  const int col = blockIdx.x * 32 + (threadIdx.x & 31);
  const int rl = threadIdx.x >> 5;
  float s = 0.f;
  if (col < ncol && J.outs.o[col / J.H] != nullptr)
    for (int b = rl; b < J.nblocks; b += 32) s += J.ws[(size_t)b * ncol + col];
  red[rl][threadIdx.x & 31] = s;
  __syncthreads();
  if (rl == 0 && col < ncol) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
    float* o = J.outs.o[col / J.H];
    if (o != nullptr) o[col % J.H] = t;
  }
}

// ======================================================== LDS-weight kernels (H = 128)
// Probe result (tools/gemm_probe.hip): with every wave streaming its own weight
// fragments from L2 the per-CU vector-memory path saturates and the fp32 MFMA pipe
// idles half of the time.  Here the 4 waves of a workgroup share ONE copy of the weights
// in LDS.  A 128x128 block is streamed as two K-halves of 32 KB (columns 0..63 / 64..127),
// double buffered (2 x 32 KB), filled by LDS-DMA (global_load_lds_dwordx4, no VGPRs) one
// half-GEMM ahead and read with ds_read_b128; 64 KB per workgroup lets TWO independent
// workgroups share a CU, so one's barriers / stores / prologue hide under the other's MFMAs
// (8 lock-stepped waves of a single workgroup measured 54 % MFMA-busy).
// Image of a half: row j at byte 256*j; 16-byte chunk ch (0..15) of the row is stored at
// chunk position ch ^ (j & 15) -- the swizzle goes on the DMA's per-lane SOURCE address (the
// LDS destination of a DMA is lane-linear), and makes the rows a b128 lane group reads hit
// different 4-bank groups (SQ_LDS_BANK_CONFLICT = 0).  One barrier per half-GEMM.
#ifdef MGN_TIMELINE
// debug build only (tools/): s_memtime stamps of wave 0 of the first workgroups, buffered in LDS
// (no VMEM traffic, position counter in a register) and dumped when the kernel ends
__device__ unsigned long long g_timeline[8][512];
__device__ int g_tlpos[8];
// placement census: per workgroup {HW_ID, XCC_ID, first stamp, last stamp} (which workgroups share a CU / SIMD)
__device__ unsigned long long g_census[1024][4];
#define TL_DECL()                                                                            \
  __shared__ unsigned long long tl_buf_[512];                                                \
  int tl_n_ = 0;                                                                             \
  if (threadIdx.x == 0 && blockIdx.x < 1024) {                                               \
    g_census[blockIdx.x][0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                     \
    g_census[blockIdx.x][1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                    \
    g_census[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();                              \
  }                                                                                          \
  const int tl_slot_ = (blockIdx.x < 4) ? (int)blockIdx.x                                    \
                       : ((blockIdx.x + 4 >= gridDim.x) ? (int)(blockIdx.x + 8 - gridDim.x) : -1)
#define TL_STAMP(tag)                                                                        \
  do {                                                                                       \
    if (threadIdx.x == 0 && tl_slot_ >= 0 && tl_n_ < 512)                                    \
      tl_buf_[tl_n_++] = (__builtin_amdgcn_s_memrealtime() << 8) | (unsigned long long)(tag);\
  } while (0)
#define TL_DUMP()                                                                            \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_census[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime(); \
    if (threadIdx.x == 0 && tl_slot_ >= 0) {                                                 \
      for (int i_ = 0; i_ < tl_n_; ++i_) g_timeline[tl_slot_][i_] = tl_buf_[i_];            \
      g_tlpos[tl_slot_] = tl_n_;                                                             \
    }                                                                                        \
  } while (0)
extern "C" int mgn_debug_census(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_census), sizeof(unsigned long long) * 1024 * 4) != hipSuccess;
}
extern "C" int mgn_debug_timeline(unsigned long long* out, int* pos) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 8 * 512) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(pos, HIP_SYMBOL(g_tlpos), sizeof(int) * 8) != hipSuccess) return 1;
  return 0;
}
#else
#define TL_DECL() ((void)0)
#define TL_STAMP(tag) ((void)0)
#define TL_DUMP() ((void)0)
#endif
// Stagger the second dispatch round (blocks >= 256 share CUs with blocks < 256) so that the two
// co-resident workgroups are half a GEMM period out of phase instead of in lock-step.
__device__ __forceinline__ void stagger_start(int cycles) {
#ifdef MGN_EXP_STAGGER
  if (blockIdx.x >= 256) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)cycles) __builtin_amdgcn_s_sleep(8);
  }
#else
  (void)cycles;
#endif
}
#ifdef MGN_EXP_NOSYNC
#define MGN_SYNC() ((void)0)
#else
#define MGN_SYNC() __syncthreads()
#endif
// arr[i] for a small kernel-argument pointer array WITHOUT a dynamic (SMEM) load: every
// element is read with a constant index (hoisted to SGPRs once) and picked by a select
// chain.  A scalar load inside the GEMM loop is poison: SMEM returns out of order, so while
// one is pending hipcc turns every LDS wait into s_waitcnt lgkmcnt(0) and the LDS operand
// prefetch ring stops overlapping (measured 63 % vs 91 % MFMA-busy).
template <typename T>
__device__ __forceinline__ T pick3(T a0, T a1, T a2, int i) {
  T r = a0;
  r = (i == 1) ? a1 : r;
  r = (i == 2) ? a2 : r;
  return r;
}
template <typename T>
__device__ __forceinline__ T pick4(T a0, T a1, T a2, T a3, int i) {
  T r = pick3(a0, a1, a2, i);
  r = (i == 3) ? a3 : r;
  return r;
}
#define LDS_MAX_NL 4

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
#define WBUF_BYTES 32768

// One LDS-DMA of 1 KB: lane L fetches 16 bytes at sbase + voff(L) into LDS byte
// lds_addr + 16*L.  Written in inline asm on purpose: hipcc models a pending
// __builtin_amdgcn_global_load_lds as a FLAT access that also occupies the LGKM counter, and
// with mixed event types pending it degrades EVERY LDS wait to s_waitcnt lgkmcnt(0) -- the
// ds_read operand ring of the GEMM then stops overlapping (63 % vs 91 % MFMA-busy).  Hidden
// in asm the DMA is invisible to that bookkeeping; the price is that WE must drain it
// (dma_drain) before the barrier that publishes the buffer.  M0 (the DMA's LDS base) is
// compiler-reserved: saved, set and restored inside the one statement.
__device__ __forceinline__ void glds16(const float* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}
// make a (really) wave-uniform pointer PROVABLY uniform, so that it may bind to an "s" operand
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Counted drain: called right after a half-GEMM, whose VMEM order is DMA pieces 0..7 (steps
// 0..7) interleaved with the 4*MT "next tensor" loads, of which the last 2*MT (blocks 2 and 3,
// steps 11 and 15) are YOUNGER than every DMA piece.  vmcnt counts in issue order, so
// allowing 2*MT outstanding operations guarantees all DMA pieces have landed while the
// youngest prefetch loads stay in flight across the barrier.  Must be called before any
// further VMEM operation is issued (more young operations would only over-wait, fewer would
// under-wait -- there are never fewer: the loads are issued unconditionally).
template <int N>
__device__ __forceinline__ void dma_drain_counted() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ unsigned lds_addr_of(lds_char* p) { return (unsigned)(size_t)p; }

// The next half's weight DMA, issued one 1 KB instruction every other GEMM step instead of
// eight back to back: a burst stalls the issuing wave for 1.2-4.2k cycles on VMEM
// back-pressure (s_memtime timeline), spread out it rides under the MFMAs.
struct DmaJob {
  const float* su0;   // wave-uniform: first row of this wave's share, column offset included
  int stride;         // floats between consecutive instructions' rows (4 * ldw)
  unsigned lds0;      // LDS byte address of this wave's first instruction
  unsigned vo[4];     // per-lane byte offsets (see dma_prepare)
  bool on;
};
__device__ __forceinline__ DmaJob dma_prepare(const float* __restrict__ Wk, int ldw, int hk, lds_char* buf, int wv, int lane, bool on) {
  // Instruction i of wave wv fills rows 4q..4q+3, q = 8*wv+i (1 KB, lane-linear): lane -> row
  // 4q + lane/16, chunk position lane%16 holding global chunk (lane%16) ^ (row%16).  With
  // row%16 = 4*(i%4) + lane/16 the per-lane part of the address takes only 4 values.
  DmaJob j;
  const int lg = lane >> 4, bg = (lane & 15) ^ lg;
#pragma unroll
  for (int k = 0; k < 4; ++k) j.vo[k] = 4u * (unsigned)(lg * ldw + 4 * (bg ^ (4 * k)));
  j.su0 = uniform_ptr(Wk + (size_t)(32 * wv) * ldw + 64 * hk);
  j.stride = 4 * ldw;
  j.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(buf)) + 8 * wv * 1024;
  j.on = on;
  return j;
}
__device__ __forceinline__ void dma_issue(const DmaJob& j, int i) {
  if (j.on) glds16(j.su0 + (size_t)i * j.stride, j.vo[i & 3], j.lds0 + i * 1024);
}
__device__ __forceinline__ void dma_weights(const float* __restrict__ Wk, int ldw, int hk, lds_char* buf, int wv, int lane) {
  const DmaJob j = dma_prepare(Wk, ldw, hk, buf, wv, lane, true);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_issue(j, i);
}

// acc += W[:, 64*HK .. 64*HK+63](LDS) * in[4*HK .. 4*HK+3]; the consumed input blocks are
// refilled from nxt (prefetch of the next tensor, see gemm_full); `job` = the DMA of the half
// after this one, interleaved.
template <int MT, int HK>
__device__ __forceinline__ void gemm_lds_half(f32x4 (&acc)[MT][8], f32x4 (&in)[MT][8], lds_char* wbuf,
                                              const int (&off)[4], const float* const (&nxt)[MT], const DmaJob& job) {
  constexpr int NI = 4, NS = 16;
  f32x4 w[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) w[0][q] = *(lds_cf32x4*)(wbuf + off[0] + q * 4096);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int kbh = s / NI, ibp = s % NI;
    if (s + 1 < NS) {
      const int s2 = s + 1, kb2 = s2 / NI, ib2 = (s2 % NI) * 2;
#pragma unroll
      for (int q = 0; q < 2; ++q) w[s2 & 1][q] = *(lds_cf32x4*)(wbuf + off[kb2] + (ib2 + q) * 4096);
    }
    if (s < 8) dma_issue(job, s);  // early in the half: the last piece gets >= 8 steps to land
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[t][ibp * 2 + q] = MFMA16(w[s & 1][q][r], in[t][4 * HK + kbh][r], acc[t][ibp * 2 + q]);
    if (ibp == NI - 1) {
#pragma unroll
      for (int t = 0; t < MT; ++t) in[t][4 * HK + kbh] = ld4(nxt[t] + 16 * (4 * HK + kbh));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Persistent: a workgroup walks tiles blockIdx.x, +gridDim.x, ... as ONE continuous stream of
// half-GEMMs.  The weight DMA ring keeps rolling across tile boundaries, the last GEMM of a
// tile prefetches the next tile's phase-0 rows, gather indices are fetched a tile ahead, and
// biases / norm scale live in LDS -- so between two GEMMs a wave only does register work.
// Per tile: GEMMs of layer 0 (one per input phase; optional gathered adds = split first
// layer), layers 1..NL-1, epilogue (RMSNorm, residual, stores), optional post-products of the
// freshly computed output rows.
// LDS: [2 x 32 KB weight halves][b0..b3, scale: 5 x 512 B].
#define FWD_LDS_BYTES (2 * WBUF_BYTES + 5 * 512)
template <int MT>
__global__ void __launch_bounds__(256, 2) k_mlp_fwd_lds(const mgn_mlp_fwd_args a) {
  constexpr int HB = 8, H = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TL_DECL();
  lds_char* wl = (lds_char*)smem;
  lds_char* cst = wl + 2 * WBUF_BYTES;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const long ntiles = (a.M + 64 * MT - 1) / (64 * MT);
  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  stagger_start(6000);
  const long my_tiles = (ntiles - tile + gridDim.x - 1) / gridDim.x;
  int off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) off[j] = c * 256 + (((4 * j + g) ^ c) & 15) * 16;

  const int GL = a.nphase + a.NL - 1;  // GEMMs of the MLP proper: phases of layer 0, layers 1..NL-1
  const int G = GL + a.n_post;         // + post-products
  const int ldw0 = (a.ldw0 > 0) ? a.ldw0 : H * a.nphase;
  const long total_halves = my_tiles * 2 * G;
  // kernel-argument pointers in SGPRs, picked by select chains (no SMEM in the loop)
  const float *W0 = a.W[0], *W1 = a.W[1], *W2 = a.W[2], *W3 = a.W[3];
  const float *PW0 = a.post_W[0], *PW1 = a.post_W[1];
  float *sH0 = a.saveH[0], *sH1 = a.saveH[1], *sH2 = a.saveH[2];
  const float *src0 = a.src[0], *src1 = a.src[1], *src2 = a.src[2];
  const int32_t *idx0 = a.idx[0], *idx1 = a.idx[1], *idx2 = a.idx[2];
  const float *as0 = a.add_src[0], *as1 = a.add_src[1];
  const int32_t *ai0 = a.add_idx[0], *ai1 = a.add_idx[1];
  auto job_for = [&](long j) -> DmaJob {  // DMA of half j of this workgroup's stream -> buffer j&1
    const int jj = (int)(j % (2 * G)), k = jj >> 1;
    const float* Wk;
    int ldw;
    if (k < a.nphase) {
      Wk = W0 + H * k;
      ldw = ldw0;
    } else if (k < GL) {
      Wk = pick4(W0, W1, W2, W3, k - a.nphase + 1);
      ldw = H;
    } else {
      Wk = (k == GL) ? PW0 : PW1;
      ldw = a.post_ldw;
    }
    return dma_prepare(Wk, ldw, jj & 1, wl + (j & 1) * WBUF_BYTES, wv, lane, j < total_halves);
  };
  {
    const DmaJob j0 = job_for(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_issue(j0, i);
  }
  // biases and scale -> LDS (zeros where absent)
  for (int i = threadIdx.x; i < 5 * H; i += 256) {
    const int l = i >> 7, jx = i & 127;
    const float* bp = (l == 4) ? a.scale : pick4(a.b[0], a.b[1], a.b[2], a.b[3], l);
    ((__attribute__((address_space(3))) float*)cst)[i] = (bp != nullptr && (l == 4 || l < a.NL)) ? bp[jx] : 0.f;
  }
  auto lds_bias = [&](f32x4 (&acc)[MT][HB], int l) {
#pragma unroll
    for (int ib = 0; ib < HB; ++ib) {
      const f32x4 bv = *(lds_cf32x4*)(cst + l * 512 + 64 * ib + 16 * g);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][ib] = bv;
    }
  };

  // rows of a tile and their (gathered) source rows: rid[0..2] = input phases, rid[3..4] = adds
  auto rows_of = [&](long tl, long (&mm)[MT], bool (&valid)[MT], int (&rid)[5][MT]) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const long m = (tl * 4 + wv) * (16 * MT) + 16 * t + c;
      valid[t] = m < a.M;
      mm[t] = valid[t] ? m : a.M - 1;  // waves past M still take part in the DMA / barriers
      rid[0][t] = idx0 ? idx0[mm[t]] : (int)mm[t];
      rid[1][t] = (a.nphase > 1) ? (idx1 ? idx1[mm[t]] : (int)mm[t]) : 0;
      rid[2][t] = (a.nphase > 2) ? (idx2 ? idx2[mm[t]] : (int)mm[t]) : 0;
      rid[3][t] = (a.n_add > 0) ? (ai0 ? ai0[mm[t]] : (int)mm[t]) : 0;
      rid[4][t] = (a.n_add > 1) ? (ai1 ? ai1[mm[t]] : (int)mm[t]) : 0;
    }
  };
  long mm[MT];
  bool valid[MT];
  int rid[5][MT];
  rows_of(tile, mm, valid, rid);
  f32x4 in[MT][HB], acc[MT][HB];
  const float* dummy[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) dummy[t] = W0 + 4 * g;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int kb = 0; kb < HB; ++kb) in[t][kb] = ld4(src0 + (long)rid[0][t] * H + 4 * g + 16 * kb);
  dma_drain();      // prologue burst of half 0 (and the first input rows)
  __syncthreads();  // constants visible

  long j = 0;
  f32x4 pa[MT][HB];  // split first layer: one gathered projection at a time rides under a half-GEMM
  bool add_mid = false;
  auto gemm_pair = [&](const float* const (&nx)[MT]) {  // one full GEMM = two half-GEMMs of the stream
    TL_STAMP(3);
    __syncthreads();  // this half's weights landed everywhere; the other buffer is free again
    TL_STAMP(4);
    {
      const DmaJob job = job_for(j + 1);
      gemm_lds_half<MT, 0>(acc, in, wl + (j & 1) * WBUF_BYTES, off, nx, job);
    }
    TL_STAMP(5);
    dma_drain_counted<2 * MT>();
    TL_STAMP(6);
    ++j;
    if (add_mid) {  // first projection landed under half 0: add it, send for the second one
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) {
          acc[t][ib] += pa[t][ib];
          if (a.n_add > 1) pa[t][ib] = ld4(as1 + (long)rid[4][t] * H + 4 * g + 16 * ib);
        }
    }
    __syncthreads();
    {
      const DmaJob job = job_for(j + 1);
      gemm_lds_half<MT, 1>(acc, in, wl + (j & 1) * WBUF_BYTES, off, nx, job);
    }
    dma_drain_counted<2 * MT>();
    ++j;
  };

  for (; tile < ntiles; tile += gridDim.x) {
    const long ntile = tile + gridDim.x;
    const bool has_next = ntile < ntiles;
    long mmn[MT];
    bool validn[MT];
    int ridn[5][MT];
    rows_of(has_next ? ntile : tile, mmn, validn, ridn);  // index loads land a tile ahead of use
    TL_STAMP(1);
    lds_bias(acc, 0);
    // split first layer: the gathered projections travel while the phase GEMM runs
    if (a.n_add > 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) pa[t][kb] = ld4(as0 + (long)rid[3][t] * H + 4 * g + 16 * kb);
    }
    f32x4 rs[MT][HB];  // residual rows, fetched under the last layer's GEMM
    for (int k = 0; k < GL; ++k) {
      const float* nx[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) nx[t] = dummy[t];
      if (k < a.nphase) {
        if (k + 1 < a.nphase) {  // prefetch the gathered rows of the next phase
          const float* sp = pick3(src0, src1, src2, k + 1);
#pragma unroll
          for (int t = 0; t < MT; ++t) nx[t] = sp + (long)((k == 0) ? rid[1][t] : rid[2][t]) * H + 4 * g;
        }
      } else {  // layer l >= 1: operand = ReLU(previous accumulator)
        const int l = k - a.nphase + 1;
        if (l == 1 && a.n_add > 1) {  // second projection (landed under half 1 of the phase GEMM)
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int ib = 0; ib < HB; ++ib) acc[t][ib] += pa[t][ib];
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int ib = 0; ib < HB; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) in[t][ib][r] = fmaxf(acc[t][ib][r], 0.f);
        float* sh = pick3(sH0, sH1, sH2, l - 1);
        if (sh != nullptr) store_tl_stream<HB, MT>(sh, in, mm, valid, g);
        lds_bias(acc, l);
      }
      if (k == GL - 1 && a.n_post == 0 && has_next) {  // last GEMM of the tile: next tile's phase-0 rows
#pragma unroll
        for (int t = 0; t < MT; ++t) nx[t] = src0 + (long)ridn[0][t] * H + 4 * g;
      }
      TL_STAMP(2);
      if (k == GL - 1 && a.resid != nullptr) {
        int nk_;
        load_tl<HB, MT, false>(rs, a.resid, nullptr, H, mm, g, nk_);
      }
      add_mid = (k == a.nphase - 1) && (a.n_add > 0);  // under the last phase GEMM of layer 0
      gemm_pair(nx);
      add_mid = false;
      TL_STAMP(7);
    }
    if (a.NL == 1 && a.n_add > 1) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) acc[t][ib] += pa[t][ib];
    }
    // ---- epilogue: RMSNorm (reference epsilon placement), residual, stores.  Without
    //      post-products `in` already holds the next tile's rows and is not touched; with them
    //      the output rows go to `in` (they are the operand of the post GEMMs).
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      float inv = 1.f;
      if (a.scale != nullptr) {
        float ss = 0.f;
#pragma unroll
        for (int ib = 0; ib < HB; ++ib)
#pragma unroll
          for (int r = 0; r < 4; ++r) ss = fmaf(acc[t][ib][r], acc[t][ib][r], ss);
        ss = rowsum4(ss);
        const float rms = sqrtf(ss) / sqrtf((float)H);
        inv = 1.0f / (rms + a.eps);  // one reciprocal per row; u = z * inv (<= 1 ulp from z / den)
        if (a.saveR != nullptr && valid[t] && g == 0) st1(a.saveR + mm[t], rms);
      }
      const long ro = mm[t] * H + 4 * g;
#pragma unroll
      for (int ib = 0; ib < HB; ++ib) {
        f32x4 y = acc[t][ib];
        if (a.scale != nullptr) {
          const f32x4 u = y * inv;
          if (a.saveU != nullptr && valid[t]) st4_stream(a.saveU + ro + 16 * ib, u);
          y = *(lds_cf32x4*)(cst + 4 * 512 + 64 * ib + 16 * g) * u;
        }
        if (a.y_out != nullptr && valid[t]) st4(a.y_out + ro + 16 * ib, y);
        if (a.resid != nullptr) y = rs[t][ib] + y;
        if (valid[t]) st4(a.out + ro + 16 * ib, y);
        if (a.n_post > 0) in[t][ib] = y;
      }
    }
    TL_STAMP(8);
    // ---- post-products of the fresh output rows (next round's node projections)
    for (int q = 0; q < a.n_post; ++q) {
      const float* nx[MT];
      const bool last = (q == a.n_post - 1);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        // not last: re-read the rows just stored, `in` keeps its value; last: next tile's phase 0
        nx[t] = (last && has_next) ? src0 + (long)ridn[0][t] * H + 4 * g : a.out + mm[t] * H + 4 * g;
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) acc[t][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      gemm_pair(nx);
      store_tl<HB, MT, false>((q == 0) ? a.post_out[0] : a.post_out[1], acc, H, mm, valid, g);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      mm[t] = mmn[t];
      valid[t] = validn[t];
#pragma unroll
      for (int p = 0; p < 5; ++p) rid[p][t] = ridn[p][t];
    }
  }
  TL_DUMP();
}

// Backward chain, persistent like the forward: a workgroup walks its tiles as one stream of
// half-GEMMs.  While a tile computes, the NEXT tile's operands are already on their way:
// its dOut rows refill dz under the last GEMM, its U rows / gathered dOut2 rows / rms are
// loaded into spare registers right after the current ones were consumed.  dscale is summed
// per lane across all tiles of the workgroup and reduced across lanes once at the end; bias
// gradients (column sums of dZ) are left to the weight-gradient kernel, where dZ is read in
// N-layout anyway (2 adds per 16 MFMAs there versus 128 shuffles per tile here) -- db[] is
// still honoured (slow path) when a caller asks for it.
// LDS: [2 x 32 KB weight halves][scale 512 B][4 waves x nslot x 512 B column-sum partials].
template <int MT>
__global__ void __launch_bounds__(256, 2) k_mlp_bwd_lds(const mgn_mlp_bwd_args a) {
  constexpr int HB = 8, H = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* wl = (lds_char*)smem;
  lds_char* cst = wl + 2 * WBUF_BYTES;                          // scale
  float* lds = (float*)(smem + 2 * WBUF_BYTES + 512);           // [4 waves][nslot][H]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int nslot = a.NL + 1;
  float* lds_w = lds + wv * nslot * H;
  for (int i = lane; i < nslot * H; i += 64) lds_w[i] = 0.f;
  for (int i = threadIdx.x; i < H; i += 256)
    ((__attribute__((address_space(3))) float*)cst)[i] = (a.scale != nullptr) ? a.scale[i] : 0.f;
  const long ntiles = (a.M + 64 * MT - 1) / (64 * MT);
  long tile = blockIdx.x;
  const long my_tiles = (tile < ntiles) ? (ntiles - tile + gridDim.x - 1) / gridDim.x : 0;
  int off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) off[j] = c * 256 + (((4 * j + g) ^ c) & 15) * 16;
  const int G = (a.NL - 1) + a.n_din;  // per tile: chain GEMMs (WT[NL-1]..WT[1]) then the input-grad GEMM
  const long total_halves = my_tiles * 2 * G;
  // kernel-argument pointers in SGPRs, picked by select chains (no SMEM in the loop)
  const float *WT1 = a.WT[1], *WT2 = a.WT[2], *WT3 = a.WT[3];
  const float* WT00 = a.WT0[0];
  const float *Hs0 = a.Hs[0], *Hs1 = a.Hs[1], *Hs2 = a.Hs[2];
  float *dZ0 = a.dZ[0], *dZ1 = a.dZ[1], *dZ2 = a.dZ[2], *dZ3 = a.dZ[3];
  float *db0 = a.db[0], *db1 = a.db[1], *db2 = a.db[2], *db3 = a.db[3];
  const float* dr0 = a.din_resid[0];
  float* dI0 = a.dIn[0];
  auto job_for = [&](long j) -> DmaJob {
    const int jj = (int)(j % (2 * G)), k = jj >> 1;
    const float* Wk = (k < a.NL - 1) ? pick4(WT1, WT1, WT2, WT3, a.NL - 1 - k) : WT00;
    return dma_prepare(Wk, H, jj & 1, wl + (j & 1) * WBUF_BYTES, wv, lane, j < total_halves);
  };
  auto rows_of = [&](long tl, long (&mm)[MT], bool (&valid)[MT]) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const long m = (tl * 4 + wv) * (16 * MT) + 16 * t + c;
      valid[t] = m < a.M;
      mm[t] = valid[t] ? m : a.M - 1;
    }
  };
  f32x4 dsum[HB];  // per-lane partial of dscale = sum_rows dY * U
#pragma unroll
  for (int ib = 0; ib < HB; ++ib) dsum[ib] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (my_tiles > 0) {
    {
      const DmaJob j0 = job_for(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) dma_issue(j0, i);
    }
    long mm[MT];
    bool valid[MT];
    rows_of(tile, mm, valid);
    f32x4 dz[MT][HB], acc[MT][HB], pu[MT][HB], pd2[MT][HB];
    float prms[MT];
    // first tile: operands loaded directly
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const long r2 = (a.dOut2 != nullptr) ? (a.idx2 ? (long)a.idx2[mm[t]] : mm[t]) : 0;
      prms[t] = (a.scale != nullptr) ? ld1(a.R + mm[t]) : 0.f;
#pragma unroll
      for (int kb = 0; kb < HB; ++kb) {
        dz[t][kb] = ld4(a.dOut + mm[t] * H + 4 * g + 16 * kb);
        pd2[t][kb] = (a.dOut2 != nullptr) ? ld4(a.dOut2 + r2 * H + 4 * g + 16 * kb) : f32x4{0.f, 0.f, 0.f, 0.f};
        pu[t][kb] = (a.scale != nullptr) ? ld4(a.U + mm[t] * H + 4 * g + 16 * kb) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    dma_drain();
    __syncthreads();  // scale visible

    long j = 0;
    for (; tile < ntiles; tile += gridDim.x) {
      const long ntile = tile + gridDim.x;
      const bool has_next = ntile < ntiles;
      long mmn[MT];
      bool validn[MT];
      rows_of(has_next ? ntile : tile, mmn, validn);
      // ---- dY, RMSNorm backward:  dz = g/(rms+eps) - u * <g,u> / (H*rms),  g = scale*dY
#pragma unroll
      for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) dz[t][ib] = valid[t] ? dz[t][ib] + pd2[t][ib] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.scale != nullptr) {
          float dot = 0.f;
#pragma unroll
          for (int ib = 0; ib < HB; ++ib) {
            dsum[ib] += dz[t][ib] * pu[t][ib];
            const f32x4 gg = *(lds_cf32x4*)(cst + 64 * ib + 16 * g) * dz[t][ib];
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(gg[r], pu[t][ib][r], dot);
            dz[t][ib] = gg;
          }
          dot = rowsum4(dot);
          const float rms = prms[t];
          const float inv = 1.0f / (rms + a.eps);
          const float k2 = (rms > 0.f) ? dot / ((float)H * rms) : 0.f;
#pragma unroll
          for (int ib = 0; ib < HB; ++ib) dz[t][ib] = dz[t][ib] * inv - pu[t][ib] * k2;
        }
      }
      // the next tile's U / dOut2 / rms start travelling now (consumed one tile later)
      if (has_next) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const long r2 = (a.dOut2 != nullptr) ? (a.idx2 ? (long)a.idx2[mmn[t]] : mmn[t]) : 0;
          if (a.scale != nullptr) prms[t] = ld1(a.R + mmn[t]);
#pragma unroll
          for (int kb = 0; kb < HB; ++kb) {
            if (a.dOut2 != nullptr) pd2[t][kb] = ld4(a.dOut2 + r2 * H + 4 * g + 16 * kb);
            if (a.scale != nullptr) pu[t][kb] = ld4(a.U + mmn[t] * H + 4 * g + 16 * kb);
          }
        }
      }
      {
        float* dzl = pick4(dZ0, dZ1, dZ2, dZ3, a.NL - 1);
        if (dzl != nullptr) store_tl<HB, MT, false>(dzl, dz, H, mm, valid, g);
        if (pick4(db0, db1, db2, db3, a.NL - 1) != nullptr) colsum_to_lds_add<HB, MT>(lds_w + (a.NL - 1) * H, dz, c, g);
      }
      for (int k = 0; k < G; ++k) {
        const bool chain = k < a.NL - 1;
        const int l = a.NL - 1 - k;  // chain: layer whose W^T is applied
        const float* nx[MT];
        if (chain) {
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int ib = 0; ib < HB; ++ib) acc[t][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
          const float* hs = pick3(Hs0, Hs1, Hs2, l - 1);
#pragma unroll
          for (int t = 0; t < MT; ++t) nx[t] = hs + mm[t] * H + 4 * g;  // dz <- h_l on the way out
        } else {  // input-grad GEMM (the last of the tile): dz is dead afterwards -> next tile's dOut
          if (dr0 != nullptr) {
            int nk_;
            load_tl<HB, MT, false>(acc, dr0, nullptr, H, mm, g, nk_);
          } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
              for (int ib = 0; ib < HB; ++ib) acc[t][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int t = 0; t < MT; ++t) nx[t] = a.dOut + mmn[t] * H + 4 * g;
        }
        __syncthreads();
        {
          const DmaJob job = job_for(j + 1);
          gemm_lds_half<MT, 0>(acc, dz, wl + (j & 1) * WBUF_BYTES, off, nx, job);
        }
        dma_drain_counted<2 * MT>();
        ++j;
        __syncthreads();
        {
          const DmaJob job = job_for(j + 1);
          gemm_lds_half<MT, 1>(acc, dz, wl + (j & 1) * WBUF_BYTES, off, nx, job);
        }
        dma_drain_counted<2 * MT>();
        ++j;
        if (chain) {
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int ib = 0; ib < HB; ++ib)
#pragma unroll
              for (int r = 0; r < 4; ++r) dz[t][ib][r] = (valid[t] && dz[t][ib][r] > 0.f) ? acc[t][ib][r] : 0.f;
          float* dzp = pick3(dZ0, dZ1, dZ2, l - 1);
          if (dzp != nullptr) store_tl<HB, MT, false>(dzp, dz, H, mm, valid, g);
          if (pick3(db0, db1, db2, l - 1) != nullptr) colsum_to_lds_add<HB, MT>(lds_w + (l - 1) * H, dz, c, g);
        } else {
          store_tl<HB, MT, false>(dI0, acc, H, mm, valid, g);
        }
      }
      if (a.n_din == 0 && has_next) {  // no input-grad GEMM to hide it under: load the next dOut rows directly
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int kb = 0; kb < HB; ++kb) dz[t][kb] = ld4(a.dOut + mmn[t] * H + 4 * g + 16 * kb);
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        mm[t] = mmn[t];
        valid[t] = validn[t];
      }
    }
  }
  // dscale: one cross-lane reduction per workgroup
  if (a.scale != nullptr && a.dscale != nullptr) {
#pragma unroll
    for (int ib = 0; ib < HB; ++ib) {
      f32x4 v = dsum[ib];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = colsum16(v[r]);
      if (c == 0) *(f32x4*)(lds_w + a.NL * H + 16 * ib + 4 * g) = v;
    }
  }
  __syncthreads();
  float* ws = (float*)a.red_ws + (size_t)blockIdx.x * nslot * H;
  for (int i = threadIdx.x; i < nslot * H; i += 256)
    ws[i] = ((lds[i] + lds[nslot * H + i]) + lds[2 * nslot * H + i]) + lds[3 * nslot * H + i];
}

#define TB_MAX 128
// ===================================================================== weight grads
struct WgradLaunch {
  int njobs;
  mgn_wgrad_job job[MGN_MAX_WGRAD_JOBS];
  int wg0[MGN_MAX_WGRAD_JOBS + 1];  // first workgroup of each job
  float* partial;                   // [total_wg][H*H + H]: dW partial, then the db partial
  int H;
  // [r5] row-vector launches whose jobs all walk the same rows (the 64-column slabs of one wide weight gradient: they share an
  // operand): interleave = workgroups per job (a multiple of 8, the same for every job).  Workgroup b then serves job (b % (8 njobs)) / 8,
  // share 8 (b / (8 njobs)) + b % 8 -- the workgroups that read the SAME rows of the shared operand are 8 apart in blockIdx, i.e. on the
  // same XCD and resident together, so two of three (five of six) of those reads are L2 hits.  0: contiguous ranges per job (wg0).
  int interleave;
};
__device__ __forceinline__ void wgrad_where(const WgradLaunch& L, int& j, int& wg, int& nwg) {
  if (L.interleave > 0) {
    const int per = 8 * L.njobs, b = (int)blockIdx.x;
    j = (b % per) >> 3;
    wg = 8 * (b / per) + (b & 7);
    nwg = L.interleave;
    return;
  }
  j = 0;
  while (j + 1 < L.njobs && (int)blockIdx.x >= L.wg0[j + 1]) ++j;
  nwg = L.wg0[j + 1] - L.wg0[j];
  wg = blockIdx.x - L.wg0[j];
}

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

  float cs[HB];  // wave 0: per-lane column sums of A (bias gradient)
#pragma unroll
  for (int jb = 0; jb < HB; ++jb) cs[jb] = 0.f;
  const int kb0 = wv * KPW;
  if (kb0 < J.nkb && t0 < t1) {
    // operands of one 16-row tile in N-layout; tile t+1 is loaded while tile t multiplies
    float av[2][HB][4], bv[2][KPW][4];
    auto load_tile = [&](long tile, float (&a)[HB][4], float (&b)[KPW][4]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = tile * 16 + 4 * g + r;
        const bool ok = row < J.M;
        const float* ap = J.A + (ok ? row : 0) * J.lda + c;
        const float* bp = J.B + (ok ? row : 0) * J.ldb + c;
#pragma unroll
        for (int jb = 0; jb < HB; ++jb) a[jb][r] = (ok && jb < J.nja) ? ld1(ap + 16 * jb) : 0.f;
#pragma unroll
        for (int kk = 0; kk < KPW; ++kk) {
          const int col = 16 * (kb0 + kk) + c;
          b[kk][r] = (ok && col < J.kw) ? ld1(bp + 16 * (kb0 + kk)) : 0.f;
        }
      }
    };
    auto mac_tile = [&](const float (&a)[HB][4], const float (&b)[KPW][4]) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int kk = 0; kk < KPW; ++kk)
#pragma unroll
          for (int jb = 0; jb < HB; ++jb) acc[kk][jb] = MFMA16(a[jb][r], b[kk][r], acc[kk][jb]);
      if (wv == 0 && J.db != nullptr) {
#pragma unroll
        for (int jb = 0; jb < HB; ++jb) cs[jb] += (a[jb][0] + a[jb][1]) + (a[jb][2] + a[jb][3]);
      }
    };
    load_tile(t0, av[0], bv[0]);
    long tile = t0;
    for (; tile + 2 <= t1 - 1; tile += 2) {  // two tiles per trip keeps buffer indices static
      load_tile(tile + 1, av[1], bv[1]);
      mac_tile(av[0], bv[0]);
      load_tile(tile + 2, av[0], bv[0]);
      mac_tile(av[1], bv[1]);
    }
    if (tile + 1 <= t1 - 1) {
      load_tile(tile + 1, av[1], bv[1]);
      mac_tile(av[0], bv[0]);
      mac_tile(av[1], bv[1]);
    } else {
      mac_tile(av[0], bv[0]);
    }
  }
  // D layout: lane (c,g), reg q -> dW[16*jb + 4g + q][16*kb + c]
  float* P = L.partial + (size_t)blockIdx.x * (H * H + H);
  if (wv == 0 && J.db != nullptr) {
#pragma unroll
    for (int jb = 0; jb < HB; ++jb) {
      const float v = rowsum4(cs[jb]);
      if (g == 0 && jb < J.nja) P[H * H + 16 * jb + c] = v;
    }
  }
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

// (k_wgrad_row64, the row-vector variant for jobs up to 64 x 64, follows the mgn_x6.inc include: its bf16 form uses the bf16 MFMA helpers)

// LDS-staged variant for full 128 x 128 jobs (lda = ldb = 128): the workgroup streams
// 32-row tiles of A (= dZ) and B (= X) through LDS by DMA, double buffered; every element is
// fetched from L2/HBM once per workgroup instead of once per wave.  Tile image: row r at
// byte 512*r, 16-byte chunk p of the row holds global chunk p ^ (4*(r&1)): rows m and m+1
// then sit 16 banks apart, so the N-layout ds_read_b32 (16 lanes x 4 rows) is conflict-free.
#define WG_TILE_ROWS 32
#define WG_TILE_BYTES (WG_TILE_ROWS * 512)

// one 1 KB piece (2 rows) of a 32-row tile of X -> LDS; piece q = 4*wv + i, i in 0..3
__device__ __forceinline__ void dma_row_piece(const float* __restrict__ X, long row0, long M, unsigned lds0, int wv, int lane, int i) {
  const int q = 4 * wv + i;
  const int r = 2 * q + (lane >> 5);
  long row = row0 + r;
  row = row < M ? row : M - 1;  // tail rows are masked at the MFMA
  const int gch = (lane & 31) ^ (4 * (r & 1));
  const unsigned voff = (unsigned)((row - row0) * 512 + 16 * gch);  // bytes from the tile's first row
  glds16(uniform_ptr(X + row0 * 128), voff, lds0 + q * 1024);
}

__global__ void __launch_bounds__(256, 2) k_wgrad_lds(const WgradLaunch L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TL_DECL();
  lds_char* sm = (lds_char*)smem;  // [2 buffers][A tile | B tile]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  int j = 0;
  while (j + 1 < L.njobs && (int)blockIdx.x >= L.wg0[j + 1]) ++j;
  const mgn_wgrad_job J = L.job[j];
  const int nwg = L.wg0[j + 1] - L.wg0[j];
  const int wg = blockIdx.x - L.wg0[j];
  const long ntiles = (J.M + WG_TILE_ROWS - 1) / WG_TILE_ROWS;
  const long t0 = ntiles * wg / nwg, t1 = ntiles * (wg + 1) / nwg;
  const unsigned sm0 = __builtin_amdgcn_readfirstlane(lds_addr_of(sm));

  f32x4 acc[2][8];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) acc[kk][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float cs[8];  // wave 0 only: per-lane column sums of A (bias gradient)
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) cs[jb] = 0.f;
  const bool want_db = (J.db != nullptr) && (wv == 0);
  const int kb0 = 2 * wv;
  // lane (c,g) reads feature 16*b + c of row 4*quad + g; rows of odd parity are stored with
  // their 64-byte chunks pair-swapped (XOR 64 on the byte offset), and 4*quad is even, so the
  // swizzle is a per-lane constant: two base offsets (even / odd feature block) + immediates
  const int lb = g * 512 + 4 * c;
  const int sw = (g & 1) << 6;
  auto ld_quad = [&](lds_char* ta, lds_char* tb, int quad, float (&av)[8], float (&bv)[2]) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
      av[jb] = *(__attribute__((address_space(3))) const float*)(ta + quad * 2048 + ((lb + 64 * jb) ^ sw));
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      bv[kk] = *(__attribute__((address_space(3))) const float*)(tb + quad * 2048 + ((lb + 64 * (kb0 + kk)) ^ sw));
  };
  if (t0 < t1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma_row_piece(J.A, t0 * WG_TILE_ROWS, J.M, sm0, wv, lane, i);
      dma_row_piece(J.B, t0 * WG_TILE_ROWS, J.M, sm0 + WG_TILE_BYTES, wv, lane, i);
    }
  }
  for (long tile = t0; tile < t1; ++tile) {
    const int buf = (int)((tile - t0) & 1);
    TL_STAMP(1);
    dma_drain();
    TL_STAMP(3);
    __syncthreads();  // tile landed; the other buffer is free again
    TL_STAMP(4);
    const bool more = tile + 1 < t1;
    const unsigned nb0 = sm0 + (buf ^ 1) * 2 * WG_TILE_BYTES;
    lds_char* ta = sm + buf * 2 * WG_TILE_BYTES;
    lds_char* tb = ta + WG_TILE_BYTES;
    const bool tail = (tile + 1) * WG_TILE_ROWS > J.M;  // only a job's very last tile
    float av[2][8], bv[2][2];
    ld_quad(ta, tb, 0, av[0], bv[0]);
    // rows past M (the DMA clamps them to row M-1) must not contribute: in the tail tile the A
    // operand is multiplied by a 0/1 lane mask; every other tile runs the mask-free body
    const int rows_left = (int)(J.M - tile * WG_TILE_ROWS);
    auto body = [&](auto TAIL) {
#pragma unroll
      for (int quad = 0; quad < WG_TILE_ROWS / 4; ++quad) {
        if (quad + 1 < WG_TILE_ROWS / 4) ld_quad(ta, tb, quad + 1, av[(quad + 1) & 1], bv[(quad + 1) & 1]);
#ifndef MGN_EXP_NODMA
        if (more && quad < 4) {  // next tile's DMA: two pieces per quad, early, under the MFMAs
          dma_row_piece(J.A, (tile + 1) * WG_TILE_ROWS, J.M, nb0, wv, lane, quad);
          dma_row_piece(J.B, (tile + 1) * WG_TILE_ROWS, J.M, nb0 + WG_TILE_BYTES, wv, lane, quad);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (decltype(TAIL)::value) {
          const float keep = (4 * quad + g < rows_left) ? 1.f : 0.f;
#pragma unroll
          for (int jb = 0; jb < 8; ++jb) av[quad & 1][jb] *= keep;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int jb = 0; jb < 8; ++jb) acc[kk][jb] = MFMA16(av[quad & 1][jb], bv[quad & 1][kk], acc[kk][jb]);
        if (want_db) {
#pragma unroll
          for (int jb = 0; jb < 8; ++jb) cs[jb] += av[quad & 1][jb];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (tail)
      body(std::true_type{});
    else
      body(std::false_type{});
    TL_STAMP(7);
  }
  TL_DUMP();
  float* P = L.partial + (size_t)blockIdx.x * (128 * 128 + 128);
  if (want_db) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
      const float v = rowsum4(cs[jb]);
      if (g == 0) P[128 * 128 + 16 * jb + c] = v;
    }
  }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
      for (int q = 0; q < 4; ++q) P[(16 * jb + 4 * g + q) * 128 + 16 * (kb0 + kk) + c] = acc[kk][jb][q];
}

#include "mgn_x6.inc"
#include "mgn_pp.inc"
#include "mgn_ppr.inc"
#include "mgn_fused.inc"

// Row-vector variant for jobs up to 64 x 64 (the 64 x 64 block jobs dense.py cuts the Transformer's / gated MLP's weight gradients
// into): a lane fetches FOUR consecutive features of a row in one 16-byte load -- 16 lanes cover the 256 bytes of a 64-wide row, a
// wave instruction four whole rows -- where the generic kernel above spends sixteen 4-byte loads per tile on A alone and fetches A
// once per wave.  The feature a lane holds in component q is 4c + q, so MFMA block q multiplies the STRIDED feature set
// {q, 4 + q, ..}: a permutation of the rows / columns of dW that the store undoes.  Every wave owns whole tiles (16 rows, no operand
// is loaded twice) and a full 64 x 64 accumulator; the four waves are summed through LDS, the workgroup writes one partial in the
// layout k_wgrad_red reads.  Needs lda % 4 == ldb % 4 == kw % 4 == 0 and 16-byte aligned operands (wgrad_job_row64).
// [r4] BF (bf16 matrix mode): the same loads and the same double buffering, but a tile is ONE K = 16 step of v_mfma_f32_16x16x16_bf16 --
// a lane's four rows (4g + r) are its four K values, the same assignment on both operands -- with the operands rounded to bf16 as
// the reference's autocast Linear backward sees them: 16 matrix instructions per tile instead of 64 exact-fp32 ones.  (A K = 32 form
// over pairs of tiles needs a second raw pair in flight to hide the loads: 664 bytes of scratch per lane, or 4 % slower without it.)
template <bool BF, bool A16 = false, bool B16 = false>
__global__ void __launch_bounds__(256, 2) k_wgrad_row64(const WgradLaunch L) {
  constexpr int H = 64;
  __shared__ float red[3][H * H + H];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  int j, wg, nwg;
  wgrad_where(L, j, wg, nwg);
  const mgn_wgrad_job J = L.job[j];
  const int pidx = L.wg0[j] + wg;   // the partial's slot: contiguous per job whatever the workgroup order
  const long ntiles = (J.M + 15) >> 4;
  const long t0 = ntiles * wg / nwg, t1 = ntiles * (wg + 1) / nwg;
  const bool a_on = 4 * c < 16 * J.nja, b_on = 4 * c < J.kw;

  f32x4 acc[4][4];  // [qa][qb]
#pragma unroll
  for (int qa = 0; qa < 4; ++qa)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) acc[qa][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};  // column sums of A over this lane's rows (bias gradient)
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 av[2][4], bv[2][4];
  // [r5] BF: an operand with a NEGATIVE leading dimension is two-byte bf16 rows of pitch -ld elements (what the bf16 mode's dense
  // launches write with out16: the values are bf16 numbers, the rows half the bytes)
  // (compile-time per launch: as run-time selects the two load forms cost 19 registers and 72 bytes of scratch)
  constexpr bool a16 = BF && A16, b16 = BF && B16;
  const long plda = A16 ? -(long)J.lda : J.lda, pldb = B16 ? -(long)J.ldb : J.ldb;
  auto ld16 = [&](const float* X, long off) -> f32x4 {
    const uint2 t = *(const uint2*)((const uint16_t*)X + off);
    return f32x4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)};
  };
  auto load_tile = [&](long tile, f32x4 (&a)[4], f32x4 (&b)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = tile * 16 + 4 * g + r;
      const bool ok = row < J.M;
      if constexpr (a16) a[r] = (ok && a_on) ? ld16(J.A, row * plda + 4 * c) : zero;
      else a[r] = (ok && a_on) ? ld4(J.A + row * plda + 4 * c) : zero;
      if constexpr (b16) b[r] = (ok && b_on) ? ld16(J.B, row * pldb + 4 * c) : zero;
      else b[r] = (ok && b_on) ? ld4(J.B + row * pldb + 4 * c) : zero;
    }
  };
  auto mac_tile = [&](const f32x4 (&a)[4], const f32x4 (&b)[4]) {
    if constexpr (BF) {   // the lane's four rows are the four K values of a K = 16 bf16 MFMA (v_mfma_f32_16x16x16_bf16), same on both operands
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      typedef u32 u32x2_ __attribute__((ext_vector_type(2)));
      s16x4 pa[4], pb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pa[q] = __builtin_bit_cast(s16x4, u32x2_{pk_bf16(a[0][q], a[1][q]), pk_bf16(a[2][q], a[3][q])});
        pb[q] = __builtin_bit_cast(s16x4, u32x2_{pk_bf16(b[0][q], b[1][q]), pk_bf16(b[2][q], b[3][q])});
      }
#pragma unroll
      for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) acc[qa][qb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa[qa], pb[qb], acc[qa][qb], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) cs += a[r];
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) acc[qa][qb] = MFMA16(a[r][qa], b[r][qb], acc[qa][qb]);
#pragma unroll
    for (int r = 0; r < 4; ++r) cs += a[r];
  };
  long tile = t0 + wv;
  if (tile < t1) {
    load_tile(tile, av[0], bv[0]);
    for (; tile + 8 < t1; tile += 8) {  // two tiles per trip keeps buffer indices static
      load_tile(tile + 4, av[1], bv[1]);
      mac_tile(av[0], bv[0]);
      load_tile(tile + 8, av[0], bv[0]);
      mac_tile(av[1], bv[1]);
    }
    if (tile + 4 < t1) {
      load_tile(tile + 4, av[1], bv[1]);
      mac_tile(av[0], bv[0]);
      mac_tile(av[1], bv[1]);
    } else {
      mac_tile(av[0], bv[0]);
    }
  }
  // lane (c,g), block (qa,qb), reg v  ->  dW[16g + 4v + qa][4c + qb]; components qb are consecutive columns: one 16-byte value
#pragma unroll
  for (int q = 0; q < 4; ++q) cs[q] = rowsum4(cs[q]);  // all g hold the sum over the wave's rows of features 4c..4c+3
  auto put = [&](float* dst) {
#pragma unroll
    for (int qa = 0; qa < 4; ++qa)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4 o = {acc[qa][0][v], acc[qa][1][v], acc[qa][2][v], acc[qa][3][v]};
        *(f32x4*)(dst + (16 * g + 4 * v + qa) * H + 4 * c) = o;
      }
    if (g == 0) *(f32x4*)(dst + H * H + 4 * c) = cs;
  };
  if (wv > 0) put(red[wv - 1]);
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
      for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const f32x4 o = *(const f32x4*)(red[w] + (16 * g + 4 * v + qa) * H + 4 * c);
#pragma unroll
          for (int qb = 0; qb < 4; ++qb) acc[qa][qb][v] += o[qb];
        }
      cs += *(const f32x4*)(red[w] + H * H + 4 * c);
    }
    float* P = L.partial + (size_t)pidx * (H * H + H);
#pragma unroll
    for (int qa = 0; qa < 4; ++qa)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4 o = {acc[qa][0][v], acc[qa][1][v], acc[qa][2][v], acc[qa][3][v]};
        st4(P + (16 * g + 4 * v + qa) * H + 4 * c, o);
      }
    if (g == 0 && J.db != nullptr) st4(P + H * H + 4 * c, cs);
  }
}


// [r5] The same jobs on the split-bf16 matrix path (fp32-grade: six bf16 terms, fp32 accumulate, as k_wgrad_x6): a tile is 32 rows = ONE
// K = 32 step.  Lane (c, g) fetches features 4c .. 4c + 3 of rows 8g .. 8g + 7 (eight 16-byte loads per operand: a wave instruction is four
// 256-byte row pieces), so for each of its four features it HOLDS the eight K values of an MFMA operand lane; split8 makes the three
// pieces, block q multiplies the strided feature set {q, 4 + q, ..} as in k_wgrad_row64, same accumulator / store mapping.  96 bf16 MFMAs
// of 16 cycles per 32 rows where the exact-fp32 form issues 128 of 32 cycles: the exact form was matrix-bound (three 64 x 64 jobs over
// 150 000 rows = 30 us of fp32 MFMA issue in a 57 us launch, profiles/r04_c5_fp32_kernel_stats.csv).  The next tile's loads are issued
// as soon as an operand's rows have been split (they land under the 96 MFMAs); MGN_FP32_MFMA=1 keeps the exact-fp32 kernel.
template <bool FULL>   // FULL: every job of the launch is 64 x 64 with kw = 64 (no lane is masked)
__global__ void __launch_bounds__(256, 2) k_wgrad_row64x6(const WgradLaunch L) {
  constexpr int H = 64;
  __shared__ float red[3][H * H + H];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  int j, wg, nwg;
  wgrad_where(L, j, wg, nwg);
  const mgn_wgrad_job J = L.job[j];
  const int pidx = L.wg0[j] + wg;   // the partial's slot: contiguous per job whatever the workgroup order
  const long ntiles = (J.M + 31) >> 5;
  const long t0 = ntiles * wg / nwg, t1 = ntiles * (wg + 1) / nwg;
  const bool a_on = FULL || 4 * c < 16 * J.nja, b_on = FULL || 4 * c < J.kw;
  f32x4 acc[4][4];  // [qa][qb]
#pragma unroll
  for (int qa = 0; qa < 4; ++qa)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) acc[qa][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 av[8], bv[8];
  // addressing: a wave-uniform row pointer (scalar registers) + ONE loop-invariant 32-bit lane offset per operand; rows past M read
  // row M - 1 (A's are zeroed: 0 x finite = 0), masked lanes read column 0 and are zeroed
  const unsigned voa = (unsigned)((8 * g * J.lda + (a_on ? 4 * c : 0)) * 4), vob = (unsigned)((8 * g * J.ldb + (b_on ? 4 * c : 0)) * 4);
  auto load_op = [&](const float* X, int ld, unsigned vo, bool on, bool is_a, long tile, f32x4 (&v)[8]) {
    if ((tile + 1) * 32 <= J.M) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = *(g_cf32x4*)((const char*)(X + (tile * 32 + r) * ld) + vo);
    } else {
      const long r0 = tile * 32 + 8 * g;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const long row = r0 + r < J.M ? r0 + r : J.M - 1;
        v[r] = ld4(X + row * ld + (on ? 4 * c : 0));
        if (is_a && r0 + r >= J.M) v[r] = zero;
      }
    }
    if (!FULL && !on) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = zero;
    }
  };
  // B's four blocks are split first (48 registers of pieces, its rows re-loaded at once); A's blocks one at a time, each
  // followed by its 24 MFMAs (the accumulators carry the sums of all earlier tiles: the order within a tile does not matter)
  u32x4 bp[4][3];
  auto split_blk = [&](const f32x4 (&v)[8], int q, u32x4 (&p)[3]) {
    const float x[8] = {v[0][q], v[1][q], v[2][q], v[3][q], v[4][q], v[5][q], v[6][q], v[7][q]};
    split8(x, p[0], p[1], p[2]);
  };
  long tile = t0 + wv;
  if (tile < t1) {
    load_op(J.A, J.lda, voa, a_on, true, tile, av);
    load_op(J.B, J.ldb, vob, b_on, false, tile, bv);
    for (; tile < t1; tile += 4) {
      const bool more = tile + 4 < t1;
#pragma unroll
      for (int q = 0; q < 4; ++q) split_blk(bv, q, bp[q]);
      __builtin_amdgcn_sched_barrier(0);
      if (more) load_op(J.B, J.ldb, vob, b_on, false, tile + 4, bv);
#pragma unroll
      for (int r = 0; r < 8; ++r) cs += av[r];
#pragma unroll
      for (int qa = 0; qa < 4; ++qa) {
        u32x4 ap[3];
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler hoists all four splits otherwise: 256 registers + scratch)
        split_blk(av, qa, ap);
        if (qa == 3) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) load_op(J.A, J.lda, voa, a_on, true, tile + 4, av);
        }
#pragma unroll
        for (int term = 0; term < 6; ++term) {
          const int pa = (term == 0) ? 2 : (term == 1) ? 0 : (term == 2 || term == 3) ? 1 : 0;
          const int pb = (term == 0) ? 0 : (term == 1) ? 2 : (term == 2) ? 1 : (term == 3) ? 0 : (term == 4) ? 1 : 0;
#pragma unroll
          for (int qb = 0; qb < 4; ++qb) acc[qa][qb] = MFMA_BF(ap[pa], bp[qb][pb], acc[qa][qb]);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) cs[q] = rowsum4(cs[q]);
  auto put = [&](float* dst) {
#pragma unroll
    for (int qa = 0; qa < 4; ++qa)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4 o = {acc[qa][0][v], acc[qa][1][v], acc[qa][2][v], acc[qa][3][v]};
        *(f32x4*)(dst + (16 * g + 4 * v + qa) * H + 4 * c) = o;
      }
    if (g == 0) *(f32x4*)(dst + H * H + 4 * c) = cs;
  };
  if (wv > 0) put(red[wv - 1]);
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
      for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const f32x4 o = *(const f32x4*)(red[w] + (16 * g + 4 * v + qa) * H + 4 * c);
#pragma unroll
          for (int qb = 0; qb < 4; ++qb) acc[qa][qb][v] += o[qb];
        }
      cs += *(const f32x4*)(red[w] + H * H + 4 * c);
    }
    float* P = L.partial + (size_t)pidx * (H * H + H);
#pragma unroll
    for (int qa = 0; qa < 4; ++qa)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4 o = {acc[qa][0][v], acc[qa][1][v], acc[qa][2][v], acc[qa][3][v]};
        st4(P + (16 * g + 4 * v + qa) * H + 4 * c, o);
      }
    if (g == 0 && J.db != nullptr) st4(P + H * H + 4 * c, cs);
  }
}


// dW[r,k] = sum over the job's workgroup partials.  64 outputs x 4 partial-lanes per block:
// consecutive threads read consecutive addresses of one partial (coalesced), four lanes walk
// the partials interleaved (short dependent chains), then a fixed-order LDS combine.
// (1024 threads = 64 elements x 16 groups of partials: a group sums every 16th partial, two accumulators; 4 groups of 256
//  threads walked 128 partials each through dependent 16 KB-strided loads -- 13-15 us per launch, 72 launches per Transformer step)
// [r5] WRED_GROUPS is a template argument: with at most 64 partials per job (the producer / consumer kernel's 256 workgroups shared by ten
// jobs; every launch of a one-mesh batch) four groups do -- 2 580 blocks of 1 024 threads for 165 000 sums cost 9-11 us per launch
// whatever the partial count (21 launches per training step at batch 1).  The group count follows from the launch shape alone:
// the summation order stays fixed run to run.
template <int WRED_GROUPS>
__global__ void __launch_bounds__(64 * WRED_GROUPS) k_wgrad_red(const WgradLaunch L) {
  __shared__ float red[WRED_GROUPS][64];
  const int H = L.H;
  const size_t st = (size_t)H * H + H;
  const int j = blockIdx.y;
  const mgn_wgrad_job J = L.job[j];
  const int rows = 16 * J.nja, cols = 16 * J.nkb;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int pl = threadIdx.x >> 6;
  const int nwg = L.wg0[j + 1] - L.wg0[j];
  const float* P0 = L.partial + (size_t)L.wg0[j] * st;
  const bool is_w = i < rows * cols;
  const bool is_b = !is_w && i < rows * cols + rows && J.db != nullptr;
  int r = 0, k = 0;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (is_w || is_b) {
    const float* P;
    if (is_w) {
      r = i / cols;
      k = i % cols;
      P = P0 + r * H + k;
    } else {
      r = i - rows * cols;
      P = P0 + (size_t)H * H + r;
    }
    int b = pl;
    for (; b + 3 * WRED_GROUPS < nwg; b += 4 * WRED_GROUPS) {   // four independent chains: the loads are 66 KB apart (L2 misses)
      s0 += P[(size_t)b * st];
      s1 += P[(size_t)(b + WRED_GROUPS) * st];
      s2 += P[(size_t)(b + 2 * WRED_GROUPS) * st];
      s3 += P[(size_t)(b + 3 * WRED_GROUPS) * st];
    }
    if (b < nwg) s0 += P[(size_t)b * st];
    if (b + WRED_GROUPS < nwg) s1 += P[(size_t)(b + WRED_GROUPS) * st];
    if (b + 2 * WRED_GROUPS < nwg) s2 += P[(size_t)(b + 2 * WRED_GROUPS) * st];
  }
  red[pl][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0) {
    float v = 0.f;  // fixed order: deterministic
#pragma unroll
    for (int q = 0; q < WRED_GROUPS; q += 4)
      v += (red[q][threadIdx.x] + red[q + 1][threadIdx.x]) + (red[q + 2][threadIdx.x] + red[q + 3][threadIdx.x]);
    if (is_w) {
      if (k < J.ldw) J.dW[(size_t)r * J.ldw + k] = v;
    } else if (is_b) {
      J.db[r] = v;
    }
  }
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
        v[u] = ld4(base + row * H);
      } else {
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < end) s += v[u];
  }
  st4(out + node * H + 4 * l, s);
}

// Two segment sums of the SAME source in one launch (the backward scatter of the first-layer
// gradients onto destination and source nodes): workgroups [0, half) run job 0, the rest job 1.
template <int HB, bool SRC16 = false>
__global__ void __launch_bounds__(256) k_segsum2(const float* __restrict__ src, const int32_t* __restrict__ rowptr0,
                                                 const int32_t* __restrict__ perm0, float* __restrict__ out0,
                                                 const int32_t* __restrict__ rowptr1, const int32_t* __restrict__ perm1,
                                                 float* __restrict__ out1, long N, unsigned half) {
  constexpr int H = 16 * HB;
  constexpr int LPR = H / 4;
  const bool second = blockIdx.x >= half;
  const int32_t* rowptr = second ? rowptr1 : rowptr0;
  const int32_t* perm = second ? perm1 : perm0;
  float* out = second ? out1 : out0;
  const long node = ((long)(blockIdx.x - (second ? half : 0)) * 256 + threadIdx.x) / LPR;
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
        if constexpr (SRC16) {   // two-byte rows in the packed feature order of the chain kernels (mgn_mlp_fwd_args.precision): features
                                 // 4l .. 4l + 3 are the four bf16 values at element 32 (l >> 3) + 8 (l & 3) + 4 ((l >> 2) & 1)
          const uint2 t = *(const uint2*)((const uint16_t*)src + row * H + 32 * (l >> 3) + 8 * (l & 3) + 4 * ((l >> 2) & 1));
          v[u] = f32x4{__builtin_bit_cast(float, t.x << 16), __builtin_bit_cast(float, t.x & 0xffff0000u),
                       __builtin_bit_cast(float, t.y << 16), __builtin_bit_cast(float, t.y & 0xffff0000u)};
        } else {
          v[u] = ld4(base + row * H);
        }
      } else {
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < end) s += v[u];
  }
  st4(out + node * H + 4 * l, s);
}

// Completes the segment sum fused into the split-bf16 edge kernel: segments cut by a 16-row
// wave-tile boundary are assembled from the per-tile partials in tile order; empty segments
// (nodes without incoming edges) are zeroed; segments inside one tile were written by the kernel.
__global__ void __launch_bounds__(256) k_seg_fix(const int32_t* __restrict__ rowptr, const float* __restrict__ part,
                                                 float* __restrict__ out, long N) {
  constexpr int H = 128, LPR = 32;
  const long node = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  if (node >= N) return;
  const int b = rowptr[node], e = rowptr[node + 1];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (b < e) {
    const int t0 = b >> 4, t1 = (e - 1) >> 4;
    if (t0 == t1) return;
    s = ld4(part + ((size_t)t0 * 2 + 1) * H + 4 * l);
    for (int t = t0 + 1; t <= t1; ++t) s += ld4(part + ((size_t)t * 2) * H + 4 * l);
  }
  st4(out + node * H + 4 * l, s);
}

// ================================================================ batched transpose
struct TBlocks {
  mgn_tblock b[TB_MAX];
};
// one workgroup per 32x32 tile of one block; LDS tile padded against bank conflicts
__global__ void __launch_bounds__(256) k_transpose_blocks(const TBlocks T, int H) {
  __shared__ float tile[32][33];
  const mgn_tblock B = T.b[blockIdx.y];
  const int tpr = H / 32 > 0 ? H / 32 : 1;
  const int tj = (blockIdx.x / tpr) * 32, tk = (blockIdx.x % tpr) * 32;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ly; r < 32; r += 8)
    if (tj + r < H && tk + lx < H) tile[r][lx] = B.src[(size_t)(tj + r) * B.ld_src + tk + lx];
  __syncthreads();
#pragma unroll
  for (int r = ly; r < 32; r += 8)
    if (tk + r < H && tj + lx < H) B.dst[(size_t)(tk + r) * B.ld_dst + tj + lx] = tile[lx][r];
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

struct MlpPlan {
  bool lds;       // LDS-weight kernel (H = 128, full widths)
  int mt;         // 16-row tiles per wave
  unsigned grid;  // workgroups
  size_t smem;    // dynamic LDS bytes
};

static bool fwd_ragged(const mgn_mlp_fwd_args& a) {
  bool r = (a.out_w != a.H);
  for (int p = 0; p < a.nphase; ++p) r = r || (a.kw[p] != a.H);
  return r;
}

static MlpPlan plan_mlp(int64_t M, int H, int NL, bool ragged, bool bwd, int act = MGN_ACT_RELU) {
  MlpPlan p;
  p.lds = (H == 128) && !ragged && NL >= (bwd ? 2 : 1) && NL <= LDS_MAX_NL;
  if (getenv("MGN_NO_LDS") != nullptr) p.lds = false;
  if (act == MGN_ACT_GELU) p.lds = false;  // GELU lives on the generic kernels only (a stand-alone build_mlp option)
  p.mt = (M >= (int64_t)64 * 2048) ? 2 : 1;
  // measured on MI355X (tools/kbench.py, E = 180k): with LDS-shared weights the forward is
  // faster at 16 rows per wave (172 VGPRs, no spills: 352 vs 368 us), the backward chain at 32
  if (p.lds) p.mt = 1;
  if (const char* e = getenv("MGN_MT")) p.mt = (atoi(e) == 2) ? 2 : 1;
  const int rows = 64 * p.mt;
  p.grid = (unsigned)((M + rows - 1) / rows);
  if (p.lds && p.grid > 512) p.grid = 512;  // persistent: 2 workgroups per CU walk the tiles
  if (const char* e = getenv("MGN_GRID")) { if (p.lds && !bwd && atoi(e) > 0 && (unsigned)atoi(e) < p.grid) p.grid = (unsigned)atoi(e); }
  p.smem = p.lds ? (bwd ? (size_t)2 * WBUF_BYTES + 512 + (size_t)4 * (NL + 1) * H * sizeof(float) : (size_t)FWD_LDS_BYTES) : 0;
  return p;
}

template <typename K>
static int set_smem(K kern, size_t bytes) {
  static thread_local const void* done[16];
  static thread_local int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done[i] == (const void*)kern) return 0;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * WBUF_BYTES + 4 * 7 * 128 * 4)) != hipSuccess) return 1;
  (void)bytes;
  if (ndone < 16) done[ndone++] = (const void*)kern;
  return 0;
}

// split-bf16 kernels: every GEMM unit of the launch comes packed (and the fp32 override is off)
static bool fwd_x6(const mgn_mlp_fwd_args& a) {
  if (a.wpk[0] == nullptr || getenv("MGN_FP32_MFMA") != nullptr) return false;
  const int G = a.nphase + a.NL - 1 + a.n_post;
  if (G > X6_MAX_UNITS || a.NL > LDS_MAX_NL) return false;
  // register sharing inside the kernel: gathered adds ride in the next-phase buffer
  if (a.n_add > 0 && (a.nphase != 1 || (a.NL == 1 && a.resid != nullptr))) return false;
  if (a.seg_out != nullptr && (a.seg_key == nullptr || a.seg_rowptr == nullptr || a.seg_part == nullptr || a.n_post > 0)) return false;
  for (int u = 0; u < G; ++u)
    if (a.wpk[u] == nullptr) return false;
  return true;
}
template <int TERMS, int NW, int ACT, class SH = ShDyn>
static int set_fwd_x6_attr() {
  return hipFuncSetAttribute((const void*)k_mlp_fwd_x6<TERMS, NW, ACT, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_FWD_LDS_BYTES(NW)) != hipSuccess;
}
// The static-shape instantiations (mgn_x6.inc: unrolled unit loop, untracked operand loads) take a launch only
// when it matches their shape field by field; MGN_X6_STATIC=0 keeps every launch on the dynamic kernel (A/B).
static int fwd_static_shape(const mgn_mlp_fwd_args& a) {
  const char* env = getenv("MGN_X6_STATIC");  // read per launch: the cross-check test flips it inside one process
  const int mode = env != nullptr ? atoi(env) : 1;  // 0 none, 2 edge only
  if (mode == 0 || a.act != MGN_ACT_RELU || a.NL != 4 || a.resid == nullptr || a.scale == nullptr) return 0;
  if (a.nphase == 1 && a.n_add == 2 && a.n_post == 0 && a.idx[0] == nullptr && a.add_idx[0] != nullptr && a.add_idx[1] != nullptr &&
      a.seg_out != nullptr && a.seg_key == a.add_idx[0])
    return 1;  // ShEdge
  // ShNode (node update): only where a workgroup runs several tiles.  On a one-tile launch (30 k node rows = 472
  // tiles) its 28 straight-line quarters are instruction fetch with no reuse: 64 us against 46 us for the rolled loop.
  if (mode != 2 && a.nphase == 2 && a.n_add == 0 && a.idx[0] == nullptr && a.idx[1] == nullptr && a.seg_out == nullptr && a.M >= 4 * 512 * 64) {
    if (a.n_post == 2) return 2;  // ShNode<2>
    if (a.n_post == 0) return 3;  // ShNode<0>
  }
  return 0;
}

// The ping-pong edge update (mgn_pp.inc) takes a ShEdge launch when it is fp32-grade, its four units lie back to back,
// the messages are not written (fused aggregation only), the saves are all there or all absent, and every CU gets at
// least two 128-row tiles (on a one-tile launch the alternation only doubles the tile's latency).
// MGN_PP unset: INFERENCE-mode launches only (no saves: measured 127-130 us against 132-145 us at the bench size; the
// training-mode instance is slower than the x6 kernel); 0: off; 1: both modes; 2: both modes at any size (tests).
static bool fwd_pp_ok(const mgn_mlp_fwd_args& a) {
  const char* env = getenv("MGN_PP");
  const int mode = (env == nullptr) ? -1 : atoi(env);
  if (mode == 0) return false;
  const int64_t min_rows = (mode == 2) ? 1 : 2 * 128 * 256;
  if (a.precision != 0 || a.y_out != nullptr || a.M < min_rows || a.M * 512 >= (int64_t)1 << 32) return false;  // (32-bit row offsets)
  for (int u = 1; u < 4; ++u)
    if ((const char*)a.wpk[u] != (const char*)a.wpk[0] + (size_t)u * MGN_WPACK_BYTES) return false;
  const bool all = a.saveU && a.saveR && a.saveH[0] && a.saveH[1] && a.saveH[2] && a.saveM[0] && a.saveM[1] && a.saveM[2];
  const bool none = !a.saveU && !a.saveR && !a.saveH[0] && !a.saveH[1] && !a.saveH[2] && !a.saveM[0] && !a.saveM[1] && !a.saveM[2];
  if (mode < 0 && !none) return false;
  // rows past M are computed as copies of row M - 1 and stored there again: the outputs must alias no input
  return (all || none) && a.resid != nullptr && a.scale != nullptr && a.out != a.resid && a.out != a.src[0];
}

// The register-resident-weights edge update (mgn_ppr.inc) takes a ShEdge launch under the ping-pong kernel's conditions (fp32-grade,
// units back to back, no message output, saves all there or all absent, outputs alias no input) from 65 536 rows (every CU gets
// >= 8 groups of 32 rows: below that the two-slot lag of its second half is not amortised).  MGN_PPR: unset = on, both modes;
// 0 = off; 2 = at any size (tests); MGN_PPR_XCD=0: plain hand-out of the groups instead of XCD by XCD (A/B).
static bool fwd_ppr_ok(const mgn_mlp_fwd_args& a) {
  const char* env = getenv("MGN_PPR");
  const int mode = (env == nullptr) ? 1 : atoi(env);
  if (mode == 0) return false;
  const int64_t min_rows = (mode == 2) ? 1 : 65536;
  if (a.precision != 0 || a.y_out != nullptr || a.M < min_rows || a.M * 512 >= (int64_t)1 << 32) return false;  // (32-bit row offsets)
  for (int u = 1; u < 4; ++u)
    if ((const char*)a.wpk[u] != (const char*)a.wpk[0] + (size_t)u * MGN_WPACK_BYTES) return false;
  const bool all = a.saveU && a.saveR && a.saveH[0] && a.saveH[1] && a.saveH[2] && a.saveM[0] && a.saveM[1] && a.saveM[2];
  const bool none = !a.saveU && !a.saveR && !a.saveH[0] && !a.saveH[1] && !a.saveH[2] && !a.saveM[0] && !a.saveM[1] && !a.saveM[2];
  // rows past M are computed as copies of row M - 1 and stored there again: the outputs must alias no input
  return (all || none) && a.resid != nullptr && a.scale != nullptr && a.out != a.resid && a.out != a.src[0];
}

template <int HB>
static int launch_fwd(const mgn_mlp_fwd_args& a, hipStream_t s) {
  const bool ragged = fwd_ragged(a);
  const MlpPlan p = plan_mlp(a.M, a.H, a.NL, ragged, false, a.act);
  if (p.lds && fwd_x6(a)) {
    // 8-wave workgroups (one per CU) when every CU still gets a tile; MGN_NW=4/8 overrides
    int nw = (a.M >= 128 * 256) ? X6_FWD_NW_LARGE : 4;
    bool nw6 = false;  // MGN_NW=6: the three-waves-per-SIMD instance of the edge update (experiment)
    if (const char* e = getenv("MGN_NW")) {
      nw = (atoi(e) == 8) ? 8 : 4;
      nw6 = atoi(e) == 6;
    }
    if (a.seg_out != nullptr) nw = 4;  // (the 8-wave instance has no fused segment sum)
    const bool silu = (a.act == MGN_ACT_SILU);
    if (silu) nw = 4;  // the SiLU variant is built for 4-wave workgroups only
    static thread_local bool attr_done = false;
    if (!attr_done) {
      if (set_fwd_x6_attr<6, 4, 0>() || set_fwd_x6_attr<1, 4, 0>() || set_fwd_x6_attr<6, 8, 0>() || set_fwd_x6_attr<1, 8, 0>() ||
          set_fwd_x6_attr<6, 4, 1>() || set_fwd_x6_attr<1, 4, 1>() || set_fwd_x6_attr<6, 4, 0, ShEdge>() ||
          set_fwd_x6_attr<6, 4, 0, ShNode<2>>() || set_fwd_x6_attr<6, 4, 0, ShNode<0>>() || set_fwd_x6_attr<1, 4, 0, ShEdge>() ||
          set_fwd_x6_attr<1, 4, 0, ShNode<2>>() || set_fwd_x6_attr<1, 4, 0, ShNode<0>>() || set_fwd_x6_attr<6, 6, 0, ShEdge>())
        return 1;
      attr_done = true;
    }
    const int rows = 16 * nw;
    unsigned grid = (unsigned)((a.M + rows - 1) / rows);
    const unsigned cap = (nw == 8) ? 256u : 512u;
    if (grid > cap) grid = cap;
    if (const char* e = getenv("MGN_GRID")) {  // occupancy experiments: fewer persistent workgroups
      if (atoi(e) > 0 && (unsigned)atoi(e) < grid) grid = (unsigned)atoi(e);
    }
    const int shape = (nw == 4 && !silu) ? fwd_static_shape(a) : 0;
    if (shape == 1 && !nw6 && fwd_ppr_ok(a)) {
      static thread_local bool ppr_attr = false;
      if (!ppr_attr) {
        if (hipFuncSetAttribute((const void*)k_edge_fwd_ppr<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PPR_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_edge_fwd_ppr<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PPR_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_edge_fwd_ppr<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PPR_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_edge_fwd_ppr<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PPR_LDS_BYTES) != hipSuccess)
          return 1;
        ppr_attr = true;
      }
      const int64_t ngroups = ((a.M + 15) / 16 + PPR_R - 1) / PPR_R;
      unsigned gp = (ngroups < 256) ? (unsigned)ngroups : 256u;
      const char* ex = getenv("MGN_PPR_XCD");
      const bool xcd = ex == nullptr || atoi(ex) != 0;
      if (a.saveU != nullptr) {
        if (xcd) hipLaunchKernelGGL((k_edge_fwd_ppr<true, true>), dim3(gp), dim3(512), PPR_LDS_BYTES, s, a);
        else hipLaunchKernelGGL((k_edge_fwd_ppr<true, false>), dim3(gp), dim3(512), PPR_LDS_BYTES, s, a);
      } else {
        if (xcd) hipLaunchKernelGGL((k_edge_fwd_ppr<false, true>), dim3(gp), dim3(512), PPR_LDS_BYTES, s, a);
        else hipLaunchKernelGGL((k_edge_fwd_ppr<false, false>), dim3(gp), dim3(512), PPR_LDS_BYTES, s, a);
      }
      return 0;
    }
    if (shape == 1 && !nw6 && fwd_pp_ok(a)) {
      static thread_local bool pp_attr = false;
      if (!pp_attr) {
        if (hipFuncSetAttribute((const void*)k_edge_fwd_pp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_edge_fwd_pp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES) != hipSuccess)
          return 1;
        pp_attr = true;
      }
      unsigned gp = (unsigned)((a.M + 127) / 128);
      if (gp > 256u) gp = 256u;
      if (a.saveU != nullptr)
        hipLaunchKernelGGL((k_edge_fwd_pp<true>), dim3(gp), dim3(512), PP_LDS_BYTES, s, a);
      else
        hipLaunchKernelGGL((k_edge_fwd_pp<false>), dim3(gp), dim3(512), PP_LDS_BYTES, s, a);
      return 0;
    }
    if (shape == 1 && nw6 && a.precision == 0) {
      unsigned g6 = (unsigned)((a.M + 95) / 96);
      if (g6 > 512u) g6 = 512u;
      hipLaunchKernelGGL((k_mlp_fwd_x6<6, 6, 0, ShEdge>), dim3(g6), dim3(384), X6_FWD_LDS_BYTES(6), s, a);
    } else if (shape == 1) {
      if (a.precision >= 1)
        hipLaunchKernelGGL((k_mlp_fwd_x6<1, 4, 0, ShEdge>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
      else
        hipLaunchKernelGGL((k_mlp_fwd_x6<6, 4, 0, ShEdge>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
    } else if (shape == 2) {
      if (a.precision >= 1)
        hipLaunchKernelGGL((k_mlp_fwd_x6<1, 4, 0, ShNode<2>>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
      else
        hipLaunchKernelGGL((k_mlp_fwd_x6<6, 4, 0, ShNode<2>>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
    } else if (shape == 3) {
      if (a.precision >= 1)
        hipLaunchKernelGGL((k_mlp_fwd_x6<1, 4, 0, ShNode<0>>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
      else
        hipLaunchKernelGGL((k_mlp_fwd_x6<6, 4, 0, ShNode<0>>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
    } else if (silu) {
      if (a.precision >= 1)
        hipLaunchKernelGGL((k_mlp_fwd_x6<1, 4, 1>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
      else
        hipLaunchKernelGGL((k_mlp_fwd_x6<6, 4, 1>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
    } else if (nw == 8) {
      if (a.precision >= 1)
        hipLaunchKernelGGL((k_mlp_fwd_x6<1, 8, 0>), dim3(grid), dim3(512), X6_FWD_LDS_BYTES(8), s, a);
      else
        hipLaunchKernelGGL((k_mlp_fwd_x6<6, 8, 0>), dim3(grid), dim3(512), X6_FWD_LDS_BYTES(8), s, a);
    } else {
      if (a.precision >= 1)
        hipLaunchKernelGGL((k_mlp_fwd_x6<1, 4, 0>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
      else
        hipLaunchKernelGGL((k_mlp_fwd_x6<6, 4, 0>), dim3(grid), dim3(256), X6_FWD_LDS_BYTES(4), s, a);
    }
    return 0;
  }
  if (p.lds) {
    if (p.mt == 2) {
      if (set_smem(k_mlp_fwd_lds<2>, p.smem)) return 1;
      hipLaunchKernelGGL((k_mlp_fwd_lds<2>), dim3(p.grid), dim3(256), p.smem, s, a);
    } else {
      if (set_smem(k_mlp_fwd_lds<1>, p.smem)) return 1;
      hipLaunchKernelGGL((k_mlp_fwd_lds<1>), dim3(p.grid), dim3(256), p.smem, s, a);
    }
    return 0;
  }
  if (ragged) {  // encoders / decoder: guarded generic first / last layer
    if (p.mt == 2)
      hipLaunchKernelGGL((k_mlp_fwd<HB, 2, true>), dim3(p.grid), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_mlp_fwd<HB, 1, true>), dim3(p.grid), dim3(256), 0, s, a);
  } else {
    if (p.mt == 2)
      hipLaunchKernelGGL((k_mlp_fwd<HB, 2, false>), dim3(p.grid), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_mlp_fwd<HB, 1, false>), dim3(p.grid), dim3(256), 0, s, a);
  }
  return 0;
}

static bool bwd_x6(const mgn_mlp_bwd_args& a) {
  if (a.wpk[0] == nullptr || getenv("MGN_FP32_MFMA") != nullptr) return false;
  const int G = a.n_front + a.NL - 1 + a.n_din;
  if (G > 8 || G < 1 || a.n_front < 0 || a.n_front > MGN_MAX_PHASES) return false;
  if (a.n_front > 0 && a.dOut2 != nullptr) return false;
  for (int u = 0; u < G; ++u)
    if (a.wpk[u] == nullptr) return false;
  if (a.seg_out != nullptr && (a.seg_key == nullptr || a.seg_rowptr == nullptr || a.seg_part == nullptr || a.dZ[0] == nullptr ||
                               a.n_front > 0 || a.act != MGN_ACT_RELU || a.precision != 0 || a.NL < 2))
    return false;
  if (a.act == MGN_ACT_SILU) {
    if (a.n_front > 0 || (a.n_din == 0 && a.dOut2 != nullptr)) return false;  // see k_mlp_bwd_x6: pd2 carries the z rows
    for (int l = 1; l < a.NL; ++l)
      if (a.Zs[l - 1] == nullptr) return false;
    return true;
  }
  for (int l = 1; l < a.NL; ++l)
    if (a.Ms[l - 1] == nullptr) return false;  // the split-bf16 chain reads ReLU masks as bits
  return true;
}

// The register-resident-weights edge backward chain (mgn_ppr.inc) takes an SbEdge launch when it is fp32-grade, its four units lie
// back to back, only dscale is asked for as a column sum, every dZ is written as fp32 rows and the outputs alias no input (rows past
// M are computed as copies of row M - 1 and stored there again), from 65 536 rows.  MGN_PPR as for the forward kernel; MGN_PPR_BWD=0
// keeps the x6 chain (A/B).
static bool bwd_ppr_ok(const mgn_mlp_bwd_args& a) {
  const char* env = getenv("MGN_PPR");
  const int mode = (env == nullptr) ? 1 : atoi(env);
  const char* eb = getenv("MGN_PPR_BWD");
  if (mode == 0 || (eb != nullptr && atoi(eb) == 0)) return false;
  const int64_t min_rows = (mode == 2) ? 1 : 65536;
  if (a.precision != 0 || a.M < min_rows || a.M * 512 >= (int64_t)1 << 32 || a.dscale == nullptr) return false;
  for (int u = 1; u < 4; ++u)
    if ((const char*)a.wpk[u] != (const char*)a.wpk[0] + (size_t)u * MGN_WPACK_BYTES) return false;
  for (int l = 0; l < 4; ++l)
    if (a.db[l] != nullptr || a.dZ[l] == nullptr) return false;
  if (a.dIn[0] == nullptr || a.Ms[0] == nullptr || a.Ms[1] == nullptr || a.Ms[2] == nullptr) return false;
  const void* ins[4] = {a.dOut, a.U, a.din_resid[0], a.dOut2};
  const void* outs[5] = {a.dZ[0], a.dZ[1], a.dZ[2], a.dZ[3], a.dIn[0]};
  for (const void* o : outs)
    for (const void* i : ins)
      if (o == i) return false;
  return true;
}

template <int HB>
static int launch_bwd(const mgn_mlp_bwd_args& a, hipStream_t s) {
  MlpPlan p = plan_mlp(a.M, a.H, a.NL, a.out_w != a.H || a.n_din > 1, true, a.act);
  if (p.lds && bwd_x6(a)) {
    static thread_local bool attr_done = false;
    if (!attr_done) {
      const int lds = X6_BWD_LDS_BYTES(LDS_MAX_NL + 1);
      if (hipFuncSetAttribute((const void*)k_mlp_bwd_x6<6, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_mlp_bwd_x6<1, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_mlp_bwd_x6<6, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_mlp_bwd_x6<1, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_mlp_bwd_x6<6, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_mlp_bwd_x6<1, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_mlp_bwd_x6<6, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_mlp_bwd_x6<6, false, 0, false, SbEdge>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_mlp_bwd_x6<1, false, 0, false, SbEdge>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return 1;
      attr_done = true;
    }
    int nused = (a.scale != nullptr && a.dscale != nullptr) ? 1 : 0;
    for (int l = 0; l < a.NL; ++l) nused += (a.db[l] != nullptr) ? 1 : 0;
    const size_t lds = X6_BWD_LDS_BYTES(nused);
    const bool front = a.n_front > 0;
    const char* st_env = getenv("MGN_X6_STATIC");
    const bool static_off = st_env != nullptr && atoi(st_env) == 0;
    // static-shape instantiation (mgn_x6.inc, SbEdge): the edge chain of a round, matched field by field
    const bool sb_edge = !static_off && !front && a.seg_out == nullptr && a.act == MGN_ACT_RELU && a.NL == 4 &&
                         a.n_din == 1 && a.din_resid[0] != nullptr && a.scale != nullptr && a.R != nullptr && a.U != nullptr &&
                         a.dOut2 != nullptr && a.idx2 != nullptr;
    if (sb_edge && bwd_ppr_ok(a)) {
      static thread_local bool ppr_attr = false;
      if (!ppr_attr) {
        if (hipFuncSetAttribute((const void*)k_edge_bwd_ppr<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PPB_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_edge_bwd_ppr<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PPB_LDS_BYTES) != hipSuccess)
          return 1;
        ppr_attr = true;
      }
      const int64_t ngroups = ((a.M + 15) / 16 + PPR_R - 1) / PPR_R;
      unsigned gp = (ngroups < 256) ? (unsigned)ngroups : 256u;
      if (gp > p.grid) gp = p.grid;  // (the column reduction reads p.grid partials: never more workgroups than that)
      const char* ex = getenv("MGN_PPR_XCD");
      if (ex == nullptr || atoi(ex) != 0)
        hipLaunchKernelGGL((k_edge_bwd_ppr<true>), dim3(gp), dim3(512), PPB_LDS_BYTES, s, a, (int)p.grid);
      else
        hipLaunchKernelGGL((k_edge_bwd_ppr<false>), dim3(gp), dim3(512), PPB_LDS_BYTES, s, a, (int)p.grid);
    } else if (sb_edge) {
      if (a.precision >= 1)
        hipLaunchKernelGGL((k_mlp_bwd_x6<1, false, 0, false, SbEdge>), dim3(p.grid), dim3(256), lds, s, a);
      else
        hipLaunchKernelGGL((k_mlp_bwd_x6<6, false, 0, false, SbEdge>), dim3(p.grid), dim3(256), lds, s, a);
    } else if (a.seg_out != nullptr) {
      hipLaunchKernelGGL((k_mlp_bwd_x6<6, false, 0, true>), dim3(p.grid), dim3(256), lds, s, a);
    } else if (a.act == MGN_ACT_SILU) {
      if (a.precision >= 1)
        hipLaunchKernelGGL((k_mlp_bwd_x6<1, false, 1>), dim3(p.grid), dim3(256), lds, s, a);
      else
        hipLaunchKernelGGL((k_mlp_bwd_x6<6, false, 1>), dim3(p.grid), dim3(256), lds, s, a);
    } else if (a.precision >= 1) {
      if (front)
        hipLaunchKernelGGL((k_mlp_bwd_x6<1, true, 0>), dim3(p.grid), dim3(256), lds, s, a);
      else
        hipLaunchKernelGGL((k_mlp_bwd_x6<1, false, 0>), dim3(p.grid), dim3(256), lds, s, a);
    } else {
      if (front)
        hipLaunchKernelGGL((k_mlp_bwd_x6<6, true, 0>), dim3(p.grid), dim3(256), lds, s, a);
      else
        hipLaunchKernelGGL((k_mlp_bwd_x6<6, false, 0>), dim3(p.grid), dim3(256), lds, s, a);
    }
    return 0;
  }
  if (p.lds) {
    if (p.mt == 2) {
      if (set_smem(k_mlp_bwd_lds<2>, p.smem)) return 1;
      hipLaunchKernelGGL((k_mlp_bwd_lds<2>), dim3(p.grid), dim3(256), p.smem, s, a);
    } else {
      if (set_smem(k_mlp_bwd_lds<1>, p.smem)) return 1;
      hipLaunchKernelGGL((k_mlp_bwd_lds<1>), dim3(p.grid), dim3(256), p.smem, s, a);
    }
    return 0;
  }
  if (a.out_w != a.H) {
    if (p.mt == 2)
      hipLaunchKernelGGL((k_mlp_bwd<HB, 2, true>), dim3(p.grid), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_mlp_bwd<HB, 1, true>), dim3(p.grid), dim3(256), 0, s, a);
  } else {
    if (p.mt == 2)
      hipLaunchKernelGGL((k_mlp_bwd<HB, 2, false>), dim3(p.grid), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_mlp_bwd<HB, 1, false>), dim3(p.grid), dim3(256), 0, s, a);
  }
  return 0;
}

extern "C" {

int mgn_version(void) { return 136; }
const char* mgn_last_error(void) { return g_err; }

size_t mgn_csr_workspace_bytes(int64_t E, int64_t N) {
  // cnt/cursor[N] + nbig (16 ints) | worklist[N] | tmp[E] | 16 spare ints (error flag of mgn_csr_build)
  return (size_t)(2 * N + E + 32) * sizeof(int);
}

// the launches of one CSR build, no synchronisation; *err_dev is set to 1 on a key outside [0, N)
static int csr_enqueue(const int64_t* key, int64_t E, int64_t N, int32_t* rowptr, int32_t* perm, void* ws, int* err_dev, hipStream_t s) {
  int* cnt = (int*)ws;
  if (hipMemsetAsync(ws, 0, (size_t)(N + 16) * sizeof(int), s) != hipSuccess) return fail(2, "mgn_csr_build: memset");
  if (E > 0) hipLaunchKernelGGL(k_csr_hist, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, key, (long)E, (long)N, cnt, err_dev);
  hipLaunchKernelGGL(k_csr_scan, dim3(1), dim3(1024), 0, s, cnt, (long)N, rowptr);
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
  return check_launch("mgn_csr_build");
}

int mgn_csr_build(const int64_t* key, int64_t E, int64_t N, int32_t* rowptr, int32_t* perm, void* ws,
                  size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (N < 0 || E < 0 || E > 2147483647LL || N > 2147483646LL) return fail(1, "mgn_csr_build: size out of int32 range");
  if (ws_bytes < mgn_csr_workspace_bytes(E, N)) return fail(1, "mgn_csr_build: workspace too small");
  int* err = (int*)ws + 2 * N + E + 24;  // a word of the workspace tail the launches do not use
  if (hipMemsetAsync(err, 0, sizeof(int), s) != hipSuccess) return fail(2, "mgn_csr_build: memset");
  if (int rc = csr_enqueue(key, E, N, rowptr, perm, ws, err, s)) return rc;
  int herr = 0;
  if (hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return fail(2, "mgn_csr_build: memcpy");
  if (hipStreamSynchronize(s) != hipSuccess) return fail(2, "mgn_csr_build: sync failed");
  if (herr) return fail(3, "mgn_csr_build: edge index outside [0, N)");
  return 0;
}

// src_s / dst_s (int32, dst-sorted order) and the int64 keys of the second CSR
__global__ void k_topo_gather(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int32_t* __restrict__ perm,
                              const int32_t* __restrict__ rowptr, long E, long N, int32_t* __restrict__ src_s, int32_t* __restrict__ dst_s,
                              int64_t* __restrict__ key2) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  // An index outside [0, N) is REPORTED through the error flag (read by the host after a synchronisation, or lazily by
  // mgn_topology_build_async's caller); until then every array this build hands out must be safe to compute on: rows
  // past the valid count (their edges were dropped by the histogram) point at edge 0, stray node ids at node 0.
  int e = perm[k];
  if (k >= rowptr[N] || e < 0 || e >= E) {
    e = 0;
    perm[k] = 0;
  }
  int64_t a = src[e], b = dst[e];
  a = (a < 0 || a >= N) ? 0 : a;
  b = (b < 0 || b >= N) ? 0 : b;
  src_s[k] = (int32_t)a;
  dst_s[k] = (int32_t)b;
  key2[k] = (k < rowptr[N]) ? src[e] : (int64_t)-1;  // dropped rows take no part in the source-side CSR either
}
// out[0] = max in-degree, out[1] = max out-degree (one atomicMax per wave)
__global__ void k_topo_maxdeg(const int32_t* __restrict__ rp0, const int32_t* __restrict__ rp1, long N, int* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int d0 = 0, d1 = 0;
  if (i < N) {
    d0 = rp0[i + 1] - rp0[i];
    d1 = rp1[i + 1] - rp1[i];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    d0 = max(d0, __shfl_xor(d0, m));
    d1 = max(d1, __shfl_xor(d1, m));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&out[0], d0);  // integer maxima: order-independent
    atomicMax(&out[1], d1);
  }
}

size_t mgn_topology_workspace_bytes(int64_t E, int64_t N) {
  return 2 * mgn_csr_workspace_bytes(E, N) + (size_t)E * sizeof(int64_t) + 256;
}

static int topology_enqueue(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int32_t* rowptr_dst, int32_t* perm_dst,
                            int32_t* src_s, int32_t* dst_s, int32_t* rowptr_src, int32_t* perm_src, int* flags, void* ws, size_t ws_bytes,
                            hipStream_t s) {
  if (N < 0 || E < 0 || E > 2147483647LL || N > 2147483646LL) return fail(1, "mgn_topology_build: size out of int32 range");
  if (ws_bytes < mgn_topology_workspace_bytes(E, N)) return fail(1, "mgn_topology_build: workspace too small");
  const size_t csr_b = (mgn_csr_workspace_bytes(E, N) + 63) & ~(size_t)63;
  char* w = (char*)ws;
  int64_t* key2 = (int64_t*)(w + 2 * csr_b);
  if (hipMemsetAsync(flags, 0, 4 * sizeof(int), s) != hipSuccess) return fail(2, "mgn_topology_build: memset");
  if (int rc = csr_enqueue(dst, E, N, rowptr_dst, perm_dst, w, flags, s)) return rc;
  if (E > 0) {
    hipLaunchKernelGGL(k_csr_hist, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, src, (long)E, (long)N, (int*)(w + csr_b), flags);  // range check of src
    hipLaunchKernelGGL(k_topo_gather, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, src, dst, perm_dst, rowptr_dst, (long)E, (long)N,
                       src_s, dst_s, key2);
  }
  if (int rc = csr_enqueue(key2, E, N, rowptr_src, perm_src, w + csr_b, flags, s)) return rc;
  if (N > 0) hipLaunchKernelGGL(k_topo_maxdeg, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, rowptr_dst, rowptr_src, (long)N, flags + 1);
  return check_launch("mgn_topology_build");
}

int mgn_topology_build(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int32_t* rowptr_dst, int32_t* perm_dst,
                       int32_t* src_s, int32_t* dst_s, int32_t* rowptr_src, int32_t* perm_src, int32_t* max_degree_host,
                       void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (ws_bytes < mgn_topology_workspace_bytes(E < 0 ? 0 : E, N < 0 ? 0 : N)) return fail(1, "mgn_topology_build: workspace too small");
  const size_t csr_b = (mgn_csr_workspace_bytes(E, N) + 63) & ~(size_t)63;
  int* flags = (int*)((char*)ws + 2 * csr_b + (size_t)E * sizeof(int64_t));  // [0] err, [1] max in-degree, [2] max out-degree
  if (int rc = topology_enqueue(src, dst, E, N, rowptr_dst, perm_dst, src_s, dst_s, rowptr_src, perm_src, flags, ws, ws_bytes, s)) return rc;
  int h[3] = {0, 0, 0};
  if (hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess) return fail(2, "mgn_topology_build: memcpy");
  if (hipStreamSynchronize(s) != hipSuccess) return fail(2, "mgn_topology_build: sync failed");
  if (h[0]) return fail(3, "mgn_topology_build: edge index outside [0, N)");
  if (max_degree_host != nullptr) {
    max_degree_host[0] = h[1];
    max_degree_host[1] = h[2];
  }
  return 0;
}

int mgn_topology_build_async(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int32_t* rowptr_dst, int32_t* perm_dst,
                             int32_t* src_s, int32_t* dst_s, int32_t* rowptr_src, int32_t* perm_src, int32_t* flags_dev,
                             void* ws, size_t ws_bytes, void* stream) {
  if (flags_dev == nullptr) return fail(1, "mgn_topology_build_async: flags_dev is required");
  return topology_enqueue(src, dst, E, N, rowptr_dst, perm_dst, src_s, dst_s, rowptr_src, perm_src, flags_dev, ws, ws_bytes, (hipStream_t)stream);
}

#define FZ_GRID 256  /* one workgroup per CU (the accumulators take the whole register file) */
size_t mgn_edge_bwd_fused_workspace_bytes(void) { return (size_t)FZ_GRID * FZ_PART_FLOATS * sizeof(float); }

int mgn_edge_bwd_fused(const mgn_edge_bwd_fused_args* args, void* stream) {
  const mgn_edge_bwd_fused_args& a = *args;
  hipStream_t s = (hipStream_t)stream;
  if (a.M < 0 || a.dOut == nullptr || a.dAgg == nullptr || a.idx == nullptr || a.U == nullptr || a.R == nullptr || a.scale == nullptr ||
      a.dIn == nullptr)
    return fail(1, "mgn_edge_bwd_fused: missing operand");
  for (int l = 0; l < 4; ++l)
    if (a.X[l] == nullptr || a.wpk[l] == nullptr || (a.dW[l] != nullptr && a.ldw[l] < 128)) return fail(1, "mgn_edge_bwd_fused: missing layer operand");
  for (int l = 0; l < 3; ++l)
    if (a.Ms[l] == nullptr) return fail(1, "mgn_edge_bwd_fused: missing ReLU mask words");
  if (a.precision != 0 && a.precision != 1) return fail(1, "mgn_edge_bwd_fused: precision must be 0 or 1");
  if (a.ws == nullptr || a.ws_bytes < mgn_edge_bwd_fused_workspace_bytes()) return fail(1, "mgn_edge_bwd_fused: workspace too small");
  static thread_local bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)k_edge_bwd_fused<6>, hipFuncAttributeMaxDynamicSharedMemorySize, FZ_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_edge_bwd_fused<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FZ_LDS_BYTES) != hipSuccess)
      return fail(2, "mgn_edge_bwd_fused: cannot reserve LDS");
    attr_done = true;
  }
  const long ntiles = (a.M + 63) / 64;
  unsigned grid = (unsigned)(ntiles < FZ_GRID ? ntiles : FZ_GRID);
  if (grid == 0) grid = 1;  // M == 0: one workgroup writes zero partials
  if (a.precision == 1)
    hipLaunchKernelGGL((k_edge_bwd_fused<1>), dim3(grid), dim3(256), FZ_LDS_BYTES, s, a);
  else
    hipLaunchKernelGGL((k_edge_bwd_fused<6>), dim3(grid), dim3(256), FZ_LDS_BYTES, s, a);
  hipLaunchKernelGGL(k_fused_red, dim3((FZ_PART_FLOATS + 63) / 64), dim3(256), 0, s, a, (int)grid);
  return check_launch("mgn_edge_bwd_fused");
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

int mgn_seg_fix(const int32_t* rowptr, const float* part, float* out, int64_t N, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_seg_fix, dim3((unsigned)((N * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr, part, out, (long)N);
  return check_launch("mgn_seg_fix");
}

int mgn_segsum2(const float* src, const int32_t* rowptr0, const int32_t* perm0, float* out0, const int32_t* rowptr1,
                const int32_t* perm1, float* out1, int64_t N, int H, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) return 0;
  if (H != 128) return fail(1, "mgn_segsum2: H must be 128");
  const unsigned half = (unsigned)((N * (H / 4) + 255) / 256);
  hipLaunchKernelGGL(k_segsum2<8>, dim3(2 * half), dim3(256), 0, s, src, rowptr0, perm0, out0, rowptr1, perm1, out1, (long)N, half);
  return check_launch("mgn_segsum2");
}

int mgn_segsum2_b16(const uint16_t* src, const int32_t* rowptr0, const int32_t* perm0, float* out0, const int32_t* rowptr1,
                    const int32_t* perm1, float* out1, int64_t N, int H, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) return 0;
  if (H != 128) return fail(1, "mgn_segsum2_b16: H must be 128");
  const unsigned half = (unsigned)((N * (H / 4) + 255) / 256);
  hipLaunchKernelGGL((k_segsum2<8, true>), dim3(2 * half), dim3(256), 0, s, (const float*)src, rowptr0, perm0, out0, rowptr1, perm1, out1,
                     (long)N, half);
  return check_launch("mgn_segsum2_b16");
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
  if (a.precision < 0 || a.precision > 2) return fail(1, "mgn_mlp_fwd: precision must be 0 (fp32-grade), 1 (bf16) or 2 (bf16, two-byte saves)");
  if (a.precision == 2 && (!fwd_x6(a) || !plan_mlp(a.M, a.H, a.NL, fwd_ragged(a), false, a.act).lds || a.act != MGN_ACT_RELU))
    return fail(1, "mgn_mlp_fwd: precision 2 (two-byte saves) needs the packed split-bf16 path (H = 128, wpk[]) and ReLU");
  if (a.seg_out != nullptr && !(plan_mlp(a.M, a.H, a.NL, fwd_ragged(a), false).lds && fwd_x6(a)))
    return fail(1, "mgn_mlp_fwd: the fused segment sum needs the packed split-bf16 path (and seg_key / seg_rowptr / seg_part, no post-products)");
  if (a.out_relu && (a.scale != nullptr || a.resid != nullptr || a.wpk[0] != nullptr ||
                     plan_mlp(a.M, a.H, a.NL, fwd_ragged(a), false).lds))
    return fail(1, "mgn_mlp_fwd: out_relu is for a plain ragged-input launch (no norm / residual / packed path)");
  if (a.act != MGN_ACT_RELU && a.act != MGN_ACT_SILU && a.act != MGN_ACT_GELU) return fail(1, "mgn_mlp_fwd: act must be MGN_ACT_RELU, _SILU or _GELU");
  if (a.act == MGN_ACT_GELU && (a.wpk[0] != nullptr || a.precision != 0 || a.seg_out != nullptr || a.n_add > 0 || a.n_post > 0 || a.ldw0 > 0))
    return fail(1, "mgn_mlp_fwd: GELU runs on the generic kernels only (no packed weights / gathers / post-products)");
  if (a.act == MGN_ACT_SILU && plan_mlp(a.M, a.H, a.NL, fwd_ragged(a), false).lds && !fwd_x6(a) && a.NL > 1)
    return fail(1, "mgn_mlp_fwd: SiLU is not available on the exact-fp32 LDS generation (pass packed weights)");
  if (a.precision >= 1 && plan_mlp(a.M, a.H, a.NL, fwd_ragged(a), false, a.act).lds && !fwd_x6(a))
    return fail(1, "mgn_mlp_fwd: at H = 128 with full widths the bf16 matrix mode needs the packed split-bf16 path (wpk); other shapes run it on the generic kernels");
  if (a.nphase < 1 || a.nphase > MGN_MAX_PHASES) return fail(1, "mgn_mlp_fwd: nphase out of range");
  for (int p = 0; p < a.nphase; ++p)
    if (a.kw[p] < 1 || a.kw[p] > a.H) return fail(1, "mgn_mlp_fwd: phase width out of range");
  if (a.scale != nullptr && a.out_w != a.H) return fail(1, "mgn_mlp_fwd: RMSNorm needs out_w == H");
  if (a.n_add < 0 || a.n_add > 2 || a.n_post < 0 || a.n_post > 2) return fail(1, "mgn_mlp_fwd: n_add / n_post out of range");
  if ((a.n_add > 0 || a.n_post > 0 || a.ldw0 > 0) && !plan_mlp(a.M, a.H, a.NL, fwd_ragged(a), false).lds)
    return fail(1, "mgn_mlp_fwd: ldw0 / n_add / n_post need the H = 128 full-width kernel");
  if (a.M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  switch (a.H) {
    case 128: rc = launch_fwd<8>(a, s); break;
    case 64: rc = launch_fwd<4>(a, s); break;
    case 32: rc = launch_fwd<2>(a, s); break;
    default: rc = launch_fwd<1>(a, s); break;
  }
  if (rc) return fail(2, "mgn_mlp_fwd: cannot reserve LDS");
  return check_launch("mgn_mlp_fwd");
}


size_t mgn_mlp_bwd_workspace_bytes(int64_t M, int H, int NL) {
  // sized for the kernel with the most workgroups (64 rows each)
  return (size_t)((M + 63) / 64 + 1) * (NL + 1) * H * sizeof(float);
}


int mgn_mlp_bwd(const mgn_mlp_bwd_args* args, void* stream) {
  const mgn_mlp_bwd_args& a = *args;
  if (int rc = check_mlp_common(a.H, a.NL, a.out_w, "mgn_mlp_bwd")) return rc;
  if (a.n_din < 0 || a.n_din > MGN_MAX_PHASES) return fail(1, "mgn_mlp_bwd: n_din out of range");
  if (a.n_din > 1 && a.dZ[0] == nullptr) return fail(1, "mgn_mlp_bwd: n_din > 1 needs dZ[0]");
  if (a.precision < 0 || a.precision > 3) return fail(1, "mgn_mlp_bwd: precision must be 0 (fp32-grade), 1 (bf16), 2 (bf16, two-byte dZ[1..]) or 3 (dZ[0] too)");
  if (a.precision >= 2 && (!bwd_x6(a) || !plan_mlp(a.M, a.H, a.NL, a.out_w != a.H || a.n_din > 1, true).lds || a.act != MGN_ACT_RELU || a.n_front != 0))
    return fail(1, "mgn_mlp_bwd: precision 2 (two-byte dZ rows) needs the packed split-bf16 path (H = 128, wpk, Ms), ReLU and no front stage");
  if (a.act != MGN_ACT_RELU && a.act != MGN_ACT_SILU && a.act != MGN_ACT_GELU) return fail(1, "mgn_mlp_bwd: act must be MGN_ACT_RELU, _SILU or _GELU");
  if (a.act == MGN_ACT_GELU && (a.wpk[0] != nullptr || a.precision != 0 || a.seg_out != nullptr || a.n_front != 0))
    return fail(1, "mgn_mlp_bwd: GELU runs on the generic kernels only");
  if (a.act != MGN_ACT_RELU) {
    for (int l = 1; l < a.NL; ++l)
      if (a.Zs[l - 1] == nullptr) return fail(1, "mgn_mlp_bwd: SiLU / GELU need the saved pre-activations Zs[]");
  }
  if (a.act == MGN_ACT_SILU) {
    if (plan_mlp(a.M, a.H, a.NL, a.out_w != a.H || a.n_din > 1, true).lds && !bwd_x6(a))
      return fail(1, "mgn_mlp_bwd: SiLU is not available on the exact-fp32 LDS generation (pass packed weights)");
  }
  if (a.n_front != 0 && !(plan_mlp(a.M, a.H, a.NL, a.out_w != a.H || a.n_din > 1, true).lds && bwd_x6(a)))
    return fail(1, "mgn_mlp_bwd: the front stage needs the packed split-bf16 path (H = 128, full widths, wpk, Ms, no dOut2)");
  if (a.precision >= 1 && plan_mlp(a.M, a.H, a.NL, a.out_w != a.H || a.n_din > 1, true, a.act).lds && !bwd_x6(a))
    return fail(1, "mgn_mlp_bwd: at H = 128 with full widths the bf16 matrix mode needs the packed split-bf16 path (wpk, Ms); other shapes run it on the generic kernels");
  if (a.seg_out != nullptr && !(plan_mlp(a.M, a.H, a.NL, a.out_w != a.H || a.n_din > 1, true).lds && bwd_x6(a)))
    return fail(1, "mgn_mlp_bwd: the fused segment sum of dZ[0] needs the packed split-bf16 path (fp32-grade, ReLU, dZ[0], no front stage)");
  if (a.M == 0) return 0;
  if (a.red_ws_bytes < mgn_mlp_bwd_workspace_bytes(a.M, a.H, a.NL)) return fail(1, "mgn_mlp_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int lrc;
  switch (a.H) {
    case 128: lrc = launch_bwd<8>(a, s); break;
    case 64: lrc = launch_bwd<4>(a, s); break;
    case 32: lrc = launch_bwd<2>(a, s); break;
    default: lrc = launch_bwd<1>(a, s); break;
  }
  if (lrc) return fail(2, "mgn_mlp_bwd: cannot reserve LDS");
  if (int rc = check_launch("mgn_mlp_bwd")) return rc;
  // column reductions: db[l], dscale
  const unsigned grid = plan_mlp(a.M, a.H, a.NL, a.out_w != a.H || a.n_din > 1, true, a.act).grid;
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
  if (!any || a.defer_reduce) return 0;  // deferred: the caller finishes with mgn_colred_batch
  const int n = nslot * a.H;
  hipLaunchKernelGGL(k_colred, dim3((n + 31) / 32), dim3(1024), 0, s, (const float*)a.red_ws, (int)grid, nslot, a.H, outs);
  return check_launch("mgn_mlp_bwd/colred");
}

int mgn_colred_batch(int n, const mgn_colred_job* jobs, void* stream) {
  if (n < 0) return fail(1, "mgn_colred_batch: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += CR_MAX) {
    ColredBatch B;
    const int m = (n - i0 < CR_MAX) ? n - i0 : CR_MAX;
    int maxcol = 0;
    for (int i = 0; i < m; ++i) {
      const mgn_colred_job& q = jobs[i0 + i];
      if (q.red_ws == nullptr || q.NL < 1 || q.NL > MGN_MAX_LAYERS) return fail(1, "mgn_colred_batch: bad job");
      ColredJob& J = B.j[i];
      J.ws = (const float*)q.red_ws;
      J.nblocks = (int)plan_mlp(q.M, q.H, q.NL, q.out_w != q.H || q.n_din > 1, true).grid;
      J.nslot = q.NL + 1;
      J.H = q.H;
      for (int l = 0; l <= MGN_MAX_LAYERS; ++l) J.outs.o[l] = nullptr;
      for (int l = 0; l < q.NL; ++l) J.outs.o[l] = q.db[l];
      J.outs.o[q.NL] = q.dscale;
      if (J.nslot * J.H > maxcol) maxcol = J.nslot * J.H;
    }
    hipLaunchKernelGGL(k_colred_batch, dim3((maxcol + 31) / 32, m), dim3(1024), 0, s, B);
  }
  return check_launch("mgn_colred_batch");
}

static bool wgrad_job_row64(const mgn_wgrad_job& j) {
  return j.nja <= 4 && j.nkb <= 4 && j.kw <= 64 && (j.kw & 3) == 0 && (j.lda & 3) == 0 && (j.ldb & 3) == 0 &&
         (((uintptr_t)j.A | (uintptr_t)j.B) & 15) == 0;
}
static int wgrad_plan(int njobs, const mgn_wgrad_job* jobs, int* wg0, bool lds, int lds_budget = 512) {
  // A fixed budget of workgroups -- at most what is co-resident (512 = 2 per CU for the LDS
  // kernel; the generic one could hold 1024) -- shared out in proportion to the rows of each job.
  // The total must NEVER exceed the budget: four workgroups too many start a second
  // scheduling round and the kernel takes 1.4x as long (measured).  Floor shares first, the
  // remainder goes to the jobs with the most tiles per workgroup.
  const int rows = lds ? WG_TILE_ROWS : 16;
  // (generic kernel: 1024 workgroups are co-resident, but half as many leave half the partials to k_wgrad_red -- neutral on the
  //  headline, -3.5 % on the Transformer step whose weight gradients are all small matrices; MGN_WGRAD_GENERIC_BUDGET overrides)
  int budget = lds ? lds_budget : 512;
  if (!lds) {
    if (const char* e = getenv("MGN_WGRAD_GENERIC_BUDGET")) {
      const int b = atoi(e);
      if (b >= 64 && b <= 1024) budget = b;
    }
  }
  int64_t tiles[MGN_MAX_WGRAD_JOBS], tot = 0;
  int n[MGN_MAX_WGRAD_JOBS];
  for (int j = 0; j < njobs; ++j) {
    tiles[j] = (jobs[j].M + rows - 1) / rows;
    if (tiles[j] < 1) tiles[j] = 1;
    tot += tiles[j];
  }
  int used = 0;
  for (int j = 0; j < njobs; ++j) {
    int64_t s = tiles[j] * budget / tot;  // floor
    const int64_t cap = (tiles[j] + 3) / 4;  // at least 4 tiles per workgroup
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    n[j] = (int)s;
    used += n[j];
  }
  while (used < budget) {  // hand out what is left, one at a time, to the most loaded job
    int best = -1;
    double load = 4.0;  // do not go below 4 tiles per workgroup
    for (int j = 0; j < njobs; ++j) {
      const double l = (double)tiles[j] / n[j];
      if (l > load) {
        load = l;
        best = j;
      }
    }
    if (best < 0) break;
    ++n[best];
    ++used;
  }
  int acc = 0;
  for (int j = 0; j < njobs; ++j) {
    wg0[j] = acc;
    acc += n[j];
  }
  wg0[njobs] = acc;
  return acc;
}

size_t mgn_wgrad_workspace_bytes(int njobs, const mgn_wgrad_job* jobs) {
  if (njobs < 1 || njobs > MGN_MAX_WGRAD_JOBS) return 0;
  // upper bound over the two launches (LDS-staged + generic)
  return (size_t)(512 + 1024 + 2 * MGN_MAX_WGRAD_JOBS) * (128 * 128 + 128) * sizeof(float);
}

static bool wgrad_job_full(const mgn_wgrad_job& j) {
  // (ldb == -128: B holds the forward's two-byte saves, split-bf16 kernel in the bf16 matrix mode only -- checked by mgn_wgrad_p)
  return j.nja == 8 && j.nkb == 8 && (j.lda == 128 || j.lda == -128) && (j.ldb == 128 || j.ldb == -128) && j.kw == 128 && j.M >= 1 && getenv("MGN_NO_LDS") == nullptr;
}

int mgn_wgrad(int njobs, const mgn_wgrad_job* jobs, void* ws, size_t ws_bytes, void* stream) {
  return mgn_wgrad_p(njobs, jobs, ws, ws_bytes, 0, stream);
}

int mgn_wgrad_p(int njobs, const mgn_wgrad_job* jobs, void* ws, size_t ws_bytes, int precision, void* stream) {
  if (precision != 0 && precision != 1) return fail(1, "mgn_wgrad: precision must be 0 (fp32-grade) or 1 (bf16)");
  if (njobs < 1 || njobs > MGN_MAX_WGRAD_JOBS) return fail(1, "mgn_wgrad: njobs out of range");
  for (int j = 0; j < njobs; ++j) {   // two-byte operands: only what k_wgrad_x6<1> reads
    if (jobs[j].lda < 0 || jobs[j].ldb < 0) {
      // full 128 x 128 jobs (k_wgrad_x6<1>) or, [r5], jobs of the row-vector kernel (k_wgrad_row64<true>: up to 64 x 64)
      const bool r64 = !wgrad_job_full(jobs[j]) && wgrad_job_row64(jobs[j]) && getenv("MGN_WGRAD_NO_ROW64") == nullptr &&
                       ((((uintptr_t)jobs[j].A | (uintptr_t)jobs[j].B) & 15) == 0);
      if (precision != 1 || !(wgrad_job_full(jobs[j]) || r64) || getenv("MGN_FP32_MFMA") != nullptr)
        return fail(1, "mgn_wgrad: a negative leading dimension (bf16 rows) needs precision 1 and a full 128 x 128 job with ld = -128 or a job of the row-vector kernel");
    }
  }
  hipStream_t s = (hipStream_t)stream;
  // two launches at most: full 128x128 jobs on the LDS-staged kernel, the others generic
  size_t ws_off = 0;
  for (int pass = 0; pass < 2; ++pass) {
    WgradLaunch L;
    L.njobs = 0;
    int maxb = 1;
    for (int j = 0; j < njobs; ++j) {
      if (jobs[j].nja < 1 || jobs[j].nkb < 1 || jobs[j].nja > 8 || jobs[j].nkb > 8) return fail(1, "mgn_wgrad: block counts out of range");
      if (wgrad_job_full(jobs[j]) != (pass == 0)) continue;
      L.job[L.njobs++] = jobs[j];
      if (jobs[j].nja > maxb) maxb = jobs[j].nja;
      if (jobs[j].nkb > maxb) maxb = jobs[j].nkb;
    }
    if (L.njobs == 0) continue;
    int HB = maxb <= 1 ? 1 : maxb <= 2 ? 2 : maxb <= 4 ? 4 : 8;
    // jobs up to 64 x 64 with 16-byte addressable rows: the row-vector kernel (MGN_WGRAD_NO_ROW64: the generic one, for A/B)
    bool row64 = pass == 1 && HB <= 4 && getenv("MGN_WGRAD_NO_ROW64") == nullptr;
    for (int j = 0; row64 && j < L.njobs; ++j) row64 = wgrad_job_row64(L.job[j]);
    if (row64) HB = 4;
    if (!row64 && pass == 1)   // (only k_wgrad_x6<1> and k_wgrad_row64<true> read two-byte rows)
      for (int j = 0; j < L.njobs; ++j)
        if (L.job[j].lda < 0 || L.job[j].ldb < 0) return fail(1, "mgn_wgrad: bf16-row jobs below 128 x 128 need every job of the launch on the row-vector kernel");
    L.H = 16 * HB;
    // [r5] fp32-row full jobs: the producer / consumer kernel (one 512-thread workgroup per CU); MGN_WGRAD_PC=0: k_wgrad_x6<6>
    bool pc = pass == 0 && precision == 0 && getenv("MGN_FP32_MFMA") == nullptr;
    if (pc) {
      const char* e = getenv("MGN_WGRAD_PC");
      pc = e == nullptr || atoi(e) != 0;
      for (int j = 0; pc && j < L.njobs; ++j) pc = L.job[j].lda == 128 && L.job[j].ldb == 128;
    }
    int total = wgrad_plan(L.njobs, L.job, L.wg0, pass == 0, pc ? 256 : 512);
    L.interleave = 0;
    if (row64 && L.njobs >= 2 && L.njobs <= 8 && getenv("MGN_WGRAD_NO_INTERLEAVE") == nullptr) {
      bool same = true;
      for (int j = 1; j < L.njobs; ++j) same = same && L.job[j].M == L.job[0].M;
      const int64_t tiles = (L.job[0].M + 15) / 16;
      int per = (512 / L.njobs) & ~7;                       // workgroups per job: a multiple of 8, at most the usual budget together
      while (per > 8 && (int64_t)per * 4 > tiles) per -= 8;   // (at least ~4 tiles per workgroup)
      if (same && per >= 8 && tiles >= 4 * per) {
        for (int j = 0; j <= L.njobs; ++j) L.wg0[j] = j * per;
        L.interleave = per;
        total = L.njobs * per;
      }
    }
    const size_t need = (size_t)total * (L.H * L.H + L.H) * sizeof(float);
    if (ws_bytes < ws_off + need) return fail(1, "mgn_wgrad: workspace too small");
    L.partial = (float*)((char*)ws + ws_off);
    ws_off += need;
    if (pc) {
      static thread_local bool attr_pc_done = false;
      if (!attr_pc_done) {
        if (hipFuncSetAttribute((const void*)k_wgrad_pc, hipFuncAttributeMaxDynamicSharedMemorySize, WPC_LDS_BYTES) != hipSuccess)
          return fail(2, "mgn_wgrad: cannot reserve LDS");
        attr_pc_done = true;
      }
      hipLaunchKernelGGL(k_wgrad_pc, dim3(total), dim3(512), WPC_LDS_BYTES, s, L);
    } else if (pass == 0) {
      const size_t smem = 4 * WG_TILE_BYTES;
      static thread_local bool attr_done = false;
      if (!attr_done) {
        if (hipFuncSetAttribute((const void*)k_wgrad_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
          return fail(2, "mgn_wgrad: cannot reserve LDS");
        attr_done = true;
      }
      if (getenv("MGN_FP32_MFMA") == nullptr) {
        static thread_local bool attr6_done = false;
        if (!attr6_done) {
          if (hipFuncSetAttribute((const void*)k_wgrad_x6<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
              hipFuncSetAttribute((const void*)k_wgrad_x6<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return fail(2, "mgn_wgrad: cannot reserve LDS");
          attr6_done = true;
        }
        if (precision == 1)
          hipLaunchKernelGGL(k_wgrad_x6<1>, dim3(total), dim3(256), smem, s, L);
        else
          hipLaunchKernelGGL(k_wgrad_x6<6>, dim3(total), dim3(256), smem, s, L);
      } else {
        hipLaunchKernelGGL(k_wgrad_lds, dim3(total), dim3(256), smem, s, L);
      }
    } else if (row64) {
      if (precision == 1) {
        // two-byte operand rows: one pattern per launch (every job's A, every job's B alike)
        const bool a16 = L.job[0].lda < 0, b16 = L.job[0].ldb < 0;
        for (int j = 1; j < L.njobs; ++j)
          if ((L.job[j].lda < 0) != a16 || (L.job[j].ldb < 0) != b16) return fail(1, "mgn_wgrad: the jobs of a row-vector launch must agree on which operand is bf16 rows");
        if (a16 && b16) hipLaunchKernelGGL((k_wgrad_row64<true, true, true>), dim3(total), dim3(256), 0, s, L);
        else if (a16) hipLaunchKernelGGL((k_wgrad_row64<true, true, false>), dim3(total), dim3(256), 0, s, L);
        else if (b16) hipLaunchKernelGGL((k_wgrad_row64<true, false, true>), dim3(total), dim3(256), 0, s, L);
        else hipLaunchKernelGGL((k_wgrad_row64<true>), dim3(total), dim3(256), 0, s, L);
      }
      else if (getenv("MGN_FP32_MFMA") == nullptr && getenv("MGN_WGRAD_ROW64_EXACT") == nullptr) {   // [r5] fp32-grade on the split-bf16 matrix path
        bool full = true;
        for (int j = 0; full && j < L.njobs; ++j) full = L.job[j].nja == 4 && L.job[j].kw == 64;
        if (full)
          hipLaunchKernelGGL(k_wgrad_row64x6<true>, dim3(total), dim3(256), 0, s, L);
        else
          hipLaunchKernelGGL(k_wgrad_row64x6<false>, dim3(total), dim3(256), 0, s, L);
      }
      else
        hipLaunchKernelGGL(k_wgrad_row64<false>, dim3(total), dim3(256), 0, s, L);
    } else {
      switch (HB) {
        case 8: hipLaunchKernelGGL(k_wgrad<8>, dim3(total), dim3(256), 0, s, L); break;
        case 4: hipLaunchKernelGGL(k_wgrad<4>, dim3(total), dim3(256), 0, s, L); break;
        case 2: hipLaunchKernelGGL(k_wgrad<2>, dim3(total), dim3(256), 0, s, L); break;
        default: hipLaunchKernelGGL(k_wgrad<1>, dim3(total), dim3(256), 0, s, L); break;
      }
    }
    if (int rc = check_launch("mgn_wgrad")) return rc;
    int max_nwg = 0;
    for (int j = 0; j < L.njobs; ++j) max_nwg = (L.wg0[j + 1] - L.wg0[j] > max_nwg) ? L.wg0[j + 1] - L.wg0[j] : max_nwg;
    if (max_nwg <= 64)
      hipLaunchKernelGGL(k_wgrad_red<4>, dim3((L.H * L.H + L.H + 63) / 64, L.njobs), dim3(64 * 4), 0, s, L);
    else
      hipLaunchKernelGGL(k_wgrad_red<16>, dim3((L.H * L.H + L.H + 63) / 64, L.njobs), dim3(64 * 16), 0, s, L);
    if (int rc = check_launch("mgn_wgrad/reduce")) return rc;
  }
  return 0;
}

// Resident workgroups per CU of the persistent kernels at their launch configuration (HIP occupancy
// query): what the 512-workgroup plans assume.  out[0..5] = fwd_x6<6,4>, fwd_x6<1,4>, bwd_x6<6> with one
// column-sum slot, bwd_x6<6> with five (the round-1 footprint), wgrad_x6<6>, fwd_x6<6,8>.
int mgn_debug_occupancy(int* out) {
  int n = 0;
  if (hipFuncSetAttribute((const void*)k_mlp_fwd_x6<6, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_FWD_LDS_BYTES(4)) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_mlp_fwd_x6<1, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_FWD_LDS_BYTES(4)) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_mlp_fwd_x6<6, 8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_FWD_LDS_BYTES(8)) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_mlp_bwd_x6<6, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_BWD_LDS_BYTES(LDS_MAX_NL + 1)) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_mlp_bwd_x6<1, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_BWD_LDS_BYTES(LDS_MAX_NL + 1)) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_wgrad_x6<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WG_TILE_BYTES) != hipSuccess) return 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mlp_fwd_x6<6, 4, 0>, 256, X6_FWD_LDS_BYTES(4)) != hipSuccess) return 2;
  out[0] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mlp_fwd_x6<1, 4, 0>, 256, X6_FWD_LDS_BYTES(4)) != hipSuccess) return 2;
  out[1] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mlp_bwd_x6<6, false, 0>, 256, X6_BWD_LDS_BYTES(1)) != hipSuccess) return 2;
  out[2] = n;  // one column-sum slot (dscale only: what the processor launches)
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mlp_bwd_x6<6, false, 0>, 256, X6_BWD_LDS_BYTES(5)) != hipSuccess) return 2;
  out[3] = n;  // all five slots (the round-1 footprint)
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wgrad_x6<6>, 256, 4 * WG_TILE_BYTES) != hipSuccess) return 2;
  out[4] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mlp_fwd_x6<6, 8, 0>, 512, X6_FWD_LDS_BYTES(8)) != hipSuccess) return 2;
  out[5] = n;
  return 0;
}

int mgn_wpack(int n, const mgn_wpack_block* blocks, void* stream) {
  if (n < 0) return fail(1, "mgn_wpack: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += TB_MAX) {
    WpackLaunch T;
    const int m = (n - i0 < TB_MAX) ? n - i0 : TB_MAX;
    for (int i = 0; i < m; ++i) {
      T.b[i] = blocks[i0 + i];
      if (T.b[i].src == nullptr || T.b[i].dst == nullptr || ((size_t)T.b[i].dst & 15) != 0) return fail(1, "mgn_wpack: null / misaligned block");
    }
    hipLaunchKernelGGL(k_wpack, dim3(8, m), dim3(256), 0, s, T);
  }
  return check_launch("mgn_wpack");
}

int mgn_transpose_blocks(int n, const mgn_tblock* blocks, int H, void* stream) {
  if (n < 0 || H < 1 || H > 128) return fail(1, "mgn_transpose_blocks: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int tiles = ((H + 31) / 32) * ((H + 31) / 32);
  for (int i0 = 0; i0 < n; i0 += TB_MAX) {
    TBlocks T;
    const int m = (n - i0 < TB_MAX) ? n - i0 : TB_MAX;
    for (int i = 0; i < m; ++i) T.b[i] = blocks[i0 + i];
    hipLaunchKernelGGL(k_transpose_blocks, dim3(tiles, m), dim3(256), 0, s, T, H);
  }
  return check_launch("mgn_transpose_blocks");
}

}  // extern "C"
