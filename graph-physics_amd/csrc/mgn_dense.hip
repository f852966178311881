// MI355X (gfx950) dense row kernels of the sparse-attention Transformer processor (BASELINE.json
// configs[4]): everything of a Transformer block that is not the sparse attention itself --
//     x + proj(attn(q, k, v)),  q/k/v = Linear(norm1(x));     x + W3 (act(W1 n + b1) * (W2 n + b2)) + b3,  n = norm2(x)
// (graphphysics/models/layers.py:564-697 Attention, :700-819 Transformer, :213-278 GatedMLP / build_gated_mlp,
// :73-129 RMSNorm) -- as fused launches: the RMSNorm is a PROLOGUE of the product that consumes it (the row is in
// registers anyway), the activation / gated product / bias / residual its EPILOGUE; a concatenated input
// (TemporalAttention's cat([h_pred, h_prev]), layers.py:858-887) is two input phases and is never materialised.
//
// Arithmetic: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) in the T-layout of the generic MLP kernels: a wave owns 16
// rows, lane (c,g) holds features 16 kb + 4g + {0..3} of row c -- the B operand of four K = 4 steps -- and the weight
// fragment W[16 ob + c][16 kb + 4g + r] (one 16-byte load per lane, L2-resident: a block's matrices are 16-150 KB) is the A
// operand; the accumulator comes out as output features 16 ob + 4g + {0..3} of row c = one 16-byte store.  Node-row
// work at hidden 64 (configs[4]): 6 products of 2 x 64 x 64..192 flop per row against ~1.5 KB of row traffic --
// HBM-bound, so the exact-fp32 rate (157 TF/s) is not what limits it.  precision = 1 rounds the operands and each
// result to bf16 (the reference under Lightning bf16-mixed: autocast runs nn.Linear in bf16, train.py:74-78; the
// sparse attention itself stays fp32, layers.py:49-70).
// Fourth translation unit of libmgn_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mgn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_d __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_d __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA_BF16(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_d, (a)), __builtin_bit_cast(bf16x8_d, (b)), (c), 0, 0, 0)

static thread_local char g_derr[256] = "";
extern "C" const char* mgn_dense_last_error(void) { return g_derr; }
static int dfail(int code, const char* msg) {
  snprintf(g_derr, sizeof(g_derr), "%s", msg);
  return code;
}
static int dcheck(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_derr, sizeof(g_derr), "%s: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// round to nearest-even bf16, returned as fp32 (v_cvt_pk_bf16_f32: the hardware conversion)
typedef float f32x2_d __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bf16r2(float a, float b, float& ra, float& rb) {
  const f32x2_d v = {a, b};
  const unsigned q = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_d));
  ra = __uint_as_float(q << 16);
  rb = __uint_as_float(q & 0xffff0000u);
}
// two T-layout blocks (features 16 kb + 4g + r and 16 (kb + 1) + 4g + r of a row / of a weight row) -> the 8 bf16 of one
// K = 32 MFMA operand; A (weights) and B (rows) use the same element order, so the contraction pairs feature with feature
__device__ __forceinline__ u32x4_d pack_bf16x8(const f32x4& lo, const f32x4& hi) {
  u32x4_d o;
  const f32x2_d a = {lo[0], lo[1]}, b = {lo[2], lo[3]}, c_ = {hi[0], hi[1]}, d = {hi[2], hi[3]};
  o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_d));
  o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_d));
  o[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(c_, bf16x2_d));
  o[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2_d));
  return o;
}
__device__ __forceinline__ uint2 pack_bf16x4(const f32x4& v) {   // exact for bf16-valued inputs (round to nearest even otherwise)
  const f32x2_d a = {v[0], v[1]}, b = {v[2], v[3]};
  return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_d)), __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_d)));
}
__device__ __forceinline__ f32x4 unpack_bf16x4(uint2 t) {
  return f32x4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)};
}
__device__ __forceinline__ float bf16r(float v) {
  float a, b;
  bf16r2(v, v, a, b);
  return a;
}
__device__ __forceinline__ f32x4 bf16r4(f32x4 v) {
  f32x4 o;
  float a, b;
  bf16r2(v[0], v[1], a, b);
  o[0] = a, o[1] = b;
  bf16r2(v[2], v[3], a, b);
  o[2] = a, o[3] = b;
  return o;
}
__device__ __forceinline__ float d_silu(float z) { return z / (1.0f + expf(-z)); }
__device__ __forceinline__ float d_dsilu(float z) {
  const float sg = 1.0f / (1.0f + expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}
__device__ __forceinline__ float d_gelu(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float d_dgelu(float z) {
  return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * expf(-0.5f * z * z);
}
__device__ __forceinline__ float d_act(float z, int act) {
  return act == MGN_ACT_RELU ? fmaxf(z, 0.f) : act == MGN_ACT_SILU ? d_silu(z) : act == MGN_ACT_GELU ? d_gelu(z) : z;
}
__device__ __forceinline__ float d_dact(float z, int act) {
  return act == MGN_ACT_RELU ? (z > 0.f ? 1.f : 0.f) : act == MGN_ACT_SILU ? d_dsilu(z) : act == MGN_ACT_GELU ? d_dgelu(z) : 1.f;
}
__device__ __forceinline__ float rowsum4d(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// ------------------------------------------------------------------ fused linear layer
// KB = input blocks of 16 features (all phases together); 64 rows per workgroup, 16 per wave.  BF (bf16 operand / result
// rounding) and GATE (second product) are template parameters: as run-time tests they put a branch behind every weight fragment
// load and the compiler answered with one `s_waitcnt vmcnt(0)` per fragment -- 16 serialised L2 round trips per row tile (seen
// in the ISA; 64 -> 64 at 150 000 rows: 35 us).  Straight-line now: all fragments of an output block are requested together,
// and (up to 8 input blocks) those of the NEXT output block before this block's MFMAs.
// LDSW: the weight matrix (both of a gated product) is staged ONCE per workgroup into LDS (rows padded by 16 bytes: the 16 rows a
// fragment read touches then start 17 chunks apart, 4-5 LDS cycles per ds_read_b128) and persistent workgroups walk the row
// tiles -- without it every wave re-reads the matrix from L1 / L2 next to its row traffic (64 -> 64 at 600 000 rows: 124 us, of
// which 40 us are those loads; timing-only builds).  Used when the image fits 64 KB (3+ workgroups per CU).
// [ob0, ob1): the output blocks of this launch.  An LDSW launch stages only their slice of the matrices, so a product whose padded image
// exceeds 64 KB (the gated 64 -> 192 pair: 104 KB) runs as 2 or 3 launches with an LDS-resident slice each (the rows are re-read per
// launch: L2 / Infinity-Cache hits for the most part; the prologue's side outputs are written by the first).  As an inner loop over
// slices the same thing cost every instance 10-16 registers (occupancy 7 -> 5 waves per SIMD, 192 -> 64: 54 -> 72 us).
template <int KB, bool BF, bool GATE, bool LDSW>
__global__ void __launch_bounds__(256) k_linear(const mgn_linear_args a, const int ob0, const int ob1) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  constexpr int SP = 16 * KB + 4;   // padded row stride of the LDS image (floats)
  extern __shared__ __attribute__((aligned(16))) float lw_[];
  const int NC = 16 * (ob1 - ob0);   // matrix rows of this launch
  if (LDSW) {
    const int nvec = NC * (KB * 4);    // f32x4 elements of one matrix slice
    for (int i = threadIdx.x; i < nvec; i += 256) {
      const int n = i / (KB * 4), j = i % (KB * 4);
      *(f32x4*)(lw_ + n * SP + 4 * j) = *(const f32x4*)(a.W + (size_t)(16 * ob0 + n) * a.ldw + 4 * j);
      if (GATE) *(f32x4*)(lw_ + (NC + n) * SP + 4 * j) = *(const f32x4*)(a.W2 + (size_t)(16 * ob0 + n) * a.ldw + 4 * j);
    }
    __syncthreads();
  }
  const long ntiles = (a.M + 63) / 64;
  const int K = 16 * KB, kb1 = a.K1 >> 4, kb2 = kb1 + (a.K2 >> 4);
  // rows of the (up to three) input phases; a phase with an index row is GATHERED (x[dst], x[src] of an edge row: the
  // concatenation cat[e, x_i, x_j] of GraphNetBlock.edge_update, layers.py:1044-1060, is never materialised)
  auto load_rows = [&](long tl, f32x4 (&dst)[KB]) {
    const long m_ = (tl * 4 + wv) * 16 + c;
    const long mm_ = m_ < a.M ? m_ : a.M - 1;
    const long r1 = a.idx != nullptr ? (long)a.idx[mm_] : mm_;
    const long r2 = a.idx2 != nullptr ? (long)a.idx2[mm_] : mm_;
    const long r3 = a.idx3 != nullptr ? (long)a.idx3[mm_] : mm_;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
      dst[kb] = (kb < kb1)   ? *(const f32x4*)(a.x + r1 * a.ldx + 16 * kb + 4 * g)
                : (kb < kb2) ? *(const f32x4*)(a.x2 + r2 * a.ldx2 + 16 * (kb - kb1) + 4 * g)
                             : *(const f32x4*)(a.x3 + r3 * a.ldx3 + 16 * (kb - kb2) + 4 * g);
  };
  // (requesting the NEXT tile's rows before this tile's products -- one more register set -- measured slower: 64 -> 64 23.8 -> 26.7 us,
  //  192 -> 64 54 -> 71 us at 150 000 rows; the launches keep 3-8 workgroups per CU and the other waves cover the latency)
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const long m = (tile * 4 + wv) * 16 + c;
  const bool valid = m < a.M;
  const long mm = valid ? m : a.M - 1;
  f32x4 in[KB];
  load_rows(tile, in);
  if (a.norm_scale != nullptr) {  // RMSNorm prologue, reference epsilon placement: scale * x / (||x|| / sqrt(K) + eps)
    float ss = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ss = fmaf(in[kb][r], in[kb][r], ss);
    ss = rowsum4d(ss);
    const float inv = 1.0f / (sqrtf(ss) / sqrtf((float)K) + a.eps);
    if (a.inv_out != nullptr && valid && g == 0 && ob0 == 0) a.inv_out[mm] = inv;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) in[kb] = *(const f32x4*)(a.norm_scale + 16 * kb + 4 * g) * (in[kb] * inv);
    if (a.n_out != nullptr && valid) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) *(f32x4*)(a.n_out + mm * K + 16 * kb + 4 * g) = in[kb];
    }
  }
  // [r4] bf16 matrix mode on the bf16 MATRIX pipe: with an even number of input blocks a product is KB / 2 MFMAs of
  // 16x16x32 (bf16 operands packed from the same registers, fp32 accumulate) instead of 4 KB exact-fp32 MFMAs of twice the
  // issue time each on operands rounded by three VALU instructions per pair -- the mode was SLOWER than fp32 before
  // (c5: 22.4 vs 21.0 ms per training step).  Same operand values (round to nearest even), fp32 accumulation.
  constexpr bool BFM = BF && (KB % 2 == 0);
  u32x4_d xb[BFM ? KB / 2 : 1];
  if (BFM) {
#pragma unroll
    for (int s_ = 0; s_ < KB / 2; ++s_) xb[s_] = pack_bf16x8(in[2 * s_], in[2 * s_ + 1]);
  } else if (BF) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) in[kb] = bf16r4(in[kb]);
  }
  const int NB = ob1, OB0 = ob0;
  constexpr bool PF = KB <= 8 && !LDSW;   // prefetch the next output block's fragments from L2 (registers allow it)
  const float* w1 = a.W + (size_t)c * a.ldw + 4 * g;
  const float* w2 = GATE ? a.W2 + (size_t)c * a.ldw + 4 * g : nullptr;
  const float* l1 = lw_ + c * SP + 4 * g;
  const float* l2 = lw_ + (NC + c) * SP + 4 * g;
  f32x4 wa[KB], wb[GATE ? KB : 1];
  if (PF) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      wa[kb] = *(const f32x4*)(w1 + 16 * kb);
      if (GATE) wb[kb] = *(const f32x4*)(w2 + 16 * kb);
    }
  }
  for (int ob = OB0; ob < NB; ++ob) {
    const int n0 = 16 * ob + 4 * g;
    f32x4 acc = (a.b != nullptr) ? *(const f32x4*)(a.b + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc2 = (GATE && a.b2 != nullptr) ? *(const f32x4*)(a.b2 + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (BF) acc = bf16r4(acc), acc2 = bf16r4(acc2);
    f32x4 ca[KB], cb[GATE ? KB : 1];
    if (PF) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        ca[kb] = wa[kb];
        if (GATE) cb[kb] = wb[kb];
      }
      const size_t nxt = (size_t)16 * ((ob + 1 < NB) ? ob + 1 : ob) * a.ldw;   // (the last block re-requests its own: an unconditional load)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#ifndef LIN_EXP_NOW
        wa[kb] = *(const f32x4*)(w1 + nxt + 16 * kb);
        if (GATE) wb[kb] = *(const f32x4*)(w2 + nxt + 16 * kb);
#endif
      }
    } else if (LDSW) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        ca[kb] = *(const f32x4*)(l1 + 16 * (ob - OB0) * SP + 16 * kb);
        if (GATE) cb[kb] = *(const f32x4*)(l2 + 16 * (ob - OB0) * SP + 16 * kb);
      }
    } else {
      const size_t cur = (size_t)16 * ob * a.ldw;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        ca[kb] = *(const f32x4*)(w1 + cur + 16 * kb);
        if (GATE) cb[kb] = *(const f32x4*)(w2 + cur + 16 * kb);
      }
    }
    if (BFM) {
#pragma unroll
      for (int s_ = 0; s_ < KB / 2; ++s_) {
        acc = MFMA_BF16(pack_bf16x8(ca[2 * s_], ca[2 * s_ + 1]), xb[s_], acc);
        if (GATE) acc2 = MFMA_BF16(pack_bf16x8(cb[2 * s_], cb[2 * s_ + 1]), xb[s_], acc2);
      }
    } else {
    if (BF) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        ca[kb] = bf16r4(ca[kb]);
        if (GATE) cb[kb] = bf16r4(cb[kb]);
      }
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#ifdef LIN_EXP_NOMFMA   // timing experiment only
        acc[r] += ca[kb][r] * in[kb][r];
#else
        acc = MFMA16(ca[kb][r], in[kb][r], acc);
#endif
        if (GATE) acc2 = MFMA16(cb[kb][r], in[kb][r], acc2);
      }
    }
    }
    if (BF) acc = bf16r4(acc), acc2 = bf16r4(acc2);   // a bf16 nn.Linear returns bf16
    if (valid) {
      if (a.saveZ1 != nullptr) *(f32x4*)(a.saveZ1 + mm * a.N + n0) = acc;
      if (GATE && a.saveZ2 != nullptr) *(f32x4*)(a.saveZ2 + mm * a.N + n0) = acc2;
    }
    f32x4 y;
#pragma unroll
#ifdef LIN_EXP_NOACT   // timing experiment only: what the activation (erf / exp per element) costs
    for (int r = 0; r < 4; ++r) y[r] = acc[r];
#else
    for (int r = 0; r < 4; ++r) y[r] = d_act(acc[r], a.act);
#endif
    if (BF && a.act >= 0) y = bf16r4(y);
    if (GATE) {
      y = y * acc2;
      if (BF) y = bf16r4(y);
    }
    if (a.resid != nullptr) y = *(const f32x4*)(a.resid + mm * a.ldr + n0) + y;   // the residual stream stays fp32
#ifdef LIN_EXP_NOSTORE
    if (valid && y[0] == 12345.678f) *(f32x4*)(a.out + mm * a.ldo + n0) = y;
#else
    if (valid) *(f32x4*)(a.out + mm * a.ldo + n0) = y;
#endif
  }
  }  // tiles
}

// ------------------------------------------------------------------ [r5] the same launch on the split-bf16 matrix path
// precision 0 at row counts where the LDS-resident weight image pays: every fp32 operand as three bf16 pieces, a product as the six
// terms of order <= 2^-16 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- the arithmetic of the H = 128 chain kernels (DESIGN 4.1;
// 5.6e-7 against fp64 where the exact-fp32 MFMA gives 5.6e-7).  16x16x4_f32 issues at 32 cycles for 2 048 flop, a K = 32 slice costs
// 8 of them = 256 cycles; its six bf16 terms cost 96.  At configs[4]'s 150 000 rows the exact-fp32 products were 40 % of a launch's time.
//   * 512 threads: eight waves (16 rows each, 128 rows per tile) share ONE piece image of the weights, staged and split once per
//     persistent workgroup: row n = [piece][K-slice][g] 16-byte chunks (lane (c, g) of output block ob reads chunk (p, s, g) of row
//     16 ob + c = the A operand of slice s), rows padded by one chunk (odd chunk stride: the 16 rows of a fragment read spread
//     over the banks).  6 bytes per weight: 64 -> 192 = 77 KB, two workgroups per CU.
//   * a wave splits its rows once per tile (44 vector instructions per 8 values) and keeps the pieces as B operands for all output
//     blocks; per output block and slice: three ds_read_b128 and six MFMAs on two accumulators (small terms / large terms).
// Same prologue / epilogue as k_linear (RMSNorm, gathers, saves, activation, gated product, residual), same [ob0, ob1) slicing.
__device__ __forceinline__ void split3_bf16x8(const f32x4& lo, const f32x4& hi, u32x4_d& p1, u32x4_d& p2, u32x4_d& p3) {
  p1 = pack_bf16x8(lo, hi);
  f32x4 rl, rh;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    rl[2 * d] = lo[2 * d] - __uint_as_float(p1[d] << 16);
    rl[2 * d + 1] = lo[2 * d + 1] - __uint_as_float(p1[d] & 0xffff0000u);
    rh[2 * d] = hi[2 * d] - __uint_as_float(p1[2 + d] << 16);
    rh[2 * d + 1] = hi[2 * d + 1] - __uint_as_float(p1[2 + d] & 0xffff0000u);
  }
  p2 = pack_bf16x8(rl, rh);
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    rl[2 * d] -= __uint_as_float(p2[d] << 16);
    rl[2 * d + 1] -= __uint_as_float(p2[d] & 0xffff0000u);
    rh[2 * d] -= __uint_as_float(p2[2 + d] << 16);
    rh[2 * d + 1] -= __uint_as_float(p2[2 + d] & 0xffff0000u);
  }
  p3 = pack_bf16x8(rl, rh);
}
// BF (precision 1, the reference under bf16-mixed): ONE piece = the operands rounded to bf16, bias / result / activation roundings
// as in k_linear<.., BF = true>; the same 512-thread workgroup and image layout with a third of the bytes.
// (second launch bound = waves per SIMD: two workgroups per CU -> at most 128 registers up to 192 inputs)
template <int KB, bool GATE, bool BF, bool X16 = false>
__global__ void __launch_bounds__(512, (KB <= 12 ? 4 : 2)) k_linear_x6(const mgn_linear_args a, const int ob0, const int ob1) {
  static_assert(KB % 2 == 0, "K = 32 matrix steps");
  constexpr int S = KB / 2;               // K = 32 slices
  constexpr int NP = BF ? 1 : 3;          // pieces
  constexpr int RB = (4 * NP * S + 1) * 16;   // bytes of an image row: pieces x S slices x 4 lane groups, + one chunk of padding
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  extern __shared__ __attribute__((aligned(16))) char li_[];
  const int NC = 16 * (ob1 - ob0);
  constexpr int K_ = 16 * KB;
  const long ntiles = (a.M + 127) / 128;
  const int K = 16 * KB, kb1 = a.K1 >> 4, kb2 = kb1 + (a.K2 >> 4);
  const char* l1 = li_ + (size_t)c * RB + g * 16;
  const char* l2 = li_ + (size_t)(NC + c) * RB + g * 16;
  // (the first tile's rows requested BEFORE the staging: in[] then lives across the loop and the instances grow from 79 / 116 to
  //  107 / 205 registers -- one workgroup per CU instead of two; not done)
  if (a.w_transposed) {
    // the array handed over is [K][ldw]: element (image row n, input k) = W[k][16 ob0 + n].  An item = four consecutive rows n x the
    // eight inputs of one chunk (k = 32 s + 4 g + r and + 16): eight 16-byte loads ALONG n (consecutive lanes = consecutive quads of n:
    // coalesced), transposed in registers, then the ordinary split and three 16-byte chunks per row
    const int nq = NC / 4, items = nq * S * 4;
    for (int i = threadIdx.x; i < items * (GATE ? 2 : 1); i += 512) {
      const int mat = i >= items, j = mat ? i - items : i;
      const int q = j % nq, sg = j / nq;   // sg = 4 s + g
      const float* w = (mat ? a.W2 : a.W) + (size_t)(32 * (sg >> 2) + 4 * (sg & 3)) * a.ldw + 16 * ob0 + 4 * q;
      f32x4 v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = *(const f32x4*)(w + (size_t)r * a.ldw), v[4 + r] = *(const f32x4*)(w + (size_t)(16 + r) * a.ldw);
      char* d = li_ + (size_t)(mat * NC + 4 * q) * RB + sg * 16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 lo_ = {v[0][e], v[1][e], v[2][e], v[3][e]}, hi_ = {v[4][e], v[5][e], v[6][e], v[7][e]};
        u32x4_d p1, p2, p3;
        split3_bf16x8(lo_, hi_, p1, p2, p3);
        *(u32x4_d*)(d + (size_t)e * RB) = p1;
        if (!BF) {
          *(u32x4_d*)(d + (size_t)e * RB + S * 64) = p2;
          *(u32x4_d*)(d + (size_t)e * RB + 2 * S * 64) = p3;
        }
      }
    }
    __syncthreads();
  } else {
    const int items = NC * S * 4;   // (row n, slice s, lane group g): 8 weights -> three chunks
    for (int i = threadIdx.x; i < items * (GATE ? 2 : 1); i += 512) {
      const int mat = i >= items, j = mat ? i - items : i;
      const int n = j / (S * 4), s_ = (j / 4) % S, g_ = j & 3;
      const float* w = (mat ? a.W2 : a.W) + (size_t)(16 * ob0 + n) * a.ldw + 32 * s_ + 4 * g_;
      u32x4_d p1, p2, p3;
      split3_bf16x8(*(const f32x4*)w, *(const f32x4*)(w + 16), p1, p2, p3);
      char* d = li_ + (size_t)(mat * NC + n) * RB + (s_ * 4 + g_) * 16;
      *(u32x4_d*)d = p1;
      if (!BF) {
        *(u32x4_d*)(d + S * 64) = p2;
        *(u32x4_d*)(d + 2 * S * 64) = p3;
      }
    }
    __syncthreads();
  }
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long m = (tile * 8 + wv) * 16 + c;
    const bool valid = m < a.M;
    const long mm = valid ? m : a.M - 1;
    f32x4 in[X16 ? 1 : KB];
    if constexpr (!X16) {
      const long r1 = a.idx != nullptr ? (long)a.idx[mm] : mm;
      const long r2 = a.idx2 != nullptr ? (long)a.idx2[mm] : mm;
      const long r3 = a.idx3 != nullptr ? (long)a.idx3[mm] : mm;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        in[kb] = (kb < kb1)   ? *(const f32x4*)(a.x + r1 * a.ldx + 16 * kb + 4 * g)
                 : (kb < kb2) ? *(const f32x4*)(a.x2 + r2 * a.ldx2 + 16 * (kb - kb1) + 4 * g)
                              : *(const f32x4*)(a.x3 + r3 * a.ldx3 + 16 * (kb - kb2) + 4 * g);
    }
    if constexpr (!X16) {
    if (a.norm_scale_outer != nullptr) {  // a norm in front of the norm prologue: the same formula on the raw row first
      float ss = 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ss = fmaf(in[kb][r], in[kb][r], ss);
      ss = rowsum4d(ss);
      const float inv = 1.0f / (sqrtf(ss) / sqrtf((float)K) + a.eps);
      if (a.inv_outer_out != nullptr && valid && g == 0 && ob0 == 0) a.inv_outer_out[mm] = inv;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) in[kb] = *(const f32x4*)(a.norm_scale_outer + 16 * kb + 4 * g) * (in[kb] * inv);
    }
    if (a.norm_scale != nullptr) {  // RMSNorm prologue (as k_linear)
      float ss = 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ss = fmaf(in[kb][r], in[kb][r], ss);
      ss = rowsum4d(ss);
      const float inv = 1.0f / (sqrtf(ss) / sqrtf((float)K) + a.eps);
      if (a.inv_out != nullptr && valid && g == 0 && ob0 == 0) a.inv_out[mm] = inv;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) in[kb] = *(const f32x4*)(a.norm_scale + 16 * kb + 4 * g) * (in[kb] * inv);
      if (a.n_out != nullptr && valid && ob0 == 0) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) *(f32x4*)(a.n_out + mm * K + 16 * kb + 4 * g) = in[kb];
      }
    }
    }
    u32x4_d x1[S], x2[BF ? 1 : S], x3[BF ? 1 : S];
    if constexpr (X16) {   // two-byte input rows: the packed operand of a K = 32 slice IS two 8-byte pieces of the row (blocks 2s, 2s + 1)
      const uint16_t* xr = (const uint16_t*)a.x + mm * a.ldx + 4 * g;
#pragma unroll
      for (int s_ = 0; s_ < S; ++s_) {
        const uint2 lo_ = *(const uint2*)(xr + 32 * s_), hi_ = *(const uint2*)(xr + 32 * s_ + 16);
        x1[s_] = u32x4_d{lo_.x, lo_.y, hi_.x, hi_.y};
      }
    } else {
#pragma unroll
    for (int s_ = 0; s_ < S; ++s_) {
      if (BF) x1[s_] = pack_bf16x8(in[X16 ? 0 : 2 * s_], in[X16 ? 0 : 2 * s_ + 1]);
      else split3_bf16x8(in[X16 ? 0 : 2 * s_], in[X16 ? 0 : 2 * s_ + 1], x1[s_], x2[BF ? 0 : s_], x3[BF ? 0 : s_]);
    }
    }
    for (int ob = ob0; ob < ob1; ++ob) {
      const int n0 = 16 * ob + 4 * g;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      f32x4 hi = (a.b != nullptr) ? *(const f32x4*)(a.b + n0) : zero, lo = zero;
      f32x4 hi2 = (GATE && a.b2 != nullptr) ? *(const f32x4*)(a.b2 + n0) : zero, lo2 = zero;
      const char* w1 = l1 + (size_t)16 * (ob - ob0) * RB;
      const char* w2 = l2 + (size_t)16 * (ob - ob0) * RB;
      if (BF) {
        hi = bf16r4(hi), hi2 = bf16r4(hi2);
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {   // slices alternate between the two accumulators
          if (s_ & 1) lo = MFMA_BF16(*(const u32x4_d*)(w1 + s_ * 64), x1[s_], lo);
          else hi = MFMA_BF16(*(const u32x4_d*)(w1 + s_ * 64), x1[s_], hi);
          if (GATE) {
            if (s_ & 1) lo2 = MFMA_BF16(*(const u32x4_d*)(w2 + s_ * 64), x1[s_], lo2);
            else hi2 = MFMA_BF16(*(const u32x4_d*)(w2 + s_ * 64), x1[s_], hi2);
          }
        }
      } else {
#pragma unroll
      for (int s_ = 0; s_ < S; ++s_) {
        const u32x4_d a1 = *(const u32x4_d*)(w1 + s_ * 64), a2 = *(const u32x4_d*)(w1 + (S + s_) * 64), a3 = *(const u32x4_d*)(w1 + (2 * S + s_) * 64);
        lo = MFMA_BF16(a3, x1[s_], lo);
        hi = MFMA_BF16(a2, x1[s_], hi);
        lo = MFMA_BF16(a2, x2[s_], lo);
        hi = MFMA_BF16(a1, x2[s_], hi);
        lo = MFMA_BF16(a1, x3[s_], lo);
        hi = MFMA_BF16(a1, x1[s_], hi);
        if (GATE) {
          const u32x4_d c1 = *(const u32x4_d*)(w2 + s_ * 64), c2 = *(const u32x4_d*)(w2 + (S + s_) * 64), c3 = *(const u32x4_d*)(w2 + (2 * S + s_) * 64);
          lo2 = MFMA_BF16(c3, x1[s_], lo2);
          hi2 = MFMA_BF16(c2, x1[s_], hi2);
          lo2 = MFMA_BF16(c2, x2[s_], lo2);
          hi2 = MFMA_BF16(c1, x2[s_], hi2);
          lo2 = MFMA_BF16(c1, x3[s_], lo2);
          hi2 = MFMA_BF16(c1, x1[s_], hi2);
        }
      }
      }
      f32x4 acc = hi + lo, acc2 = hi2 + lo2;
      if (BF) acc = bf16r4(acc), acc2 = bf16r4(acc2);   // a bf16 nn.Linear returns bf16
      if (valid) {
        if (BF && a.z16) {   // two-byte rows: the values are bf16 numbers already
          if (a.saveZ1 != nullptr) *(uint2*)((uint16_t*)a.saveZ1 + mm * a.N + n0) = pack_bf16x4(acc);
          if (GATE && a.saveZ2 != nullptr) *(uint2*)((uint16_t*)a.saveZ2 + mm * a.N + n0) = pack_bf16x4(acc2);
        } else {
          if (a.saveZ1 != nullptr) *(f32x4*)(a.saveZ1 + mm * a.N + n0) = acc;
          if (GATE && a.saveZ2 != nullptr) *(f32x4*)(a.saveZ2 + mm * a.N + n0) = acc2;
        }
      }
      if (!GATE && a.gb_z1 != nullptr) {   // the gated product's backward as the epilogue (mgn_act_gate_bwd on the accumulator)
        f32x4 z1, z2;
        if (BF && a.z16) {
          z1 = unpack_bf16x4(*(const uint2*)((const uint16_t*)a.gb_z1 + mm * a.N + n0));
          z2 = unpack_bf16x4(*(const uint2*)((const uint16_t*)a.gb_z2 + mm * a.N + n0));
        } else {
          z1 = *(const f32x4*)(a.gb_z1 + mm * a.N + n0), z2 = *(const f32x4*)(a.gb_z2 + mm * a.N + n0);
        }
        f32x4 a1, a2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float h = d_act(z1[r], a.act);
          if (BF) h = bf16r(h);
          a1[r] = acc[r] * z2[r] * d_dact(z1[r], a.act);
          a2[r] = acc[r] * h;
        }
        if (BF) a1 = bf16r4(a1), a2 = bf16r4(a2);
        if (valid) {
          if (BF && a.out16) {
            *(uint2*)((uint16_t*)a.out + mm * a.ldo + n0) = pack_bf16x4(a1);
            *(uint2*)((uint16_t*)a.out2 + mm * a.ldo + n0) = pack_bf16x4(a2);
          } else {
            *(f32x4*)(a.out + mm * a.ldo + n0) = a1;
            *(f32x4*)(a.out2 + mm * a.ldo + n0) = a2;
          }
        }
        continue;
      }
      f32x4 y;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = d_act(acc[r], a.act);
      if (BF && a.act >= 0) y = bf16r4(y);
      if (GATE) {
        y = y * acc2;
        if (BF) y = bf16r4(y);
      }
      if (a.resid != nullptr) y = *(const f32x4*)(a.resid + mm * a.ldr + n0) + y;
      if (valid) {
        if (BF && a.out16) *(uint2*)((uint16_t*)a.out + mm * a.ldo + n0) = pack_bf16x4(y);   // (a bf16 number: no residual here)
        else *(f32x4*)(a.out + mm * a.ldo + n0) = y;
      }
    }
  }
}
// number of launches (groups of output blocks) of the k_linear_x6 form, 0 = the launch stays on k_linear
static int linear_x6_plan(long M, int KB, int N, bool gated, int precision, size_t* img) {
  static const bool x6_on = [] { const char* e = getenv("MGN_LINEAR_X6"); return e == nullptr || e[0] != '0'; }();
  if (!x6_on || (KB & 1) || (M + 63) / 64 < 1024) return 0;
  const int NB = N >> 4;
  const size_t rowb = (size_t)(4 * (precision == 1 ? 1 : 3) * (KB / 2) + 1) * 16;
  const size_t full = (size_t)(gated ? 2 : 1) * N * rowb;
  int nc = 1;
  while (nc <= NB && (full / nc > 79 * 1024 || NB % nc != 0)) ++nc;
  if (nc > NB) return 0;
  *img = full / nc;
  return nc;
}
extern "C" int mgn_linear_accepts_transposed(int64_t M, int K, int N, int gated, int precision) {
  size_t img;
  return (K >= 32 && K <= 384 && (K & 31) == 0 && N >= 16 && (N & 15) == 0) ? (linear_x6_plan((long)M, K >> 4, N, gated != 0, precision, &img) > 0) : 0;
}
template <int KB>
static int launch_linear_x6(const mgn_linear_args& a, unsigned grid, size_t lds, int nchunk, hipStream_t s) {
  if constexpr (KB % 2 == 0) {
    const bool gate = a.W2 != nullptr, bf = a.precision == 1;
    const int nbc = (a.N >> 4) / nchunk;
#define LINX_GO(GATE_, BF_)                                                                                                                  \
  do {                                                                                                                                       \
    if (lds > 48 * 1024 &&                                                                                                                   \
        hipFuncSetAttribute((const void*)k_linear_x6<KB, GATE_, BF_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)   \
      return 2;                                                                                                                              \
    for (int ch = 0; ch < nchunk; ++ch)                                                                                                      \
      hipLaunchKernelGGL((k_linear_x6<KB, GATE_, BF_>), dim3(grid), dim3(512), lds, s, a, ch * nbc, (ch + 1) * nbc);                         \
  } while (0)
    if (bf && a.x16 && !gate) {   // two-byte input rows: an instance of its own (no fp32 row registers)
      if (lds > 48 * 1024 &&
          hipFuncSetAttribute((const void*)k_linear_x6<KB, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return 2;
      for (int ch = 0; ch < nchunk; ++ch)
        hipLaunchKernelGGL((k_linear_x6<KB, false, true, true>), dim3(grid), dim3(512), lds, s, a, ch * nbc, (ch + 1) * nbc);
      return 0;
    }
    if (gate && bf) LINX_GO(true, true);
    else if (gate) LINX_GO(true, false);
    else if (bf) LINX_GO(false, true);
    else LINX_GO(false, false);
#undef LINX_GO
    return 0;
  } else {
    return 2;
  }
}

template <int KB, bool LDSW>
static int launch_linear_v(const mgn_linear_args& a, unsigned grid, size_t lds, int nchunk, hipStream_t s) {
  const bool bf = a.precision == 1, gate = a.W2 != nullptr;
  const int nbc = (a.N >> 4) / nchunk;   // output blocks per launch
#define LIN_GO(BF_, GATE_)                                                                                                            \
  do {                                                                                                                                \
    if (LDSW && lds > 48 * 1024) {                                                                                                    \
      if (hipFuncSetAttribute((const void*)k_linear<KB, BF_, GATE_, LDSW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return 2;                                                                                                                     \
    }                                                                                                                                 \
    for (int ch = 0; ch < nchunk; ++ch)                                                                                               \
      hipLaunchKernelGGL((k_linear<KB, BF_, GATE_, LDSW>), dim3(grid), dim3(256), lds, s, a, ch * nbc, (ch + 1) * nbc);               \
  } while (0)
  if (bf && gate) LIN_GO(true, true);
  else if (bf) LIN_GO(true, false);
  else if (gate) LIN_GO(false, true);
  else LIN_GO(false, false);
#undef LIN_GO
  return 0;
}
template <int KB>
static int launch_linear(const mgn_linear_args& a, hipStream_t s) {
  const long ntiles = (a.M + 63) / 64;
  const size_t full = (size_t)(a.W2 != nullptr ? 2 : 1) * a.N * (16 * KB + 4) * sizeof(float);
  // the LDS-resident weight image pays once every workgroup walks several tiles and three or more workgroups fit a CU; an image
  // above 64 KB is cut into 2 or 3 groups of output blocks (MGN_LINEAR_NO_CHUNK: such launches stay on the L2 path, for A/B)
  const int NB = a.N >> 4;
  // [r5] an even number of input blocks, 65 536 rows or more: k_linear_x6 -- precision 0 as the six-term split-bf16 form, precision 1
  // as its one-piece form (MGN_LINEAR_X6=0: k_linear everywhere, for A/B).  The piece image is 6 (2) bytes per weight; it is cut into
  // groups of output blocks until two 512-thread workgroups fit a CU.
  {
    size_t imgx;
    const int nc = linear_x6_plan(a.M, KB, a.N, a.W2 != nullptr, a.precision, &imgx);
    if (nc > 0) {
      const long nt128 = (a.M + 127) / 128;
      int per_cu = (int)((160 * 1024) / imgx);
      if (per_cu > 4) per_cu = 4;
      unsigned grid = 256u * (unsigned)per_cu;
      if ((long)grid > nt128) grid = (unsigned)nt128;
      return launch_linear_x6<KB>(a, grid, imgx, nc, s);
    }
  }
  if (a.w_transposed || a.gb_z1 != nullptr || a.norm_scale_outer != nullptr || a.z16 || a.x16 || a.out16) return 3;
  int nchunk = 1;
  while (nchunk < 4 && (full / nchunk > 64 * 1024 || NB % nchunk != 0)) ++nchunk;
  if (nchunk > 1 && getenv("MGN_LINEAR_NO_CHUNK") != nullptr) nchunk = 4;
  const size_t img = full / (nchunk < 4 ? nchunk : 1);
  if (nchunk < 4 && ntiles >= 1024 && getenv("MGN_LINEAR_NO_LDS") == nullptr) {
    int per_cu = (int)((160 * 1024) / img);
    if (per_cu > 8) per_cu = 8;
    unsigned grid = 256u * (unsigned)per_cu;
    if ((long)grid > ntiles) grid = (unsigned)ntiles;
    return launch_linear_v<KB, true>(a, grid, img, nchunk, s);
  }
  return launch_linear_v<KB, false>(a, (unsigned)ntiles, 0, 1, s);
}

extern "C" int mgn_linear_fwd(const mgn_linear_args* args, void* stream) {
  const mgn_linear_args& a = *args;
  if (a.M < 0 || a.x == nullptr || a.W == nullptr || a.out == nullptr) return dfail(1, "mgn_linear_fwd: missing operand");
  const int K = a.K1 + a.K2 + a.K3;
  if (a.K1 < 16 || (a.K1 & 15) || a.K2 < 0 || (a.K2 & 15) || a.K3 < 0 || (a.K3 & 15) || K > 384 || (a.K2 > 0 && a.x2 == nullptr) ||
      (a.K3 > 0 && (a.x3 == nullptr || a.K2 == 0)))
    return dfail(1, "mgn_linear_fwd: input widths must be multiples of 16, at most 384 together");
  if (a.K3 > 0 && (a.ldx3 < a.K3 || (a.ldx3 & 3))) return dfail(1, "mgn_linear_fwd: bad leading dimension of the third phase");
  if (a.N < 16 || (a.N & 15) || a.N > 1024) return dfail(1, "mgn_linear_fwd: output width must be a multiple of 16");
  if (a.ldx < a.K1 || (a.ldx & 3) || (a.K2 > 0 && (a.ldx2 < a.K2 || (a.ldx2 & 3))) || a.ldw < (a.w_transposed ? a.N : K) || (a.ldw & 3) || a.ldo < a.N || (a.ldo & 3) ||
      (a.resid != nullptr && (a.ldr < a.N || (a.ldr & 3))))
    return dfail(1, "mgn_linear_fwd: leading dimensions must cover the widths and keep rows 16-byte aligned");
  if (a.act < -1 || a.act > MGN_ACT_GELU) return dfail(1, "mgn_linear_fwd: act must be MGN_ACT_NONE, _RELU, _SILU or _GELU");
  if (a.norm_scale_outer != nullptr && (a.norm_scale == nullptr || a.K2 > 0 || a.idx != nullptr))
    return dfail(1, "mgn_linear_fwd: norm_scale_outer needs norm_scale and a single ungathered input phase");
  if ((a.z16 || a.x16 || a.out16) && a.precision != 1) return dfail(1, "mgn_linear_fwd: two-byte rows (z16 / x16 / out16) need precision 1");
  if (a.x16 && (a.norm_scale != nullptr || a.K2 > 0 || a.idx != nullptr || a.W2 != nullptr))
    return dfail(1, "mgn_linear_fwd: x16 takes one ungathered input phase without a norm prologue or a gated product");
  if (a.out16 && a.resid != nullptr) return dfail(1, "mgn_linear_fwd: out16 has no residual epilogue");
  if (a.gb_z1 != nullptr && (a.gb_z2 == nullptr || a.out2 == nullptr || a.W2 != nullptr || a.b != nullptr || a.resid != nullptr || a.saveZ1 != nullptr ||
                             a.ldo != a.N))
    return dfail(1, "mgn_linear_fwd: the gated-backward epilogue takes gb_z1, gb_z2, out2, dense outputs, no W2 / bias / residual / saves");
  if (a.precision != 0 && a.precision != 1) return dfail(1, "mgn_linear_fwd: precision must be 0 or 1");
  if (a.M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  switch (K >> 4) {
#define LIN_CASE(KB_)                                                                                                                \
  case KB_: {                                                                                                                        \
    const int rc_ = launch_linear<KB_>(a, s);                                                                                        \
    if (rc_ == 3) return dfail(1, "mgn_linear_fwd: a transposed weight / the gated-backward epilogue / an outer norm need the LDS-staged form (mgn_linear_accepts_transposed)");  \
    if (rc_) return dfail(2, "mgn_linear_fwd: cannot reserve LDS");                                                                  \
  } break;
    LIN_CASE(1) LIN_CASE(2) LIN_CASE(3) LIN_CASE(4) LIN_CASE(6) LIN_CASE(8) LIN_CASE(12) LIN_CASE(16) LIN_CASE(24)
#undef LIN_CASE
    default: return dfail(1, "mgn_linear_fwd: total input width must be 16, 32, 48, 64, 96, 128, 192, 256 or 384");
  }
  return dcheck("mgn_linear_fwd");
}

// ------------------------------------------------------------------ activation / gated-product backward
// dZ1 = dP * (Z2 or 1) * act'(Z1);  dZ2 = dP * act(Z1)      (elementwise over [M, N], 16-byte lanes)
__global__ void __launch_bounds__(256) k_act_gate_bwd(const float* __restrict__ dP, const float* __restrict__ Z1, const float* __restrict__ Z2,
                                                     long n4, int act, int bf, float* __restrict__ dZ1, float* __restrict__ dZ2) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 d = ((const f32x4*)dP)[i], z1 = ((const f32x4*)Z1)[i];
  f32x4 z2 = f32x4{1.f, 1.f, 1.f, 1.f};
  if (Z2 != nullptr) z2 = ((const f32x4*)Z2)[i];
  f32x4 a1, a2;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float h = d_act(z1[r], act);
    if (bf) h = bf16r(h);
    a1[r] = d[r] * z2[r] * d_dact(z1[r], act);
    a2[r] = d[r] * h;
  }
  if (bf) a1 = bf16r4(a1), a2 = bf16r4(a2);
  ((f32x4*)dZ1)[i] = a1;
  if (dZ2 != nullptr) ((f32x4*)dZ2)[i] = a2;
}

extern "C" int mgn_act_gate_bwd(const float* dP, const float* Z1, const float* Z2, int64_t M, int N, int act, int precision, float* dZ1,
                                float* dZ2, void* stream) {
  if (M < 0 || N < 4 || (N & 3) || dP == nullptr || Z1 == nullptr || dZ1 == nullptr || (Z2 != nullptr && dZ2 == nullptr) || act < -1 ||
      act > MGN_ACT_GELU)
    return dfail(1, "mgn_act_gate_bwd: bad arguments");
  const long n4 = (long)M * N / 4;
  if (n4 == 0) return 0;
  hipLaunchKernelGGL(k_act_gate_bwd, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dP, Z1, Z2, n4, act, precision, dZ1, dZ2);
  return dcheck("mgn_act_gate_bwd");
}

// ------------------------------------------------------------------ stand-alone RMSNorm (layers.py:73-129)
// y = scale * x / (||x|| / sqrt(K) + eps) per row; one wave per row pass (the Transformer's norm2 feeds a gated MLP that
// starts with a norm of its own, layers.py:256-278: the outer one cannot be a prologue)
template <int KPL>
__global__ void __launch_bounds__(256) k_rownorm_fwd(const float* __restrict__ x, int ldx, int K, const float* __restrict__ scale, float eps, long M,
                                                    float* __restrict__ y, float* __restrict__ inv_out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (long m = (long)blockIdx.x * 4 + wv; m < M; m += (long)gridDim.x * 4) {
    float xv[KPL];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int k = lane + 64 * j;
      xv[j] = (k < K) ? x[m * ldx + k] : 0.f;
      ss = fmaf(xv[j], xv[j], ss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float inv = 1.0f / (sqrtf(ss) / sqrtf((float)K) + eps);
    if (inv_out != nullptr && lane == 0) inv_out[m] = inv;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int k = lane + 64 * j;
      if (k < K) y[m * K + k] = scale[k] * (xv[j] * inv);
    }
  }
}
extern "C" int mgn_rownorm_fwd(const float* x, int ldx, int K, const float* scale, float eps, int64_t M, float* y, float* inv_out, void* stream) {
  if (M < 0 || x == nullptr || scale == nullptr || y == nullptr || K < 1 || K > 384 || ldx < K) return dfail(1, "mgn_rownorm_fwd: bad arguments (at most 384 columns)");
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  unsigned grid = (unsigned)((M + 3) / 4);
  if (grid > 2048) grid = 2048;
  switch ((K + 63) / 64) {
#define RF_CASE(P_) case P_: hipLaunchKernelGGL(k_rownorm_fwd<P_>, dim3(grid), dim3(256), 0, s, x, ldx, K, scale, eps, (long)M, y, inv_out); break;
    RF_CASE(1) RF_CASE(2) RF_CASE(3) RF_CASE(4) RF_CASE(5) RF_CASE(6)
#undef RF_CASE
    default: return dfail(1, "mgn_rownorm_fwd: width out of range");
  }
  return dcheck("mgn_rownorm_fwd");
}

// ------------------------------------------------------------------ RMSNorm-prologue backward
// n = s * x c, c = 1 / (||x|| / sqrt(K) + eps):   dx = c g - x c^2 <g, x> / (K r),  g = s dn,  r = ||x|| / sqrt(K);
// dscale partial of the workgroup's rows: sum_rows dn * x c.   One wave per row pass, lane l owns columns l, l + 64, ...
struct RnPhases {
  const float* x[3];
  const int32_t* idx[3];
  int ld[3];
  float* dx[3];
  int lddx[3];
  const float* acc[3];
  int k0[4];   // first column of each phase, k0[3] = K
};
template <int KPL>  // columns per lane (K / 64 rounded up)
__global__ void __launch_bounds__(256) k_rownorm_bwd(const float* __restrict__ dn, const RnPhases P, int K, const float* __restrict__ inv,
                                                    const float* __restrict__ scale, float eps, long M, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float ds[KPL];
  float sc[KPL];
  int ph[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    ds[j] = 0.f;
    const int k = lane + 64 * j;
    sc[j] = (k < K) ? scale[k] : 0.f;
    ph[j] = (k < P.k0[1]) ? 0 : (k < P.k0[2]) ? 1 : 2;
  }
  for (long m = (long)blockIdx.x * 4 + wv; m < M; m += (long)gridDim.x * 4) {
    float xv[KPL], gv[KPL];
    float dot = 0.f;
    const float c = inv[m];
    long row[3];
    row[0] = (P.idx[0] != nullptr) ? (long)P.idx[0][m] : m;
    row[1] = (P.idx[1] != nullptr) ? (long)P.idx[1][m] : m;
    row[2] = (P.idx[2] != nullptr) ? (long)P.idx[2][m] : m;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int k = lane + 64 * j;
      xv[j] = 0.f, gv[j] = 0.f;
      if (k < K) {
        const int p = ph[j];
        // (constant indices + selects: a kernel-argument array indexed by a register is copied to scratch)
        const float* xp = (p == 0) ? P.x[0] : (p == 1) ? P.x[1] : P.x[2];
        const long rw = (p == 0) ? row[0] : (p == 1) ? row[1] : row[2];
        const int ldp = (p == 0) ? P.ld[0] : (p == 1) ? P.ld[1] : P.ld[2];
        const int k0p = (p == 0) ? P.k0[0] : (p == 1) ? P.k0[1] : P.k0[2];
        xv[j] = xp[rw * ldp + (k - k0p)];
        const float d = dn[m * K + k];
        ds[j] += d * xv[j] * c;
        gv[j] = sc[j] * d;
        dot = fmaf(gv[j], xv[j], dot);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    const float r = 1.0f / c - eps;                       // ||x|| / sqrt(K)
    const float k2 = (r > 0.f) ? c * c * dot / ((float)K * r) : 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int k = lane + 64 * j;
      if (k < K) {
        const int p = ph[j];
        float* dp = (p == 0) ? P.dx[0] : (p == 1) ? P.dx[1] : P.dx[2];
        const float* ap = (p == 0) ? P.acc[0] : (p == 1) ? P.acc[1] : P.acc[2];
        const int ldd = (p == 0) ? P.lddx[0] : (p == 1) ? P.lddx[1] : P.lddx[2];
        const int k0p = (p == 0) ? P.k0[0] : (p == 1) ? P.k0[1] : P.k0[2];
        float v = c * gv[j] - xv[j] * k2;   // per EDGE row: the caller sums gathered phases over their segments
        if (ap != nullptr) v = ap[m * ldd + (k - k0p)] + v;
        dp[m * ldd + (k - k0p)] = v;
      }
    }
  }
  __shared__ float red[4][64 * KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) red[wv][lane + 64 * j] = ds[j];
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) part[(size_t)blockIdx.x * K + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}
// [r5] two stacked RMSNorms (n = s_i * h c_i, h = s_o * x c_o) backward in one pass over the rows: h is recomputed from x (never stored),
// dx = acc + ..., partial column sums of both scale gradients side by side ([ds_inner | ds_outer], 2 K columns per workgroup)
template <int KPL>
__global__ void __launch_bounds__(256) k_rownorm2_bwd(const float* __restrict__ dn, const float* __restrict__ x, int ldx, int K,
                                                     const float* __restrict__ inv_o, const float* __restrict__ sc_o, const float* __restrict__ inv_i,
                                                     const float* __restrict__ sc_i, float eps, long M, const float* __restrict__ acc,
                                                     float* __restrict__ dx, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float dsi[KPL], dso[KPL], so[KPL], si[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    const int k = lane + 64 * j;
    dsi[j] = dso[j] = 0.f;
    so[j] = (k < K) ? sc_o[k] : 0.f;
    si[j] = (k < K) ? sc_i[k] : 0.f;
  }
  for (long m = (long)blockIdx.x * 4 + wv; m < M; m += (long)gridDim.x * 4) {
    float xv[KPL], hv[KPL], g0[KPL], av[KPL];
    const float co = inv_o[m], ci = inv_i[m];
    float dot0 = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int k = lane + 64 * j;
      xv[j] = hv[j] = g0[j] = av[j] = 0.f;
      if (k < K) {
        xv[j] = x[m * ldx + k];
        const float d = dn[m * K + k];
        if (acc != nullptr) av[j] = acc[m * K + k];
        hv[j] = so[j] * (xv[j] * co);
        dsi[j] += d * hv[j] * ci;
        g0[j] = si[j] * d;
        dot0 = fmaf(g0[j], hv[j], dot0);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot0 += __shfl_xor(dot0, o);
    const float r0 = 1.0f / ci - eps;
    const float k0 = (r0 > 0.f) ? ci * ci * dot0 / ((float)K * r0) : 0.f;
    float g2[KPL];
    float dot2 = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const float dh = ci * g0[j] - hv[j] * k0;
      dso[j] += dh * xv[j] * co;
      g2[j] = so[j] * dh;
      dot2 = fmaf(g2[j], xv[j], dot2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot2 += __shfl_xor(dot2, o);
    const float r2 = 1.0f / co - eps;
    const float k2 = (r2 > 0.f) ? co * co * dot2 / ((float)K * r2) : 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int k = lane + 64 * j;
      if (k < K) dx[m * K + k] = av[j] + (co * g2[j] - xv[j] * k2);
    }
  }
  __shared__ float red[4][128 * KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) red[wv][lane + 64 * j] = dsi[j], red[wv][64 * KPL + lane + 64 * j] = dso[j];
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * K; k += 256) {
    const int q = (k < K) ? k : 64 * KPL + (k - K);
    part[(size_t)blockIdx.x * 2 * K + k] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
  }
}
// column sums of the workgroups' partials: 64 columns per block, 4 slices of the partial list per column, combined in a fixed
// order (deterministic)
__global__ void __launch_bounds__(256) k_colsum_parts(const float* __restrict__ part, int nparts, int chunk, int K, float* __restrict__ out) {
  // block (x, y): columns 64 x .. 64 x + 63 of the partials [y chunk, (y + 1) chunk) -> out[y][k]; a long partial list is summed in
  // two launches (chunks, then the chunk sums) so that no thread walks more than a few dozen dependent loads
  __shared__ float red[4][64];
  const int k = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  const int lo = blockIdx.y * chunk;
  const int hi = (lo + chunk < nparts) ? lo + chunk : nparts;
  float s0 = 0.f, s1 = 0.f;
  if (k < K) {
    int p = lo + sl;
    for (; p + 4 < hi; p += 8) {
      s0 += part[(size_t)p * K + k];
      s1 += part[(size_t)(p + 4) * K + k];
    }
    if (p < hi) s0 += part[(size_t)p * K + k];
  }
  red[sl][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (sl == 0 && k < K) out[(size_t)blockIdx.y * K + k] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

#define RN_GRID 2048
#define RN_CHUNK 64
extern "C" size_t mgn_rownorm_bwd_workspace_bytes(int K) { return (size_t)(RN_GRID + RN_GRID / RN_CHUNK + 1) * (K > 0 ? K : 0) * sizeof(float); }
extern "C" int mgn_rownorm_bwd(const float* dn, const mgn_rownorm_phase* phases, int nphase, const float* inv, const float* scale, float eps,
                               int64_t M, float* dscale, void* ws, size_t ws_bytes, void* stream) {
  if (M < 0 || dn == nullptr || phases == nullptr || nphase < 1 || nphase > 3 || inv == nullptr || scale == nullptr || dscale == nullptr)
    return dfail(1, "mgn_rownorm_bwd: bad arguments");
  RnPhases P;
  int K = 0;
  for (int p = 0; p < 3; ++p) {
    const bool on = p < nphase;
    if (on && (phases[p].x == nullptr || phases[p].dx == nullptr || phases[p].K < 1 || phases[p].ldx < phases[p].K || phases[p].lddx < phases[p].K))
      return dfail(1, "mgn_rownorm_bwd: bad phase");
    P.x[p] = on ? phases[p].x : phases[0].x;
    P.idx[p] = on ? phases[p].idx : nullptr;
    P.ld[p] = on ? phases[p].ldx : 0;
    P.dx[p] = on ? phases[p].dx : phases[0].dx;
    P.lddx[p] = on ? phases[p].lddx : 0;
    P.acc[p] = on ? phases[p].acc : nullptr;
    P.k0[p] = K;
    K += on ? phases[p].K : 0;
  }
  P.k0[3] = K;
  for (int p = nphase; p < 3; ++p) P.k0[p] = K;
  if (K > 384) return dfail(1, "mgn_rownorm_bwd: at most 384 columns");
  if (ws == nullptr || ws_bytes < mgn_rownorm_bwd_workspace_bytes(K)) return dfail(1, "mgn_rownorm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  unsigned grid = (unsigned)((M + 15) / 16);   // >= 4 rows per wave: enough loads in flight per wave, enough waves per CU
  if (grid > RN_GRID) grid = RN_GRID;
  if (grid == 0) grid = 1;
  float* part = (float*)ws;
  switch ((K + 63) / 64) {
#define RN_CASE(P_) case P_: hipLaunchKernelGGL(k_rownorm_bwd<P_>, dim3(grid), dim3(256), 0, s, dn, P, K, inv, scale, eps, (long)M, part); break;
    RN_CASE(1) RN_CASE(2) RN_CASE(3) RN_CASE(4) RN_CASE(5) RN_CASE(6)
#undef RN_CASE
    default: return dfail(1, "mgn_rownorm_bwd: width out of range");
  }
  const unsigned kb_ = (unsigned)((K + 63) / 64);
  if (grid > RN_CHUNK) {
    const unsigned nch = (grid + RN_CHUNK - 1) / RN_CHUNK;
    float* part2 = part + (size_t)RN_GRID * K;
    hipLaunchKernelGGL(k_colsum_parts, dim3(kb_, nch), dim3(256), 0, s, (const float*)part, (int)grid, RN_CHUNK, K, part2);
    hipLaunchKernelGGL(k_colsum_parts, dim3(kb_, 1), dim3(256), 0, s, (const float*)part2, (int)nch, (int)nch, K, dscale);
  } else {
    hipLaunchKernelGGL(k_colsum_parts, dim3(kb_, 1), dim3(256), 0, s, (const float*)part, (int)grid, (int)grid, K, dscale);
  }
  return dcheck("mgn_rownorm_bwd");
}

extern "C" int mgn_rownorm2_bwd(const float* dn, const float* x, int ldx, int K, const float* inv_outer, const float* scale_outer,
                                const float* inv_inner, const float* scale_inner, float eps, int64_t M, const float* acc, float* dx,
                                float* dscale_io, void* ws, size_t ws_bytes, void* stream) {
  if (M < 0 || dn == nullptr || x == nullptr || inv_outer == nullptr || scale_outer == nullptr || inv_inner == nullptr || scale_inner == nullptr ||
      dx == nullptr || dscale_io == nullptr || K < 1 || K > 192 || ldx < K)
    return dfail(1, "mgn_rownorm2_bwd: bad arguments (at most 192 columns)");
  if (ws == nullptr || ws_bytes < mgn_rownorm_bwd_workspace_bytes(2 * K)) return dfail(1, "mgn_rownorm2_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  unsigned grid = (unsigned)((M + 15) / 16);
  if (grid > RN_GRID) grid = RN_GRID;
  if (grid == 0) grid = 1;
  float* part = (float*)ws;
  float* both = dscale_io;   // [ds_inner | ds_outer]
  if (M == 0) {
    hipMemsetAsync(dscale_io, 0, (size_t)2 * K * sizeof(float), s);
    return dcheck("mgn_rownorm2_bwd");
  }
  switch ((K + 63) / 64) {
#define RN2_CASE(P_) \
  case P_: hipLaunchKernelGGL(k_rownorm2_bwd<P_>, dim3(grid), dim3(256), 0, s, dn, x, ldx, K, inv_outer, scale_outer, inv_inner, scale_inner, eps, (long)M, acc, dx, part); break;
    RN2_CASE(1) RN2_CASE(2) RN2_CASE(3)
#undef RN2_CASE
    default: return dfail(1, "mgn_rownorm2_bwd: width out of range");
  }
  const int K2 = 2 * K;
  const unsigned kb_ = (unsigned)((K2 + 63) / 64);
  if (grid > RN_CHUNK) {
    const unsigned nch = (grid + RN_CHUNK - 1) / RN_CHUNK;
    float* part2 = part + (size_t)RN_GRID * K2;
    hipLaunchKernelGGL(k_colsum_parts, dim3(kb_, nch), dim3(256), 0, s, (const float*)part, (int)grid, RN_CHUNK, K2, part2);
    hipLaunchKernelGGL(k_colsum_parts, dim3(kb_, 1), dim3(256), 0, s, (const float*)part2, (int)nch, (int)nch, K2, both);
  } else {
    hipLaunchKernelGGL(k_colsum_parts, dim3(kb_, 1), dim3(256), 0, s, (const float*)part, (int)grid, (int)grid, K2, both);
  }
  return dcheck("mgn_rownorm2_bwd");
}
