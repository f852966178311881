// MI355X (gfx950) kernels of the sparse-attention Transformer processor: edge-masked
// scaled dot-product attention over the mesh adjacency,
//     score[e,h] = sum_d q[i_e, d, h] k[j_e, d, h] / sqrt(D),   attn = softmax over the edges of row i,
//     y[i, d, h] = sum_{e in row i} attn[e,h] v[j_e, d, h]
// (reference: scaled_query_key_softmax / scaled_dot_product_attention, graphphysics/models/layers.py:
// 493-559, through DGL's bsddmm -> SparseMatrix.softmax -> bspmm with the adjacency
// dglsp.spmatrix(indices=edge_index), processors.py:352 -- rows are edge_index[0], columns edge_index[1]).
// Head layout as the reference reshapes it: q.reshape(N, head_dim, num_heads) (layers.py:673-675), i.e.
// the HEAD index is the fastest axis: feature f = d * num_heads + h.
//
// HBM-bound gather work (two 4*hidden-byte rows per edge), no matrix cores: one group of hidden/4 lanes
// per row, 16-byte lanes, online softmax (one pass over the row's edges), CSR order => deterministic,
// atomics-free.  The backward is two passes: by ROW (dq, per-edge attn / dscore) and by COLUMN (dk, dv)
// through the column-grouped CSR of the same edges.
// Third translation unit of libmgn_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mgn_hip.h"

static thread_local char g_aerr[256] = "";
extern "C" const char* mgn_attn_last_error(void) { return g_aerr; }
static int afail(int code, const char* msg) {
  snprintf(g_aerr, sizeof(g_aerr), "%s", msg);
  return code;
}
static int acheck(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_aerr, sizeof(g_aerr), "%s: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// Lane l of a row group holds features 4l .. 4l+3; feature f belongs to head f % NH.  head_reduce turns the
// per-feature partial products p[r] into the full per-HEAD sums, delivered to every lane for the heads of
// its own four features: butterfly over the lanes that share the same heads (stride >= NH/4), then a fold
// inside the lane when NH < 4.  LPR = lanes per row (hidden / 4), a power of two <= 32.
// p + (p of lane l ^ M) inside groups of LPR lanes, on the DPP network (no LDS round trip): quad permutes for M = 1, 2; rotations
// of the 16-lane DPP row for M = 8 and, when the group IS that row, for M = 4 (the operand already has period 8 there, so the
// rotation reads the same value as the exchange would -- bit-identical sums); 8-lane groups take M = 4 as two bank-masked
// rotations; M = 16 (128-wide rows) stays a shuffle.
template <int M, int LPR>
__device__ __forceinline__ float xor_add(float x) {
  const int xi = __builtin_bit_cast(int, x);
  int t;
  if constexpr (M == 1) {
    t = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xf, 0xf, true);           // quad_perm [1,0,3,2]
  } else if constexpr (M == 2) {
    t = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xf, 0xf, true);           // quad_perm [2,3,0,1]
  } else if constexpr (M == 4 && LPR >= 16) {
    t = __builtin_amdgcn_update_dpp(0, xi, 0x124, 0xf, 0xf, true);          // row_ror:4 (after the M = 8 step)
  } else if constexpr (M == 4) {
    t = __builtin_amdgcn_update_dpp(0, xi, 0x12C, 0xf, 0x5, false);         // row_ror:12 (lane l reads l + 4) into lanes 0-3, 8-11
    t = __builtin_amdgcn_update_dpp(t, xi, 0x124, 0xf, 0xa, false);         // row_ror:4 (lane l reads l - 4) into lanes 4-7, 12-15
  } else if constexpr (M == 8) {
    t = __builtin_amdgcn_update_dpp(0, xi, 0x128, 0xf, 0xf, true);          // row_ror:8 == l ^ 8
  } else {
    return x + __shfl_xor(x, M, LPR);
  }
  return x + __builtin_bit_cast(float, t);
}

// The same step for the four partial products of a lane at once, as v_add_f32_dpp (operand 0 through the DPP network, one
// instruction per value: hipcc keeps update_dpp as a v_mov_b32_dpp + zero fill + add, three).  The leading s_nop covers the
// VALU-write -> DPP-read hazard the assembler does not see inside an asm block; within and between the blocks a register is
// read four instructions after it was written.
#define ATTN_DPP4(CTRL)                                                             \
  asm("s_nop 1\n\t"                                                                 \
      "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
      "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
      "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
      "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1"       \
      : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]))
template <int M, int LPR>
__device__ __forceinline__ void xor_add4(float (&p)[4]) {
  if constexpr (M == 1) {
    ATTN_DPP4("quad_perm:[1,0,3,2]");
  } else if constexpr (M == 2) {
    ATTN_DPP4("quad_perm:[2,3,0,1]");
  } else if constexpr (M == 4 && LPR >= 16) {
    ATTN_DPP4("row_ror:4");
  } else if constexpr (M == 8) {
    ATTN_DPP4("row_ror:8");
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = xor_add<M, LPR>(p[r]);
  }
}

// GS = lanes of a row group that hold one copy of every head between them (max(NH / 4, 1)): the butterfly runs over the strides
// >= GS, top down (M = 8 before M = 4: what the rotation form of M = 4 relies on).
template <int LPR, int GS>
__device__ __forceinline__ void head_reduce(float (&p)[4], int NH) {
  static_assert(LPR <= 32 && GS >= 1, "row groups of at most 32 lanes");
  if constexpr (LPR >= 32 && GS <= 16) xor_add4<16, LPR>(p);
  if constexpr (LPR >= 16 && GS <= 8) xor_add4<8, LPR>(p);
  if constexpr (LPR >= 8 && GS <= 4) xor_add4<4, LPR>(p);
  if constexpr (LPR >= 4 && GS <= 2) xor_add4<2, LPR>(p);
  if constexpr (LPR >= 2 && GS <= 1) xor_add4<1, LPR>(p);
  if (NH == 2) {
    const float a = p[0] + p[2], b = p[1] + p[3];
    p[0] = p[2] = a, p[1] = p[3] = b;
  } else if (NH == 1) {
    const float a = (p[0] + p[1]) + (p[2] + p[3]);
    p[0] = p[1] = p[2] = p[3] = a;
  }
}

// [r5] Four heads (the reference's default, coarse-aneurysm.json): a lane's four features are the four heads of one head-dim index, so
// after head_reduce ALL lanes of a row hold the same four scores and the per-head scalar work -- the exponentials above all: 9
// instructions each, these kernels are bound by vector-instruction issue -- was done four times per lane, identically.  Q4: lane l
// does it for head l & 3 only and the quad hands the results round (v_mov_b32_dpp quad_perm:[r,r,r,r]); the per-feature updates
// stay per lane; the score itself comes from head_reduce_own below.
template <int R>
__device__ __forceinline__ float quad_bcast(float x) {
  constexpr int ctrl = R | (R << 2) | (R << 4) | (R << 6);
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true));
}
__device__ __forceinline__ float sel4(const float (&p)[4], int h) { return h == 0 ? p[0] : h == 1 ? p[1] : h == 2 ? p[2] : p[3]; }
// Q4: the sum over the row's lanes of p[l & 3] ONLY (what lane l needs), as a reduce-scatter inside the quad -- pairs exchange the two
// heads the partner keeps (2 adds), then the one head (1 add) -- followed by the strides >= 4 on that one value: 6 selects + 3..6 DPP
// adds where the all-lanes butterfly of head_reduce takes 16 adds (+ 3 selects to pick the head).  Another association of the same
// 16 terms than head_reduce's (results differ in the last bit; forward and backward use the same one).
#define ATTN_DPP_ADD(MINE, GIVE, CTRL) \
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(MINE) : "v"(GIVE))
template <int LPR>
__device__ __forceinline__ float head_reduce_own(const float (&p)[4], int l) {
  const bool b0 = l & 1, b1 = l & 2;
  float m0 = b0 ? p[1] : p[0], m1 = b0 ? p[3] : p[2];
  const float g0 = b0 ? p[0] : p[1], g1 = b0 ? p[2] : p[3];
  ATTN_DPP_ADD(m0, g0, "quad_perm:[1,0,3,2]");
  ATTN_DPP_ADD(m1, g1, "quad_perm:[1,0,3,2]");
  float m = b1 ? m1 : m0;
  const float g = b1 ? m0 : m1;
  ATTN_DPP_ADD(m, g, "quad_perm:[2,3,0,1]");
  if constexpr (LPR >= 32) m = xor_add<16, LPR>(m);
  if constexpr (LPR >= 16) m = xor_add<8, LPR>(m);
  if constexpr (LPR >= 8) m = xor_add<4, LPR>(m);
  return m;
}

// [r5] XCD-aware block order: consecutive workgroups go to consecutive XCDs, each with an L2 of its own, so with rows handed out in
// blockIdx order every L2 saw rows from everywhere.  Block b serves row block (b & 7) * (gridDim / 8) + (b >> 3) instead: an XCD
// walks ONE contiguous eighth of the (Morton-ordered) rows, and the k / v (q / dy) rows its workgroups gather are mostly rows its own
// L2 already holds.  -DATTN_NO_XCD_ORDER: blockIdx order.
__device__ __forceinline__ long attn_block() {
#ifdef ATTN_NO_XCD_ORDER
  return (long)blockIdx.x;
#else
  const unsigned per = gridDim.x >> 3, b = blockIdx.x;
  return (b < 8 * per) ? (long)(b & 7) * per + (b >> 3) : (long)b;
#endif
}

// The edge loops below are software-pipelined by hand: the column index of edge e + 2 and the gathered rows of edge e + 1 are
// requested before edge e is processed (a lane walks ONE row's edges serially, and index -> row -> arithmetic is a dependent
// chain of two memory latencies per edge otherwise; rows of a wave have different degrees, so the compiler does not do it).
__device__ __forceinline__ float4 ldrow(const float* p, size_t row, int H, int l) { return *(const float4*)(p + row * H + 4 * l); }

// bf16 matrix mode (the *_b16 entry points): the key / value rows are STORED as bf16 (8 bytes per lane instead of 16: they are the
// outputs of bf16-mode projections, so the narrowing is exact), the scaled query, y and the incoming dy are rounded to bf16 where the
// reference holds bf16 tensors (layers.py:509-510; scores / softmax / AV stay fp32, layers.py:49-70).
__device__ __forceinline__ float bf16r(float x) {
  unsigned u = __builtin_bit_cast(unsigned, x);
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  return __builtin_bit_cast(float, u);
}
// exp(x) for the softmax terms (x <= 0 up to rounding; a very negative FINITE x gives 0): expf's own evaluation -- t = fl(x log2 e),
// the part of x log2 e that t lost (the product's rounding and log2 e's tail) carried in r, v_exp_f32 on the REDUCED argument
// (t - round(t)) + r in [-0.5, 0.5], then ldexp -- without its range checks (9 instructions instead of 14).  Skipping the reduction
// (v_exp_f32 on t itself, corrected to first order in r: 6 instructions) was measured 4x noisier on the cancelling key-bias
// gradient and is not used.  These kernels are bound by vector-instruction issue, not by bytes ([r4]).
__device__ __forceinline__ float exp_sm(float x) {
#ifdef ATTN_EXPF
  return expf(x);
#endif
  constexpr float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
  // [r5] exp(-150) is an exact 0 after the ldexp; clamping keeps n inside int range and r bounded whatever the score magnitude
  // (the -1e37 start of the running maximum, or |score| >= 3e29, used to reach fptosi's undefined range); a NaN stays a NaN
  x = x < -150.f ? -150.f : x;
  const float t = x * L2E_HI;
  const float r = __builtin_fmaf(x, L2E_LO, __builtin_fmaf(x, L2E_HI, -t));
  const float n = __builtin_rintf(t);
  const float e = __builtin_amdgcn_exp2f((t - n) + r);
  return __builtin_amdgcn_ldexpf(e, (int)n);
}
constexpr float ATTN_M0 = -1e37f;   // the running maximum before the first edge: finite, so exp_sm(-(p - M0)) is an exact 0, no NaN

template <bool B16>
__device__ __forceinline__ float4 ldkv(const void* p, size_t row, int H, int l) {   // H = the row pitch in elements here
  if constexpr (B16) {
    const uint2 t = *(const uint2*)((const uint16_t*)p + row * H + 4 * l);
    return make_float4(__builtin_bit_cast(float, t.x << 16), __builtin_bit_cast(float, t.x & 0xffff0000u),
                       __builtin_bit_cast(float, t.y << 16), __builtin_bit_cast(float, t.y & 0xffff0000u));
  } else {
    return ldrow((const float*)p, row, H, l);
  }
}
template <bool B16>
__device__ __forceinline__ float qscaled(float q, float scale, float sd) { return B16 ? bf16r(q / sd) : q * scale; }

// Row pitches (in elements) of the strided operands: q / k / v may be column slabs of one [N, 3H] projection output and
// dq / dk / dv slabs of its gradient (the *_s entry points); y, lse, dy are dense [N, H].
struct AttnLd {
  int q, k, v, dq, dk, dv;
};

// y[i] and lse[i] (per feature: log-sum-exp of its head's scores) for every row i
template <int LPR, int GS, int B16, bool Q4>
__global__ void __launch_bounds__(256) k_attn_fwd(const float* __restrict__ q, const void* __restrict__ k, const void* __restrict__ v,
                                                 const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, long N, int NH,
                                                 float scale, float sd, float* __restrict__ y, float* __restrict__ lse, float* __restrict__ y_raw, AttnLd ld) {
  constexpr int H = 4 * LPR;
  const long i = (attn_block() * 256 + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  if (i >= N) return;
  const float4 qv = *(const float4*)(q + (size_t)i * ld.q + 4 * l);
  const float qq[4] = {qscaled<(B16 != 0)>(qv.x, scale, sd), qscaled<(B16 != 0)>(qv.y, scale, sd), qscaled<(B16 != 0)>(qv.z, scale, sd), qscaled<(B16 != 0)>(qv.w, scale, sd)};
  float m[4] = {ATTN_M0, ATTN_M0, ATTN_M0, ATTN_M0}, s[4] = {0.f, 0.f, 0.f, 0.f}, acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int hm = l & 3;               // Q4: the head this lane keeps the softmax state of
  float m_ = ATTN_M0, s_ = 0.f;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  if (e0 < e1) {
    float4 kv = ldkv<B16 == 1>(k, (size_t)col[e0], ld.k, l), vv = ldkv<B16 == 1>(v, (size_t)col[e0], ld.v, l);
    int jn = (e0 + 1 < e1) ? col[e0 + 1] : 0;
    for (int e = e0; e < e1; ++e) {
      float4 kn = kv, vn = vv;
      if (e + 1 < e1) kn = ldkv<B16 == 1>(k, (size_t)jn, ld.k, l), vn = ldkv<B16 == 1>(v, (size_t)jn, ld.v, l);
      if (e + 2 < e1) jn = col[e + 2];
      float p[4] = {qq[0] * kv.x, qq[1] * kv.y, qq[2] * kv.z, qq[3] * kv.w};
      if constexpr (!Q4) head_reduce<LPR, GS>(p, NH);
      const float vr[4] = {vv.x, vv.y, vv.z, vv.w};
      if constexpr (Q4) {
        const float pm = head_reduce_own<LPR>(p, l);
        const bool up = pm > m_;
        const float t = exp_sm(-fabsf(pm - m_));
        s_ = up ? __builtin_fmaf(s_, t, 1.f) : s_ + t;
        m_ = up ? pm : m_;
        const float ts = up ? -t : t;   // the factor with "the maximum moved" in its sign bit (t >= 0; -0 keeps the bit)
#define ATTN_Q4_ACC(R)                                                                                  \
  {                                                                                                     \
    const float tb = quad_bcast<R>(ts);                                                                 \
    const float tt = fabsf(tb);                                                                         \
    acc[R] = (__builtin_bit_cast(int, tb) < 0) ? __builtin_fmaf(acc[R], tt, vr[R]) : __builtin_fmaf(tt, vr[R], acc[R]); \
  }
        ATTN_Q4_ACC(0) ATTN_Q4_ACC(1) ATTN_Q4_ACC(2) ATTN_Q4_ACC(3)
#undef ATTN_Q4_ACC
      } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // online softmax with ONE exponential per head and edge: of the two factors exp(m - nm), exp(p - nm) one is exactly 1, so the
        // running sums are rescaled only where the maximum moved (no multiply by an exact 1).  236 -> ~190 us per launch against the
        // two-exponential form; same value, other rounding (outputs move by <= 1e-6 of the scale; the three parity readings of the
        // 10-block model for both forms on three seeds: profiles/r03_attn_one_exp_readings.txt)
        const bool up = p[r] > m[r];
        const float t = exp_sm(-fabsf(p[r] - m[r]));      // first edge: exp(-1e37) = 0
        s[r] = up ? __builtin_fmaf(s[r], t, 1.f) : s[r] + t;
        acc[r] = up ? __builtin_fmaf(acc[r], t, vr[r]) : __builtin_fmaf(t, vr[r], acc[r]);
        m[r] = up ? p[r] : m[r];
      }
      }
      kv = kn, vv = vn;
    }
  }
  if constexpr (Q4) {
    s[0] = quad_bcast<0>(s_), s[1] = quad_bcast<1>(s_), s[2] = quad_bcast<2>(s_), s[3] = quad_bcast<3>(s_);
    m[0] = quad_bcast<0>(m_), m[1] = quad_bcast<1>(m_), m[2] = quad_bcast<2>(m_), m[3] = quad_bcast<3>(m_);
  }
  float4 o, ls;
  float* op = &o.x;
  float* lp = &ls.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    op[r] = (s[r] > 0.f) ? acc[r] / s[r] : 0.f;        // a row without edges attends to nothing: zeros
    lp[r] = (s[r] > 0.f) ? m[r] + logf(s[r]) : 0.f;
  }
  if (B16) {  // y leaves as a bf16 tensor; the backward's D = sum_d dy y is the fp32 softmax's own (the unrounded rows)
    if (y_raw != nullptr) *(float4*)(y_raw + (size_t)i * H + 4 * l) = o;
#pragma unroll
    for (int r = 0; r < 4; ++r) op[r] = bf16r(op[r]);
  }
  *(float4*)(y + (size_t)i * H + 4 * l) = o;
  if (lse != nullptr) *(float4*)(lse + (size_t)i * H + 4 * l) = ls;
}

// backward, pass A (by row): dq[i]; per edge and head the attention weight a and the score gradient ds
//   D[h] = sum_d dy[i,d,h] y[i,d,h];  a = exp(score - lse);  dA = sum_d dy[i,d,h] v[j,d,h];  ds = a (dA - D)
//   dq[i,f] = scale * sum_e ds[e,h(f)] k[j_e,f]
template <int LPR, int GS, int B16, bool Q4>
__global__ void __launch_bounds__(256) k_attn_bwd_row(const float* __restrict__ q, const void* __restrict__ k, const void* __restrict__ v,
                                                     const float* __restrict__ y, const float* __restrict__ lse, const float* __restrict__ dy,
                                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, long N, int NH,
                                                     float scale, float sd, float* __restrict__ dq, float* __restrict__ a_out,
                                                     float* __restrict__ ds_out, AttnLd ld, uint16_t* __restrict__ q16,
                                                     uint16_t* __restrict__ g16) {
  constexpr int H = 4 * LPR;
  const long i = (attn_block() * 256 + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  if (i >= N) return;
  const size_t ro = (size_t)i * H + 4 * l;
  const float4 qv = *(const float4*)(q + (size_t)i * ld.q + 4 * l), yv = *(const float4*)(y + ro), lv = *(const float4*)(lse + ro);
  float4 gv = *(const float4*)(dy + ro);
  if (B16) gv = make_float4(bf16r(gv.x), bf16r(gv.y), bf16r(gv.z), bf16r(gv.w));
  const float qq[4] = {qscaled<(B16 != 0)>(qv.x, scale, sd), qscaled<(B16 != 0)>(qv.y, scale, sd), qscaled<(B16 != 0)>(qv.z, scale, sd), qscaled<(B16 != 0)>(qv.w, scale, sd)};
  const float g[4] = {gv.x, gv.y, gv.z, gv.w}, ls[4] = {lv.x, lv.y, lv.z, lv.w};
  if (B16 && q16 != nullptr) {   // the column pass gathers these rows per edge: hand it the bf16 tensors themselves (both ARE bf16 values)
    auto pk2 = [](float a_, float b_) { return (__builtin_bit_cast(unsigned, a_) >> 16) | (__builtin_bit_cast(unsigned, b_) & 0xffff0000u); };
    *(uint2*)(q16 + (size_t)i * H + 4 * l) = make_uint2(pk2(qq[0], qq[1]), pk2(qq[2], qq[3]));
    *(uint2*)(g16 + (size_t)i * H + 4 * l) = make_uint2(pk2(g[0], g[1]), pk2(g[2], g[3]));
  }
  float D[4] = {gv.x * yv.x, gv.y * yv.y, gv.z * yv.z, gv.w * yv.w};
  head_reduce<LPR, GS>(D, NH);
  const float ls_m = sel4(ls, l & 3), D_m = sel4(D, l & 3);   // Q4
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int gs = (NH >= 4) ? (NH >> 2) : 1;     // lanes 0 .. gs-1 hold one copy of every head between them
  const int nr = (NH >= 4) ? 4 : NH;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  if (e0 < e1) {
    float4 kv = ldkv<B16 == 1>(k, (size_t)col[e0], ld.k, l), vv = ldkv<B16 == 1>(v, (size_t)col[e0], ld.v, l);
    int jn = (e0 + 1 < e1) ? col[e0 + 1] : 0;
    for (int e = e0; e < e1; ++e) {
      float4 kn = kv, vn = vv;
      if (e + 1 < e1) kn = ldkv<B16 == 1>(k, (size_t)jn, ld.k, l), vn = ldkv<B16 == 1>(v, (size_t)jn, ld.v, l);
      if (e + 2 < e1) jn = col[e + 2];
      const float kr[4] = {kv.x, kv.y, kv.z, kv.w};
      float p[4] = {qq[0] * kv.x, qq[1] * kv.y, qq[2] * kv.z, qq[3] * kv.w};
      float dA[4] = {g[0] * vv.x, g[1] * vv.y, g[2] * vv.z, g[3] * vv.w};
      float a4[4], d4[4];
      if constexpr (Q4) {   // head l & 3 per lane, the score gradient handed round the quad; lanes 0..3 store their head's pair
        const float a_m = exp_sm(head_reduce_own<LPR>(p, l) - ls_m);
        const float d_m = a_m * (head_reduce_own<LPR>(dA, l) - D_m);
        acc[0] += quad_bcast<0>(d_m) * kr[0];
        acc[1] += quad_bcast<1>(d_m) * kr[1];
        acc[2] += quad_bcast<2>(d_m) * kr[2];
        acc[3] += quad_bcast<3>(d_m) * kr[3];
        if (l < 4) {
          a_out[(size_t)e * 4 + l] = a_m;
          ds_out[(size_t)e * 4 + l] = d_m;
        }
        kv = kn, vv = vn;
        continue;
      }
      head_reduce<LPR, GS>(p, NH);
      head_reduce<LPR, GS>(dA, NH);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a4[r] = exp_sm(p[r] - ls[r]);
        d4[r] = a4[r] * (dA[r] - D[r]);
        acc[r] += d4[r] * kr[r];
      }
      if (NH == 4) {  // lane 0 holds one copy of the four heads: one 16-byte store each
        if (l == 0) {
          *(float4*)(a_out + (size_t)e * 4) = make_float4(a4[0], a4[1], a4[2], a4[3]);
          *(float4*)(ds_out + (size_t)e * 4) = make_float4(d4[0], d4[1], d4[2], d4[3]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (l < gs && r < nr) {  // head h = 4l + r (NH >= 4) or r (NH < 4)
            a_out[(size_t)e * NH + 4 * l + r] = a4[r];
            ds_out[(size_t)e * NH + 4 * l + r] = d4[r];
          }
      }
      kv = kn, vv = vn;
    }
  }
  if (B16) {  // back through float(bf16(q / sd)) * sd: the gradient is a bf16 tensor between the two casts
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = bf16r(acc[r] * scale * sd) / (sd * scale);
  }
  *(float4*)(dq + (size_t)i * ld.dq + 4 * l) = make_float4(acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale);
}

// backward, pass B (by column j through the column-grouped order of the same edges: the t-th edge of that order is the
// row-sorted edge cperm[t], whose row is crow[t]):
//   dk[j,f] = scale * sum_e ds[e,h(f)] q[i_e,f];   dv[j,f] = sum_e a[e,h(f)] dy[i_e,f]
template <int LPR, int GS, int B16, bool Q4_UNUSED>
__global__ void __launch_bounds__(256) k_attn_bwd_col(const float* __restrict__ q, const float* __restrict__ dy, const float* __restrict__ a_in,
                                                     const float* __restrict__ ds_in, const int32_t* __restrict__ cptr,
                                                     const int32_t* __restrict__ cperm, const int32_t* __restrict__ crow, long N, int NH,
                                                     float scale, float sd, float* __restrict__ dk, float* __restrict__ dv, AttnLd ld,
                                                     const uint16_t* __restrict__ q16, const uint16_t* __restrict__ g16) {
  constexpr int H = 4 * LPR;
  const long j = (attn_block() * 256 + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  if (j >= N) return;
  float ak[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
  const bool sd_pow2 = (__builtin_bit_cast(unsigned, sd) & 0x7fffffu) == 0u;   // then scale = 1 / sd is exact as well
  const int t0 = cptr[j], t1 = cptr[j + 1];
  auto heads = [&](const float* __restrict__ src, size_t e, float (&o)[4]) {
    if (NH == 4) {
      const float4 t = *(const float4*)(src + e * 4);
      o[0] = t.x, o[1] = t.y, o[2] = t.z, o[3] = t.w;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = src[e * NH + (4 * l + r) % NH];
    }
  };
  if (t0 < t1) {
    // two edges in flight behind the one being summed, their indices one further ahead: nothing but latency bounds this pass
    // (no reductions, four adds per edge), so the depth of the request queue is its speed ([r4]: 251 -> ~200 us at depth 2)
    struct Row {
      float4 q, g;
      float ds[4], aw[4];
    };
    auto fetch = [&](int i_, int e_) -> Row {
      Row r;
      if (B16 && q16 != nullptr) {   // bf16(q / sd) and bf16(dy) as the row pass stored them: half the gathered bytes
        r.q = ldkv<true>(q16, (size_t)i_, H, l), r.g = ldkv<true>(g16, (size_t)i_, H, l);
      } else {
        r.q = ldrow(q, (size_t)i_, ld.q, l), r.g = ldrow(dy, (size_t)i_, H, l);
      }
      heads(ds_in, (size_t)e_, r.ds), heads(a_in, (size_t)e_, r.aw);
      return r;
    };
    Row cur = fetch(crow[t0], cperm[t0]), nx1 = cur;
    int i2 = 0, e2 = 0;
    if (t0 + 1 < t1) nx1 = fetch(crow[t0 + 1], cperm[t0 + 1]);
    if (t0 + 2 < t1) i2 = crow[t0 + 2], e2 = cperm[t0 + 2];
    for (int t = t0; t < t1; ++t) {
      Row nx2 = nx1;
      if (t + 2 < t1) nx2 = fetch(i2, e2);
      if (t + 3 < t1) i2 = crow[t + 3], e2 = cperm[t + 3];
      float qr[4] = {cur.q.x, cur.q.y, cur.q.z, cur.q.w}, gr[4] = {cur.g.x, cur.g.y, cur.g.z, cur.g.w};
      if (B16 && q16 != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) qr[r] *= sd;   // (the stored rows are bf16(q / sd) and bf16(dy) already)
      } else if (B16) {  // the query the scores were formed from (float(bf16(q / sd)) * sd) and the bf16 dy
        if (sd_pow2) {  // a power-of-two sqrt(D) (head widths 4, 16, 64): q * (1 / sd) IS q / sd, without four divisions per edge
#pragma unroll
          for (int r = 0; r < 4; ++r) qr[r] = bf16r(qr[r] * scale) * sd, gr[r] = bf16r(gr[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) qr[r] = bf16r(qr[r] / sd) * sd, gr[r] = bf16r(gr[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ak[r] += cur.ds[r] * qr[r];
        av[r] += cur.aw[r] * gr[r];
      }
      cur = nx1, nx1 = nx2;
    }
  }
  *(float4*)(dk + (size_t)j * ld.dk + 4 * l) = make_float4(ak[0] * scale, ak[1] * scale, ak[2] * scale, ak[3] * scale);
  *(float4*)(dv + (size_t)j * ld.dv + 4 * l) = make_float4(av[0], av[1], av[2], av[3]);
}

// attention weights per edge and head (return_attention=True, layers.py:543-559): a[e,h] = exp(score[e,h] - lse[i_e,h]),
// written at out_pos[e] (the edge's position in the caller's edge_index; NULL: the row-sorted position itself)
template <int LPR, int GS, int B16_UNUSED, bool Q4_UNUSED>
__global__ void __launch_bounds__(256) k_attn_weights(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ lse,
                                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     const int32_t* __restrict__ out_pos, long N, int NH, float scale, float* __restrict__ a_out) {
  constexpr int H = 4 * LPR;
  const long i = (attn_block() * 256 + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  if (i >= N) return;
  const size_t ro = (size_t)i * H + 4 * l;
  const float4 qv = *(const float4*)(q + ro), lv = *(const float4*)(lse + ro);
  const float qq[4] = {qv.x * scale, qv.y * scale, qv.z * scale, qv.w * scale}, ls[4] = {lv.x, lv.y, lv.z, lv.w};
  const int gs = (NH >= 4) ? (NH >> 2) : 1;
  const int nr = (NH >= 4) ? 4 : NH;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const size_t j = (size_t)col[e];
    const float4 kv = *(const float4*)(k + j * H + 4 * l);
    float p[4] = {qq[0] * kv.x, qq[1] * kv.y, qq[2] * kv.z, qq[3] * kv.w};
    head_reduce<LPR, GS>(p, NH);
    const size_t o = (size_t)(out_pos != nullptr ? out_pos[e] : e) * NH;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (l < gs && r < nr) a_out[o + 4 * l + r] = exp_sm(p[r] - ls[r]);
  }
}

static int attn_args_ok(int64_t N, int H, int NH) {
  if (N < 0) return 0;
  if (!(H == 16 || H == 32 || H == 64 || H == 128)) return 0;
  if (!(NH == 1 || NH == 2 || NH == 4 || NH == 8 || NH == 16) || H % NH != 0 || NH > H / 4 * 4) return 0;
  return 1;
}

// MGN_ATTN_Q4=0: four heads on the every-lane form (A/B; the two forms agree to the last bits: another association of the head sums)
static bool attn_q4_on() {
  static const bool on = [] { const char* e = getenv("MGN_ATTN_Q4"); return e == nullptr || e[0] != '0'; }();
  return on;
}
#define ATTN_DISPATCH_GS(KERNEL, GS_, B16, Q4_, ...)                                                                       \
  do {                                                                                                           \
    const unsigned grid = (unsigned)(((long)N * (H / 4) + 255) / 256);                                           \
    switch (H) {                                                                                                 \
      case 128: hipLaunchKernelGGL((KERNEL<32, GS_, B16, Q4_>), dim3(grid), dim3(256), 0, s, __VA_ARGS__); break;     \
      case 64: hipLaunchKernelGGL((KERNEL<16, GS_, B16, Q4_>), dim3(grid), dim3(256), 0, s, __VA_ARGS__); break;      \
      case 32: hipLaunchKernelGGL((KERNEL<8, GS_, B16, Q4_>), dim3(grid), dim3(256), 0, s, __VA_ARGS__); break;       \
      default: hipLaunchKernelGGL((KERNEL<4, GS_, B16, Q4_>), dim3(grid), dim3(256), 0, s, __VA_ARGS__); break;       \
    }                                                                                                            \
  } while (0)
// the reduction stride is a template argument: 1 (up to 4 heads), 2 (8 heads), 4 (16 heads)
#define ATTN_DISPATCH(KERNEL, B16, ...)                                    \
  do {                                                                     \
    if (num_heads == 4 && attn_q4_on())                                    \
      ATTN_DISPATCH_GS(KERNEL, 1, B16, true, __VA_ARGS__);                 \
    else if (num_heads <= 4)                                               \
      ATTN_DISPATCH_GS(KERNEL, 1, B16, false, __VA_ARGS__);                \
    else if (num_heads == 8)                                               \
      ATTN_DISPATCH_GS(KERNEL, 2, B16, false, __VA_ARGS__);                \
    else                                                                   \
      ATTN_DISPATCH_GS(KERNEL, 4, B16, false, __VA_ARGS__);                \
  } while (0)

static int attn_ld_ok(const AttnLd& ld, int H, bool grads) {
  const int p[6] = {ld.q, ld.k, ld.v, ld.dq, ld.dk, ld.dv};
  for (int i = 0; i < (grads ? 6 : 3); ++i)
    if (p[i] < H || p[i] % 4 != 0) return 0;
  return 1;
}
static AttnLd attn_dense_ld(int H) { return AttnLd{H, H, H, H, H, H}; }

static int attn_fwd_any(int b16, const float* q, const void* k, const void* v, const int32_t* rowptr, const int32_t* col, int64_t N, int H,
                        int num_heads, float* y, float* lse, float* y_raw, AttnLd ld, void* stream, const char* who) {
  if (!attn_args_ok(N, H, num_heads)) {
    snprintf(g_aerr, sizeof(g_aerr), "%s: hidden must be 16/32/64/128 and num_heads 1/2/4/8/16 dividing it", who);
    return 1;
  }
  if (!attn_ld_ok(ld, H, false)) {
    snprintf(g_aerr, sizeof(g_aerr), "%s: row pitches must be multiples of 4 elements and >= hidden", who);
    return 1;
  }
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const float sd = sqrtf((float)(H / num_heads)), scale = 1.0f / sd;
  if (b16 == 1)
    ATTN_DISPATCH(k_attn_fwd, 1, q, k, v, rowptr, col, (long)N, num_heads, scale, sd, y, lse, y_raw, ld);
  else if (b16 == 2)
    ATTN_DISPATCH(k_attn_fwd, 2, q, k, v, rowptr, col, (long)N, num_heads, scale, sd, y, lse, y_raw, ld);
  else
    ATTN_DISPATCH(k_attn_fwd, 0, q, k, v, rowptr, col, (long)N, num_heads, scale, sd, y, lse, y_raw, ld);
  return acheck(who);
}

static int attn_bwd_any(int b16, const float* q, const void* k, const void* v, const float* y, const float* lse, const float* dy,
                        const int32_t* rowptr, const int32_t* col, const int32_t* cptr, const int32_t* cperm, const int32_t* crow, int64_t N,
                        int64_t E, int H, int num_heads, float* dq, float* dk, float* dv, float* ws, size_t ws_bytes, AttnLd ld, void* stream,
                        const char* who) {
  if (!attn_args_ok(N, H, num_heads) || E < 0) {
    snprintf(g_aerr, sizeof(g_aerr), "%s: bad arguments", who);
    return 1;
  }
  if (ws_bytes < (size_t)2 * E * num_heads * sizeof(float)) {
    snprintf(g_aerr, sizeof(g_aerr), "%s: workspace too small (2 * E * num_heads floats)", who);
    return 1;
  }
  if (!attn_ld_ok(ld, H, true)) {
    snprintf(g_aerr, sizeof(g_aerr), "%s: row pitches must be multiples of 4 elements and >= hidden", who);
    return 1;
  }
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const float sd = sqrtf((float)(H / num_heads)), scale = 1.0f / sd;
  float* a_e = ws;
  float* ds_e = ws + (size_t)E * num_heads;
  // bf16 mode with N * H more floats of workspace: the row pass leaves bf16(q / sd) and bf16(dy) as two-byte rows for the column pass
  uint16_t* q16 = nullptr;
  uint16_t* g16 = nullptr;
  if (b16 == 1 && ws_bytes >= ((size_t)2 * E * num_heads + (size_t)N * H) * sizeof(float)) {
    q16 = (uint16_t*)(ws + (size_t)2 * E * num_heads);
    g16 = q16 + (size_t)N * H;
  }
#define ATTN_BWD_GO(B16_)                                                                                                                       \
  do {                                                                                                                                         \
    ATTN_DISPATCH(k_attn_bwd_row, B16_, q, k, v, y, lse, dy, rowptr, col, (long)N, num_heads, scale, sd, dq, a_e, ds_e, ld, q16, g16);            \
    ATTN_DISPATCH(k_attn_bwd_col, B16_, q, dy, (const float*)a_e, (const float*)ds_e, cptr, cperm, crow, (long)N, num_heads, scale, sd, dk, dv, \
                  ld, (const uint16_t*)q16, (const uint16_t*)g16);                                                                             \
  } while (0)
  if (b16 == 1) ATTN_BWD_GO(1);
  else if (b16 == 2) ATTN_BWD_GO(2);
  else ATTN_BWD_GO(0);
#undef ATTN_BWD_GO
  return acheck(who);
}

extern "C" int mgn_sparse_attn_fwd(const float* q, const float* k, const float* v, const int32_t* rowptr, const int32_t* col, int64_t N, int H,
                                   int num_heads, float* y, float* lse, void* stream) {
  return attn_fwd_any(0, q, k, v, rowptr, col, N, H, num_heads, y, lse, nullptr, attn_dense_ld(H), stream, "mgn_sparse_attn_fwd");
}

extern "C" int mgn_sparse_attn_bwd(const float* q, const float* k, const float* v, const float* y, const float* lse, const float* dy,
                                   const int32_t* rowptr, const int32_t* col, const int32_t* cptr, const int32_t* cperm,
                                   const int32_t* crow, int64_t N, int64_t E, int H, int num_heads, float* dq, float* dk, float* dv,
                                   float* ws, size_t ws_bytes, void* stream) {
  return attn_bwd_any(false, q, k, v, y, lse, dy, rowptr, col, cptr, cperm, crow, N, E, H, num_heads, dq, dk, dv, ws, ws_bytes, attn_dense_ld(H), stream,
                      "mgn_sparse_attn_bwd");
}

extern "C" int mgn_sparse_attn_fwd_b16(const float* q, const uint16_t* k16, const uint16_t* v16, const int32_t* rowptr, const int32_t* col,
                                       int64_t N, int H, int num_heads, float* y, float* lse, float* y_raw, void* stream) {
  return attn_fwd_any(1, q, k16, v16, rowptr, col, N, H, num_heads, y, lse, y_raw, attn_dense_ld(H), stream, "mgn_sparse_attn_fwd_b16");
}

extern "C" int mgn_sparse_attn_bwd_b16(const float* q, const uint16_t* k16, const uint16_t* v16, const float* y, const float* lse,
                                       const float* dy, const int32_t* rowptr, const int32_t* col, const int32_t* cptr, const int32_t* cperm,
                                       const int32_t* crow, int64_t N, int64_t E, int H, int num_heads, float* dq, float* dk, float* dv,
                                       float* ws, size_t ws_bytes, void* stream) {
  return attn_bwd_any(true, q, k16, v16, y, lse, dy, rowptr, col, cptr, cperm, crow, N, E, H, num_heads, dq, dk, dv, ws, ws_bytes, attn_dense_ld(H), stream,
                      "mgn_sparse_attn_bwd_b16");
}

// Strided forms: q / k / v as column slabs of one projection output (ldq / ldk / ldv = row pitches in elements; kv_bf16 = 1: k / v are
// bf16 rows and the bf16-mode roundings apply, as in the *_b16 pair; [r5] kv_bf16 = 2: the same roundings on k / v kept as FP32 rows
// whose values are bf16 numbers -- the slabs of a bf16-mode projection as they are, no narrowing copy, and the gather path of the
// fp32 kernels, which is the faster one once the rows are cache-local), dq / dk / dv as slabs of its gradient.
extern "C" int mgn_sparse_attn_fwd_s(const float* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, int kv_bf16,
                                     const int32_t* rowptr, const int32_t* col, int64_t N, int H, int num_heads, float* y, float* lse,
                                     float* y_raw, void* stream) {
  const AttnLd ld{(int)ldq, (int)ldk, (int)ldv, H, H, H};
  if (kv_bf16 < 0 || kv_bf16 > 2) return afail(1, "mgn_sparse_attn_fwd_s: kv_bf16 must be 0, 1 or 2");
  return attn_fwd_any(kv_bf16, q, k, v, rowptr, col, N, H, num_heads, y, lse, y_raw, ld, stream, "mgn_sparse_attn_fwd_s");
}

extern "C" int mgn_sparse_attn_bwd_s(const float* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, int kv_bf16,
                                     const float* y, const float* lse, const float* dy, const int32_t* rowptr, const int32_t* col,
                                     const int32_t* cptr, const int32_t* cperm, const int32_t* crow, int64_t N, int64_t E, int H,
                                     int num_heads, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* ws,
                                     size_t ws_bytes, void* stream) {
  const AttnLd ld{(int)ldq, (int)ldk, (int)ldv, (int)lddq, (int)lddk, (int)lddv};
  if (kv_bf16 < 0 || kv_bf16 > 2) return afail(1, "mgn_sparse_attn_bwd_s: kv_bf16 must be 0, 1 or 2");
  return attn_bwd_any(kv_bf16, q, k, v, y, lse, dy, rowptr, col, cptr, cperm, crow, N, E, H, num_heads, dq, dk, dv, ws, ws_bytes, ld,
                      stream, "mgn_sparse_attn_bwd_s");
}

extern "C" int mgn_sparse_attn_weights(const float* q, const float* k, const float* lse, const int32_t* rowptr, const int32_t* col,
                                       const int32_t* out_pos, int64_t N, int H, int num_heads, float* attn, void* stream) {
  if (!attn_args_ok(N, H, num_heads) || lse == nullptr || attn == nullptr) return afail(1, "mgn_sparse_attn_weights: bad arguments");
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)(H / num_heads));
  ATTN_DISPATCH(k_attn_weights, 0, q, k, lse, rowptr, col, out_pos, (long)N, num_heads, scale, attn);
  return acheck("mgn_sparse_attn_weights");
}

// =====================================================================================================================
// [r6] Attention over the HEAD axis of one node: what the reference's scaled_dot_product_attention computes when it is
// handed no adjacency (layers.py:493-559 with att_mask = None: attn = softmax(q k^T / sqrt(d)) over the LAST axis of
// [N, d, d], y = attn v) -- the TemporalAttention of an installation without DGL (processors.py:203-209,376-377 pass
// adj = None).  q / k / v rows are [d, NH] (feature f = i * NH + h): S[i][j] = sum_h q[i][h] k[j][h] / sqrt(d), softmax
// over j, y[i][h] = sum_j P[i][j] v[j][h]; no node reads another node's rows.  A few hundred flops per output value on
// rows that are read once: one thread per (node, i), the node's k / v rows broadcast out of LDS, fixed summation order
// (forward and backward are bit-reproducible; the backward's column pass recomputes the weights instead of adding
// atomically).
// =====================================================================================================================
#define HAX_THREADS 128

template <int NH>
__global__ __launch_bounds__(HAX_THREADS) void k_head_axis_attn_fwd(const float* __restrict__ q, const float* __restrict__ k,
                                                                    const float* __restrict__ v, long N, int d, float sd,
                                                                    float* __restrict__ y, float* __restrict__ lse) {
  extern __shared__ float hax_lds[];
  const int H = d * NH, nb = HAX_THREADS / d;
  const long n0 = (long)blockIdx.x * nb;
  const int nn = (int)min((long)nb, N - n0);
  float* K = hax_lds;
  float* V = hax_lds + nb * H;
  for (int t = threadIdx.x; t < nn * H; t += HAX_THREADS) {
    K[t] = k[n0 * H + t];
    V[t] = v[n0 * H + t];
  }
  __syncthreads();
  const int ln = threadIdx.x / d, i = threadIdx.x - ln * d;
  if (ln >= nn) return;
  const long n = n0 + ln;
  float qi[NH], acc[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    qi[h] = q[n * H + i * NH + h] / sd;   // layers.py:509-510: the query is divided first
    acc[h] = 0.f;
  }
  const float* Kn = K + ln * H;
  const float* Vn = V + ln * H;
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < d; ++j) {
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) s = fmaf(qi[h], Kn[j * NH + h], s);
    const float mn = fmaxf(m, s), c = expf(m - mn), p = expf(s - mn);
    l = l * c + p;
#pragma unroll
    for (int h = 0; h < NH; ++h) acc[h] = fmaf(p, Vn[j * NH + h], acc[h] * c);
    m = mn;
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int h = 0; h < NH; ++h) y[n * H + i * NH + h] = acc[h] * inv;
  lse[n * d + i] = m + logf(l);
}

template <int NH>
__global__ __launch_bounds__(HAX_THREADS) void k_head_axis_attn_bwd(const float* __restrict__ q, const float* __restrict__ k,
                                                                    const float* __restrict__ v, const float* __restrict__ y,
                                                                    const float* __restrict__ lse, const float* __restrict__ dy, long N,
                                                                    int d, float sd, float* __restrict__ dq, float* __restrict__ dk,
                                                                    float* __restrict__ dv) {
  extern __shared__ float hax_lds[];
  const int H = d * NH, nb = HAX_THREADS / d;
  const long n0 = (long)blockIdx.x * nb;
  const int nn = (int)min((long)nb, N - n0);
  float* Q = hax_lds;                 // the SCALED queries
  float* K = Q + nb * H;
  float* V = K + nb * H;
  float* G = V + nb * H;              // dy
  float* L = G + nb * H;              // lse  [nb][d]
  float* D = L + nb * d;              // sum_h dy[i][h] y[i][h]  [nb][d]
  for (int t = threadIdx.x; t < nn * H; t += HAX_THREADS) {
    Q[t] = q[n0 * H + t] / sd;
    K[t] = k[n0 * H + t];
    V[t] = v[n0 * H + t];
    G[t] = dy[n0 * H + t];
  }
  const int ln = threadIdx.x / d, i = threadIdx.x - ln * d;
  const bool live = ln < nn;
  const long n = n0 + ln;
  if (live) {
    float dsum = 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) dsum = fmaf(dy[n * H + i * NH + h], y[n * H + i * NH + h], dsum);
    D[ln * d + i] = dsum;
    L[ln * d + i] = lse[n * d + i];
  }
  __syncthreads();
  if (!live) return;
  const float* Qn = Q + ln * H;
  const float* Kn = K + ln * H;
  const float* Vn = V + ln * H;
  const float* Gn = G + ln * H;
  {  // row pass: dq[i] = sum_j dS[i][j] k[j] / sqrt(d)
    float qi[NH], gi[NH], a[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      qi[h] = Qn[i * NH + h];
      gi[h] = Gn[i * NH + h];
      a[h] = 0.f;
    }
    const float li = L[ln * d + i], di = D[ln * d + i];
    for (int j = 0; j < d; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        s = fmaf(qi[h], Kn[j * NH + h], s);
        dp = fmaf(gi[h], Vn[j * NH + h], dp);
      }
      const float ds = expf(s - li) * (dp - di);
#pragma unroll
      for (int h = 0; h < NH; ++h) a[h] = fmaf(ds, Kn[j * NH + h], a[h]);
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) dq[n * H + i * NH + h] = a[h] / sd;
  }
  {  // column pass (this thread is column j = i): dk[j] = sum_i dS[i][j] q_scaled[i], dv[j] = sum_i P[i][j] dy[i]
    const int j = i;
    float kj[NH], vj[NH], ak[NH], av[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      kj[h] = Kn[j * NH + h];
      vj[h] = Vn[j * NH + h];
      ak[h] = 0.f;
      av[h] = 0.f;
    }
    for (int r = 0; r < d; ++r) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        s = fmaf(Qn[r * NH + h], kj[h], s);
        dp = fmaf(Gn[r * NH + h], vj[h], dp);
      }
      const float p = expf(s - L[ln * d + r]), ds = p * (dp - D[ln * d + r]);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        ak[h] = fmaf(ds, Qn[r * NH + h], ak[h]);
        av[h] = fmaf(p, Gn[r * NH + h], av[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      dk[n * H + j * NH + h] = ak[h];   // Q holds q / sqrt(d) already
      dv[n * H + j * NH + h] = av[h];
    }
  }
}

static int hax_args_ok(int64_t N, int H, int NH) {
  if (N < 0 || H <= 0 || H > 1024) return 0;
  if (!(NH == 1 || NH == 2 || NH == 4 || NH == 8 || NH == 16) || H % NH != 0) return 0;
  return H / NH <= HAX_THREADS;
}

#define HAX_LAUNCH(KERNEL, LDS, ...)                                                                            \
  switch (num_heads) {                                                                                          \
    case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(grid), dim3(HAX_THREADS), LDS, s, __VA_ARGS__); break;         \
    case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(grid), dim3(HAX_THREADS), LDS, s, __VA_ARGS__); break;         \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(grid), dim3(HAX_THREADS), LDS, s, __VA_ARGS__); break;         \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(grid), dim3(HAX_THREADS), LDS, s, __VA_ARGS__); break;         \
    default: hipLaunchKernelGGL((KERNEL<16>), dim3(grid), dim3(HAX_THREADS), LDS, s, __VA_ARGS__); break;       \
  }

extern "C" int mgn_head_axis_attn_fwd(const float* q, const float* k, const float* v, int64_t N, int H, int num_heads, float* y,
                                      float* lse, void* stream) {
  if (!hax_args_ok(N, H, num_heads) || (N > 0 && (!q || !k || !v || !y || !lse)))
    return afail(1, "mgn_head_axis_attn_fwd: num_heads must be 1/2/4/8/16 and divide hidden, hidden / num_heads <= 128");
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int d = H / num_heads, nb = HAX_THREADS / d;
  const unsigned grid = (unsigned)((N + nb - 1) / nb);
  const size_t lds = (size_t)2 * nb * H * sizeof(float);
  HAX_LAUNCH(k_head_axis_attn_fwd, lds, q, k, v, (long)N, d, sqrtf((float)d), y, lse);
  return acheck("mgn_head_axis_attn_fwd");
}

extern "C" int mgn_head_axis_attn_bwd(const float* q, const float* k, const float* v, const float* y, const float* lse, const float* dy,
                                      int64_t N, int H, int num_heads, float* dq, float* dk, float* dv, void* stream) {
  if (!hax_args_ok(N, H, num_heads) || (N > 0 && (!q || !k || !v || !y || !lse || !dy || !dq || !dk || !dv)))
    return afail(1, "mgn_head_axis_attn_bwd: num_heads must be 1/2/4/8/16 and divide hidden, hidden / num_heads <= 128");
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int d = H / num_heads, nb = HAX_THREADS / d;
  const unsigned grid = (unsigned)((N + nb - 1) / nb);
  const size_t lds = ((size_t)4 * nb * H + 2 * nb * d) * sizeof(float);
  HAX_LAUNCH(k_head_axis_attn_bwd, lds, q, k, v, y, lse, dy, (long)N, d, sqrtf((float)d), dq, dk, dv);
  return acheck("mgn_head_axis_attn_bwd");
}
