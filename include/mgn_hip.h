/*
 * mgn_hip.h -- C ABI of the MI355X (gfx950) MeshGraphNet message-passing engine.
 *
 * Every entry point takes raw DEVICE pointers, plain sizes and a hipStream_t
 * (passed as void*), returns 0 on success / non-zero on error (never throws
 * across the boundary; mgn_last_error() gives the text), allocates nothing and
 * keeps no state: the caller owns outputs and workspaces.  All launches are
 * asynchronous on `stream` except mgn_csr_build (one-time topology prep, which
 * synchronises the stream to report malformed indices).
 *
 * All matrices are row-major fp32.  H (latent width) must be 16, 32, 64 or 128.
 * "T-layout" / "N-layout" in the comments are register layouts, see DESIGN.md.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   mgn_mlp_fwd        build_mlp(...) forward incl. RMSNorm
 *                        graphphysics/models/layers.py:163-210, :104-129
 *                      fused with the gathers/concats/residuals of
 *                        GraphNetBlock.forward / edge_update / update
 *                        graphphysics/models/layers.py:1015-1028,1039-1040,1044-1060,1074-1102
 *   mgn_segsum         MessagePassing.propagate(aggr="add") -> scatter-add
 *                        graphphysics/models/layers.py:926,1031-1037
 *                        (torch-geometric==2.6.1, requirements.txt:7)
 *   mgn_mlp_bwd        autograd backward of the above (activation/norm/dgrad chain)
 *   mgn_wgrad          autograd backward of nn.Linear wrt weight (dW = dZ^T X)
 *   mgn_csr_build      the dst-sorted edge order that replaces PyG's index-based
 *                        scatter (no reference counterpart: layout prep)
 *   mgn_transpose_blocks  layout prep for mgn_mlp_bwd (autograd's implicit W^T of nn.Linear)
 *   mgn_wpack          layout prep: nn.Linear weights (or their transposes) as bf16x3 MFMA images
 */
#ifndef MGN_HIP_H
#define MGN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGN_MAX_LAYERS 8   /* Linear layers per MLP (reference uses 4) */
#define MGN_MAX_PHASES 3   /* concatenated input blocks of the first layer */
#define MGN_MAX_WGRAD_JOBS 48   /* [r5] 12 before: the jobs of four rounds of the default block in one launch (ops.py, MGN_WGRAD_BATCH_MB) */

/* ABI version: major*10000 + minor*100 + patch. */
int mgn_version(void);
/* Text of the last error on this thread ("" if none). */
const char* mgn_last_error(void);

/* ------------------------------------------------------------------ CSR build
 * Stable counting sort of edge ids by key[e] in [0,N).
 *   rowptr[N+1] : segment offsets;  perm[E] : edge ids grouped by key, ascending
 *   inside a segment (the order CPU index_add_ sums in).
 * ws: at least mgn_csr_workspace_bytes(E,N) bytes of device scratch.
 * Returns 3 if any key is outside [0,N). Synchronises `stream`. */
size_t mgn_csr_workspace_bytes(int64_t E, int64_t N);
int mgn_csr_build(const int64_t* key, int64_t E, int64_t N, int32_t* rowptr, int32_t* perm,
                  void* ws, size_t ws_bytes, void* stream);

/* The whole topology of an edge_index in ONE call and ONE stream synchronisation: CSR by destination
 * (rowptr_dst / perm_dst as mgn_csr_build), the source / destination index of every dst-sorted edge row
 * (src_s / dst_s, int32), the CSR of those rows grouped by source (rowptr_src / perm_src), and on the host
 * the maximum in- and out-degree (max_degree_host[2], may be NULL) -- what the engine needs per mesh (a
 * new one every step under a shuffled loader).  Returns 3 on an index outside [0,N). */
size_t mgn_topology_workspace_bytes(int64_t E, int64_t N);
int mgn_topology_build(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int32_t* rowptr_dst, int32_t* perm_dst,
                       int32_t* src_s, int32_t* dst_s, int32_t* rowptr_src, int32_t* perm_src, int32_t* max_degree_host,
                       void* ws, size_t ws_bytes, void* stream);
/* The same build WITHOUT the host synchronisation (a shuffled loader hands the engine a new edge_index every step,
 * graphphysics/train.py:160-198: a per-step stream synchronisation costs ~1 ms of a 13 ms step).  flags_dev[4]
 * (device, int32) receives [0] 1 if an index was outside [0,N), [1] max in-degree, [2] max out-degree; the caller
 * copies it to the host asynchronously and reads it at its next natural wait.  Every array handed out is safe to
 * compute on even when [0] is set (edges with a stray index are dropped from both CSRs, rows past the valid count
 * repeat edge 0, stray node ids read node 0), so launches may be queued before the flag has been looked at. */
int mgn_topology_build_async(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int32_t* rowptr_dst, int32_t* perm_dst,
                             int32_t* src_s, int32_t* dst_s, int32_t* rowptr_src, int32_t* perm_src, int32_t* flags_dev,
                             void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- segment sum
 * out[i,:] = sum_{k=rowptr[i]}^{rowptr[i+1]-1} src[perm ? perm[k] : k, :]
 * summed sequentially in k order (deterministic, atomics-free).  H in {16,32,64,128}. */
int mgn_segsum(const float* src, const int32_t* rowptr, const int32_t* perm, float* out,
               int64_t N, int H, void* stream);

/* Second stage of the segment sum fused into mgn_mlp_fwd (seg_out / seg_part): N keys. */
int mgn_seg_fix(const int32_t* rowptr, const float* part, float* out, int64_t N, void* stream);

/* Two segment sums of the same source in one launch (H = 128): out0 over (rowptr0, perm0), out1
 * over (rowptr1, perm1) -- the backward scatter onto destination and source nodes. */
int mgn_segsum2(const float* src, const int32_t* rowptr0, const int32_t* perm0, float* out0,
                const int32_t* rowptr1, const int32_t* perm1, float* out1, int64_t N, int H, void* stream);
/* [r4] the same over TWO-BYTE source rows: dZ[0] of a backward launch with precision == 3 (128 bf16 values per row in the chain
 * kernels' packed feature order, mgn_mlp_fwd_args.precision); outputs fp32 in the true feature order. */
int mgn_segsum2_b16(const uint16_t* src, const int32_t* rowptr0, const int32_t* perm0, float* out0,
                    const int32_t* rowptr1, const int32_t* perm1, float* out1, int64_t N, int H, void* stream);

/* ------------------------------------------------------------ fused MLP forward
 * For each row m in [0,M):
 *   in  = cat_p src[p][ idx[p] ? idx[p][m] : m , 0:kw[p] ]      (p < nphase)
 *   z   = W[NL-1] act(... act(W[0] in + b[0]) ...) + b[NL-1]     (act = ReLU, or SiLU: field `act`)
 *   y   = scale ? scale * z / (||z||_2/sqrt(H) + eps) : z        (RMSNorm, layers.py:104-129)
 *   out = (resid ? resid[m] : 0) + y ;   y_out = y (optional)
 * W[0] is [Hout, ktot] with ktot = sum_p pad16(kw[p]) and phase p occupying columns
 * [sum_{q<p} pad16(kw[q]), +kw[p]) (zero padded); W[l>0] is [Hout, H].  The last
 * layer's W/b are padded to pad16(out_w) rows.  All weight pointers 16-byte aligned.
 * Optional saves for backward: saveH[l] = activation feeding layer l+1 ([M,H]),
 * saveU = z / (rms+eps) ([M,H]), saveR = rms ([M]).  Any of them may be NULL.
 */
typedef struct {
  int64_t M;
  int H, NL, nphase;
  const float* src[MGN_MAX_PHASES];
  const int32_t* idx[MGN_MAX_PHASES];
  int kw[MGN_MAX_PHASES];
  const float* W[MGN_MAX_LAYERS];
  const float* b[MGN_MAX_LAYERS];   /* NULL = no bias */
  const float* scale;               /* NULL = no RMSNorm */
  float eps;
  int out_w;                        /* width of the last layer (== H when scale != NULL) */
  const float* resid;               /* [M,out_w] or NULL */
  float* out;                       /* [M,out_w] */
  float* y_out;                     /* [M,out_w] or NULL */
  float* saveH[MGN_MAX_LAYERS];
  float* saveU;
  float* saveR;
  /* --- optional extensions (H = 128, full widths only; zero them when unused) ------------
   * ldw0   : leading dimension of W[0] when it is a column slab of a wider matrix (0 = ktot).
   * n_add  : layer-0 pre-activation += add_src[q][ add_idx[q] ? add_idx[q][m] : m , 0:H ]
   *          -- the algebraic split of the first edge layer: W0.[e,x_d,x_s] =
   *          W_e.e + (x W_d^T)[dst] + (x W_s^T)[src], the two projections being N-row products.
   * n_post : post_out[q][m,:] = post_W[q] . out[m,:]   (post_W[q] is [H,H], leading dim post_ldw)
   *          -- the next round's node projections, computed while `out` is still in registers. */
  int ldw0;
  int n_add;
  const float* add_src[2];
  const int32_t* add_idx[2];
  int n_post;
  int post_ldw;
  const float* post_W[2];
  float* post_out[2];
  /* --- split-bf16 matrix path (H = 128, full widths).  wpk[u] = the GEMM units of this launch
   * packed by mgn_wpack, in stream order: the nphase column slabs of W[0], W[1..NL-1], then the
   * n_post post_W blocks.  All of them non-NULL selects the bf16x3-operand / 6-term kernels
   * (fp32-grade accuracy at 2.67x the fp32 MFMA rate); W[] / post_W[] are then not read, b[] /
   * scale still are.  wpk[0] == NULL keeps the exact-fp32 MFMA kernels. */
  const void* wpk[8];
  /* saveM[l] ([M][4] uint32, optional, split-bf16 kernels only): sign bits of saveH[l] -- word g
   * of row m holds bit 4*ib + r for feature 16*ib + 4*g + r (set iff the activation is > 0).
   * The backward chain needs only these 16 bytes per row and layer, not the 512-byte fp32 row. */
  uint32_t* saveM[MGN_MAX_LAYERS];
  /* matrix precision of the packed path: 0 = fp32-grade (bf16x3 operands, 6 product terms);
   * 1 = bf16 (operands rounded to bf16, ONE term, fp32 accumulate; biases, RMSNorm, residuals
   * and every stored tensor stay fp32) -- the semantic of the reference under bf16-mixed
   * autocast (train.py:74-78,268-293), BASELINE configs[2].  [r4] Off the packed path (H != 128, ragged widths: the generic
   * kernels) precision = 1 rounds the row operands and every layer's result to bf16 in the kernel; the caller passes W[] / b[]
   * already rounded (mgn_mlp_bwd: WT[] / WT0[] likewise, and every dZ it writes is a bf16 value).
   * [r4] precision = 2 (packed path only): as 1, and saveH[l] receives TWO-BYTE rows -- the activation as the bf16 tensor the
   * reference's autocast holds, bit for bit what precision 1 would round at its next use -- 128 bf16 values per row in the
   * kernel's operand order (K-slice p at element 32 p; inside it, for g = 0..3: features 32p + 4g .. + 3, then
   * 32p + 16 + 4g .. + 3).  Only mgn_wgrad_p reads them back (ldb = -128). */
  int precision;
  /* out = act(z) instead of z (generic ragged-input kernel only; no norm / residual): lets an
   * encoder run its narrow first layer stand-alone and the three full layers on the packed path
   * (y_out, if given, still receives z: the pre-activation a SiLU backward needs). */
  int out_relu;
  /* Fused segment sum (split-bf16 kernel, no post-products): rows are sorted by seg_key[m] (the
   * CSR order), seg_rowptr[k] .. seg_rowptr[k+1] is the row range of key k.  The kernel adds up
   * y (before the residual) over each run of equal keys inside a 16-row wave tile; finished
   * segments go to seg_out[k,:], runs cut by a tile boundary to seg_part[tile][0|1][H]
   * (ceil(M/16) tiles) and mgn_seg_fix assembles those and zeroes the empty segments.  With it
   * the messages need not be written at all (y_out = NULL).  Fixed summation order (a 16-lane
   * scan, then tile order): deterministic, but not the sequential k-order of mgn_segsum. */
  const int32_t* seg_key;
  const int32_t* seg_rowptr;
  float* seg_out;
  float* seg_part;
  /* Activation between the layers: MGN_ACT_RELU (0) or MGN_ACT_SILU (1) -- the reference's
   * process-global switch (layers.py:132-160, JSON model.use_silu_activation).  SiLU is not
   * invertible, so its backward needs the PRE-activations: saveZ[l] ([M,H], optional) receives
   * the input of the activation that follows layer l (saveM is ReLU-only).  Supported by the
   * generic and the split-bf16 kernels (not by the exact-fp32 LDS generation). */
  int act;
  float* saveZ[MGN_MAX_LAYERS];
} mgn_mlp_fwd_args;
#define MGN_ACT_RELU 0
#define MGN_ACT_SILU 1
/* GELU (exact erf form, nn.GELU()): the reference's build_mlp(act="gelu") option (layers.py:150-160).  Generic kernels only
 * (any H, ragged widths; no packed weights, gathers or post-products); saveZ / Zs as for SiLU. */
#define MGN_ACT_GELU 2
int mgn_mlp_fwd(const mgn_mlp_fwd_args* args, void* stream);

/* ----------------------------------------------------- fused MLP backward chain
 * Given dY = dOut[m] (+ dOut2[idx2[m]]) = gradient wrt y above, computes
 *   dZ[NL-1] (through the RMSNorm), dZ[l-1] = (WT[l] dZ[l]) * (H_l > 0)   (l = NL-1..1)
 *   dIn[q]   = (din_resid[q] ? din_resid[q] : 0) + WT0[q] dZ[0]          (q < n_din)
 * and the column sums db[l] = sum_m dZ[l][m,:], dscale = sum_m dY*U.
 * WT[l] = W[l]^T, [H, pad16(out width of layer l)] row-major (l >= 1; WT[0] unused);
 * WT0[q] = (column slab q of W[0])^T, [H(in), H(out)] row-major.
 * dZ[l] are [M, pad16(width)] outputs that feed mgn_wgrad.
 * red_ws: device scratch of mgn_mlp_bwd_workspace_bytes(M,H,NL) bytes.
 */
typedef struct {
  int64_t M;
  int H, NL;
  const float* dOut;                /* [M,out_w] */
  const float* dOut2;               /* [*,H] gathered by idx2, or NULL */
  const int32_t* idx2;
  int out_w;
  const float* U; const float* R; const float* scale; float eps;   /* scale NULL = no norm */
  const float* Hs[MGN_MAX_LAYERS];  /* Hs[l-1] = saved input of layer l (l>=1) */
  const float* WT[MGN_MAX_LAYERS];
  float* dZ[MGN_MAX_LAYERS];
  int n_din;
  const float* WT0[MGN_MAX_PHASES];
  const float* din_resid[MGN_MAX_PHASES];
  float* dIn[MGN_MAX_PHASES];
  float* db[MGN_MAX_LAYERS];        /* [pad16(width)] or NULL */
  float* dscale;                    /* [H] or NULL */
  void* red_ws; size_t red_ws_bytes;
  /* split-bf16 matrix path (H = 128, full widths, n_din <= 1): the launch's GEMM units packed
   * by mgn_wpack(transpose = 1 of the forward weights) in stream order WT[NL-1], ..., WT[1],
   * then WT0[0] if n_din == 1 (preceded by the n_front front units, see below).  wpk[0] == NULL
   * keeps the exact-fp32 MFMA kernels. */
  const void* wpk[8];
  /* Ms[l-1] = saveM[l-1] of the forward launch (ReLU masks as bits).  Required by the split-bf16
   * kernel (it does not read Hs); ignored by the fp32 kernels. */
  const uint32_t* Ms[MGN_MAX_LAYERS];
  int precision;                    /* as in mgn_mlp_fwd_args; [r4] 2: as 1, and dZ[l] for l >= 1 are written as TWO-BYTE rows in the
                                     * forward's packed feature order (dZ[0] stays fp32); 3: dZ[0] too (for mgn_segsum2_b16 / lda = -128); packed path, ReLU,
                                     * no front stage */
  /* Optional front stage (split-bf16 kernel only; dOut / dOut2 are then ignored):
   *   dY[m] = (front_resid ? front_resid[m] : 0) + sum_{p < n_front} Wf_p . front_src[p][m]
   * with Wf_p = wpk[p] (packed, [H,H]); dY is stored to front_out (if not NULL) and feeds the
   * chain.  It fuses "dX = dX' + Wcat . [dZn0; Sd; Ss]" of round i with the node chain of
   * round i-1: same rows, one launch, no round trip of dX through HBM before its first use. */
  int n_front;
  const float* front_src[MGN_MAX_PHASES];
  const float* front_resid;
  float* front_out;
  /* != 0: leave the column sums (db[], dscale) as per-workgroup partials in red_ws -- which must
   * then stay alive -- and finish many launches at once with mgn_colred_batch. */
  int defer_reduce;
  /* activation as in mgn_mlp_fwd_args; for MGN_ACT_SILU Zs[l-1] = saveZ[l-1] of the forward launch
   * (pre-activations) replaces Hs / Ms:  dZ[l-1] = (WT[l] dZ[l]) * silu'(Zs[l-1]). */
  int act;
  const float* Zs[MGN_MAX_LAYERS];
  /* Fused segment sum of dZ[0] (split-bf16 kernel, fp32-grade / ReLU, no front stage; dZ[0] != NULL): the
   * rows are sorted by seg_key (the CSR order) and the kernel adds dZ[0] up over every run of equal keys
   * exactly as mgn_mlp_fwd does for its output (seg_out / seg_part, finish with mgn_seg_fix) -- the
   * backward scatter of the first-layer gradients onto DESTINATION nodes without re-reading dZ[0]. */
  const int32_t* seg_key;
  const int32_t* seg_rowptr;
  float* seg_out;
  float* seg_part;
} mgn_mlp_bwd_args;
size_t mgn_mlp_bwd_workspace_bytes(int64_t M, int H, int NL);
int mgn_mlp_bwd(const mgn_mlp_bwd_args* args, void* stream);

/* Deferred column reductions of mgn_mlp_bwd launches (defer_reduce != 0), all in one launch.
 * M, H, NL, out_w, n_din as in the launch that filled red_ws (they fix its partial count). */
typedef struct {
  const void* red_ws;
  int64_t M;
  int H, NL, out_w, n_din;
  float* db[MGN_MAX_LAYERS];
  float* dscale;
} mgn_colred_job;
int mgn_colred_batch(int n, const mgn_colred_job* jobs, void* stream);

/* ------------------------------------------------------------------ weight grads
 * For each job: dW[j, k] = sum_m A[m, j] * B[m, k],  j < 16*nja, k < 16*nkb
 * (B columns >= kw read as zero).  A is [M,lda], B is [M,ldb], dW is [16*nja, ldw].
 * db (optional, [16*nja]): db[j] = sum_m A[m, j] -- the bias gradient of the same Linear,
 * a by-product of reading A (= dZ) here.
 * ws: device scratch of mgn_wgrad_workspace_bytes(njobs, jobs) bytes.
 * [r4] ldb == -128 / lda == -128 (mgn_wgrad_p with precision 1, full 128 x 128 jobs only): B / A points to TWO-BYTE rows -- the
 * saves a forward launch with precision == 2 wrote, the dZ[1..] rows of a backward launch with precision == 2 (rows of 128 bf16
 * values in those launches' packed feature order, see mgn_mlp_fwd_args.precision). */
typedef struct {
  const float* A; const float* B; float* dW;
  int64_t M;
  int lda, ldb, ldw;
  int nja, nkb, kw;
  float* db;
} mgn_wgrad_job;
size_t mgn_wgrad_workspace_bytes(int njobs, const mgn_wgrad_job* jobs);
int mgn_wgrad(int njobs, const mgn_wgrad_job* jobs, void* ws, size_t ws_bytes, void* stream);
/* same with an explicit matrix precision for the full 128x128 jobs (0 / 1 as in mgn_mlp_fwd_args) */
int mgn_wgrad_p(int njobs, const mgn_wgrad_job* jobs, void* ws, size_t ws_bytes, int precision, void* stream);

/* ------------------------------------------------ fused edge backward of a round (chain + weight gradients)
 * The backward of GraphNetBlock's edge update (graphphysics/models/layers.py:1044-1060 under autograd: edge_block =
 * build_mlp :163-210, 4 Linear / ReLU / RMSNorm :104-129, residual :1039) for the dst-sorted edge rows, as ONE kernel:
 *   dY   = dOut + dAgg[idx]                     (gradient of e' = e + m and of agg = sum m, layers.py:1031-1040)
 *   dZ3  = RMSNorm backward of dY (U, R, scale);   dZ_{l-1} = (W_l^T dZ_l) masked by Ms[l-1]   (l = 3, 2, 1)
 *   dIn  = dOut + We0^T dZ0                       (the input gradient wrt e)
 *   dW_l = dZ_l^T X_l,  db_l = column sums of dZ_l   (X_0 = the round's input e, X_l = saved activation H_l),
 *   dscale = sum_rows dY * U;   dZ0 is also stored (the node-side scatters of the split first layer read it).
 * It replaces mgn_mlp_bwd (edge chain) + the four E-row jobs of mgn_wgrad of a round; dZ1..dZ3 never reach memory.
 * wpk[0..3]: packed W3^T, W2^T, W1^T, We0^T (mgn_wpack).  dW[l]: [128, ldw[l]] destinations (dW[0] is the first slab
 * of the [128, 384] first-layer gradient: ldw[0] = 384).  ws: mgn_edge_bwd_fused_workspace_bytes() bytes of device
 * scratch (per-workgroup partials, summed in a fixed order: deterministic).  H = 128, 4 layers, ReLU only.
 * precision as in mgn_mlp_fwd_args. */
typedef struct {
  int64_t M;
  const float* dOut;
  const float* dAgg;
  const int32_t* idx;
  const float* U;
  const float* R;
  const float* scale;
  float eps;
  const float* X[4];
  const uint32_t* Ms[3];
  const void* wpk[4];
  float* dIn;
  float* dZ0;
  float* dW[4];
  int ldw[4];
  float* db[4];
  float* dscale;
  void* ws;
  size_t ws_bytes;
  int precision;
} mgn_edge_bwd_fused_args;
size_t mgn_edge_bwd_fused_workspace_bytes(void);
int mgn_edge_bwd_fused(const mgn_edge_bwd_fused_args* args, void* stream);

/* Diagnostic: resident workgroups per CU (HIP occupancy query) of the persistent kernels at their launch
 * configuration -- out[0..5] = forward chain (fp32-grade, 4 waves), forward chain (bf16), backward chain
 * with one column-sum slot, backward chain with five, weight gradients, forward chain (8 waves). */
int mgn_debug_occupancy(int* out);

/* ------------------------------------------------------- batched block transpose
 * dst[k, j] = src[j, k] for n square H x H blocks (leading dimensions ld_src / ld_dst; a block
 * may be a column slab of a wider matrix).  Prepares the W^T operands of mgn_mlp_bwd for all
 * rounds in one or two launches instead of ~10 small copies per round. */
typedef struct {
  const float* src; float* dst;
  int ld_src, ld_dst;
} mgn_tblock;
int mgn_transpose_blocks(int n, const mgn_tblock* blocks, int H, void* stream);

/* ------------------------------------------------------- split-bf16 weight packing
 * Packs n 128x128 fp32 blocks (src[o*ld_src + f], or src[f*ld_src + o] when transpose != 0;
 * a block may be a slab of a wider matrix) into the 96 KB bf16x3 image the split-bf16 kernels
 * stream through LDS: [K-slice 0..3][piece 0..2][out block 0..7][lane 0..63][8 bf16], where
 * w = w1 + w2 + w3 (bf16 each, round to nearest) and element i of lane (c,g) is
 * W[16*ob + c][32*j + 16*(i>>2) + 4*g + (i&3)].  dst must be 16-byte aligned. */
#define MGN_WPACK_BYTES 98304
typedef struct {
  const float* src; void* dst;
  int ld_src, transpose;
} mgn_wpack_block;
int mgn_wpack(int n, const mgn_wpack_block* blocks, void* stream);

/* ================================================================================
 * Either side of the path -- csrc/mgn_prep.hip
 * ================================================================================ */

/* ------------------------------------------------------------- faces -> edges (N2)
 * T.FaceToEdge(remove_faces=False) + to_undirected of the reference's preprocessing
 * (graphphysics/dataset/preprocessing.py:421-424; torch-geometric==2.6.1): face is [K,F]
 * int64 (K = 3 triangles / 4 tetrahedra, PyG layout); every pair of corners in both directions,
 * coalesced = sorted by (src,dst), duplicates and self loops removed.  src / dst have capacity
 * F*K*(K-1); *n_edges (device) receives E.  Synchronises `stream`; returns 3 on a corner
 * outside [0,N).  Needs N*N < 2^63. */
size_t mgn_faces_to_edges_workspace_bytes(int64_t F, int K);
int mgn_faces_to_edges(const int64_t* face, int K, int64_t F, int64_t N, int64_t* src, int64_t* dst,
                       int64_t* n_edges, void* ws, size_t ws_bytes, void* stream);

/* --------------------------------------------------------------- world edges (N2)
 * add_world_edges (preprocessing.py:92-140): node pairs whose world positions
 * x[:, pos_start:pos_start+D] are within `radius` (<=, evaluated in double like scipy's
 * cKDTree.query_pairs) with one end OBSTACLE and the other NORMAL, in both directions, merged
 * with the given edges -- taken in both directions, as to_undirected does -- and coalesced (sorted
 * by (src,dst), unique).  src / dst have capacity 2*E + 2*max_world_pairs + 1; returns 4 if more pairs than max_world_pairs match, 3 on an edge index
 * outside [0,N).  Synchronises `stream`. */
#define MGN_NODE_OBSTACLE 1
size_t mgn_world_edges_workspace_bytes(int64_t E, int64_t max_world_pairs);
int mgn_add_world_edges(const float* x, int x_w, int pos_start, int D, int type_idx, int64_t N, double radius,
                        const int64_t* src_in, const int64_t* dst_in, int64_t E, int64_t max_world_pairs,
                        int64_t* src, int64_t* dst, int64_t* n_edges, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------- edge features (N2)
 * T.Cartesian(norm=False) + T.Distance(norm=False) (preprocessing.py:16-23):
 * edge_attr[e] = [pos[src]-pos[dst] (D values), ||pos[dst]-pos[src]||_2],  D = 2 or 3. */
int mgn_edge_features(const float* pos, int D, const int64_t* src, const int64_t* dst, int64_t E,
                      float* edge_attr, void* stream);

/* ------------------------------------------------------------ locality renumbering of the nodes
 * No reference counterpart (PyTorch gathers x[col], x[row] in whatever numbering the dataset has,
 * graphphysics/models/layers.py:1017-1018): the engine may renumber the nodes of a mesh along a Morton curve of
 * their positions so that the 512-byte rows the edge kernels gather sit close together, and un-does it on exit
 * (ops.Topology(renumber=...), processors.EncodeProcessDecode.forward).  pos: [N, ld] floats, the first D (2 or 3)
 * columns are coordinates.  order[i] = old id at new position i (ties by old id), rank[old] = new.  Everything on
 * the device, no host synchronisation. */
size_t mgn_morton_order_workspace_bytes(int64_t N);
int mgn_morton_order(const float* pos, int ld, int D, int64_t N, int32_t* order, int32_t* rank, void* ws, size_t ws_bytes,
                     void* stream);

/* ------------------------------------------------------------ noise injection (N2)
 * add_noise (graphphysics/dataset/preprocessing.py:177-238, wired by build_preprocessing :421-435):
 * x[n, c] += scales[r] * z for every NORMAL node n and column c in [starts[r], ends[r]), r < n_ranges
 * (<= 8), in place; z ~ N(0,1) from a counter-based stream any implementation can reproduce:
 *   Philox4x32-10(key = seed, counter = (n_lo, n_hi, c | r << 16, offset)) -> r0, r1;
 *   u1 = ((r0 >> 8) + 1) 2^-24, u2 = (r1 >> 8) 2^-24, z = sqrt(-2 ln u1) cos(2 pi u2).
 * `offset` = the call / training-step counter.  The curriculum scale 10*std*(1+cos(t*pi)) (:226) is
 * applied by the caller to scales[]. */
int mgn_add_noise(float* x, int x_w, int64_t N, int n_ranges, const int* starts, const int* ends, const float* scales,
                  int type_idx, uint64_t seed, uint32_t offset, void* stream);

/* ------------------------------------------------- Simulator pre / post processing (N1)
 * Simulator._build_input_graph / build_outputs (graphphysics/models/simulator.py:112-191) with
 * the three Normalizers (models/layers.py:331-391) in one pass per tensor.  Streams:
 *   0 node features  cat[x[:, feat_start:feat_end], one_hot(x[:, type_idx], 9)]
 *   1 target delta   y[:, 0:out_w] - x[:, out_start:out_start+out_w]
 *   2 edge features  edge_attr
 * accumulate[s] != 0 (and *num_acc[s] < max_accumulations): first add this batch's column sums /
 * sums of squares / row count to the running buffers (two-stage, atomics-free), then normalise
 * with the updated statistics -- Normalizer.forward's order.  out pointers may be NULL (stream
 * skipped).  Returns 1 when a width is inconsistent: feat_end / out_start+out_w / type_idx beyond
 * x_w, out_w beyond y_w, or a stream width different from norm_w[s]. */
#define MGN_NODE_TYPES 9     /* NodeType.SIZE (graphphysics/utils/nodetype.py) */
#define MGN_NODE_NORMAL 0
#define MGN_NODE_OUTFLOW 5
typedef struct {
  int64_t N, E;
  const float* x; int x_w;
  const float* y; int y_w;
  const float* edge_attr; int edge_w;
  int feat_start, feat_end, out_start, out_w, type_idx;
  float* acc_sum[3]; float* acc_sumsq[3]; float* acc_count[3]; float* num_acc[3];
  int accumulate[3];
  float std_eps;
  float* node_out; float* target_out; float* edge_out;
  /* widths of the three normalisers' acc_sum / acc_sumsq buffers; mgn_sim_pre rejects (code 1)
   * a stream whose width differs -- the reference raises a shape error there. */
  int norm_w[3];
  /* Normalizer._max_accumulations (layers.py:311,345-349): accumulate[s] asks for accumulation,
   * the kernels decide from the DEVICE counter *num_acc[s] < max_accumulations -- a hipGraph
   * replay of a captured training step therefore stops accumulating like the reference does. */
  float max_accumulations;
  /* optional device int: set to 1 when a node-type code is outside [0, MGN_NODE_TYPES) (F.one_hot
   * raises there); never cleared by the engine. */
  int* type_err;
} mgn_sim_desc;
size_t mgn_sim_workspace_bytes(void);
int mgn_sim_pre(const mgn_sim_desc* desc, void* ws, size_t ws_bytes, void* stream);
/* pred = x[:, out_start:+O] + (net_out * std + mean) of the output normaliser; mask_truth != 0
 * re-imposes y on the nodes whose type is not NORMAL / OUTFLOW (rollout,
 * training/lightning_module.py:27-35,375-409). */
int mgn_sim_post(const float* x, int x_w, int out_start, int type_idx, const float* y, int y_w,
                 const float* net_out, int O, const float* acc_sum, const float* acc_sumsq,
                 const float* acc_count, float std_eps, int mask_truth, int64_t N, float* pred, void* stream);
/* ------------------------------------------- fused gradient clipping + AdamW (harness, R8)
 * Trainer(gradient_clip_val=1.0) (train.py:288) + AdamW(betas=(0.9,0.95), weight_decay=1e-4)
 * (training/lightning_module.py:494-511) over n tensors in 2 * ceil(n/96) launches:
 *   norm = sqrt(sum_t ||g_t||^2);  g *= min(1, max_norm / (norm + 1e-6))   (max_norm <= 0: no clip)
 *   *step += 1;  p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   p -= (lr / (1 - b1^step)) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
 * lr and step are DEVICE scalars (a hipGraph replay reads their current values); *grad_norm_out
 * (device, optional) receives the pre-clip norm.  The tensor table is a HOST array. */
typedef struct {
  float* p; float* g; float* m; float* v;
  int64_t n;
} mgn_opt_tensor;
size_t mgn_clip_adamw_workspace_bytes(int n, const mgn_opt_tensor* tensors);
int mgn_clip_adamw(int n, const mgn_opt_tensor* tensors, float max_norm, const float* lr, float* step,
                   float beta1, float beta2, float eps, float weight_decay, float* grad_norm_out,
                   void* ws, size_t ws_bytes, void* stream);
/* [r3] The loss of the training step (graphphysics/training/loss.py:70-75 with the node-type masks of lightning_module.py:27-35):
 *   loss = mean over the rows whose type (type[n * ldty], compared as a float like the reference's `node_type == NodeType.X`) is one
 *   of types[0..ntypes) (ntypes <= 4), and over their O columns, of (out - target)^2
 * -- two launches (part: 512 floats of scratch; loss and inv = 1 / (rows x O) are device scalars), one launch for
 * d_out = g * 2 (out - target) * w * inv.  `types` is a HOST array.  Deterministic (fixed-order partials). */
int mgn_masked_mse_fwd(const float* out, int ldo, const float* target, int ldt, const float* type, int ldty, int64_t N, int O,
                       const float* types, int ntypes, float* part, float* loss, float* inv, void* stream);
int mgn_masked_mse_bwd(const float* out, int ldo, const float* target, int ldt, const float* type, int ldty, int64_t N, int O,
                       const float* types, int ntypes, const float* inv, const float* g, float* d_out, void* stream);
/* [r3] The same tail in TWO launches whatever the number of tensors (up to 480): the fields that do not change from step to step
 * (p, m, v, n) live in a device table built once per parameter set -- mgn_clip_adamw_table: a BLOCKING host-to-device copy, not to be
 * called under stream capture; table_bytes from mgn_clip_adamw_table_bytes (0: too many tensors, use mgn_clip_adamw) -- and
 * mgn_clip_adamw_t takes the step's gradient pointers (tensors[i].g; the other fields must equal the table's) as kernel arguments.
 * Same arithmetic in the same order as mgn_clip_adamw: bit-identical results. */
size_t mgn_clip_adamw_table_bytes(int n, const mgn_opt_tensor* tensors);
int mgn_clip_adamw_table(int n, const mgn_opt_tensor* tensors, void* table, size_t table_bytes);
int mgn_clip_adamw_t(int n, const mgn_opt_tensor* tensors, const void* table, float max_norm, const float* lr, float* step,
                     float beta1, float beta2, float eps, float weight_decay, float* grad_norm_out,
                     void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------- halo exchange
 * One-hop halo of the node-partitioned large mesh (no reference counterpart: the reference's
 * Cluster-GCN sub-meshing, graphphysics/utils/torch_graph.py:108-135, drops the cut edges).
 * mgn_gather_rows: out[i,:] = src[idx[i],:], i < n -- packs the rows a peer needs.
 * mgn_halo_unpack_add: dst[nodes[j],:] += sum_{k in [rowptr[j], rowptr[j+1])} rows[perm[k],:], k
 * ascending -- the backward of the exchange: ghost-row gradients summed into their owners in a
 * fixed order (the send list grouped by node; atomics-free => bit-deterministic).  H % 4 == 0. */
int mgn_gather_rows(const float* src, const int32_t* idx, int64_t n, int H, float* out, void* stream);
int mgn_halo_unpack_add(const float* rows, const int32_t* nodes, const int32_t* rowptr, const int32_t* perm,
                        int64_t n_nodes, int H, float* dst, void* stream);

/* ------------------------------------------------- GraphNetBlock variants
 * Sigmoid gate on the aggregated messages (graphphysics/models/layers.py:1091-1098,
 * JSON model.use_gated_attention):  gate = sigmoid(G + phi[n] * gate_pos[j]) with G = gate_proj(x)
 * (an mgn_mlp_fwd launch), agg_out = agg * gate.  phi / gate_pos may be NULL (no positional term),
 * gate_out may be NULL (inference).  Backward: dAgg = dAggG * gate (may alias dAggG),
 * dG = dAggG * agg * gate * (1 - gate). */
int mgn_gate_fwd(const float* G, const float* phi, const float* gate_pos, const float* agg, int64_t N, int H,
                 float* gate_out, float* agg_out, void* stream);
int mgn_gate_bwd(const float* dAggG, const float* agg, const float* gate, int64_t N, int H, float* dAgg, float* dG,
                 void* stream);
/* Relative RoPE on the source-node features (layers.py:1020-1026,1104-1149, JSON
 * model.use_rope_embeddings): out[k,:] = RoPE(x[src[k],:], pos[src[k]] - pos[dst[k]]); the first
 * 2*pair_count*axes channels are rotated pairwise, pair i of axis a by (delta pos)[a] * inv_freq[i]
 * (inv_freq: the block's _rope_inv_freq buffer, layers.py:972-976), the rest pass through.
 * mgn_rope_scatter is its transpose summed over the edges of each source node (src-grouped CSR over
 * the same edge rows): out[j,:] = resid[j,:] + sum_{k: src[k] = j} RoPE^T(T[k,:]), fixed order. */
int mgn_rope_gather(const float* x, const float* pos, int pos_w, const float* inv_freq, int pair_count, int axes,
                    const int32_t* src, const int32_t* dst, int64_t E, int H, float* out, void* stream);
int mgn_rope_scatter(const float* T, const float* pos, int pos_w, const float* inv_freq, int pair_count, int axes,
                     const int32_t* src, const int32_t* dst, const int32_t* rowptr_src, const int32_t* perm_src,
                     int64_t N, int H, const float* resid, float* out, void* stream);

/* text of the last error of the entry points in this section */
const char* mgn_prep_last_error(void);

/* ================================================================================
 * Sparse-attention Transformer processor -- csrc/mgn_attn.hip
 * ================================================================================
 * Edge-masked scaled dot-product attention (graphphysics/models/layers.py:493-559 through DGL's
 * bsddmm -> SparseMatrix.softmax -> bspmm over dglsp.spmatrix(indices=edge_index), processors.py:352):
 *   score[e,h] = sum_d q[i_e,d,h] k[j_e,d,h] / sqrt(D);  attn = softmax over the edges of row i;
 *   y[i,d,h]   = sum_{e in row i} attn[e,h] v[j_e,d,h]
 * q, k, v, y are [N, H] row-major with the reference's head layout reshape(N, head_dim, num_heads)
 * (layers.py:673-675): feature f = d * num_heads + h.  The edges come as a CSR grouped by ROW
 * (rowptr[N+1], col[E] = j_e; rows = edge_index[0], columns = edge_index[1]).  lse ([N,H], optional)
 * receives per feature the log-sum-exp of its head's scores (what the backward needs).  Rows without
 * edges produce zeros.  H in {16,32,64,128}; num_heads in {1,2,4,8,16} dividing H.
 * Backward: dq, dk, dv from dy; cptr[N+1] / cperm[E] group the same (row-sorted) edge positions by
 * COLUMN, crow[E] = the row of the t-th edge of that column-grouped order (= i of edge cperm[t]; [r3]: it was
 * row_of_edge[E] indexed by the row-sorted edge before -- one dependent load more per edge); ws = 2 * E * num_heads
 * floats.  Deterministic, atomics-free. */
int mgn_sparse_attn_fwd(const float* q, const float* k, const float* v, const int32_t* rowptr, const int32_t* col,
                        int64_t N, int H, int num_heads, float* y, float* lse, void* stream);
int mgn_sparse_attn_bwd(const float* q, const float* k, const float* v, const float* y, const float* lse, const float* dy,
                        const int32_t* rowptr, const int32_t* col, const int32_t* cptr, const int32_t* cperm,
                        const int32_t* crow, int64_t N, int64_t E, int H, int num_heads,
                        float* dq, float* dk, float* dv, float* ws, size_t ws_bytes, void* stream);
/* bf16 matrix mode of the same pair ([r4]; the reference under torch.autocast(bfloat16): q / k / v are bf16 tensors,
 * the scaled query q / sqrt(D) is a bf16 tensor (layers.py:509-510), scores / softmax / AV run in fp32 (the shims at
 * layers.py:49-70) and y returns in v's dtype).  k16 / v16 are the key / value rows STORED as bf16 ([N, H] uint16, half the
 * gather bytes; exact when k, v came out of bf16-mode projections); q is the fp32 projection, rounded in the kernel as
 * bf16(q / sqrt(D)); y is rounded to bf16 values (fp32 storage) and y_raw (optional) receives the unrounded rows, which is
 * what the backward takes as its y (D = sum_d dy y is the fp32 softmax's own); the backward rounds dy on load and returns dq
 * through the two casts (bf16(dq_scaled * sqrt(D)) / sqrt(D)); dk, dv are fp32.  Same CSR arguments and determinism; ws as above, and
 * with N * H more floats of it the row pass hands the column pass bf16(q / sqrt(D)) and bf16(dy) as two-byte rows (half the gathered
 * bytes of that pass, same results bit for bit). */
int mgn_sparse_attn_fwd_b16(const float* q, const uint16_t* k16, const uint16_t* v16, const int32_t* rowptr, const int32_t* col,
                            int64_t N, int H, int num_heads, float* y, float* lse, float* y_raw, void* stream);
int mgn_sparse_attn_bwd_b16(const float* q, const uint16_t* k16, const uint16_t* v16, const float* y, const float* lse,
                            const float* dy, const int32_t* rowptr, const int32_t* col, const int32_t* cptr, const int32_t* cperm,
                            const int32_t* crow, int64_t N, int64_t E, int H, int num_heads,
                            float* dq, float* dk, float* dv, float* ws, size_t ws_bytes, void* stream);
/* Strided forms of both pairs ([r4]): q / k / v are column slabs of ONE [N, 3H] projection output (Attention's three Linears,
 * layers.py:606-616,668-671, issued as a single launch over the concatenated weights) -- ldq / ldk / ldv are the row pitches in
 * elements (multiples of 4, >= H; of the uint16 rows when kv_bf16) -- and dq / dk / dv slabs of that output's gradient, so neither
 * side needs a copy or a concatenation.  kv_bf16 = 1: the *_b16 semantic (k, v bf16 rows; y_raw as there); kv_bf16 = 2: the same
 * roundings with k, v left as fp32 rows holding bf16-representable values (the slabs of a bf16-mode projection as they are).
 * y, lse, dy dense. */
int mgn_sparse_attn_fwd_s(const float* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, int kv_bf16,
                          const int32_t* rowptr, const int32_t* col, int64_t N, int H, int num_heads, float* y, float* lse,
                          float* y_raw, void* stream);
int mgn_sparse_attn_bwd_s(const float* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, int kv_bf16,
                          const float* y, const float* lse, const float* dy, const int32_t* rowptr, const int32_t* col,
                          const int32_t* cptr, const int32_t* cperm, const int32_t* crow, int64_t N, int64_t E, int H,
                          int num_heads, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* ws,
                          size_t ws_bytes, void* stream);
/* The attention weights themselves (Attention.forward(..., return_attention=True), layers.py:543-559,688-697: the values of
 * the softmax-ed sparse matrix): attn[out_pos[e], h] = exp(score[e,h] - lse[i_e,h]) for the row-sorted edge e; out_pos
 * (optional) = the position of that edge in the caller's edge_index (the CSR build's perm), so attn [E, num_heads] lines up
 * with edge_index.  lse from mgn_sparse_attn_fwd. */
int mgn_sparse_attn_weights(const float* q, const float* k, const float* lse, const int32_t* rowptr, const int32_t* col,
                            const int32_t* out_pos, int64_t N, int H, int num_heads, float* attn, void* stream);
/* [r6] Attention over the HEAD axis of each node -- scaled_dot_product_attention(q, k, v, att_mask=None) on [N, d, num_heads]
 * operands (/root/reference/graphphysics/models/layers.py:493-559 with no adjacency: softmax(q k^T / sqrt(d)) over the last
 * axis of [N, d, d], then @ v), what TemporalAttention.forward (layers.py:858-887) runs when the installation has no DGL
 * (processors.py:203-209 and :376-377 hand it adj = None).  q / k / v / y / dy / dq / dk / dv: [N, H] fp32 rows, feature
 * i * num_heads + h = (i, h) of the [d, num_heads] view, d = H / num_heads <= 128; lse: [N, d] (written by _fwd, read by
 * _bwd).  No node reads another node's rows.  Returns 0, 1 (bad arguments), 2 (HIP error: mgn_attn_last_error()). */
int mgn_head_axis_attn_fwd(const float* q, const float* k, const float* v, int64_t N, int H, int num_heads, float* y, float* lse,
                           void* stream);
int mgn_head_axis_attn_bwd(const float* q, const float* k, const float* v, const float* y, const float* lse, const float* dy,
                           int64_t N, int H, int num_heads, float* dq, float* dk, float* dv, void* stream);
const char* mgn_attn_last_error(void);

/* ================================================================================
 * Dense row work of the Transformer processor -- csrc/mgn_dense.hip
 * ================================================================================
 * One fused launch per Linear of a Transformer block (graphphysics/models/layers.py:564-697 Attention, :700-819
 * Transformer, :213-278 GatedMLP / build_gated_mlp) and of TemporalAttention (:822-887):
 *   out = [resid +] epi( n W^T + b ),   n = RMSNorm(cat[x, x2]) if norm_scale else cat[x, x2]      (RMSNorm :73-129)
 *   epi(z) = act(z)                       (act = MGN_ACT_NONE / _RELU / _SILU / _GELU)
 *          = act(z) * (n W2^T + b2)       when W2 is given  (the gated product of GatedMLP, :249-253)
 * x [*, ldx] (first K1 columns), x2 / x3 (optional further phases: a concatenation that is never materialised), each optionally
 * GATHERED through an int32 index row (the cat[e, x[dst], x[src]] of GraphNetBlock.edge_update, layers.py:1044-1060);
 * K1, K2, K3 multiples of 16, their sum in {16,32,48,64,96,128,192,256,384}; W, W2 [N, ldw] row-major (nn.Linear layout),
 * N a multiple of 16.  Optional outputs for the backward pass: inv_out [M] = 1 / (rms + eps), n_out [M, K1 + K2] = the
 * normalised input (the B operand of the weight gradient), saveZ1 / saveZ2 [M, N] = the pre-activations.
 * precision 1 = the reference under Lightning bf16-mixed (autocast runs nn.Linear in bf16, train.py:74-78): operands and
 * every result rounded to bf16, fp32 accumulation; the residual add stays fp32.
 * The backward pass is built from the same launch (dX = dZ W: pass W^T as the weight), mgn_act_gate_bwd, mgn_rownorm_bwd
 * and mgn_wgrad jobs. */
#define MGN_ACT_NONE (-1)
typedef struct {
  int64_t M;
  const float* x; int ldx; int K1;
  const float* x2; int ldx2; int K2;
  const float* x3; int ldx3; int K3;            /* third phase (optional; needs the second) */
  const int32_t* idx; const int32_t* idx2; const int32_t* idx3;   /* optional gather rows per phase (int32 [M]): phase p reads x_p[idx_p[m]] */
  const float* norm_scale; float eps;
  float* inv_out;
  float* n_out;
  const float* W; int ldw;
  const float* b;
  const float* W2;
  const float* b2;
  int act;
  int N;
  const float* resid; int ldr;
  float* out; int ldo;
  float* saveZ1;
  float* saveZ2;
  int precision;
  int w_transposed;   /* W (and W2) are [K, ldw] row-major, ldw >= N: the launch multiplies by their TRANSPOSE -- the input gradient
                         dX = dZ W of a Linear straight from its nn.Linear weight.  Accepted where the weights are staged through LDS
                         (mgn_linear_accepts_transposed); elsewhere mgn_linear_fwd returns 1. */
  /* [r5] gated-product BACKWARD as the epilogue (same shapes as w_transposed; W2, b, resid NULL): with P = n W^T the gradient of the
   * gated product's output (P = dY W3 of the Linear behind a GatedMLP, layers.py:249-253,274-277), the launch writes
   *   out = P * Z2 * act'(Z1)  (= dZ1)   and   out2 = P * act(Z1)  (= dZ2)      -- mgn_act_gate_bwd without materialising P;
   * gb_z1 / gb_z2 [M, N] = the saved pre-activations, act = the product's activation.  NULL: off. */
  const float* gb_z1;
  const float* gb_z2;
  float* out2;
  /* [r5] a second RMSNorm in FRONT of the norm prologue (the Transformer block's norm2 before build_gated_mlp's own norm,
   * layers.py:256-278,700-819): n = RMSNorm(RMSNorm(x; norm_scale_outer); norm_scale); inv_outer_out [M] optional.  Same shapes as
   * w_transposed (the LDS-staged form); needs norm_scale. */
  const float* norm_scale_outer;
  float* inv_outer_out;
  /* [r5] precision 1 only: saveZ1 / saveZ2 (written) and gb_z1 / gb_z2 (read) are TWO-BYTE rows [M, N] of bf16 -- a bf16 Linear's
   * result is a bf16 number, so the narrowing is exact and the gated half's largest tensors take half the bytes. */
  int z16;
  int x16;     /* precision 1: the rows of the (single, ungathered, un-normed) input phase are two-byte bf16 rows, ldx = their pitch in elements */
  int out16;   /* precision 1: out (and out2) are written as two-byte bf16 rows, ldo = their pitch in elements; no residual */
} mgn_linear_args;
int mgn_linear_fwd(const mgn_linear_args* args, void* stream);
/* 1 when mgn_linear_fwd takes w_transposed for this shape (M rows, K = K1 + K2 + K3 inputs, N outputs, gated product or not) */
int mgn_linear_accepts_transposed(int64_t M, int K, int N, int gated, int precision);
/* dZ1 = dP * (Z2 or 1) * act'(Z1);  dZ2 = dP * act(Z1)   over [M, N] (Z2 / dZ2 NULL: a plain activation) */
int mgn_act_gate_bwd(const float* dP, const float* Z1, const float* Z2, int64_t M, int N, int act, int precision, float* dZ1,
                     float* dZ2, void* stream);
/* stand-alone RMSNorm (layers.py:73-129): y [M, K] = scale * x / (||x|| / sqrt(K) + eps), inv_out [M] optional */
int mgn_rownorm_fwd(const float* x, int ldx, int K, const float* scale, float eps, int64_t M, float* y, float* inv_out, void* stream);
/* backward of the RMSNorm prologue: from dn [M, K] (gradient of the normalised concatenated input), the raw rows of the (up to three,
 * optionally gathered) phases, inv and scale: per phase dx [M, lddx] IN ROW SPACE (a gathered phase's rows are summed over their
 * segments by the caller: mgn_segsum), dscale [K] (fixed summation order). */
typedef struct {
  const float* x; int ldx; int K;
  const int32_t* idx;
  float* dx; int lddx;
  const float* acc;   /* [r5] optional [M, lddx]: dx = acc + (the norm's input gradient) -- the gradient a residual connection carries
                         past the norm, added here instead of by an autograd node of its own */
} mgn_rownorm_phase;
size_t mgn_rownorm_bwd_workspace_bytes(int K);
int mgn_rownorm_bwd(const float* dn, const mgn_rownorm_phase* phases, int nphase, const float* inv, const float* scale, float eps,
                    int64_t M, float* dscale, void* ws, size_t ws_bytes, void* stream);
/* [r5] backward of the two stacked RMSNorms above in one pass: from dn [M, K] (gradient of n), the raw rows x, both inv vectors and both
 * scales: dx [M, K] = acc + d x (acc optional: the gradient a residual connection carries past the norms), dscale_io [2 K] = the inner
 * scale's gradient followed by the outer one's (fixed summation order).  K <= 192.  ws: mgn_rownorm_bwd_workspace_bytes(2 K). */
int mgn_rownorm2_bwd(const float* dn, const float* x, int ldx, int K, const float* inv_outer, const float* scale_outer,
                     const float* inv_inner, const float* scale_inner, float eps, int64_t M, const float* acc, float* dx,
                     float* dscale_io, void* ws, size_t ws_bytes, void* stream);
const char* mgn_dense_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MGN_HIP_H */
