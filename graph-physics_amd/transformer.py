"""Sparse-attention Transformer processor (BASELINE.json configs[4]): ``Attention``,
``Transformer``, ``TemporalAttention`` and ``EncodeTransformDecode`` with the reference's constructor
signatures, attribute names and ``state_dict`` keys (graphphysics/models/layers.py:562-887,
processors.py:218-384).

The attention itself -- edge-masked QK^T, per-row softmax over the mesh adjacency, AV -- is the SPARSE
half and runs on the HIP kernels of csrc/mgn_attn.hip (CSR by row, online softmax, atomics-free two-pass
backward).  It follows the reference's DGL branch (``HAS_DGL_SPARSE``): adjacency
``dglsp.spmatrix(indices=edge_index)`` (rows = edge_index[0]), ``bsddmm`` -> ``softmax`` -> ``bspmm`` with
the head index as the fastest axis of the hidden dimension.  Encoder / decoder are the engine's fused MLP
kernels.  The DENSE half of a block -- q / k / v / gate / output projections, both RMSNorms, the gated MLP, the
mixers of ``TemporalAttention`` -- runs on the fused Linear launches of csrc/mgn_dense.hip (``dense.py``): norm as
prologue, activation / gated product / bias / residual as epilogue, concatenated inputs as two phases; no
``F.linear`` and no ``torch.cat`` on the path.  ``ops.set_matrix_precision("bf16")`` runs those Linears in bf16
(the reference under Lightning bf16-mixed; the sparse attention stays fp32, layers.py:49-70).  CUDA tensors only."""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import _capi, ops
from .dense import ACT_IDS, GatedMlpResidualFn, SigmoidGateFn, dense, rms_norm
from .gated import build_gated_mlp
from .layers import RMSNorm, build_mlp


class AttnTopology:
    """CSR of the attention mask: rows = edge_index[0] (the attending node), columns = edge_index[1].
    ``renumber`` ("morton" with ``pos``, or None): built over node ids renumbered along a Morton curve; ``node_order`` /
    ``node_rank`` (int64) map new -> old / old -> new (see :class:`ops.Topology`).  ``perm`` still names the caller's edge ids."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, renumber: Optional[str] = None, pos: Optional[torch.Tensor] = None):
        t = ops.Topology(edge_index.flip(0), num_nodes, renumber=renumber, pos=pos)  # "dst"-sorted by edge_index[0]
        self.N, self.E = t.N, t.E
        self.node_order, self.node_rank = t.node_order, t.node_rank
        self.rowptr, self.col, self.row = t.rowptr_dst, t.src_s, t.dst_s
        self.cptr, self.cperm = t.rowptr_src, t.perm_src
        self.crow = self.row.index_select(0, self.cperm.long()).contiguous()   # row of the t-th edge of the column-grouped order
        self.perm = t.perm_dst   # row-sorted position -> edge id in the caller's edge_index


#: the attention kernels gather two (backward: four) 256-byte rows per edge out of [N, hidden] matrices: from this many nodes on
#: EncodeTransformDecode.forward renumbers the nodes along a Morton curve of graph.pos (ops.set_node_renumbering "auto" / "on";
#: "off": never) so that a row's neighbours sit in the same cache lines of L2 -- the reference has no counterpart (it hands DGL the
#: dataset's numbering, layers.py:486-561); node rows are permuted on entry and back on exit.  configs[4] (150 000 randomly
#: numbered nodes): training step 16.6 -> 15.0 ms.
ATTN_RENUMBER_MIN_NODES = 32768
_attn_cache: dict = {}


def want_attn_renumbering(num_nodes: int, pos) -> bool:
    mode = ops.get_node_renumbering()
    ok = pos is not None and pos.dim() == 2 and pos.shape[1] >= 2 and pos.shape[0] == num_nodes and pos.is_cuda
    return ok and (mode == "on" or (mode == "auto" and num_nodes >= ATTN_RENUMBER_MIN_NODES))


def get_attn_topology(edge_index: torch.Tensor, num_nodes: int, pos: Optional[torch.Tensor] = None, renumber: bool = False) -> AttnTopology:
    """cached per ``edge_index`` tensor.  ``renumber``: let the engine renumber the nodes for locality when
    :func:`want_attn_renumbering` says so -- the order is computed from the positions seen at the FIRST call with this edge_index
    (it only has to be spatially coherent: a deforming mesh keeps it)."""
    import weakref
    ren = "morton" if (renumber and want_attn_renumbering(int(num_nodes), pos)) else None
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes), str(edge_index.device), ren)
    hit = _attn_cache.get(key)
    if hit is not None and hit[0]() is edge_index:
        return hit[1]
    topo = AttnTopology(edge_index, num_nodes, renumber=ren, pos=pos if ren else None)
    if len(_attn_cache) > 64:
        _attn_cache.clear()
    _attn_cache[key] = (weakref.ref(edge_index), topo)
    return topo


class SparseAttentionFn(torch.autograd.Function):
    """y = softmax_rows(mask . q k^T / sqrt(D)) v on the HIP kernels (mgn_sparse_attn_fwd / _bwd).
    ``b16`` (bf16 matrix mode, k / v straight out of bf16-mode projections): the *_b16 entry points -- k / v are gathered (and
    saved for the backward) as bf16 rows, the roundings of the scaled query, y, dy and dq happen inside the kernels."""

    @staticmethod
    def forward(ctx, q, k, v, topo: AttnTopology, num_heads: int, b16: bool = False):
        ops._require_device(q, k, v)
        q = q.float().contiguous()
        k, v = ((t.contiguous().to(torch.bfloat16) if b16 else t.float().contiguous()) for t in (k, v))
        N, H = q.shape
        y = torch.empty_like(q)
        lse = torch.empty_like(q)
        need_raw = b16 and any(ctx.needs_input_grad[:3])
        y_raw = torch.empty_like(q) if need_raw else None     # the unrounded rows: the backward's y
        with torch.cuda.device(q.device):
            if b16:
                rc = _capi.lib().mgn_sparse_attn_fwd_b16(q.data_ptr(), k.data_ptr(), v.data_ptr(), topo.rowptr.data_ptr(), topo.col.data_ptr(),
                                                         N, H, num_heads, y.data_ptr(), lse.data_ptr(),
                                                         y_raw.data_ptr() if need_raw else None, ops._stream(q.device))
            else:
                rc = _capi.lib().mgn_sparse_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), topo.rowptr.data_ptr(), topo.col.data_ptr(),
                                                     N, H, num_heads, y.data_ptr(), lse.data_ptr(), ops._stream(q.device))
        _capi.check(rc, "mgn_sparse_attn_fwd", attn=True)
        ctx.save_for_backward(q, k, v, y_raw if need_raw else y, lse)
        ctx.topo, ctx.num_heads, ctx.b16 = topo, num_heads, b16
        ctx.mark_non_differentiable(lse)
        return y, lse

    @staticmethod
    def backward(ctx, dy, _dlse=None):
        q, k, v, y, lse = ctx.saved_tensors
        topo, nh = ctx.topo, ctx.num_heads
        dy = dy.float().contiguous()
        N, H = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        ws = torch.empty(max(2 * topo.E * nh + (N * H if ctx.b16 else 0), 1), dtype=torch.float32, device=q.device)
        fn = _capi.lib().mgn_sparse_attn_bwd_b16 if ctx.b16 else _capi.lib().mgn_sparse_attn_bwd
        with torch.cuda.device(q.device):
            rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), y.data_ptr(), lse.data_ptr(), dy.data_ptr(),
                    topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.cptr.data_ptr(), topo.cperm.data_ptr(),
                    topo.crow.data_ptr(), N, topo.E, H, nh, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                    ws.data_ptr(), ws.numel() * 4, ops._stream(q.device))
        _capi.check(rc, "mgn_sparse_attn_bwd", attn=True)
        return dq, dk, dv, None, None, None


class _StackRows(torch.autograd.Function):
    """[W_1; W_2; ...] (and the biases alike) of Linears that read the same rows, as ONE matrix for one launch: a multi-tensor copy
    of the parameters (a few KB) into a fresh buffer; the backward hands each parameter its row slab of the stacked gradient."""

    @staticmethod
    def forward(ctx, *ws):
        rows = [int(w.shape[0]) for w in ws]
        out = torch.empty((sum(rows),) + tuple(ws[0].shape[1:]), dtype=torch.float32, device=ws[0].device)
        torch._foreach_copy_(list(out.split(rows, dim=0)), [w.detach() for w in ws])
        ctx.rows = rows
        return out

    @staticmethod
    def backward(ctx, d):
        return tuple(d.split(ctx.rows, dim=0))


class PackedAttentionFn(torch.autograd.Function):
    """Sparse attention over q | k | v held as column slabs of ONE [N, 3H] projection output (mgn_sparse_attn_fwd_s / _bwd_s): no
    copies out of it, and dq | dk | dv written as slabs of its gradient.  ``b16``: bf16 matrix mode -- the *_b16 semantic (roundings of the
    scaled query, y, dy, dq inside the kernels) on the k | v slabs AS THEY ARE: fp32 rows whose values are bf16 numbers, what a
    bf16-mode projection writes (kv_bf16 = 2; MGN_ATTN_KV16=1: narrowed to two-byte rows first, kv_bf16 = 1, the round-4 form).  The
    caller guarantees that k and v hold bf16-representable values in this mode."""

    @staticmethod
    def forward(ctx, qkv, topo: AttnTopology, num_heads: int, b16: bool):
        ops._require_device(qkv)
        qkv = ops._f32c(qkv)
        N, H3 = qkv.shape
        H = H3 // 3
        mode = (1 if os.environ.get("MGN_ATTN_KV16") == "1" else 2) if b16 else 0
        kv = qkv[:, H:].to(torch.bfloat16) if mode == 1 else qkv[:, H:]         # [N, 2H]: k | v
        es, ekv = qkv.element_size(), kv.element_size()
        y, lse = torch.empty(N, H, dtype=torch.float32, device=qkv.device), torch.empty(N, H, dtype=torch.float32, device=qkv.device)
        need_raw = b16 and ctx.needs_input_grad[0]
        y_raw = torch.empty_like(y) if need_raw else None
        with torch.cuda.device(qkv.device):
            rc = _capi.lib().mgn_sparse_attn_fwd_s(qkv.data_ptr(), H3, kv.data_ptr(), int(kv.stride(0)), kv.data_ptr() + H * ekv, int(kv.stride(0)),
                                                   mode, topo.rowptr.data_ptr(), topo.col.data_ptr(), N, H, num_heads, y.data_ptr(),
                                                   lse.data_ptr(), y_raw.data_ptr() if need_raw else None, ops._stream(qkv.device))
        _capi.check(rc, "mgn_sparse_attn_fwd_s", attn=True)
        ctx.save_for_backward(qkv, kv if mode == 1 else qkv.new_empty(0), y_raw if need_raw else y, lse)
        ctx.topo, ctx.num_heads, ctx.b16 = topo, num_heads, mode
        return y

    @staticmethod
    def backward(ctx, dy):
        qkv, kv16, y, lse = ctx.saved_tensors
        topo, nh, mode = ctx.topo, ctx.num_heads, ctx.b16
        dy = ops._f32c(dy)
        N, H3 = qkv.shape
        H = H3 // 3
        kv = kv16 if mode == 1 else qkv[:, H:]
        ekv = kv.element_size()
        d = torch.empty_like(qkv)
        ws = torch.empty(max(2 * topo.E * nh + (N * H if mode == 1 else 0), 1), dtype=torch.float32, device=qkv.device)
        with torch.cuda.device(qkv.device):
            rc = _capi.lib().mgn_sparse_attn_bwd_s(qkv.data_ptr(), H3, kv.data_ptr(), int(kv.stride(0)), kv.data_ptr() + H * ekv, int(kv.stride(0)),
                                                   mode, y.data_ptr(), lse.data_ptr(), dy.data_ptr(), topo.rowptr.data_ptr(),
                                                   topo.col.data_ptr(), topo.cptr.data_ptr(), topo.cperm.data_ptr(), topo.crow.data_ptr(),
                                                   N, topo.E, H, nh, d.data_ptr(), H3, d.data_ptr() + 4 * H, H3, d.data_ptr() + 8 * H, H3,
                                                   ws.data_ptr(), ws.numel() * 4, ops._stream(qkv.device))
        _capi.check(rc, "mgn_sparse_attn_bwd_s", attn=True)
        return d, None, None, None


class HeadAxisAttentionFn(torch.autograd.Function):
    """``scaled_dot_product_attention(q, k, v, att_mask=None)`` on [N, d, num_heads] operands (layers.py:493-559 without an adjacency):
    per node, softmax(q k^T / sqrt(d)) over the last axis of [d, d], then @ v -- mgn_head_axis_attn_fwd / _bwd."""

    @staticmethod
    def forward(ctx, q, k, v, num_heads: int):
        ops._require_device(q, k, v)
        q, k, v = (t.float().contiguous() for t in (q, k, v))
        N, H = q.shape
        y = torch.empty_like(q)
        lse = torch.empty(N, H // num_heads, dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            rc = _capi.lib().mgn_head_axis_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), N, H, num_heads, y.data_ptr(), lse.data_ptr(),
                                                    ops._stream(q.device))
        _capi.check(rc, "mgn_head_axis_attn_fwd", attn=True)
        ctx.save_for_backward(q, k, v, y, lse)
        ctx.num_heads = num_heads
        return y

    @staticmethod
    def backward(ctx, dy):
        q, k, v, y, lse = ctx.saved_tensors
        dy = dy.float().contiguous()
        N, H = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        with torch.cuda.device(q.device):
            rc = _capi.lib().mgn_head_axis_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), y.data_ptr(), lse.data_ptr(), dy.data_ptr(),
                                                    N, H, ctx.num_heads, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ops._stream(q.device))
        _capi.check(rc, "mgn_head_axis_attn_bwd", attn=True)
        return dq, dk, dv, None


def head_axis_attention(q, k, v, num_heads: int):
    """the reference's attention when no adjacency is handed over: [N, H] rows viewed as [N, H / num_heads, num_heads]"""
    return HeadAxisAttentionFn.apply(q, k, v, num_heads)


def sparse_attention(q, k, v, topo: AttnTopology, num_heads: int, return_attention: bool = False, b16: bool = False):
    """``return_attention``: also the per-edge attention weights [E, num_heads] in the order of the caller's edge_index
    (the values of the reference's softmax-ed sparse matrix, layers.py:543-559); no gradient flows through them."""
    if b16 and return_attention:
        raise ValueError("sparse_attention: b16 and return_attention do not combine (the weights kernel reads fp32 q / k)")
    y, lse = SparseAttentionFn.apply(q, k, v, topo, num_heads, b16)
    if not return_attention:
        return y
    qd, kd = q.detach().float().contiguous(), k.detach().float().contiguous()
    attn = torch.empty(topo.E, num_heads, dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        rc = _capi.lib().mgn_sparse_attn_weights(qd.data_ptr(), kd.data_ptr(), lse.data_ptr(), topo.rowptr.data_ptr(), topo.col.data_ptr(),
                                                 topo.perm.data_ptr(), qd.shape[0], qd.shape[1], num_heads, attn.data_ptr(),
                                                 ops._stream(q.device))
    _capi.check(rc, "mgn_sparse_attn_weights", attn=True)
    return y, attn


def _make_inv_freq(m: int, base: float) -> torch.Tensor:
    """layers.py:410-417"""
    if m <= 0:
        return torch.empty(0, dtype=torch.float32)
    step = math.log(base) / max(m, 1)
    return torch.exp(-torch.arange(m, dtype=torch.float32) * step)


def _apply_rope_with_inv(q, k, pos, inv_freq):
    """Absolute RoPE on q / k in the [N, head_dim, num_heads] layout (layers.py:420-490): per position axis
    a, pair i of the next 2m head dims rotates by pos[n, a] * inv_freq[i], the same for every head.
    Elementwise device ops on N x hidden values."""
    N, D, Hh = q.shape
    pd = pos.shape[1]
    m = D // (pd * 2)
    if m == 0 or inv_freq.numel() == 0:
        return q, k
    d_rope = pd * 2 * m
    ang = pos[:, :pd].to(torch.float32).unsqueeze(-1) * inv_freq.to(pos.device, torch.float32).view(1, 1, m)
    cos, sin = torch.cos(ang).unsqueeze(-1), torch.sin(ang).unsqueeze(-1)   # [N, pd, m, 1]

    def app(x):
        part = x[:, :d_rope, :].reshape(N, pd, m, 2, Hh)
        even, odd = part[..., 0, :], part[..., 1, :]
        rot = torch.stack((even * cos - odd * sin, even * sin + odd * cos), dim=3).reshape(N, d_rope, Hh)
        return torch.cat([rot, x[:, d_rope:, :]], dim=1) if D > d_rope else rot

    return app(q), app(k)


class Attention(nn.Module):
    """layers.py:562-697"""

    def __init__(self, input_dim=512, output_dim=512, num_heads=4, pos_dimension: int = 3, use_proj_bias: bool = True,
                 use_separate_proj_weight: bool = True, use_rope_embeddings: bool = False, use_gated_attention: bool = False,
                 rope_base: float = 10000.0):
        super().__init__()
        assert output_dim % num_heads == 0, "Output dimension must be divisible by number of heads."
        self.hidden_size, self.num_heads, self.head_dim = output_dim, num_heads, output_dim // num_heads
        self.use_rope_embeddings, self.use_gated_attention = use_rope_embeddings, use_gated_attention
        self.pos_dimension, self.rope_base = pos_dimension, rope_base
        self.q_proj = nn.Linear(input_dim, output_dim, bias=use_proj_bias)
        self.k_proj = nn.Linear(input_dim, output_dim, bias=use_proj_bias)
        self.v_proj = nn.Linear(input_dim, output_dim, bias=use_proj_bias)
        self.proj = nn.Linear(output_dim, output_dim, bias=use_proj_bias)
        if self.use_rope_embeddings:
            self.m = self.head_dim // max(self.pos_dimension * 2, 1)
            self.register_buffer("rope_inv_freq", _make_inv_freq(self.m, self.rope_base), persistent=True)
        else:
            self.m = 0
            self.register_buffer("rope_inv_freq", torch.empty(0, dtype=torch.float32), persistent=False)
        self.gate_proj = nn.Linear(input_dim, output_dim, bias=use_proj_bias) if use_gated_attention else None
        if not use_separate_proj_weight:
            with torch.no_grad():
                self.k_proj.weight = self.q_proj.weight
                self.v_proj.weight = self.q_proj.weight

    def forward(self, x: torch.Tensor, adj, pos: Optional[torch.Tensor] = None, return_attention: bool = False,
                _norm_scale: Optional[torch.Tensor] = None, _resid: Optional[torch.Tensor] = None):
        """``adj``: an :class:`AttnTopology` (or an edge_index tensor, converted and cached).
        ``_norm_scale`` / ``_resid`` (used by :class:`Transformer`): RMSNorm fused as the prologue of the q / k / v / gate
        projections, the block's residual as the epilogue of the output projection."""
        if self.use_rope_embeddings and pos is None:
            raise ValueError("RoPE embeddings require positional information when enabled.")
        ops._require_device(x)
        N = x.size(0)
        topo = adj if isinstance(adj, AttnTopology) else get_attn_topology(adj, N)
        lin = lambda m: dense(x, m.weight, m.bias, norm_scale=_norm_scale)  # noqa: E731
        rope = self.use_rope_embeddings and self.rope_inv_freq.numel() > 0
        bf16 = ops.get_matrix_precision() == "bf16"
        if not rope and not return_attention and 3 * self.hidden_size <= 384:
            # the three projections as ONE launch over [W_q; W_k; W_v] (the norm prologue runs once, x is read once), attention
            # straight on the column slabs of its output ([r4])
            projs = (self.q_proj, self.k_proj, self.v_proj)
            Wc = _StackRows.apply(*(m.weight for m in projs))
            bc = _StackRows.apply(*(m.bias for m in projs)) if self.q_proj.bias is not None else None
            if _resid is x and _norm_scale is not None and torch.is_grad_enabled() and x.requires_grad:
                # the block's residual rides through the projection node: its gradient is added inside that node's norm backward
                # (dense(pass_x=True)) instead of by an autograd add over the two uses of x
                qkv, xr = dense(x, Wc, bc, norm_scale=_norm_scale, pass_x=True)
                if self.use_gated_attention and self.gate_proj is not None:
                    y = SigmoidGateFn.apply(PackedAttentionFn.apply(qkv, topo, self.num_heads, bf16), lin(self.gate_proj))
                    return dense(y, self.proj.weight, self.proj.bias, resid=_resid)   # (x has a third use: the plain path)
                return dense(PackedAttentionFn.apply(qkv, topo, self.num_heads, bf16), self.proj.weight, self.proj.bias, resid=xr)
            y = PackedAttentionFn.apply(dense(x, Wc, bc, norm_scale=_norm_scale), topo, self.num_heads, bf16)
            if self.use_gated_attention and self.gate_proj is not None:
                y = SigmoidGateFn.apply(y, lin(self.gate_proj))
            return dense(y, self.proj.weight, self.proj.bias, resid=_resid)
        q, k, v = lin(self.q_proj), lin(self.k_proj), lin(self.v_proj)
        if rope:
            q3, k3 = _apply_rope_with_inv(q.reshape(N, self.head_dim, self.num_heads), k.reshape(N, self.head_dim, self.num_heads),
                                          pos, self.rope_inv_freq)
            q, k = q3.reshape(N, -1), k3.reshape(N, -1)
        # bf16 mode, k / v straight out of the (bf16-rounding) projections: the *_b16 kernels gather bf16 rows and do the
        # roundings below themselves (no elementwise launches around the attention)
        fused16 = bf16 and not rope and not return_attention
        if bf16 and not fused16:  # the scaled query is a bf16 tensor in the reference (q / sqrt(d), layers.py:509-510)
            s_ = math.sqrt(self.head_dim)
            q = (q / s_).bfloat16().float() * s_
        attn = None
        if return_attention:   # (out, attn): attn [E, num_heads] lines up with edge_index (layers.py:688-697)
            y, attn = sparse_attention(q, k, v, topo, self.num_heads, return_attention=True)
        else:
            y = sparse_attention(q, k, v, topo, self.num_heads, b16=fused16)
        if bf16 and not fused16:  # scores / softmax / AV ran in fp32 (layers.py:49-70); y returns in v's dtype
            y = y.bfloat16().float()
        if self.use_gated_attention and self.gate_proj is not None:
            y = SigmoidGateFn.apply(y, lin(self.gate_proj))   # same flat layout as reshape(N, head_dim, num_heads)
        out = dense(y, self.proj.weight, self.proj.bias, resid=_resid)
        return (out, attn) if return_attention else out


class Transformer(nn.Module):
    """layers.py:700-819: x + Attention(norm1(x)); x + gated_mlp(norm2(x))"""

    def __init__(self, input_dim: int, output_dim: int, num_heads: int, activation_layer=nn.ReLU, use_proj_bias: bool = True,
                 use_separate_proj_weight: bool = True, use_rope_embeddings: bool = False, use_gated_attention: bool = False,
                 pos_dimension: int = 3, rope_base: float = 10000.0):
        super().__init__()
        self.use_rope_embeddings, self.use_gated_attention, self.pos_dimension = use_rope_embeddings, use_gated_attention, pos_dimension
        self.attention = Attention(input_dim=input_dim, output_dim=output_dim, num_heads=num_heads, pos_dimension=pos_dimension,
                                   use_proj_bias=use_proj_bias, use_separate_proj_weight=use_separate_proj_weight,
                                   use_rope_embeddings=use_rope_embeddings, use_gated_attention=use_gated_attention, rope_base=rope_base)
        self.activation = activation_layer()
        self.norm1, self.norm2 = RMSNorm(output_dim), RMSNorm(output_dim)
        self.gated_mlp = build_gated_mlp(in_size=output_dim, hidden_size=output_dim, out_size=output_dim)
        self.use_adjacency = True

    def forward(self, x: torch.Tensor, adj, pos: Optional[torch.Tensor] = None, return_attention: bool = False) -> torch.Tensor:
        if self.use_rope_embeddings and pos is None:
            raise ValueError("Transformer blocks require node positions when use_rope_embeddings=True.")
        # x + Attention(norm1(x)): norm1 is the prologue of the projections, the residual the epilogue of the output projection
        a = self.attention(x, adj, pos=pos, return_attention=return_attention, _norm_scale=self.norm1.scale, _resid=x)
        x, attn = a if return_attention else (a, None)
        # x + gated_mlp(norm2(x)); build_gated_mlp starts with a norm of its own (layers.py:256-278), so norm2 is a launch of
        # its own and the inner one the prologue of the two gate Linears
        gm = self.gated_mlp
        act = "silu" if isinstance(gm[1].activation, nn.SiLU) else "gelu"
        prec = 1 if ops.get_matrix_precision() == "bf16" else 0
        if GatedMlpResidualFn.usable(x, gm[1].linear1.weight, gm[2].weight, prec):
            # large meshes: the whole half as one autograd node (dense.GatedMlpResidualFn: the gated product's backward as the epilogue of
            # the launch that forms its incoming gradient, the residual's gradient added inside the last norm backward)
            ops._call.grad = torch.is_grad_enabled()
            try:
                x = GatedMlpResidualFn.apply(x, self.norm2.scale, gm[0].scale, gm[1].linear1.weight, gm[1].linear1.bias, gm[1].linear2.weight,
                                             gm[1].linear2.bias, gm[2].weight, gm[2].bias, ACT_IDS[act], prec)
            finally:
                ops._call.grad = True
            return (x, attn) if return_attention else x
        h = rms_norm(x, self.norm2.scale)
        p_ = dense(h, gm[1].linear1.weight, gm[1].linear1.bias, W2=gm[1].linear2.weight, b2=gm[1].linear2.bias, norm_scale=gm[0].scale, act=act)
        x = dense(p_, gm[2].weight, gm[2].bias, resid=x)
        return (x, attn) if return_attention else x


class TemporalAttention(nn.Module):
    """Temporal corrector as sparse cross-attention (layers.py:822-887): queries / values from the predicted
    state, keys from the previous one; optional sigmoid gate; SiLU mixer."""

    def __init__(self, hidden_size: int, num_heads: int = 4, use_gate: bool = True):
        super().__init__()
        assert hidden_size % num_heads == 0, "hidden_size must be divisible by num_heads"
        self.h, self.H, self.d, self.use_gate = hidden_size, num_heads, hidden_size // num_heads, use_gate
        self.q_proj = nn.Linear(self.h, self.h, bias=True)
        self.k_proj = nn.Linear(self.h, self.h, bias=True)
        self.v_proj = nn.Linear(self.h, self.h, bias=True)
        self.out_proj = nn.Linear(self.h, self.h, bias=True)
        if use_gate:
            self.gate = nn.Sequential(nn.Linear(2 * self.h, self.h), nn.SiLU(), nn.Linear(self.h, self.h), nn.Sigmoid())
        self.mixer = nn.Sequential(nn.Linear(2 * self.h, self.h), nn.SiLU(), nn.Linear(self.h, self.h))

    def forward(self, h_prev: torch.Tensor, h_pred: torch.Tensor, adj=None) -> torch.Tensor:
        ops._require_device(h_prev, h_pred)
        q = dense(h_pred, self.q_proj.weight, self.q_proj.bias)
        k = dense(h_prev, self.k_proj.weight, self.k_proj.bias)
        v = dense(h_pred, self.v_proj.weight, self.v_proj.bias)
        if adj is None:
            # an installation without DGL hands the block no adjacency (processors.py:203-209, :376-377) and
            # scaled_dot_product_attention (layers.py:493-559) then attends over the head axis of each node
            y = head_axis_attention(q, k, v, self.H)
        else:
            topo = adj if isinstance(adj, AttnTopology) else get_attn_topology(adj, h_prev.size(0))
            y = sparse_attention(q, k, v, topo, self.H)
        out = dense(y, self.out_proj.weight, self.out_proj.bias)
        if self.use_gate:   # sigmoid(Linear(SiLU(Linear(cat[h_pred, h_prev])))) * out: the concatenation is two input phases
            g1 = dense(h_pred, self.gate[0].weight, self.gate[0].bias, x2=h_prev, act="silu")
            out = SigmoidGateFn.apply(out, dense(g1, self.gate[2].weight, self.gate[2].bias))
        h_corr = h_prev + out
        m1 = dense(h_corr, self.mixer[0].weight, self.mixer[0].bias, x2=h_prev, act="silu")
        return dense(m1, self.mixer[2].weight, self.mixer[2].bias, resid=h_corr)


class TransformerConv(nn.Module):
    """``torch_geometric.nn.TransformerConv(in_channels, out_channels, heads, concat=False, beta=True)`` -- the block the reference builds
    when DGL is absent (processors.py:303-314; its own CI runs that branch) -- with PyG's parameter names (``lin_key``, ``lin_query``,
    ``lin_value``, ``lin_skip``, ``lin_beta``: a checkpoint of that branch loads).  PyG 2.6.1 is not installable here, so the
    arithmetic follows its published algorithm (oracle/mgn_oracle.py::transformer_conv: parity unpinned).  Projections on the
    engine's fused Linear launches, one sparse-attention call per head (PyG keeps head h in columns h*C .. h*C + C - 1 and, without
    ``concat``, every head is ``out_channels`` wide), softmax over the IN-edges of a node (rows = ``edge_index[1]``)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = False, beta: bool = True):
        super().__init__()
        if concat or not beta:
            raise NotImplementedError("TransformerConv: only the reference's configuration (concat=False, beta=True)")
        self.in_channels, self.out_channels, self.heads, self.concat, self.beta = in_channels, out_channels, heads, concat, beta
        self.lin_key = nn.Linear(in_channels, heads * out_channels)
        self.lin_query = nn.Linear(in_channels, heads * out_channels)
        self.lin_value = nn.Linear(in_channels, heads * out_channels)
        self.lin_skip = nn.Linear(in_channels, out_channels)
        self.lin_beta = nn.Linear(3 * out_channels, 1, bias=False)

    def forward(self, x: torch.Tensor, edge_index) -> torch.Tensor:
        ops._require_device(x)
        N, C, Hh = x.size(0), self.out_channels, self.heads
        topo = edge_index if isinstance(edge_index, AttnTopology) else get_attn_topology(_flipped(edge_index), N)
        q, k, v = (dense(x, m.weight, m.bias) for m in (self.lin_query, self.lin_key, self.lin_value))
        out = None
        for h in range(Hh):   # one head = one [N, C] problem (1 / sqrt(C) scaling inside the kernel)
            sl = slice(h * C, (h + 1) * C)
            y = sparse_attention(q[:, sl].contiguous(), k[:, sl].contiguous(), v[:, sl].contiguous(), topo, 1)
            out = y if out is None else out + y
        out = out / Hh
        r = dense(x, self.lin_skip.weight, self.lin_skip.bias)
        w = self.lin_beta.weight.view(3, C)
        beta = torch.sigmoid((out * w[0] + r * w[1] + (out - r) * w[2]).sum(dim=-1, keepdim=True))
        return beta * r + (1 - beta) * out


_flip_cache: dict = {}


def _flipped(edge_index: torch.Tensor) -> torch.Tensor:
    """edge_index with its two rows swapped, the same tensor object for the same input (the attention topology is cached per object)"""
    import weakref
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape))
    hit = _flip_cache.get(key)
    if hit is not None and hit[0]() is edge_index:
        return hit[1]
    f = edge_index.flip(0).contiguous()
    if len(_flip_cache) > 64:
        _flip_cache.clear()
    _flip_cache[key] = (weakref.ref(edge_index), f)
    return f


def default_attention_backend() -> str:
    """"dgl" (the sparse-attention Transformer blocks, what the reference builds when DGL is importable) unless
    MGN_ATTENTION_BACKEND=pyg asks for the TransformerConv branch of a DGL-less installation (processors.py:303-314).  The
    reference's own GRAPH_PHYSICS_ASSUME_NO_DGL only silences its prompt and is set by anyone importing the reference
    unattended -- it does not say which branch a checkpoint was trained with, so it is deliberately NOT read here."""
    import os
    v = os.getenv("MGN_ATTENTION_BACKEND", "dgl").strip().lower()
    if v not in ("dgl", "pyg"):
        raise ValueError("MGN_ATTENTION_BACKEND must be 'dgl' or 'pyg'")
    return v


class EncodeTransformDecode(nn.Module):
    """processors.py:218-384 (the DGL branch: Transformer blocks over the sparse adjacency; ``attention_backend="pyg"`` or
    MGN_ATTENTION_BACKEND=pyg: the non-DGL branch, TransformerConv blocks)"""

    def __init__(self, message_passing_num: int, node_input_size: int, output_size: int, hidden_size: int = 128, num_heads: int = 4,
                 only_processor: bool = False, use_proj_bias: bool = True, use_separate_proj_weight: bool = True,
                 use_rope_embeddings: bool = False, use_gated_attention: bool = False, rope_pos_dimension: int = 3,
                 rope_base: float = 10000.0, use_temporal_block: bool = False, attention_backend: Optional[str] = None):
        super().__init__()
        self.attention_backend = attention_backend if attention_backend is not None else default_attention_backend()
        if self.attention_backend not in ("dgl", "pyg"):
            raise ValueError("attention_backend must be 'dgl' or 'pyg'")
        pyg = self.attention_backend == "pyg"
        self.hidden_size, self.only_processor, self.d = hidden_size, only_processor, output_size
        # (the reference drops RoPE / the gate without DGL, with a warning: processors.py:267,317-326)
        self.use_rope_embeddings, self.use_gated_attention = use_rope_embeddings and not pyg, use_gated_attention
        self._requested_rope, self.use_temporal_block = use_rope_embeddings, use_temporal_block
        if not self.only_processor:
            self.nodes_encoder = build_mlp(node_input_size, hidden_size, hidden_size)
            self.decode_module = build_mlp(hidden_size, hidden_size, output_size, layer_norm=False)
        if pyg:
            self.processor_list = nn.ModuleList([TransformerConv(hidden_size, hidden_size, heads=num_heads, concat=False, beta=True)
                                                 for _ in range(message_passing_num)])
        else:
            self.processor_list = nn.ModuleList([
                Transformer(input_dim=hidden_size, output_dim=hidden_size, num_heads=num_heads, use_proj_bias=use_proj_bias,
                            use_separate_proj_weight=use_separate_proj_weight, use_rope_embeddings=use_rope_embeddings,
                            use_gated_attention=use_gated_attention, pos_dimension=rope_pos_dimension, rope_base=rope_base)
                for _ in range(message_passing_num)])
        self.temporal_block = TemporalAttention(hidden_size=hidden_size, num_heads=num_heads) if use_temporal_block else None

    def forward(self, graph) -> torch.Tensor:
        pos = getattr(graph, "pos", None)
        if self.use_rope_embeddings and pos is None:
            raise ValueError("use_rope_embeddings=True requires 'pos' attribute in the input graph.")
        n = graph.x.shape[0]
        pyg = self.attention_backend == "pyg"
        topo = None if pyg else getattr(graph, "mgn_attn_topology", None)
        if topo is None:
            # cached per edge_index; large meshes are renumbered for locality (ATTN_RENUMBER_MIN_NODES): node rows are permuted
            # here on entry and back on exit.  pyg branch: rows = the node that aggregates = edge_index[1]
            topo = get_attn_topology(_flipped(graph.edge_index) if pyg else graph.edge_index, n, pos=pos, renumber=True)
        order, rank = topo.node_order, topo.node_rank
        x_in = graph.x if order is None else graph.x.index_select(0, order)
        if order is not None and pos is not None:
            pos = pos.index_select(0, order)
        x = x_in if self.only_processor else self.nodes_encoder(x_in)
        if pyg:   # processors.py:372-377: x = block(x, edge_index); the temporal block is handed adj = None (no DGL: no adjacency)
            prev_x = last_x = x
            for block in self.processor_list:
                prev_x = x
                last_x = block(prev_x, topo)
                x = last_x
            if self.use_temporal_block and self.temporal_block is not None:
                x = self.temporal_block(prev_x, last_x, None)
        else:
            prev_x = last_x = x
            for block in self.processor_list:
                prev_x = x
                last_x = block(prev_x, topo, pos=pos)
                x = last_x
            if self.use_temporal_block and self.temporal_block is not None:
                x = self.temporal_block(prev_x, last_x, topo)
        out = x if self.only_processor else self.decode_module(x)
        return out if rank is None else out.index_select(0, rank)   # back to the caller's numbering (the decoder is row-wise)
