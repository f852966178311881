"""``use_gated_mlp`` variant of GraphNetBlock (graphphysics/models/layers.py:213-278,932-942): edge /
node blocks = RMSNorm(in) -> GatedMLP (act(W1 x) * (W2 x), 3x expansion) -> Linear.

Hybrid by design: the SPARSE half of the round -- gathers in the engine's dst-sorted edge order, the
relative RoPE, the segment-sum aggregation and its gather backward -- runs on the HIP kernels (atomics-free,
deterministic, same CSR as the default path); the three plain dense GEMMs of a gated block
([3H x 3H], [3H x 3H], [3H x H]: no fusion partner, no gather) go to rocBLAS through
``torch.nn.functional.linear`` -- the "library GEMM for plain GEMMs" rule -- with PyTorch autograd
differentiating them.  CUDA tensors only, like every other path of the package.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .layers import RMSNorm, use_silu_activation


class GatedMLP(nn.Module):
    """left = act(linear1(x)); right = linear2(x); left * right   (layers.py:213-253; GELU unless the
    global SiLU switch is set)"""

    def __init__(self, in_size: int, hidden_size: int, expansion_factor: int):
        super().__init__()
        self.linear1 = nn.Linear(in_size, expansion_factor * hidden_size)
        self.linear2 = nn.Linear(in_size, expansion_factor * hidden_size)
        self.activation = nn.SiLU() if use_silu_activation() else nn.GELU()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ops._require_device(x)
        return self.activation(self.linear1(x)) * self.linear2(x)


def build_gated_mlp(in_size: int, hidden_size: int, out_size: int, expansion_factor: int = 3) -> nn.Module:
    """layers.py:256-278"""
    return nn.Sequential(RMSNorm(in_size), GatedMLP(in_size, hidden_size, expansion_factor),
                         nn.Linear(hidden_size * expansion_factor, out_size))


class _Gather(torch.autograd.Function):
    """rows gathered by the dst- or src- index of the dst-sorted edges; the backward is the CSR segment sum of
    the same grouping (no atomics, hub-safe)."""

    @staticmethod
    def forward(ctx, x, topo, by):
        ctx.aux = (topo, by, x.shape[0])
        return ops.gather_rows(x.contiguous(), topo.dst_s if by == "dst" else topo.src_s)

    @staticmethod
    def backward(ctx, g):
        topo, by, n = ctx.aux
        out = torch.empty(n, g.shape[1], dtype=torch.float32, device=g.device)
        return ops.segsum_topo(g.contiguous(), topo, by, out), None, None


class _SegSum(torch.autograd.Function):
    """agg[i] = sum of the (dst-sorted) message rows of node i; backward = gather by dst."""

    @staticmethod
    def forward(ctx, m, topo):
        ctx.topo = topo
        out = torch.empty(topo.N, m.shape[1], dtype=torch.float32, device=m.device)
        return ops.segsum_topo(m.contiguous(), topo, "dst", out)

    @staticmethod
    def backward(ctx, g):
        return ops.gather_rows(g.contiguous(), ctx.topo.dst_s), None


class _Rope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, inv_freq, topo, axes):
        ctx.aux = (pos, inv_freq, topo, axes, x.shape[0])
        out = torch.empty(topo.E, x.shape[1], dtype=torch.float32, device=x.device)
        ops.rope_gather(x.contiguous(), pos, inv_freq, topo, axes, out)
        return out

    @staticmethod
    def backward(ctx, g):
        pos, inv_freq, topo, axes, n = ctx.aux
        out = torch.empty(n, g.shape[1], dtype=torch.float32, device=g.device)
        ops.rope_scatter(g.contiguous(), pos, inv_freq, topo, axes, None, out)
        return out, None, None, None, None


def gated_block_forward(block, x: torch.Tensor, e_sorted: torch.Tensor, topo, pos: Optional[torch.Tensor],
                        phi: Optional[torch.Tensor]):
    """GraphNetBlock.forward (layers.py:989-1042) for ``use_gated_mlp`` blocks; ``e_sorted`` / the
    returned edge latents are in the topology's dst-sorted order."""
    ops._require_device(x, e_sorted)
    x = x.float().contiguous()
    x_i = _Gather.apply(x, topo, "dst")
    if block.use_rope:
        if pos is None:
            raise ValueError("Node positions `pos` must be provided when use_rope=True.")
        x_j = _Rope.apply(x, pos.float().contiguous(), block._rope_inv_freq, topo, block.rope_axes)
    else:
        x_j = _Gather.apply(x, topo, "src")
    m = block.edge_block(torch.cat([e_sorted, x_i, x_j], dim=-1))          # edge_update, layers.py:1044-1060
    agg = _SegSum.apply(m, topo)                                           # propagate(aggr="add"), :1031-1037
    if block.use_gate:                                                     # update, :1091-1098
        logits = block.gate_proj(x)
        if phi is not None:
            logits = logits + phi.view(-1, 1).to(logits.dtype) * block.gate_pos.view(1, -1)
        agg = agg * torch.sigmoid(logits)
    upd = block.node_block(torch.cat([x, agg], dim=-1))                    # :1100-1101
    return x + upd, e_sorted + m                                           # residuals, :1039-1040
