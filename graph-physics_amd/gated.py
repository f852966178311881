"""``use_gated_mlp`` variant of GraphNetBlock (graphphysics/models/layers.py:213-278,932-942): edge /
node blocks = RMSNorm(in) -> GatedMLP (act(W1 x) * (W2 x), 3x expansion) -> Linear.

On the engine end to end: the gathers ``x[dst]`` / ``x[src]`` are gathered input PHASES of the fused Linear launch
(``mgn_linear_fwd``: the ``cat[e, x_i, x_j]`` of edge_update is never materialised), the block's leading RMSNorm its prologue,
``act(W1 n + b1) * (W2 n + b2)`` its epilogue; the relative RoPE, the segment-sum aggregation and the gate are the HIP kernels of
the default path.  Backward: ``dense.DenseFn`` (input gradients of gathered phases = CSR segment sums, atomics-free).  No
``F.linear`` and no ``torch.cat`` on the block's path.  CUDA tensors only, like every other path of the package.
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
        k, n = x.shape[-1], self.linear1.weight.shape[0]
        if x.dim() == 2 and k in (16, 32, 48, 64, 96, 128, 192, 256, 384) and n % 16 == 0 and n <= 384:
            from .dense import dense   # one fused launch: both products, the activation and the gated product
            return dense(x, self.linear1.weight, self.linear1.bias, W2=self.linear2.weight, b2=self.linear2.bias, act=_gated_act(self))
        return self.activation(self.linear1(x)) * self.linear2(x)   # widths outside the fused kernel: plain library GEMMs on the device


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


def _gated_act(gm) -> str:
    return "silu" if isinstance(gm.activation, nn.SiLU) else "gelu"


def gated_block_forward(block, x: torch.Tensor, e_sorted: torch.Tensor, topo, pos: Optional[torch.Tensor],
                        phi: Optional[torch.Tensor]):
    """GraphNetBlock.forward (layers.py:989-1042) for ``use_gated_mlp`` blocks; ``e_sorted`` / the
    returned edge latents are in the topology's dst-sorted order."""
    from .dense import SigmoidGateFn, dense

    ops._require_device(x, e_sorted)
    x = x.float().contiguous()
    eb, nb = block.edge_block, block.node_block
    # edge_update (layers.py:1044-1060): m = edge_block(cat[e, x_i, x_j]) -- three input phases, two of them gathered
    if block.use_rope:
        if pos is None:
            raise ValueError("Node positions `pos` must be provided when use_rope=True.")
        x_j = _Rope.apply(x, pos.float().contiguous(), block._rope_inv_freq, topo, block.rope_axes)   # rotated per edge: rows of its own
        third, by = x_j, ("dst", None)
    else:
        third, by = x, ("dst", "src")
    p_e = dense(e_sorted, eb[1].linear1.weight, eb[1].linear1.bias, x2=x, x3=third, W2=eb[1].linear2.weight, b2=eb[1].linear2.bias,
                norm_scale=eb[0].scale, act=_gated_act(eb[1]), gather=(topo, (None,) + by))
    m = dense(p_e, eb[2].weight, eb[2].bias)
    agg = _SegSum.apply(m, topo)                                           # propagate(aggr="add"), :1031-1037
    if block.use_gate:                                                     # update, :1091-1098
        logits = dense(x, block.gate_proj.weight, block.gate_proj.bias)
        if phi is not None:
            logits = logits + phi.view(-1, 1).to(logits.dtype) * block.gate_pos.view(1, -1)
        agg = SigmoidGateFn.apply(agg, logits)
    # update (:1100-1101) and the node residual (:1040): x + node_block(cat[x, agg]) -- two phases, residual as the epilogue
    p_n = dense(x, nb[1].linear1.weight, nb[1].linear1.bias, x2=agg, W2=nb[1].linear2.weight, b2=nb[1].linear2.bias,
                norm_scale=nb[0].scale, act=_gated_act(nb[1]))
    x_new = dense(p_n, nb[2].weight, nb[2].bias, resid=x)
    return x_new, e_sorted + m                                             # edge residual, :1039
