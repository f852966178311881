"""Multi-GPU plumbing (one process per GPU, ``torch.distributed``; backend "nccl"
is RCCL on ROCm, "gloo" in the CPU tests).

Small meshes (CylinderFlow) are data-parallel replicas: every rank steps its own
batch and the only exchange is ONE flat all-reduce of the 2.87 M-parameter
gradient (11.5 MB fp32) per step -- a single bucket, because over point-to-point
xGMI a ring all-reduce is per-link latency bound at this size and splitting it
only adds hops.  The partitioned large-mesh path lives in ``partition.py``.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, world, local_rank); initialises the default group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MGN_DIST_BACKEND=gloo: rehearsal of the multi-process path on a box with fewer GPUs than ranks
            backend = os.environ.get("MGN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if os.environ.get("MGN_SHARE_GPU") and torch.cuda.is_available():
        local = local % torch.cuda.device_count()  # rehearsal only: several ranks on one device (not with RCCL)
    return rank, world, local


class GradAllReduce:
    """callable(params): average ``.grad`` over the group through one flat buffer."""

    def __init__(self, group=None, average: bool = True):
        self.group = group
        self.average = average  # False: sum (partitioned mesh, the loss is already global)
        self._flat: Optional[torch.Tensor] = None

    def __call__(self, params: Iterable[torch.nn.Parameter]):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        ps = [p for p in params if p.grad is not None]
        n = sum(p.grad.numel() for p in ps)
        if self._flat is None or self._flat.numel() != n or self._flat.device != ps[0].grad.device:
            self._flat = torch.empty(n, dtype=torch.float32, device=ps[0].grad.device)
        torch._foreach_copy_(list(torch.split(self._flat, [p.grad.numel() for p in ps])),
                             [p.grad.reshape(-1) for p in ps])
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            self._flat.div_(dist.get_world_size(self.group))
        # one multi-tensor launch back (296 single copies cost ~0.6 ms of the step on the GPU)
        sizes = [p.grad.numel() for p in ps]
        torch._foreach_copy_([p.grad for p in ps],
                             [v.view_as(p.grad) for v, p in zip(torch.split(self._flat, sizes), ps)])


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Make every replica start from rank ``src``'s weights and normaliser buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


# ----------------------------------------------------------------------------------
# Partitioned large-mesh path: one-hop halo exchange of ghost-node latents per round.
# ----------------------------------------------------------------------------------
class HaloExchange(torch.autograd.Function):
    """ghost rows <- owners' rows (forward); ghost-row gradients summed back into the
    owners' rows (backward).  One ``all_to_all_single`` each way -- a neighbour exchange
    of ~O(sqrt(N/P)) rows per peer, latency-bound over xGMI."""

    @staticmethod
    def forward(ctx, x_own, plan, group):
        ctx.plan, ctx.group, ctx.n_own = plan, group, x_own.shape[0]
        idx = plan.send_idx.to(x_own.device)
        ctx.idx = idx
        ctx.local = plan.world == 1 or not dist.is_initialized()
        send = x_own.detach().index_select(0, idx).contiguous()
        recv = torch.empty(plan.n_ghost, x_own.shape[1], dtype=x_own.dtype, device=x_own.device)
        if ctx.local:  # single process: nothing to exchange (a world-1 plan has no ghosts)
            return recv.zero_()
        dist.all_to_all_single(recv, send, output_split_sizes=plan.recv_counts, input_split_sizes=plan.send_counts, group=group)
        return recv

    @staticmethod
    def backward(ctx, d_ghost):
        plan = ctx.plan
        d_ghost = d_ghost.contiguous()
        back = torch.empty(ctx.idx.numel(), d_ghost.shape[1], dtype=d_ghost.dtype, device=d_ghost.device)
        if ctx.local:
            return torch.zeros(ctx.n_own, d_ghost.shape[1], dtype=d_ghost.dtype, device=d_ghost.device), None, None
        dist.all_to_all_single(back, d_ghost, output_split_sizes=plan.send_counts, input_split_sizes=plan.recv_counts, group=ctx.group)
        d_own = torch.zeros(ctx.n_own, d_ghost.shape[1], dtype=d_ghost.dtype, device=d_ghost.device)
        d_own.index_add_(0, ctx.idx, back)
        return d_own, None, None


class HipBackend:
    """Compute backend of the partitioned model: the HIP engine (default, GPU only)."""

    def prepare(self, edge_index, n_local):
        from . import ops
        return ops.Topology(edge_index, n_local)

    def mlp(self, module, x):
        return module(x)

    def order_edges(self, edge_attr, ctx):
        return edge_attr[ctx.perm_dst.long()]

    def block(self, block, x, e, ctx):
        from . import ops
        from .layers import _block_params
        return ops.processor_apply(x, e, ctx, 1, *_block_params(block))


class PartitionedEPD(torch.nn.Module):
    """EncodeProcessDecode over a node-partitioned mesh.  Each rank holds its owned nodes,
    the ghost sources of its edges, and the edges whose destination it owns; results on the
    owned nodes equal the un-partitioned forward / backward (the parity oracle of section 8e).

    forward(x_in_own[n_own,F_n], edge_attr_loc[E_loc,F_e]) -> out_own[n_own,O]
    """

    def __init__(self, model, plan, group=None, backend=None):
        super().__init__()
        self.model, self.plan, self.group = model, plan, group
        self.backend = backend if backend is not None else HipBackend()
        self._ctx = None

    def forward(self, x_in_own: torch.Tensor, edge_attr_loc: torch.Tensor) -> torch.Tensor:
        plan, be, m = self.plan, self.backend, self.model
        dev = x_in_own.device
        if self._ctx is None:
            self._ctx = be.prepare(plan.edge_index.to(dev), plan.n_own + plan.n_ghost)
        x_own = be.mlp(m.nodes_encoder, x_in_own)
        e = be.mlp(m.edges_encoder, be.order_edges(edge_attr_loc, self._ctx))
        for blk in m.processor_list:
            x_gh = HaloExchange.apply(x_own, plan, self.group)
            x_full = torch.cat([x_own, x_gh], dim=0)
            x_full, e = be.block(blk, x_full, e, self._ctx)
            x_own = x_full[: plan.n_own]
        return be.mlp(m.decode_module, x_own)


def partitioned_loss(net_out_own, target_own, node_type_own, group=None):
    """Masked L2 over ALL ranks' NORMAL|OUTFLOW nodes (same value on every rank); its local
    gradient, SUMMED over ranks (GradAllReduce(average=False)), is the global gradient."""
    from .nodetype import NodeType
    mask = (node_type_own == int(NodeType.NORMAL)) | (node_type_own == int(NodeType.OUTFLOW))
    w = mask.to(net_out_own.dtype).unsqueeze(1)
    num = ((net_out_own - target_own) ** 2 * w).sum()
    cnt = (w.sum() * net_out_own.shape[1]).detach()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(cnt, group=group)
    return num / cnt
