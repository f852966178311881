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
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradAllReduce:
    """callable(params): average ``.grad`` over the group through one flat buffer."""

    def __init__(self, group=None):
        self.group = group
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
        self._flat.div_(dist.get_world_size(self.group))
        off = 0
        for p in ps:
            k = p.grad.numel()
            p.grad.copy_(self._flat[off:off + k].view_as(p.grad))
            off += k


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Make every replica start from rank ``src``'s weights and normaliser buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
