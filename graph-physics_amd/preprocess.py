"""On-device input construction: what the reference's CPU
dataloader does per sample with PyG transforms -- ``T.FaceToEdge`` + ``to_undirected``,
``T.Cartesian(norm=False)``, ``T.Distance(norm=False)`` (graphphysics/dataset/preprocessing.py:
16-23,421-424; torch-geometric==2.6.1) -- as HIP kernels on the MI355X (csrc/mgn_prep.hip).
There is no CPU path here: CPU tensors raise ``RuntimeError``."""
from __future__ import annotations

from typing import Optional

import torch

from . import _capi
from .ops import _ptr, _require_device, _stream


def faces_to_edges(face: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """face [K,F] (K = 3 triangles or 4 tetrahedra, PyG ``data.face`` layout) -> edge_index [2,E]
    int64: every pair of corners in both directions, sorted by (src,dst), duplicates and self
    loops removed.  One-time topology prep: synchronises once to read E."""
    _require_device(face)
    if face.dim() != 2 or face.shape[0] not in (3, 4):
        raise ValueError("face must have shape [3, F] or [4, F]")
    L = _capi.lib()
    dev = face.device
    f = face.to(torch.int64).contiguous()
    K, F = int(f.shape[0]), int(f.shape[1])
    cap = F * K * (K - 1)
    out = torch.empty(2, max(cap, 1), dtype=torch.int64, device=dev)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(L.mgn_faces_to_edges_workspace_bytes(F, K), 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.mgn_faces_to_edges(_ptr(f), K, F, int(num_nodes), out[0].data_ptr(), out[1].data_ptr(), _ptr(n),
                                  _ptr(ws), ws.numel(), _stream(dev))
    if rc == 3:
        raise IndexError(f"face has corners outside [0, {num_nodes})")
    _capi.check(rc, "mgn_faces_to_edges", prep=True)
    E = int(n.item())
    return out[:, :E].contiguous()


def edge_features(pos: torch.Tensor, edge_index: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """edge_attr [E, D+1] = [pos[src]-pos[dst], ||pos[dst]-pos[src]||_2] (Cartesian then Distance)."""
    _require_device(pos, edge_index)
    D = int(pos.shape[1])
    if D not in (2, 3):
        raise ValueError("pos must be [N,2] or [N,3]")
    p = pos.to(torch.float32).contiguous()
    ei = edge_index.to(torch.int64).contiguous()
    E = int(ei.shape[1])
    if out is None:
        out = torch.empty(E, D + 1, dtype=torch.float32, device=pos.device)
    with torch.cuda.device(pos.device):
        rc = _capi.lib().mgn_edge_features(_ptr(p), D, ei[0].data_ptr(), ei[1].data_ptr(), E, _ptr(out), _stream(pos.device))
    _capi.check(rc, "mgn_edge_features", prep=True)
    return out


def add_world_edges(x: torch.Tensor, edge_index: torch.Tensor, world_pos_index_start: int, world_pos_index_end: int,
                    node_type_index: int, radius: float = 0.03, max_world_pairs: Optional[int] = None) -> torch.Tensor:
    """``add_world_edges`` of the reference (preprocessing.py:92-140): OBSTACLE-NORMAL node pairs
    within ``radius`` in world position, both directions, merged with ``edge_index`` and coalesced.
    ``max_world_pairs`` bounds the output buffer (default 16 per node)."""
    _require_device(x, edge_index)
    L = _capi.lib()
    dev = x.device
    xx = x.to(torch.float32).contiguous()
    N, D = int(xx.shape[0]), int(world_pos_index_end - world_pos_index_start)
    if D not in (2, 3):
        raise ValueError("world positions must be 2-D or 3-D")
    ei = edge_index.to(torch.int64).contiguous()
    E = int(ei.shape[1])
    mw = int(max_world_pairs if max_world_pairs is not None else 16 * N)
    out = torch.empty(2, 2 * E + 2 * mw + 1, dtype=torch.int64, device=dev)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(L.mgn_world_edges_workspace_bytes(E, mw), 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.mgn_add_world_edges(_ptr(xx), int(xx.shape[1]), int(world_pos_index_start), D, int(node_type_index), N, float(radius),
                                   ei[0].data_ptr(), ei[1].data_ptr(), E, mw, out[0].data_ptr(), out[1].data_ptr(), _ptr(n),
                                   _ptr(ws), ws.numel(), _stream(dev))
    if rc == 3:
        raise IndexError(f"edge_index has entries outside [0, {N})")
    if rc == 4:
        raise RuntimeError("more world-edge pairs than max_world_pairs; pass a larger bound")
    _capi.check(rc, "mgn_add_world_edges", prep=True)
    return out[:, :int(n.item())].contiguous()
