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


def add_noise(graph, noise_index_start, noise_index_end, noise_scale, node_type_index: int, t: Optional[float] = None,
              seed: int = 0, offset: int = 0):
    """``add_noise`` of the reference (preprocessing.py:177-238), on the device and in place: Gaussian noise
    on the given feature column ranges of the NORMAL nodes of ``graph.x``; with ``t`` the curriculum
    scale ``10 * std * (1 + cos(t * pi))``.  Same argument conventions and errors as the reference.
    The noise stream is counter-based (``mgn_add_noise``: Philox4x32-10 keyed by ``seed``, indexed by
    (row, column, range, ``offset``)) -- pass the training-step counter as ``offset``."""
    import ctypes as C
    import math

    if isinstance(noise_index_start, int):
        noise_index_start = [noise_index_start]
    if isinstance(noise_index_end, int):
        noise_index_end = [noise_index_end]
    if isinstance(noise_scale, (float, int)):
        noise_scale = [float(noise_scale)] * len(noise_index_start)
    if len(noise_index_start) != len(noise_index_end):
        raise ValueError("noise_index_start and noise_index_end must have the same length.")
    if len(noise_scale) != len(noise_index_start):
        raise ValueError("noise_scale must have the same length as noise_index_start and noise_index_end.")
    x = graph.x
    _require_device(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("graph.x must be a contiguous float32 tensor (noise is added in place)")
    n = len(noise_index_start)
    scales = [(10 * s * (1 + math.cos(t * math.pi)) if t is not None else s) for s in noise_scale]
    st, en, sc = (C.c_int * n)(*noise_index_start), (C.c_int * n)(*noise_index_end), (C.c_float * n)(*scales)
    with torch.cuda.device(x.device):
        rc = _capi.lib().mgn_add_noise(_ptr(x), int(x.shape[1]), int(x.shape[0]), n, st, en, sc, int(node_type_index),
                                       int(seed) & (2 ** 64 - 1), int(offset) & 0xFFFFFFFF, _stream(x.device))
    _capi.check(rc, "mgn_add_noise", prep=True)
    return graph


def add_obstacles_next_pos(graph, world_pos_index_start: int, world_pos_index_end: int, node_type_index: int):
    """``add_obstacles_next_pos`` (preprocessing.py:44-89): the obstacle's displacement to its next position
    becomes 3 extra node features (non-obstacle nodes get the obstacles' mean displacement).  A handful of
    elementwise device ops -- plumbing, no kernel of its own."""
    from .nodetype import NodeType

    _require_device(graph.x, graph.y)
    world_pos = graph.x[:, world_pos_index_start:world_pos_index_end]
    other = graph.x[:, world_pos_index_end:]
    disp = graph.y[:, world_pos_index_start:world_pos_index_end] - world_pos
    node_type = graph.x[:, node_type_index - 3]  # the index is the one AFTER the 3 columns are inserted (:78-80)
    is_obs = (node_type == int(NodeType.OBSTACLE)).unsqueeze(1)
    mean_obs = (disp * is_obs).sum(dim=0) / is_obs.sum().clamp_min(1)
    disp = torch.where(is_obs, disp, mean_obs.unsqueeze(0).expand_as(disp))
    graph.x = torch.cat([world_pos, disp, other], dim=1).contiguous()
    return graph


def add_world_pos_features(graph, world_pos_index_start: int, world_pos_index_end: int):
    """``add_world_pos_features`` (preprocessing.py:143-176): relative WORLD position and its norm appended to
    the edge features (the same kernel as the mesh-space edge features)."""
    wp = graph.x[:, world_pos_index_start:world_pos_index_end].contiguous()
    extra = edge_features(wp, graph.edge_index)
    graph.edge_attr = extra if graph.edge_attr is None else torch.cat([graph.edge_attr, extra], dim=1)
    return graph


def build_preprocessing(noise_parameters=None, world_pos_parameters=None, add_edges_features: bool = True,
                        extra_node_features=None, extra_edge_features=None, seed: int = 0):
    """Device-side ``build_preprocessing`` (preprocessing.py:380-444): the same transform ORDER as the
    reference -- extra node features, [obstacle next pos,] faces -> edges, [world edges,] edge features,
    noise inserted at position 1, extra edge features -- as one callable
    ``f(graph, step=0) -> graph`` over device tensors (``graph.face`` [K,F], ``graph.pos``, ``graph.x``,
    ``graph.y``).  Topology-only results (edge_index) can be cached by the caller across a trajectory."""
    steps = []
    if extra_node_features is not None:
        steps += list(extra_node_features) if isinstance(extra_node_features, (list, tuple)) else [extra_node_features]

    def face_to_edge(g, step):
        if g.edge_index is None:
            g.edge_index = faces_to_edges(g.face, g.x.shape[0])
        return g

    def feats(g, step):
        g.edge_attr = edge_features(g.pos, g.edge_index)
        return g

    if world_pos_parameters is not None:
        w = world_pos_parameters
        steps.append(lambda g, step: add_obstacles_next_pos(g, w["world_pos_index_start"], w["world_pos_index_end"], w["node_type_index"]))
        steps.append(face_to_edge)

        def world(g, step):
            g.edge_index = add_world_edges(g.x, g.edge_index, w["world_pos_index_start"], w["world_pos_index_end"], w["node_type_index"],
                                           radius=w.get("radius", 0.03))
            return g
        steps.append(world)
        steps.append(feats)   # (the reference's pipeline stops here: add_world_pos_features stays a stand-alone transform, :401-420)
    else:
        steps.append(face_to_edge)
        if add_edges_features:
            steps.append(feats)
    if noise_parameters is not None:
        p = noise_parameters
        steps.insert(1, lambda g, step: add_noise(g, p["noise_index_start"], p["noise_index_end"], p["noise_scale"], p["node_type_index"],
                                                  seed=seed, offset=step))
    if extra_edge_features is not None:
        steps += list(extra_edge_features) if isinstance(extra_edge_features, (list, tuple)) else [extra_edge_features]

    def run(graph, step: int = 0):
        import inspect
        for f in steps:
            graph = f(graph, step) if len(inspect.signature(f).parameters) >= 2 else f(graph)
        return graph

    return run
