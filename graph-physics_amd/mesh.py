"""Graph container and synthetic meshes that follow the hot path's input contract.

Input contract (reference
graphphysics/dataset/preprocessing.py:16-23,421-424 and PyG collate):
  * ``edge_index[2,E]`` int64 = symmetric closure of the triangle sides,
    coalesced: sorted by (src, dst), no duplicates, no self loops;
  * ``edge_attr = [pos[src]-pos[dst], ||pos[dst]-pos[src]||_2]``;
  * a batch is the block-diagonal union of graphs with node-offset indices.

There is no dataset on the GPU box, so bench / tests build CylinderFlow-shaped
meshes here (scipy Delaunay, deterministic for a given seed).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from .nodetype import NodeType


class Graph:
    """Attribute bag with the fields the reference reads off a PyG ``Data``
    (``x, y, pos, edge_index, edge_attr``); missing attributes read as None, as
    PyG's ``Data.__getattr__`` does."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __getattr__(self, name):  # only called when the attribute is missing
        if name.startswith("__"):
            raise AttributeError(name)
        return None

    def to(self, device, non_blocking: bool = False):
        out = Graph()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
        return out

    def clone(self):
        out = Graph()
        for k, v in self.__dict__.items():
            setattr(out, k, v.clone() if torch.is_tensor(v) else v)
        return out


def faces_to_edges(faces: np.ndarray, num_nodes: int) -> np.ndarray:
    """Triangles/tetrahedra -> symmetric, coalesced directed edge list [2,E]
    (what ``T.FaceToEdge`` + ``to_undirected`` yield, preprocessing.py:421)."""
    k = faces.shape[1]
    pairs = []
    for a in range(k):
        for b in range(a + 1, k):
            pairs.append(faces[:, [a, b]])
    und = np.concatenate(pairs, axis=0)
    both = np.concatenate([und, und[:, ::-1]], axis=0).astype(np.int64)
    both = both[both[:, 0] != both[:, 1]]
    key = np.unique(both[:, 0] * np.int64(num_nodes) + both[:, 1])  # sorted by (src, dst)
    return np.stack([key // num_nodes, key % num_nodes], axis=0)


def edge_features(pos: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """Cartesian(norm=False) then Distance(norm=False), preprocessing.py:16-23."""
    src, dst = edge_index[0], edge_index[1]
    cart = pos[src] - pos[dst]
    dist = torch.norm(pos[dst] - pos[src], p=2, dim=-1, keepdim=True)
    return torch.cat([cart, dist], dim=-1)


def cylinder_mesh(n_nodes: int = 1885, seed: int = 0) -> Graph:
    """CylinderFlow-like 2-D mesh: uniform
    points in [0,1.6]x[0,0.41] minus a disc r=0.05 at (0.33,0.2); Delaunay;
    x = [v_x, v_y, node_type, t]; y = next velocity."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    pts = np.zeros((0, 2))
    while pts.shape[0] < n_nodes:
        cand = rng.uniform([0.0, 0.0], [1.6, 0.41], size=(n_nodes, 2))
        keep = np.hypot(cand[:, 0] - 0.33, cand[:, 1] - 0.2) > 0.05
        pts = np.concatenate([pts, cand[keep]], axis=0)
    pts = pts[:n_nodes]
    tri = Delaunay(pts)
    simp = tri.simplices
    cen = pts[simp].mean(axis=1)
    simp = simp[np.hypot(cen[:, 0] - 0.33, cen[:, 1] - 0.2) > 0.05]  # carve the hole
    ei = faces_to_edges(simp, n_nodes)

    node_type = np.full(n_nodes, int(NodeType.NORMAL), dtype=np.float32)
    node_type[(pts[:, 1] < 0.01) | (pts[:, 1] > 0.40)] = int(NodeType.WALL_BOUNDARY)
    node_type[np.hypot(pts[:, 0] - 0.33, pts[:, 1] - 0.2) < 0.06] = int(NodeType.WALL_BOUNDARY)
    node_type[pts[:, 0] < 0.02] = int(NodeType.INFLOW)
    node_type[pts[:, 0] > 1.58] = int(NodeType.OUTFLOW)

    vel = rng.standard_normal((n_nodes, 2)).astype(np.float32)
    t = np.zeros((n_nodes, 1), dtype=np.float32)
    x = np.concatenate([vel, node_type[:, None], t], axis=1)
    y = vel + 0.01 * rng.standard_normal((n_nodes, 2)).astype(np.float32)

    pos = torch.from_numpy(pts.astype(np.float32))
    edge_index = torch.from_numpy(ei)
    return Graph(
        x=torch.from_numpy(x),
        y=torch.from_numpy(y.astype(np.float32)),
        pos=pos,
        face=torch.from_numpy(simp.T.astype(np.int64)),
        edge_index=edge_index,
        edge_attr=edge_features(pos, edge_index),
    )


def plate_mesh(n_nodes: int = 1300, seed: int = 61, radius: float = 0.1, device=None) -> Graph:
    """DeformingPlate-shaped sample (BASELINE configs[2]; training_config/plate.json:23-29): x = [world_pos(3),
    obstacle displacement(3), node_type], y = next world_pos; a block of OBSTACLE nodes hovering next to the NORMAL
    plate nodes; tetrahedra of a 3-D Delaunay.  The reference's per-sample transforms run ON THE DEVICE
    (FaceToEdge -> add_world_edges -> Cartesian + Distance, dataset/preprocessing.py:92-140,16-23), so ``device``
    must be a GPU."""
    from scipy.spatial import Delaunay

    from . import preprocess as PP

    rng = np.random.default_rng(seed)
    pts = (rng.random((n_nodes, 3)) * np.array([1.0, 0.3, 0.3])).astype(np.float32)
    types = np.where(pts[:, 0] < 0.25, float(int(NodeType.OBSTACLE)), float(int(NodeType.NORMAL))).astype(np.float32)
    types[rng.integers(0, n_nodes, 40)] = float(int(NodeType.HANDLE))
    disp = (0.01 * rng.standard_normal((n_nodes, 3))).astype(np.float32)
    x = torch.from_numpy(np.concatenate([pts, disp, types[:, None]], axis=1))
    y = torch.from_numpy((pts + 0.01 * rng.standard_normal((n_nodes, 3))).astype(np.float32))
    cells = torch.from_numpy(Delaunay(pts).simplices.T.astype(np.int64))
    xd, pos = x.to(device), torch.from_numpy(pts).to(device)
    ei = PP.add_world_edges(xd, PP.faces_to_edges(cells.to(device), n_nodes), 0, 3, 6, radius=radius)
    return Graph(x=xd, y=y.to(device), pos=pos, face=cells.to(device), edge_index=ei, edge_attr=PP.edge_features(pos, ei))


def square_mesh(n_nodes: int, seed: int = 0, spatial_sort: bool = False) -> Graph:
    """Config C4: 2-D Delaunay of ``n_nodes`` uniform points in the unit square, seed 0
    (~3 undirected edges per node), nodes numbered in the order the generator drew them -- NO locality.
    Locality renumbering is the ENGINE's job (ops.set_node_renumbering / Topology(renumber=...)), applied to any
    input and timed with the topology build.  ``spatial_sort=True`` pre-sorts the points along a Morton curve
    inside the generator (the round-2 behaviour; kept for A/B measurements only)."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    pts = rng.uniform(0.0, 1.0, size=(n_nodes, 2))
    if spatial_sort:
        pts = pts[np.argsort(morton2d(pts), kind="stable")]
    simp = Delaunay(pts).simplices
    ei = faces_to_edges(simp, n_nodes)
    node_type = np.full(n_nodes, int(NodeType.NORMAL), dtype=np.float32)
    b = (pts[:, 0] < 2e-3) | (pts[:, 0] > 1 - 2e-3) | (pts[:, 1] < 2e-3) | (pts[:, 1] > 1 - 2e-3)
    node_type[b] = int(NodeType.WALL_BOUNDARY)
    vel = rng.standard_normal((n_nodes, 2)).astype(np.float32)
    x = np.concatenate([vel, node_type[:, None], np.zeros((n_nodes, 1), np.float32)], axis=1)
    y = vel + 0.01 * rng.standard_normal((n_nodes, 2)).astype(np.float32)
    pos = torch.from_numpy(pts.astype(np.float32))
    edge_index = torch.from_numpy(ei)
    return Graph(x=torch.from_numpy(x), y=torch.from_numpy(y), pos=pos,
                 edge_index=edge_index, edge_attr=edge_features(pos, edge_index))


def morton2d(pts: np.ndarray, bits: int = 16) -> np.ndarray:
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    q = ((pts - lo) / np.maximum(hi - lo, 1e-30) * ((1 << bits) - 1)).astype(np.uint64)

    def spread(v):
        v = v & np.uint64(0xFFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
        return v

    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))


def collate(graphs: Sequence[Graph]) -> Graph:
    """Block-diagonal union, as PyG ``Batch.from_data_list`` (train.py:160-168)."""
    xs, ys, ps, eis, eas, batch = [], [], [], [], [], []
    off = 0
    for i, g in enumerate(graphs):
        xs.append(g.x)
        if g.y is not None:
            ys.append(g.y)
        ps.append(g.pos)
        eis.append(g.edge_index + off)
        eas.append(g.edge_attr)
        batch.append(torch.full((g.x.shape[0],), i, dtype=torch.int64))
        off += g.x.shape[0]
    return Graph(
        x=torch.cat(xs), y=torch.cat(ys) if ys else None, pos=torch.cat(ps),
        edge_index=torch.cat(eis, dim=1), edge_attr=torch.cat(eas), batch=torch.cat(batch),
        num_graphs=len(graphs),
    )


def cylinder_batch(batch_size: int = 16, n_nodes: int = 1885, seed0: int = 0) -> Graph:
    """Config C2: ``batch_size`` cylinder meshes with seeds seed0..seed0+b-1."""
    return collate([cylinder_mesh(n_nodes, seed0 + i) for i in range(batch_size)])
