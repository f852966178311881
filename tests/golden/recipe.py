"""Deterministic inputs / weights shared by the golden generator (which loads them
into the REFERENCE modules in the build container) and by the tests (which load
the same numbers into the oracle and into the HIP engine anywhere).  Only numpy
``default_rng`` streams and scipy Delaunay are used, so the numbers are identical
on every machine with this image."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch


def epd_param_shapes(L: int, H: int, F_n: int, F_e: int, O: int, only_processor: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys/shapes of EncodeProcessDecode (processors.py:129-160)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def mlp(prefix, fin, fout, norm=True):
        dims = [fin, H, H, H, fout]
        for n, i in enumerate((0, 2, 4, 6)):
            sh[f"{prefix}{i}.weight"] = (dims[n + 1], dims[n])
            sh[f"{prefix}{i}.bias"] = (dims[n + 1],)
        if norm:
            sh[f"{prefix}7.scale"] = (fout,)

    if not only_processor:
        mlp("nodes_encoder.", F_n, H)
        mlp("edges_encoder.", F_e, H)
        mlp("decode_module.", H, O, norm=False)
    for i in range(L):
        mlp(f"processor_list.{i}.edge_block.", 3 * H, H)
        mlp(f"processor_list.{i}.node_block.", 2 * H, H)
    return sh


def make_params(shapes: "OrderedDict[str, Tuple[int, ...]]", seed: int) -> "OrderedDict[str, torch.Tensor]":
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, s in shapes.items():
        if k.endswith("weight"):
            a = np.sqrt(6.0 / s[1])
            v = rng.uniform(-a, a, size=s)
        elif k.endswith("bias"):
            v = rng.uniform(-0.1, 0.1, size=s)
        else:  # RMSNorm scale
            v = 1.0 + 0.1 * rng.standard_normal(size=s)
        out[k] = torch.from_numpy(v.astype(np.float32))
    return out


def delaunay_graph(n: int, seed: int, dim: int = 2):
    """(pos[n,dim] f32, edge_index[2,E] i64 symmetric+coalesced, edge_attr[E,dim+1] f32)."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    pts = rng.uniform(0.0, 1.0, size=(n, dim))
    simp = Delaunay(pts).simplices
    k = simp.shape[1]
    und = np.concatenate([simp[:, [a, b]] for a in range(k) for b in range(a + 1, k)], axis=0)
    both = np.concatenate([und, und[:, ::-1]], axis=0).astype(np.int64)
    key = np.unique(both[:, 0] * np.int64(n) + both[:, 1])
    ei = np.stack([key // n, key % n], axis=0)
    pos = torch.from_numpy(pts.astype(np.float32))
    edge_index = torch.from_numpy(ei)
    src, dst = edge_index[0], edge_index[1]
    cart = pos[src] - pos[dst]
    dist = torch.norm(pos[dst] - pos[src], p=2, dim=-1, keepdim=True)
    return pos, edge_index, torch.cat([cart, dist], dim=-1)


def random_graph(n: int, e: int, seed: int) -> torch.Tensor:
    """Arbitrary edge_index as in the reference's test_processors.py:23: unsorted,
    duplicates and self loops allowed; node n-1 is left isolated (zero in-degree)."""
    rng = np.random.default_rng(seed)
    ei = rng.integers(0, n - 1, size=(2, e)).astype(np.int64)
    ei[:, 0] = (3, 3)  # a guaranteed self loop
    ei[:, 1] = ei[:, 2]  # a guaranteed duplicate
    return torch.from_numpy(ei)


def randn(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    return torch.from_numpy((scale * np.random.default_rng(seed).standard_normal(size=shape)).astype(np.float32))


CYL_INDEX = {"feature_index_start": 0, "feature_index_end": 2, "output_index_start": 0,
             "output_index_end": 2, "node_type_index": 2}


def trajectory(n: int, steps: int, seed: int):
    """Synthetic CylinderFlow-like trajectory on a fixed Delaunay mesh:
    frames x_t = [v_x, v_y, node_type, t], y_t = v_{t+1}."""
    pos, edge_index, edge_attr = delaunay_graph(n, seed)
    rng = np.random.default_rng(seed + 1000)
    node_type = np.zeros(n, dtype=np.float32)
    p = pos.numpy()
    node_type[(p[:, 1] < 0.08) | (p[:, 1] > 0.92)] = 6  # WALL_BOUNDARY
    node_type[p[:, 0] < 0.08] = 4  # INFLOW
    node_type[p[:, 0] > 0.92] = 5  # OUTFLOW
    vel = [rng.standard_normal((n, 2)).astype(np.float32)]
    for _ in range(steps):
        vel.append((vel[-1] + 0.05 * rng.standard_normal((n, 2))).astype(np.float32))
    xs, ys = [], []
    for t in range(steps):
        x = np.concatenate([vel[t], node_type[:, None], np.full((n, 1), t, np.float32)], axis=1)
        xs.append(torch.from_numpy(x))
        ys.append(torch.from_numpy(vel[t + 1]))
    return pos, edge_index, edge_attr[:, :3], xs, ys


# ---------------------------------------------------------------- GraphNetBlock variants (N3)
#: name -> options.  ``variant`` is what the oracle takes, ``act`` the global activation switch,
#: ``phi``: feed a per-node scalar to the gate.
VARIANTS = {
    "silu": dict(seed=1, act="silu", variant={}),
    "gate": dict(seed=2, act="relu", variant={"use_gate": True}),
    "gate_phi": dict(seed=3, act="relu", variant={"use_gate": True}, phi=True),
    "rope3": dict(seed=4, act="relu", variant={"use_rope": True, "rope_axes": 3, "rope_base": 10000.0}),
    "rope2": dict(seed=5, act="relu", variant={"use_rope": True, "rope_axes": 2, "rope_base": 100.0}),
    "gated": dict(seed=6, act="relu", variant={"use_gated_mlp": True}),
    "gated_silu": dict(seed=7, act="silu", variant={"use_gated_mlp": True, "use_gate": True}, phi=True),
    "l3nonorm": dict(seed=8, act="relu", variant={}, nb_of_layers=3, layer_norm=False),
    "combo": dict(seed=9, act="silu", variant={"use_gate": True, "use_rope": True, "rope_axes": 3, "rope_base": 10000.0}, phi=True),
}


def block_kwargs(v):
    """GraphNetBlock(...) keyword arguments of a variant (reference signature, layers.py:896-906)"""
    vv = v["variant"]
    kw = dict(use_rope=vv.get("use_rope", False), rope_axes=vv.get("rope_axes", 3), rope_base=vv.get("rope_base", 10000.0),
              use_gated_mlp=vv.get("use_gated_mlp", False), use_gate=vv.get("use_gate", False))
    if "nb_of_layers" in v:
        kw.update(nb_of_layers=v["nb_of_layers"], layer_norm=v["layer_norm"])
    return kw


def epd_kwargs(v):
    """EncodeProcessDecode(...) keyword arguments (processors.py:67-81; nb_of_layers / layer_norm are not
    reachable through it, so the l3nonorm variant is block-only there and runs the default EPD)"""
    vv = v["variant"]
    return dict(use_rope_embeddings=vv.get("use_rope", False), rope_pos_dimension=vv.get("rope_axes", 3),
                rope_base=vv.get("rope_base", 10000.0), use_gated_mlp=vv.get("use_gated_mlp", False),
                use_gated_attention=vv.get("use_gate", False))


def variant_params(state_dict, seed: int, key_order=None) -> "OrderedDict[str, torch.Tensor]":
    """deterministic weights for ANY module layout: shapes are read off ``state_dict`` (key order), values
    come from the numpy stream -- weights uniform(+-sqrt(6/fan_in)), biases / gate_pos uniform(+-0.1),
    norm scales 1 + 0.1 N(0,1)"""
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    order = list(key_order) if key_order is not None else list(state_dict.keys())
    assert set(order) == set(state_dict.keys()), sorted(set(order) ^ set(state_dict.keys()))
    for k in order:
        t = state_dict[k]
        if k.endswith("inv_freq"):  # persistent RoPE buffer (layers.py:617-619): a constant, not a weight
            out[k] = t.detach().clone()
            continue
        s = tuple(t.shape)
        if k.endswith("scale"):
            v = 1.0 + 0.1 * rng.standard_normal(size=s)
        elif len(s) == 2:
            a = np.sqrt(6.0 / s[1])
            v = rng.uniform(-a, a, size=s)
        else:
            v = rng.uniform(-0.1, 0.1, size=s)
        out[k] = torch.from_numpy(v.astype(np.float32))
    return out


# ------------------------------------------------ sparse-attention Transformer cases (N4)
TRANSFORMER_CASES = {
    "etd_h64": dict(model="etd", hidden=64, heads=4, L=2, N=150, seed=501),                       # coarse-aneurysm.json shape
    "etd_h128_rope_gate": dict(model="etd", hidden=128, heads=4, L=2, N=120, seed=502, rope=True, gate=True, pos_dim=3),
    "etd_h32_heads8_temporal": dict(model="etd", hidden=32, heads=8, L=3, N=100, seed=503, temporal=True),
    "etd_h64_heads2_rope2d": dict(model="etd", hidden=64, heads=2, L=2, N=100, seed=504, rope=True, pos_dim=2),
    "epd_temporal": dict(model="epd", hidden=128, heads=4, L=3, N=120, seed=505),
    # BASELINE.json configs[4] with training_config/coarse-aneurysm.json:11-22's exact model keys: 10 blocks, hidden 64, 4 heads,
    # node_input_size 14 (+ 9 one-hot = 23 inputs), output_size 3, on a 3-D tetrahedral mesh
    "etd_aneurysm": dict(model="etd", hidden=64, heads=4, L=10, N=500, seed=506, f_in=23, out=3, pos_dim=3),
}


# ------------------------------------------------ TemporalAttention WITHOUT an adjacency (an installation without DGL: adj = None)
TEMPORAL_DENSE_CASES = {
    "tmp_h128_heads4": dict(hidden=128, heads=4, N=90, seed=601),     # EncodeProcessDecode's temporal block (num_heads default)
    "tmp_h32_heads8": dict(hidden=32, heads=8, N=50, seed=602),
    "tmp_h64_heads1": dict(hidden=64, heads=1, N=40, seed=603),
    "tmp_h48_heads2_nogate": dict(hidden=48, heads=2, N=33, seed=604, gate=False),
}
EPD_TEMPORAL_NODGL = dict(hidden=128, L=3, N=120, seed=605)           # EncodeProcessDecode(use_temporal_block=True), HAS_DGL_SPARSE False
