"""CPU oracle for the MeshGraphNet message-passing path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain ``torch`` CPU ops (no PyG, no Lightning), the
algorithm of the reference's hot path so that the HIP engine can be checked
against it anywhere (the reference source never travels to the GPU box).

Who may import this: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / timed baseline only.
The product package (``graph-physics_amd/``) must never import it.

Parity status: the reference's own tests hold NO numeric golden vectors for
this path (shape / "grad exists" assertions only), so the
oracle is pinned against outputs of the reference itself, imported in the build
container by ``tests/golden/make_golden.py`` (PyG's ``MessagePassing.propagate``
-- un-vendored ``torch-geometric==2.6.1``, ``requirements.txt:7`` -- restated
there as ``zeros(N,H).index_add_(0, edge_index[1], msg)``).  The committed
fixtures under ``tests/golden/*.npz`` carry the reference's outputs; the test
``tests/test_oracle_golden.py`` checks this file against them bit-for-bit.

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

NODE_TYPE_SIZE = 9  # graphphysics/utils/nodetype.py:12  (NodeType.SIZE)
NODE_NORMAL = 0  # nodetype.py:5
NODE_OUTFLOW = 5  # nodetype.py:10


# --------------------------------------------------------------------------- R1
def rms_norm(x: torch.Tensor, scale: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """graphphysics/models/layers.py:104-129 (default p=-1, no bias).

    y = scale * x / (||x||_2 / sqrt(d) + eps)  -- eps is added to the RMS, not
    inside the square root.
    """
    d = x.shape[-1]
    norm_x = x.norm(2, dim=-1, keepdim=True)  # layers.py:116
    rms_x = norm_x / math.sqrt(d)  # layers.py:123
    x_normed = x / (rms_x + eps)  # layers.py:124
    return scale * x_normed  # layers.py:129


def rms_norm_general(x: torch.Tensor, scale: torch.Tensor, d: int, p: float = -1.0, eps: float = 1e-8,
                     offset: torch.Tensor = None) -> torch.Tensor:
    """RMSNorm.forward with every constructor option (layers.py:104-129): p in [0, 1] norms the first int(d * p) columns only
    (:113-121), bias adds ``offset`` (:126-127); eps outside the root (:123-124)."""
    if p < 0.0 or p > 1.0:
        norm_x, d_x = x.norm(2, dim=-1, keepdim=True), d
    else:
        k = int(d * p)
        norm_x, d_x = x[..., :k].norm(2, dim=-1, keepdim=True), k
    rms_x = norm_x / math.sqrt(d_x)
    y = scale * (x / (rms_x + eps))
    return y + offset if offset is not None else y


# --------------------------------------------------------------------------- R2
#: bf16-mixed evaluation (set by :func:`bf16_mixed`).  The reference trains with Lightning
#: ``precision="bf16-mixed"`` when ``training.enable_vram_optimizations`` is set (train.py:74-78,
#: 268-293) -- on its target (a GPU) that is ``torch.autocast("cuda", torch.bfloat16)``, whose op
#: lists (torch.amp docs, "CUDA Ops that can autocast to float16 / float32") cast ``linear`` inputs,
#: weight and bias to bf16 (bf16 result, fp32 accumulation inside the GEMM) and run ``norm`` in
#: fp32; everything else follows type promotion.  CPU autocast has different lists (``norm`` stays
#: bf16 there), so the oracle states the CUDA semantic explicitly instead of relying on
#: ``torch.autocast("cpu")``: Linear in bf16, activation in bf16, RMSNorm on the bf16 result in
#: fp32, residual stream / aggregation / loss in fp32.
_BF16_MIXED = False


class bf16_mixed:
    """context manager: evaluate ``mlp`` (and everything built on it) with the bf16-mixed semantic"""

    def __enter__(self):
        global _BF16_MIXED
        self._old, _BF16_MIXED = _BF16_MIXED, True

    def __exit__(self, *a):
        global _BF16_MIXED
        _BF16_MIXED = self._old


def mlp(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str, act: str = "relu",
        pre_out: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """build_mlp(nb_of_layers=4) forward, graphphysics/models/layers.py:163-210.
    ``pre_out`` (tests): receives the pre-activations of the three hidden layers.

    Sequential entries 0,2,4,6 are nn.Linear (y = x W^T + b, W[out,in]); 1,3,5
    are the activation (ReLU unless the global SiLU flag is set,
    layers.py:150-160); entry 7 is RMSNorm when present (``layer_norm=True``).
    ``p`` is a state_dict; ``prefix`` e.g. "processor_list.0.edge_block.".
    """
    f = {"relu": torch.relu, "silu": torch.nn.functional.silu, "gelu": torch.nn.functional.gelu}[act]  # layers.py:150-160
    h = x
    idx = []  # Linear entries 0, 2, 4, ... (nb_of_layers of them; 4 unless the block was built otherwise)
    while f"{prefix}{2 * len(idx)}.weight" in p:
        idx.append(2 * len(idx))
    for i in idx:
        W, b = p[f"{prefix}{i}.weight"], p[f"{prefix}{i}.bias"]
        if _BF16_MIXED:
            h = torch.nn.functional.linear(h.to(torch.bfloat16), W.to(torch.bfloat16), b.to(torch.bfloat16))
        else:
            h = torch.nn.functional.linear(h, W, b)
        if i != idx[-1]:
            if pre_out is not None:
                pre_out.append(h.detach())
            h = f(h)
    if _BF16_MIXED:
        h = h.float()  # norm runs in fp32 under CUDA autocast; bf16 / fp32 promotes to fp32 (exact widening)
    key = f"{prefix}{idx[-1] + 1}.scale"
    if key in p:
        h = rms_norm(h, p[key])
    return h


def _linear(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.Linear; under :class:`bf16_mixed` as CUDA autocast runs it: inputs, weight and bias cast to bf16, bf16 result
    (fp32 accumulation inside the GEMM)."""
    if _BF16_MIXED:
        return torch.nn.functional.linear(x.to(torch.bfloat16), W.to(torch.bfloat16), None if b is None else b.to(torch.bfloat16))
    return torch.nn.functional.linear(x, W, b)


def gated_mlp(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str, act: str = "relu") -> torch.Tensor:
    """build_gated_mlp, graphphysics/models/layers.py:256-278: Sequential(RMSNorm(in) [entry 0],
    GatedMLP [entry 1: act(linear1 x) * linear2 x, layers.py:249-253; activation SiLU when the global
    switch is set, else GELU, :235-236], Linear [entry 2]).  Under bf16_mixed: the norm in fp32, the three Linears, the
    activation and the product in bf16 (autocast op lists; the product of two bf16 tensors is bf16)."""
    h = rms_norm(x.float(), p[prefix + "0.scale"])
    a = torch.nn.functional.silu if act == "silu" else torch.nn.functional.gelu
    left = a(_linear(h, p[prefix + "1.linear1.weight"], p[prefix + "1.linear1.bias"]))
    right = _linear(h, p[prefix + "1.linear2.weight"], p[prefix + "1.linear2.bias"])
    return _linear(left * right, p[prefix + "2.weight"], p[prefix + "2.bias"])


def rope_inv_freq(hidden_size: int, rope_axes: int, rope_base: float) -> torch.Tensor:
    """GraphNetBlock.__init__, layers.py:965-976"""
    pair_count = hidden_size // (2 * rope_axes)
    inv = torch.arange(pair_count, dtype=torch.float32)
    return torch.pow(rope_base, -inv / max(float(pair_count), 1.0))


def apply_rope_rel(x_src: torch.Tensor, delta_pos: torch.Tensor, inv_freq: torch.Tensor, rope_axes: int) -> torch.Tensor:
    """GraphNetBlock._apply_rope_rel, layers.py:1104-1149: per axis, pairs (even, odd) of the next
    2*pair_count channels rotate by theta = delta[axis] * inv_freq[pair]."""
    pair_count = inv_freq.numel()
    if pair_count == 0:
        return x_src
    E = x_src.shape[0]
    rope_dim = pair_count * 2 * rope_axes
    parts, start = [], 0
    for axis in range(rope_axes):
        seg = x_src[:, start:start + 2 * pair_count].reshape(E, pair_count, 2)
        theta = delta_pos[:, axis].unsqueeze(1) * inv_freq.unsqueeze(0)
        c, s_ = torch.cos(theta), torch.sin(theta)
        even, odd = seg[..., 0], seg[..., 1]
        parts.append(torch.stack([even * c - odd * s_, even * s_ + odd * c], dim=-1).reshape(E, 2 * pair_count))
        start += 2 * pair_count
    return torch.cat(parts + [x_src[:, rope_dim:]], dim=-1)


# ------------------------------------------------------------------- R3, R4, R5
def graph_net_block(
    x: torch.Tensor,
    e: torch.Tensor,
    edge_index: torch.Tensor,
    p: Dict[str, torch.Tensor],
    prefix: str,
    act: str = "relu",
    return_intermediates: bool = False,
    variant: Optional[dict] = None,
    pos: Optional[torch.Tensor] = None,
    phi: Optional[torch.Tensor] = None,
):
    """GraphNetBlock.forward, graphphysics/models/layers.py:989-1042.

    row, col = edge_index; x_i = x[col] (target), x_j = x[row] (source)
    (layers.py:1016-1018).  Message m = edge_block(cat[e, x_i, x_j])
    (layers.py:1058-1059).  Aggregation is PyG ``aggr="add"``,
    ``flow="source_to_target"`` (layers.py:926): agg[i] = sum_{k: col[k]=i} m[k],
    summed in edge-id order on CPU (``index_add_``).  Node update
    node_block(cat[x, agg]) (layers.py:1100-1101); residuals layers.py:1039-1040
    -- note the aggregated message is the PRE-residual MLP output.
    """
    v = variant or {}
    row, col = edge_index[0], edge_index[1]
    x_i = x[col]
    x_j = x[row]
    if v.get("use_rope"):  # layers.py:1020-1026: relative RoPE on the SOURCE features
        axes = v.get("rope_axes", 3)
        delta_pos = pos[row, :axes] - pos[col, :axes]
        x_j = apply_rope_rel(x_j, delta_pos, rope_inv_freq(x.shape[1], axes, v.get("rope_base", 10000.0)), axes)
    pre_e: Optional[List[torch.Tensor]] = [] if return_intermediates else None
    pre_n: Optional[List[torch.Tensor]] = [] if return_intermediates else None
    gated = v.get("use_gated_mlp", False)
    if gated:  # layers.py:932-942
        m = gated_mlp(torch.cat([e, x_i, x_j], dim=-1), p, prefix + "edge_block.", act)
    else:
        m = mlp(torch.cat([e, x_i, x_j], dim=-1), p, prefix + "edge_block.", act, pre_e)
    agg = torch.zeros(x.shape[0], m.shape[1], dtype=m.dtype).index_add_(0, col, m)
    if v.get("use_gate"):  # update(), layers.py:1091-1098
        logits = torch.nn.functional.linear(x, p[prefix + "gate_proj.weight"], p[prefix + "gate_proj.bias"])
        if phi is not None:
            logits = logits + phi.view(-1, 1) * p[prefix + "gate_pos"].view(1, -1)
        agg = agg * torch.sigmoid(logits)
    if gated:
        upd = gated_mlp(torch.cat([x, agg], dim=-1), p, prefix + "node_block.", act)
    else:
        upd = mlp(torch.cat([x, agg], dim=-1), p, prefix + "node_block.", act, pre_n)
    e_new = e + m
    x_new = x + upd
    if return_intermediates:
        return x_new, e_new, {"m": m, "agg": agg, "upd": upd, "edge_pre": pre_e, "node_pre": pre_n}
    return x_new, e_new


# --------------------------------------------------------------------------- R6
def epd_forward(
    x_in: torch.Tensor,
    edge_attr_in: torch.Tensor,
    edge_index: torch.Tensor,
    p: Dict[str, torch.Tensor],
    message_passing_num: int,
    only_processor: bool = False,
    act: str = "relu",
    per_round: Optional[List[torch.Tensor]] = None,
    intermediates: Optional[List[dict]] = None,
    variant: Optional[dict] = None,
    pos: Optional[torch.Tensor] = None,
    phi: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """EncodeProcessDecode.forward, graphphysics/models/processors.py:162-215.  ``variant``: the block
    options the constructor forwards (:129-160: use_rope / rope_axes / rope_base / use_gate /
    use_gated_mlp); ``act`` = the global SiLU switch; ``pos`` / ``phi`` as read off the graph
    (:186-192).  The temporal block is not restated."""
    if only_processor:
        x, e = x_in, edge_attr_in  # processors.py:176-177
    else:
        x = mlp(x_in, p, "nodes_encoder.", act)  # processors.py:179
        e = mlp(edge_attr_in, p, "edges_encoder.", act)  # processors.py:180
    for i in range(message_passing_num):  # processors.py:193-202
        if intermediates is not None:
            x, e, inter = graph_net_block(x, e, edge_index, p, f"processor_list.{i}.", act, return_intermediates=True,
                                          variant=variant, pos=pos, phi=phi)
            intermediates.append(inter)
        else:
            x, e = graph_net_block(x, e, edge_index, p, f"processor_list.{i}.", act, variant=variant, pos=pos, phi=phi)
        if per_round is not None:
            per_round.append(x)
    if only_processor:
        return x  # processors.py:211-212
    return mlp(x, p, "decode_module.", act)  # processors.py:214 (no RMSNorm: :141-146)


# --------------------------------------------------------------------------- R7
class NormalizerState:
    """Running-sum normaliser, graphphysics/models/layers.py:281-391."""

    def __init__(self, size: int, max_accumulations: float = 10**5, std_epsilon: float = 1e-8):
        self.max_accumulations = max_accumulations  # layers.py:311
        self.std_epsilon = torch.tensor(std_epsilon, dtype=torch.float32)  # layers.py:312
        self.acc_count = torch.tensor(0.0)  # layers.py:315
        self.num_accumulations = torch.tensor(0.0)  # layers.py:316
        self.acc_sum = torch.zeros(1, size)  # layers.py:317
        self.acc_sum_squared = torch.zeros(1, size)  # layers.py:323

    def mean(self):  # layers.py:378-382
        return self.acc_sum / torch.max(self.acc_count, torch.tensor(1.0))

    def std(self):  # layers.py:384-391
        safe = torch.max(self.acc_count, torch.tensor(1.0))
        var = self.acc_sum_squared / safe - self.mean() ** 2
        return torch.max(torch.sqrt(torch.clamp(var, min=0.0)), self.std_epsilon)

    def __call__(self, data: torch.Tensor, accumulate: bool = True) -> torch.Tensor:  # layers.py:331-349
        if accumulate and self.num_accumulations < self.max_accumulations:
            d = data.detach()  # layers.py:363-376
            self.acc_sum += d.sum(dim=0, keepdim=True)
            self.acc_sum_squared += (d**2).sum(dim=0, keepdim=True)
            self.acc_count += d.shape[0]
            self.num_accumulations += 1
        return (data - self.mean()) / self.std()

    def inverse(self, data: torch.Tensor) -> torch.Tensor:  # layers.py:351-361
        return data * self.std() + self.mean()

    def load(self, sd: Dict[str, torch.Tensor], prefix: str):
        self.acc_count = sd[prefix + "_acc_count"].clone()
        self.num_accumulations = sd[prefix + "_num_accumulations"].clone()
        self.acc_sum = sd[prefix + "_acc_sum"].clone()
        self.acc_sum_squared = sd[prefix + "_acc_sum_squared"].clone()


class SimulatorOracle:
    """Simulator.forward, graphphysics/models/simulator.py:145-217."""

    def __init__(self, index: Dict[str, int], node_input_size: int, edge_input_size: int, output_size: int):
        self.ix = index
        self.out_norm = NormalizerState(output_size)  # simulator.py:65-67
        self.node_norm = NormalizerState(node_input_size)  # simulator.py:68-70
        self.edge_norm = NormalizerState(edge_input_size) if edge_input_size > 0 else None  # :71-75

    def build_input(self, x, y, edge_attr, training: bool):
        ix = self.ix
        pre_target = x[:, ix["output_index_start"] : ix["output_index_end"]]  # simulator.py:91
        target = None
        if y is not None:
            target = self.out_norm(y - pre_target, training)  # simulator.py:106-110
        node_type = x[:, ix["node_type_index"]]
        one_hot = torch.nn.functional.one_hot(torch.squeeze(node_type.long()), NODE_TYPE_SIZE)  # :122-125
        feats = x[:, ix["feature_index_start"] : ix["feature_index_end"]]
        node_features = torch.cat([feats, one_hot], dim=1)  # simulator.py:139-140
        xn = self.node_norm(node_features, training)  # simulator.py:162
        en = self.edge_norm(edge_attr, training) if self.edge_norm is not None else edge_attr  # :164-167
        return xn, en, target

    def build_outputs(self, x, net_out):  # simulator.py:178-191
        ix = self.ix
        return x[:, ix["output_index_start"] : ix["output_index_end"]] + self.out_norm.inverse(net_out)


# --------------------------------------------------------------------------- R8
def l2_loss(net_out: torch.Tensor, target: torch.Tensor, node_type: torch.Tensor,
            masks: Sequence[int] = (NODE_NORMAL, NODE_OUTFLOW)) -> torch.Tensor:
    """L2Loss.forward, graphphysics/utils/loss.py:37-75 with the default masks of
    lightning_module.py:48: mean over the selected ROWS' elements."""
    mask = torch.zeros_like(node_type, dtype=torch.bool)
    for t in masks:
        mask |= node_type == t
    return torch.mean(((net_out - target) ** 2)[mask])


def lr_factor(step: int, warmup: int, max_iters: int, min_lr_factor: float = 0.001) -> float:
    """CosineWarmupScheduler.get_lr_factor, graphphysics/utils/scheduler.py:51-67.
    ``step`` is last_epoch (starts at 0 after construction)."""
    epoch = step + 1
    f = 0.5 * (1 + math.cos(math.pi * epoch / max_iters))
    if epoch <= warmup:
        f *= epoch * 1.0 / warmup
    return max(f, min_lr_factor)


def train_steps(
    params: Dict[str, torch.Tensor],
    sim: SimulatorOracle,
    batches: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]],
    message_passing_num: int,
    lr: float,
    warmup: int,
    num_steps: int,
    mixed: bool = False,
    grads_out: Optional[List[Dict[str, torch.Tensor]]] = None,
):
    """Reference training semantics: LightningModule.training_step
    (training/lightning_module.py:270-320) + configure_optimizers (:494-511) +
    Trainer(gradient_clip_val=1.0) (train.py:288).

    ``params`` are leaf tensors (requires_grad) updated in place.  Returns the
    list of (loss, grad_norm_before_clip) per step.  ``mixed``: the bf16-mixed semantic
    (``training.enable_vram_optimizations``); parameters, optimiser state and the
    normalisers stay fp32 as under Lightning's mixed precision.
    """
    names = list(params.keys())
    leaves = [params[k] for k in names]
    opt = torch.optim.AdamW(leaves, lr=lr, weight_decay=0.0001, betas=(0.9, 0.95))
    log = []
    for step, (x, y, edge_attr, edge_index) in enumerate(batches):
        for g in opt.param_groups:  # LR for THIS step = base_lr * factor(last_epoch=step)
            g["lr"] = lr * lr_factor(step, warmup, num_steps)
        xn, en, target = sim.build_input(x, y, edge_attr, training=True)
        if mixed:  # Lightning bf16-mixed: forward + loss inside autocast (train.py:74-78,268-293)
            with bf16_mixed():
                out = epd_forward(xn, en, edge_index, params, message_passing_num)
        else:
            out = epd_forward(xn, en, edge_index, params, message_passing_num)
        loss = l2_loss(out, target, x[:, sim.ix["node_type_index"]])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if grads_out is not None:  # un-clipped gradients of this step (tests)
            grads_out.append({k: params[k].grad.detach().clone() for k in names})
        gn = torch.nn.utils.clip_grad_norm_(leaves, 1.0)
        opt.step()
        log.append((float(loss.detach()), float(gn)))
    return log


# --------------------------------------------------------------------------- R9
def rollout(
    params: Dict[str, torch.Tensor],
    sim: SimulatorOracle,
    frames_x: Sequence[torch.Tensor],
    frames_y: Sequence[torch.Tensor],
    edge_attr: torch.Tensor,
    edge_index: torch.Tensor,
    message_passing_num: int,
) -> List[torch.Tensor]:
    """Autoregressive rollout, LightningModule._make_prediction
    (training/lightning_module.py:375-409) with build_mask (:27-35): nodes that
    are NOT NORMAL/OUTFLOW get the ground truth re-imposed."""
    ix = sim.ix
    last = None
    preds = []
    with torch.no_grad():
        for x, y in zip(frames_x, frames_y):
            x = x.clone()
            if last is not None:
                x[:, ix["output_index_start"] : ix["output_index_end"]] = last  # :379-382
            node_type = x[:, ix["node_type_index"]]
            mask = ~((node_type == NODE_NORMAL) | (node_type == NODE_OUTFLOW))  # :27-35
            xn, en, _ = sim.build_input(x, y, edge_attr, training=False)
            net = epd_forward(xn, en, edge_index, params, message_passing_num)
            pred = sim.build_outputs(x, net)  # simulator.py:213-217
            pred[mask] = y[mask]  # lightning_module.py:398
            last = pred
            preds.append(pred.clone())
    return preds


# ------------------------------------------------------------- CSR (integer work)
def csr_by_key(key: torch.Tensor, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stable sort of edge ids by ``key`` (dst for aggregation).  Returns
    (rowptr[n+1] int32, perm[E] int32) with perm ascending in edge id inside each
    segment -- the order CPU ``index_add_`` sums in (layers.py:1031 via PyG
    scatter).  Bit-exact target for ``mgn_csr_build``."""
    key = key.to(torch.int64)
    perm = torch.argsort(key, stable=True)
    counts = torch.bincount(key, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), perm.to(torch.int32)


# --------------------------------------------------------------------------- R0 / N2
def faces_to_edges_oracle(face, num_nodes: int):
    """T.FaceToEdge(remove_faces=False) (+ its to_undirected) of the reference's preprocessing
    (graphphysics/dataset/preprocessing.py:421-424; torch-geometric==2.6.1, not installed here:
    restated from its published behaviour): face [K,F] -> edge_index [2,E] int64 holding every
    pair of corners in both directions, coalesced = sorted by (src,dst) without duplicates; self
    loops of degenerate faces dropped.  The reference's tests pin no values for
    this transform; the anchor is its own test mesh tests/mock_vtu/cylinder_0.vtu (1923 nodes,
    3612 triangles -> 11 070 directed edges, SURVEY 8 header)."""
    import numpy as np

    f = np.asarray(face, dtype=np.int64)
    K = f.shape[0]
    src, dst = [], []
    for a in range(K):
        for b in range(a + 1, K):
            src += [f[a], f[b]]
            dst += [f[b], f[a]]
    src, dst = np.concatenate(src), np.concatenate(dst)
    keep = src != dst
    key = np.unique(src[keep] * np.int64(num_nodes) + dst[keep])
    return np.stack([key // num_nodes, key % num_nodes], axis=0)


def edge_features_oracle(pos: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """T.Cartesian(norm=False) then T.Distance(norm=False) (preprocessing.py:16-23):
    cart = pos[row] - pos[col] with (row, col) = edge_index; dist = ||pos[col] - pos[row]||_2
    appended as the last column."""
    row, col = edge_index[0], edge_index[1]
    cart = pos[row] - pos[col]
    dist = torch.norm(pos[col] - pos[row], p=2, dim=-1).view(-1, 1)
    return torch.cat([cart, dist], dim=-1)


def add_world_edges_oracle(x, edge_index, world_pos_index_start: int, world_pos_index_end: int, node_type_index: int,
                           radius: float = 0.03):
    """add_world_edges, graphphysics/dataset/preprocessing.py:92-140: cKDTree.query_pairs(radius)
    on the world positions (all i<j with Euclidean distance <= radius, double arithmetic on the
    float32 coordinates), kept when one end is OBSTACLE (1) and the other NORMAL (0), concatenated
    with the mesh edges and passed through to_undirected (symmetrise + coalesce)."""
    import numpy as np

    xn = np.asarray(x, dtype=np.float32)
    N = xn.shape[0]
    wp = xn[:, world_pos_index_start:world_pos_index_end].astype(np.float64)
    t = xn[:, node_type_index].astype(np.int64)
    obs, nor = np.nonzero(t == 1)[0], np.nonzero(t == 0)[0]
    d2 = ((wp[obs][:, None, :] - wp[nor][None, :, :]) ** 2).sum(-1)
    io, jn = np.nonzero(d2 <= float(radius) ** 2)
    a, b = obs[io], nor[jn]
    ei = np.asarray(edge_index, dtype=np.int64)
    src = np.concatenate([a, b, ei[0], ei[1]])
    dst = np.concatenate([b, a, ei[1], ei[0]])
    key = np.unique(src * np.int64(N) + dst)
    return np.stack([key // N, key % N], axis=0)


def philox4x32_10(k0, k1, c0, c1, c2, c3):
    """Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), vectorised
    in numpy uint64 arithmetic; returns the first two output words.  Integer work: bit-exact target for
    the device stream of ``mgn_add_noise``."""
    import numpy as np

    M = np.uint64(0xFFFFFFFF)
    k0, k1 = np.uint64(k0), np.uint64(k1)
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & M for v in (c0, c1, c2, c3))
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M
        c1, c3, c0, c2 = p1 & M, p0 & M, n0, n2
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return c0.astype(np.uint32), c1.astype(np.uint32)


def add_noise_oracle(x, noise_index_start, noise_index_end, noise_scale, node_type_index: int, t=None, seed: int = 0, offset: int = 0):
    """add_noise, graphphysics/dataset/preprocessing.py:177-238 -- list handling (:200-217), NORMAL-only mask
    (:219-222,230-231), curriculum scale (:226), in-place column update (:233-234) -- with the
    counter-based noise stream the engine documents (include/mgn_hip.h, mgn_add_noise) in place of
    torch.randn_like.  Returns (noised x, the standard-normal draws per range)."""
    import numpy as np

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
    x = np.array(x, dtype=np.float32, copy=True)
    N = x.shape[0]
    normal = x[:, node_type_index] == np.float32(NODE_NORMAL)   # the float is compared (preprocessing.py:219-222)
    rows = np.arange(N, dtype=np.uint64)
    draws = []
    for r, (s0, s1, sc) in enumerate(zip(noise_index_start, noise_index_end, noise_scale)):
        sc = np.float32(10 * sc * (1 + math.cos(t * math.pi)) if t is not None else sc)
        cols = np.arange(s0, s1, dtype=np.uint64)
        n_, c_ = np.meshgrid(rows, cols, indexing="ij")
        r0, r1 = philox4x32_10(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, n_ & np.uint64(0xFFFFFFFF), n_ >> np.uint64(32),
                               c_ | np.uint64(r << 16), np.uint64(offset & 0xFFFFFFFF))
        u1 = ((r0 >> 8).astype(np.float32) + np.float32(1)) * np.float32(2.0 ** -24)
        u2 = (r1 >> 8).astype(np.float32) * np.float32(2.0 ** -24)
        z = np.sqrt(np.float32(-2) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
        z = z.astype(np.float32)
        draws.append(z)
        x[:, s0:s1] = np.where(normal[:, None], x[:, s0:s1] + z * sc, x[:, s0:s1]).astype(np.float32)
    return x, draws


# ------------------------------------------------------------------ N4: sparse-attention Transformer
# The reference's DGL branch (HAS_DGL_SPARSE): adj = dglsp.spmatrix(indices=edge_index, shape=(N, N))
# (processors.py:352), scores = dglsp.bsddmm(adj, q, k^T), attn = scores.softmax(), y = dglsp.bspmm(attn, v)
# (layers.py:493-559).  ``dgl`` is an OPTIONAL, un-vendored dependency of the reference (not in its
# requirements.txt; imported in a try block, layers.py:10-18) and is not installed in this image, so the
# three sparse primitives are restated from DGL's published semantics (dgl.sparse, DGL 2.x docs):
#   bsddmm(A, X1[L,M,K], X2[M,N,K]) -> values[e,k] = sum_m X1[row_e,m,k] X2[m,col_e,k] on A's non-zeros;
#   SparseMatrix.softmax()          -> softmax over the non-zeros of each ROW, per trailing batch k;
#   bspmm(A, X[N,F,K])              -> out[l,f,k] = sum_{e: row_e=l} A.val[e,k] X[col_e,f,k].
# Everything AROUND them (projections, the head layout reshape(N, head_dim, num_heads), the scale
# q / sqrt(k.size(1)), RoPE, gate, residual structure, gated MLP) is pinned against the reference itself:
# tests/golden/make_golden_transformer.py runs the unmodified reference modules over a dglsp stand-in built
# on dense masked tensors and checks this restatement against them.
def sparse_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """scaled_dot_product_attention with an adjacency mask, layers.py:493-559.  q, k, v: [N, D, Hh]."""
    row, col = edge_index[0], edge_index[1]
    N, D, Hh = q.shape
    q = q / math.sqrt(k.size(1))                                   # layers.py:509-510
    score = (q[row] * k[col]).sum(dim=1)                           # bsddmm on the non-zeros: [E, Hh]
    mx = torch.full((N, Hh), float("-inf"), dtype=q.dtype).scatter_reduce(0, row.view(-1, 1).expand(-1, Hh), score, "amax")
    ex = torch.exp(score - mx[row])
    den = torch.zeros(N, Hh, dtype=q.dtype).index_add_(0, row, ex)
    attn = ex / den[row]                                           # row-wise softmax over the non-zeros
    return torch.zeros(N, D, Hh, dtype=q.dtype).index_add_(0, row, attn.unsqueeze(1) * v[col])   # bspmm


def attn_inv_freq(head_dim: int, pos_dimension: int, base: float) -> torch.Tensor:
    """_make_inv_freq, layers.py:410-417, with m = head_dim // (2 * pos_dimension) (:617)"""
    m = head_dim // max(pos_dimension * 2, 1)
    if m <= 0:
        return torch.empty(0, dtype=torch.float32)
    return torch.exp(-torch.arange(m, dtype=torch.float32) * (math.log(base) / max(m, 1)))


def apply_rope_with_inv(q, k, pos, inv_freq):
    """_apply_rope_with_inv, layers.py:420-490 (absolute positions; the same rotation for every head)."""
    N, D, Hh = q.shape
    pd = pos.shape[1]
    m = D // (pd * 2)
    if m == 0 or inv_freq.numel() == 0:
        return q, k
    d_rope = pd * 2 * m
    ang = pos[:, :pd].to(torch.float32).unsqueeze(-1) * inv_freq.to(torch.float32).view(1, 1, m)
    cos, sin = torch.cos(ang).to(q.dtype).unsqueeze(-1), torch.sin(ang).to(q.dtype).unsqueeze(-1)

    def app(x):
        part = x[:, :d_rope, :].contiguous().view(N, pd, m, 2, Hh)
        even, odd = part[..., 0, :], part[..., 1, :]
        rot = torch.stack((even * cos - odd * sin, even * sin + odd * cos), dim=3).reshape(N, d_rope, Hh)
        return torch.cat([rot, x[:, d_rope:, :]], dim=1)

    return app(q), app(k)


def attention(x, p, prefix, edge_index, num_heads, pos=None, use_rope=False, use_gate=False, pos_dimension=3, rope_base=10000.0):
    """Attention.forward, layers.py:641-697."""
    N = x.size(0)
    lin = lambda name, t: _linear(t, p[f"{prefix}{name}.weight"], p.get(f"{prefix}{name}.bias"))  # noqa: E731
    hd = p[prefix + "q_proj.weight"].shape[0] // num_heads
    q, k, v = (lin(n, x).reshape(N, hd, num_heads) for n in ("q_proj", "k_proj", "v_proj"))
    if use_rope:
        q, k = apply_rope_with_inv(q, k, pos, attn_inv_freq(hd, pos_dimension, rope_base))
    if _BF16_MIXED:
        # training.enable_vram_optimizations sets Lightning bf16-mixed AND the memory-optimised switch (parse_parameters.py:
        # 111-112): the scaled query is a bf16 tensor (q / sqrt(d) on bf16, layers.py:509-510), bsddmm / softmax / bspmm run in
        # fp32 through the _bsddmm_fp32 / _bspmm_fp32 shims (layers.py:49-70,513-517,549-553) and y returns in v's dtype
        scale = math.sqrt(k.size(1))
        qs = (q / scale).float() * scale          # the bf16-rounded scaled query, re-expressed for sparse_attention's own division
        y = sparse_attention(qs, k.float(), v.float(), edge_index).to(torch.bfloat16)
    else:
        y = sparse_attention(q, k, v, edge_index)
    if use_gate:
        y = y * torch.sigmoid(lin("gate_proj", x)).reshape(N, hd, num_heads)
    return lin("proj", y.reshape(N, -1))


def transformer_block(x, p, prefix, edge_index, num_heads, act="relu", **kw):
    """Transformer.forward, layers.py:813-816: x + attention(norm1(x)); x + gated_mlp(norm2(x))."""
    x = x + attention(rms_norm(x, p[prefix + "norm1.scale"]), p, prefix + "attention.", edge_index, num_heads, **kw)   # fp32 + bf16 -> fp32
    return x + gated_mlp(rms_norm(x, p[prefix + "norm2.scale"]), p, prefix + "gated_mlp.", act)


def head_axis_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """scaled_dot_product_attention WITHOUT an adjacency, layers.py:493-559 (att_mask = None: the ``else`` arms of :518-522 and
    :555-556): on [N, D, Hh] operands the matrix products run over the last two axes, so every node attends over its own D rows --
    attn = softmax(q k^T / sqrt(D)) of shape [N, D, D], y = attn @ v.  Pinned by tests/golden/temporal_dense.npz (the unmodified
    reference TemporalAttention called with adj = None)."""
    q = q / math.sqrt(k.size(1))
    return torch.softmax(q @ k.transpose(-2, -1), dim=-1) @ v


def temporal_attention(h_prev, h_pred, p, prefix, edge_index, num_heads=4, use_gate=True):
    """TemporalAttention.forward, layers.py:858-887 (use_gate=True, the constructor default).  ``edge_index=None``: what an
    installation without DGL computes (processors.py:203-209, :376-377 pass adj = None)."""
    N, h = h_prev.shape
    lin = lambda name, t: torch.nn.functional.linear(t, p[f"{prefix}{name}.weight"], p[f"{prefix}{name}.bias"])  # noqa: E731
    d = h // num_heads
    q, k, v = lin("q_proj", h_pred).reshape(N, d, num_heads), lin("k_proj", h_prev).reshape(N, d, num_heads), lin("v_proj", h_pred).reshape(N, d, num_heads)
    y = head_axis_attention(q, k, v) if edge_index is None else sparse_attention(q, k, v, edge_index)
    out = lin("out_proj", y.reshape(N, h))
    if use_gate:   # layers.py:880-882
        out = torch.sigmoid(lin("gate.2", torch.nn.functional.silu(lin("gate.0", torch.cat([h_pred, h_prev], dim=-1))))) * out
    h_corr = h_prev + out
    return h_corr + lin("mixer.2", torch.nn.functional.silu(lin("mixer.0", torch.cat([h_corr, h_prev], dim=-1))))


def transformer_conv(x, p, prefix, edge_index, heads):
    """torch_geometric.nn.TransformerConv(in, out, heads, concat=False, beta=True) -- the block of the reference's NON-DGL branch
    (processors.py:303-314, called as ``block(prev_x, edge_index)`` at :372-375).  PyG is a pinned third-party dependency
    (torch-geometric==2.6.1, requirements.txt:7) that is not installable here: this is its PUBLISHED algorithm
    (torch_geometric/nn/conv/transformer_conv.py, Shi et al. 2021 eq. 3-6) restated -- **parity unpinned** for this function:
        q_i = W_q x_i + b_q,  k_j = W_k x_j + b_k,  v_j = W_v x_j + b_v            (per head: view(-1, H, C), head index SLOW)
        alpha_ij = softmax over the in-edges j -> i (i = edge_index[1], j = edge_index[0]) of  q_i . k_j / sqrt(C)
        m_i = mean over heads of  sum_j alpha_ij v_j                                  (concat=False)
        r_i = W_skip x_i + b_skip;  beta_i = sigmoid(w_beta . [m_i, r_i, m_i - r_i]);  out_i = beta_i r_i + (1 - beta_i) m_i
    (dropout 0, no edge features)."""
    src, dst = edge_index[0], edge_index[1]
    N = x.shape[0]
    lin = lambda name, t: _linear(t, p[f"{prefix}{name}.weight"], p.get(f"{prefix}{name}.bias"))  # noqa: E731
    C = p[prefix + "lin_skip.weight"].shape[0]
    q, k, v = (lin(n, x).view(N, heads, C) for n in ("lin_query", "lin_key", "lin_value"))
    score = (q[dst] * k[src]).sum(dim=-1) / math.sqrt(C)                                        # [E, H]
    mx = torch.full((N, heads), float("-inf"), dtype=x.dtype).scatter_reduce(0, dst.view(-1, 1).expand(-1, heads), score, "amax")
    ex = torch.exp(score - mx[dst])
    den = torch.zeros(N, heads, dtype=x.dtype).index_add_(0, dst, ex)
    alpha = ex / den[dst]
    out = torch.zeros(N, heads, C, dtype=x.dtype).index_add_(0, dst, alpha.unsqueeze(-1) * v[src]).mean(dim=1)
    r = lin("lin_skip", x)
    beta = torch.sigmoid(_linear(torch.cat([out, r, out - r], dim=-1), p[prefix + "lin_beta.weight"], None))
    return beta * r + (1 - beta) * out


def etd_forward(x_in, edge_index, p, message_passing_num, num_heads, act="relu", pos=None, use_rope=False, use_gate=False,
                pos_dimension=3, rope_base=10000.0, use_temporal_block=False, conv="dgl"):
    """EncodeTransformDecode.forward, processors.py:334-371 (``conv="dgl"``: the sparse-attention Transformer blocks;
    ``conv="pyg"``: the non-DGL branch, TransformerConv blocks applied as ``x = block(x, edge_index)``, :372-375)."""
    x = mlp(x_in, p, "nodes_encoder.", act)
    prev_x = last_x = x
    if conv == "pyg":
        for i in range(message_passing_num):
            prev_x = x
            x = last_x = transformer_conv(prev_x, p, f"processor_list.{i}.", edge_index, num_heads)
        if use_temporal_block:   # :376-377 with adj = None
            x = temporal_attention(prev_x, last_x, p, "temporal_block.", None, num_heads)
        return mlp(x, p, "decode_module.", act)
    for i in range(message_passing_num):
        prev_x = x
        last_x = transformer_block(prev_x, p, f"processor_list.{i}.", edge_index, num_heads, act, pos=pos, use_rope=use_rope,
                                   use_gate=use_gate, pos_dimension=pos_dimension, rope_base=rope_base)
        x = last_x
    if use_temporal_block:
        x = temporal_attention(prev_x, last_x, p, "temporal_block.", edge_index, num_heads)
    return mlp(x, p, "decode_module.", act)
