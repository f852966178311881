"""Module mirror of the reference's hot-path layers
(graphphysics/models/layers.py): same constructor signatures, attribute names and
``state_dict`` keys, so checkpoints and the Simulator/LightningModule glue carry
over; ``forward`` runs on the HIP engine (``ops``), never on a torch/CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops

# process-global activation switch, like layers.py:132-147
_USE_SILU_ACTIVATION = False


def set_use_silu_activation(use_silu: bool) -> None:
    global _USE_SILU_ACTIVATION
    _USE_SILU_ACTIVATION = bool(use_silu)


def use_silu_activation() -> bool:
    return _USE_SILU_ACTIVATION


class RMSNorm(nn.Module):
    """Trailing normalisation of ``build_mlp`` (layers.py:73-129: y = scale * x / (||x||/sqrt(d) + eps),
    eps OUTSIDE the root).  Inside an MLP the arithmetic is fused into the kernel's epilogue and this
    module only holds ``scale``; called on its own (``net.nodes_encoder[7](x)``, the leading norm of a
    gated MLP) it evaluates the same formula with elementwise device ops."""

    def __init__(self, d: int, p: float = -1.0, eps: float = 1e-8, bias: bool = False):
        super().__init__()
        self.d, self.p, self.eps, self.bias = d, p, eps, bias
        self.scale = nn.Parameter(torch.ones(d))
        if self.bias:
            self.offset = nn.Parameter(torch.zeros(d))   # layers.py:100-101 (same state_dict key)

    @property
    def is_default(self) -> bool:
        """the form build_mlp constructs (whole-row norm, no offset, eps 1e-8): the one the fused kernels implement"""
        return (not self.bias) and not (0.0 <= self.p <= 1.0) and self.eps == ops.EPS

    def forward(self, x):
        ops._require_device(x)
        if self.is_default and x.shape[-1] <= 384:   # engine kernel (csrc/mgn_dense.hip), forward and backward
            from .dense import rms_norm
            return rms_norm(x, self.scale)
        # partial (p in [0, 1]) / biased / non-default eps (layers.py:113-128): reachable only by constructing the module
        # directly (build_mlp never does) -- the reference's formula with elementwise device ops (differentiable)
        if 0.0 <= self.p <= 1.0:
            k = int(self.d * self.p)
            norm_x, d_x = x[..., :k].norm(2, dim=-1, keepdim=True), k
        else:
            norm_x, d_x = x.norm(2, dim=-1, keepdim=True), self.d
        y = self.scale * (x / (norm_x / (d_x ** 0.5) + self.eps))
        return y + self.offset if self.bias else y


class ReLU(nn.Module):
    """Sequential entries 1,3,5 of ``build_mlp``: fused inside the MLP kernel when the enclosing MLP is
    called; on its own an elementwise device op (the Sequential stays index-transparent)."""

    def forward(self, x):
        ops._require_device(x)
        return torch.relu(x)


class SiLU(ReLU):
    """nn.SiLU (``use_silu_activation``, layers.py:132-160)."""

    def forward(self, x):
        ops._require_device(x)
        return torch.nn.functional.silu(x)


class GELU(ReLU):
    """nn.GELU() (exact erf form) -- ``build_mlp(act="gelu")`` (layers.py:150-160)."""

    def forward(self, x):
        ops._require_device(x)
        return torch.nn.functional.gelu(x)


_ACT_MODULES = {"relu": ReLU, "silu": SiLU, "gelu": GELU}


class MLP(nn.Sequential):
    """``build_mlp`` result: entries 0,2,4,.. nn.Linear, odd entries activation,
    last entry RMSNorm if ``layer_norm`` (layers.py:198-210).  forward() is one
    fused HIP kernel (ops.MlpFunction)."""

    act = "relu"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lin = [m for m in self if isinstance(m, nn.Linear)]
        norm = self[len(self) - 1] if isinstance(self[len(self) - 1], RMSNorm) else None
        if norm is not None and not norm.is_default:
            raise NotImplementedError("a partial / biased RMSNorm at the end of an MLP is not fused: call the entries one by one")
        params = []
        for m in lin:
            params += [m.weight, m.bias]
        if norm is not None:
            params.append(norm.scale)
        return ops.mlp_apply(x, norm is not None, *params, act=self.act)


def build_mlp(in_size: int, hidden_size: int, out_size: int, nb_of_layers: int = 4,
              layer_norm: bool = True, act: Optional[str] = None) -> nn.Module:
    """Same signature / assertion as the reference build_mlp (layers.py:163-210)."""
    assert nb_of_layers >= 2, "The MLP must have at least 2 layers (input and output)."
    key = act if act is not None else ("silu" if _USE_SILU_ACTIVATION else "relu")
    if key not in ("relu", "gelu", "silu"):
        raise NotImplementedError(f"Activation '{key}' not supported. Available: ['relu', 'gelu', 'silu'].")
    A = _ACT_MODULES[key]
    layers = [nn.Linear(in_size, hidden_size), A()]
    for _ in range(nb_of_layers - 2):
        layers.extend([nn.Linear(hidden_size, hidden_size), A()])
    layers.append(nn.Linear(hidden_size, out_size))
    if layer_norm:
        layers.append(RMSNorm(out_size))
    mlp = MLP(*layers)
    mlp.act = key
    return mlp


def _mlp_param_list(mlp: nn.Sequential):
    out = []
    for m in mlp:
        if isinstance(m, nn.Linear):
            out += [m.weight, m.bias]
    if isinstance(mlp[len(mlp) - 1], RMSNorm):
        out.append(mlp[len(mlp) - 1].scale)
    return out


def _block_params(block: "GraphNetBlock"):
    """parameters of one block in the order ops.ProcessorFunction takes them (= state_dict order)"""
    out = _mlp_param_list(block.edge_block) + _mlp_param_list(block.node_block)
    if block.use_gate:
        out += [block.gate_proj.weight, block.gate_proj.bias, block.gate_pos]
    return out


class GraphNetBlock(nn.Module):
    """One MeshGraphNet round (layers.py:890-1042): gather -> edge MLP -> segment-sum -> node MLP ->
    residuals, on the HIP engine; optional relative RoPE on the source features (:1020-1026), sigmoid
    gate on the aggregate (:1091-1098), SiLU activations (global switch), any ``nb_of_layers >= 2``,
    ``layer_norm`` on / off, gated-MLP blocks (:213-278)."""

    def __init__(self, hidden_size: int, nb_of_layers: int = 4, layer_norm: bool = True,
                 use_rope: bool = False, rope_axes: int = 3, rope_base: float = 10000.0,
                 use_gated_mlp: bool = False, use_gate: bool = False):
        super().__init__()
        self.hidden_size = hidden_size
        self.use_gated_mlp = use_gated_mlp
        if use_gated_mlp:
            from .gated import build_gated_mlp
            if hidden_size not in (16, 32, 64, 128):
                # the gated blocks run on mgn_linear_fwd / _bwd: phases of 16/32/64/128 columns, outputs up to 384 (include/mgn_hip.h,
                # "Dense row work"); say so here instead of a C-layer error at the first forward
                raise NotImplementedError(f"use_gated_mlp needs hidden_size in (16, 32, 64, 128) on the HIP dense kernels, got {hidden_size}")
            self.edge_block = build_gated_mlp(3 * hidden_size, hidden_size, hidden_size)
            self.node_block = build_gated_mlp(2 * hidden_size, hidden_size, hidden_size)
        else:
            self.edge_block = build_mlp(3 * hidden_size, hidden_size, hidden_size, nb_of_layers, layer_norm)
            self.node_block = build_mlp(2 * hidden_size, hidden_size, hidden_size, nb_of_layers, layer_norm)
        self.use_rope, self.rope_axes, self.rope_base = use_rope, rope_axes, rope_base
        if self.use_rope:
            if rope_axes not in (2, 3):
                raise ValueError("rope_axes must be 2 or 3 when use_rope=True.")
            self._pair_count = hidden_size // (2 * rope_axes)
            self._rope_dim = self._pair_count * 2 * rope_axes
            if self._pair_count == 0:
                raise ValueError(f"hidden_size={hidden_size} too small for rope_axes={rope_axes}; "
                                 "need at least 2 * rope_axes channels.")
            inv = torch.arange(self._pair_count, dtype=torch.float32)
            inv = torch.pow(self.rope_base, -inv / max(float(self._pair_count), 1.0))  # layers.py:972-976
            self.register_buffer("_rope_inv_freq", inv, persistent=False)
        else:
            self._pair_count, self._rope_dim = 0, 0
            self.register_buffer("_rope_inv_freq", torch.zeros(0), persistent=False)  # layers.py:977-981
        self.use_gate = use_gate
        if self.use_gate:
            self.gate_proj = nn.Linear(hidden_size, hidden_size, bias=True)
            self.gate_pos = nn.Parameter(torch.zeros(hidden_size))
        self.spec = ops.BlockSpec(nb_layers=nb_of_layers, layer_norm=layer_norm,
                                  act=("silu" if _USE_SILU_ACTIVATION else "relu"), gate=use_gate, rope=use_rope, rope_axes=rope_axes)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor, size=None,
                pos: Optional[torch.Tensor] = None, phi: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.use_rope and pos is None:
            raise ValueError("Node positions `pos` must be provided when use_rope=True.")
        topo = ops.get_topology(edge_index, x.shape[0])
        e_sorted = edge_attr[topo.perm_dst_long]
        if self.use_gated_mlp:
            from .gated import gated_block_forward
            x_new, e_new = gated_block_forward(self, x, e_sorted, topo, pos, phi)
        else:
            x_new, e_new = ops.processor_apply(x, e_sorted, topo, 1, *_block_params(self), spec=self.spec,
                                               pos=pos if self.use_rope else None, phi=phi if self.use_gate else None,
                                               rope_inv_freq=self._rope_inv_freq if self.use_rope else None)
        out = (x_new, e_new[topo.inv_perm])
        if not topo.resolved:  # lazily built topology: launches are queued, now look at its flags (IndexError on a stray index)
            topo.resolve()
        return out


class Normalizer(nn.Module):
    """Online mean/std normaliser (layers.py:281-408); elementwise boundary code
    kept in PyTorch-ROCm.  Same buffers / keys."""

    def __init__(self, size: int, max_accumulations: int = 10**5, std_epsilon: float = 1e-8,
                 name: str = "Normalizer", device: Optional[Union[str, torch.device]] = "cuda"):
        super().__init__()
        self.name, self.device = name, device
        self._max_accumulations = max_accumulations
        self._std_epsilon = torch.tensor(std_epsilon, dtype=torch.float32, requires_grad=False, device=device)
        self._std_epsilon_value = float(torch.tensor(std_epsilon, dtype=torch.float32))  # host copy for the fused kernels
        self.register_buffer("_acc_count", torch.tensor(0.0, device=device))
        self.register_buffer("_num_accumulations", torch.tensor(0.0, device=device))
        self.register_buffer("_acc_sum", torch.zeros((1, size), dtype=torch.float32, device=device))
        self.register_buffer("_acc_sum_squared", torch.zeros((1, size), dtype=torch.float32, device=device))
        self._host_num_acc: Optional[int] = None  # host mirror: avoids a device sync per call

    def forward(self, batched_data: torch.Tensor, accumulate: bool = True) -> torch.Tensor:
        if accumulate:
            if self._host_num_acc is None:
                self._host_num_acc = int(self._num_accumulations.item())
            if self._host_num_acc < self._max_accumulations:
                self._accumulate(batched_data.detach())
                self._host_num_acc += 1
        return (batched_data - self._mean()) / self._std_with_epsilon()

    def inverse(self, normalized_batch_data: torch.Tensor) -> torch.Tensor:
        return normalized_batch_data * self._std_with_epsilon() + self._mean()

    def _accumulate(self, batched_data: torch.Tensor):
        # column sums of a tall skinny [rows, size] matrix: torch's strided reduction needs ~45 us
        # for [180k, 3]; a transposed copy + contiguous row reduction ~10 us.  (A GEMV with a ones
        # vector is far worse: rocBLAS takes milliseconds for this shape.)
        if batched_data.is_cuda and batched_data.dim() == 2 and batched_data.shape[0] > 4096:
            xt = batched_data.t().contiguous()
            s1 = xt.sum(dim=1).unsqueeze(0)
            s2 = (xt * xt).sum(dim=1).unsqueeze(0)
        else:
            s1 = torch.sum(batched_data, dim=0, keepdim=True)
            s2 = torch.sum(batched_data**2, dim=0, keepdim=True)
        self._acc_sum += s1
        self._acc_sum_squared += s2
        self._acc_count += batched_data.shape[0]
        self._num_accumulations += 1

    def _mean(self) -> torch.Tensor:
        return self._acc_sum / torch.clamp(self._acc_count, min=1.0)

    def _std_with_epsilon(self) -> torch.Tensor:
        safe = torch.clamp(self._acc_count, min=1.0)
        var = self._acc_sum_squared / safe - self._mean() ** 2
        return torch.max(torch.sqrt(torch.clamp(var, min=0.0)), self._std_epsilon.to(var.device))

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._host_num_acc = None

    def get_variable(self):
        return {"_max_accumulations": self._max_accumulations, "_std_epsilon": self._std_epsilon,
                "_acc_count": self._acc_count, "_num_accumulations": self._num_accumulations,
                "_acc_sum": self._acc_sum, "_acc_sum_squared": self._acc_sum_squared, "name": self.name}
