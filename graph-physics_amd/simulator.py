"""``Simulator`` boundary wrapper (graphphysics/models/simulator.py:13-217):
normalisers, one-hot node type, delta target, inverse normalisation.

On the MI355X the whole pre-processing of a step (one-hot + slice + concat, the three online
normalisers incl. their running statistics, the delta target) is ONE call into the engine
(``mgn_sim_pre``: 2-3 launches instead of ~40 small torch kernels), and the post-processing
(inverse normalisation, optional ground-truth re-imposition of the rollout) another
(``mgn_sim_post``) -- SURVEY.md section 8f row N1.  The torch statement of the same arithmetic
below stays as the module-level reference semantic for tensors the fused path does not take
(CPU tensors in host-side tests, more than 32 feature columns)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _capi
from .layers import Normalizer
from .mesh import Graph
from .nodetype import NodeType


class Simulator(nn.Module):
    def __init__(self, node_input_size: int, edge_input_size: int, output_size: int, feature_index_start: int,
                 feature_index_end: int, output_index_start: int, output_index_end: int, node_type_index: int,
                 model: nn.Module, device: torch.device, model_dir: str = "checkpoint/simulator.pth"):
        super().__init__()
        self.node_input_size = node_input_size
        self.edge_input_size = edge_input_size if edge_input_size > 0 else None
        self.output_size = output_size
        self.feature_index_start, self.feature_index_end = feature_index_start, feature_index_end
        self.node_type_index = node_type_index
        self.output_index_start, self.output_index_end = output_index_start, output_index_end
        self.model_dir = model_dir
        self.model = model.to(device)
        self._output_normalizer = Normalizer(size=output_size, name="output_normalizer", device=device)
        self._node_normalizer = Normalizer(size=node_input_size, name="node_normalizer", device=device)
        self._edge_normalizer = (Normalizer(size=edge_input_size, name="edge_normalizer", device=device)
                                 if self.edge_input_size is not None else None)
        self.device = device
        self.fused = True  # engine kernels for the pre / post processing of CUDA tensors
        self._ws = None
        self._type_err = None      # device flag: node type outside [0, 9) seen by the fused kernels
        self._types_checked = False

    # ------------------------------------------------------------------ fused path (N1)
    def _can_fuse(self, inputs) -> bool:
        x = inputs.x
        return (self.fused and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                and self.node_input_size <= 32 and self.output_size <= 32
                and (self.edge_input_size or 0) <= 32
                and (inputs.edge_attr is None or inputs.edge_attr.dtype == torch.float32)
                and (getattr(inputs, "y", None) is None or inputs.y.dtype == torch.float32))

    def _check_widths(self, x, y, ea):
        """The shape errors the reference raises from its slicing / ``cat`` / broadcasting
        (simulator.py:86-143, layers.py:331-349), raised here before the fused kernels index with
        these widths."""
        nf = self.feature_index_end - self.feature_index_start
        if nf < 0 or self.feature_index_start < 0 or self.feature_index_end > x.shape[1]:
            raise ValueError(f"feature columns [{self.feature_index_start}, {self.feature_index_end}) are outside x (width {x.shape[1]})")
        if not (0 <= self.node_type_index < x.shape[1]):
            raise ValueError(f"node_type_index {self.node_type_index} is outside x (width {x.shape[1]})")
        if nf + NodeType.SIZE != self.node_input_size:
            raise ValueError(f"node features have {nf} + {NodeType.SIZE} columns but node_input_size is {self.node_input_size} "
                             "(the node normaliser's width)")
        if self.output_index_end - self.output_index_start != self.output_size:
            raise ValueError("output_index_end - output_index_start differs from output_size")
        if y is not None:
            if self.output_index_start < 0 or self.output_index_start + self.output_size > x.shape[1]:
                raise ValueError(f"output columns [{self.output_index_start}, {self.output_index_start + self.output_size}) are outside x")
            if y.shape[1] < self.output_size:
                raise ValueError(f"y has {y.shape[1]} columns, output_size is {self.output_size}")
            if y.shape[0] != x.shape[0]:
                raise ValueError("x and y have different row counts")
        if ea is not None and ea.shape[1] != self.edge_input_size:
            raise ValueError(f"edge_attr has {ea.shape[1]} columns, edge_input_size is {self.edge_input_size}")

    def _desc(self, inputs, is_training: bool):
        d = _capi.SimDesc()
        x = inputs.x.contiguous()
        y = getattr(inputs, "y", None)
        y = y.contiguous() if y is not None else None
        ea = inputs.edge_attr.contiguous() if (self._edge_normalizer is not None and inputs.edge_attr is not None) else None
        self._check_widths(x, y, ea)
        keep = [x, y, ea]
        d.N, d.E = x.shape[0], (ea.shape[0] if ea is not None else 0)
        d.x, d.x_w = x.data_ptr(), x.shape[1]
        d.y, d.y_w = (y.data_ptr() if y is not None else None), (y.shape[1] if y is not None else 0)
        d.edge_attr, d.edge_w = (ea.data_ptr() if ea is not None else None), (ea.shape[1] if ea is not None else 0)
        d.feat_start, d.feat_end = self.feature_index_start, self.feature_index_end
        d.out_start, d.out_w, d.type_idx = self.output_index_start, self.output_size, self.node_type_index
        norms = (self._node_normalizer, self._output_normalizer, self._edge_normalizer)
        present = (True, y is not None, ea is not None)
        for s, (nz, here) in enumerate(zip(norms, present)):
            if nz is None:
                nz = self._node_normalizer  # never dereferenced: edge_w == 0
            d.acc_sum[s], d.acc_sumsq[s] = nz._acc_sum.data_ptr(), nz._acc_sum_squared.data_ptr()
            d.acc_count[s], d.num_acc[s] = nz._acc_count.data_ptr(), nz._num_accumulations.data_ptr()
            d.norm_w[s] = nz._acc_sum.shape[1]
            # Normalizer.forward (layers.py:345-349) accumulates in train() while the DEVICE counter is
            # below max_accumulations: the kernels read it themselves (no host sync, and a captured
            # hipGraph stops accumulating at the same step the reference does)
            acc = bool(is_training and here and norms[s] is not None)
            if acc:
                nz._host_num_acc = None  # the torch path's host mirror is stale from here on
            d.accumulate[s] = int(acc)
        d.max_accumulations = float(self._output_normalizer._max_accumulations)
        d.std_eps = float(self._output_normalizer._std_epsilon_value)
        if self._type_err is None or self._type_err.device != x.device:
            self._type_err = torch.zeros(1, dtype=torch.int32, device=x.device)
        d.type_err = self._type_err.data_ptr()
        return d, keep, (x, y, ea)

    def check_node_types(self) -> None:
        """Raise if a fused call met a node-type code outside [0, NodeType.SIZE) (``F.one_hot``
        raises there).  Synchronises; the first fused call of a Simulator checks itself."""
        if self._type_err is not None and int(self._type_err.item()) != 0:
            raise RuntimeError(f"node type codes must lie in [0, {NodeType.SIZE}) (class values must be smaller than num_classes)")

    def _build_input_graph_fused(self, inputs, is_training: bool):
        dev = inputs.x.device
        d, keep, (x, y, ea) = self._desc(inputs, is_training)
        Wn = (self.feature_index_end - self.feature_index_start) + NodeType.SIZE
        xn = torch.empty(x.shape[0], Wn, dtype=torch.float32, device=dev)
        tgt = torch.empty(x.shape[0], self.output_size, dtype=torch.float32, device=dev) if y is not None else None
        en = torch.empty_like(ea) if ea is not None else inputs.edge_attr
        d.node_out, d.target_out = xn.data_ptr(), (tgt.data_ptr() if tgt is not None else None)
        d.edge_out = en.data_ptr() if ea is not None else None
        L = _capi.lib()
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(L.mgn_sim_workspace_bytes(), dtype=torch.uint8, device=dev)
        import ctypes as C
        with torch.cuda.device(dev):
            rc = L.mgn_sim_pre(C.byref(d), self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(rc, "mgn_sim_pre", prep=True)
        if not self._types_checked and not torch.cuda.is_current_stream_capturing():
            self._types_checked = True  # one sync, on the first step only
            self.check_node_types()
        graph = Graph(x=xn, pos=inputs.pos, edge_attr=en, edge_index=inputs.edge_index)
        topo = getattr(inputs, "mgn_topology", None)
        if topo is not None:
            graph.mgn_topology = topo
        return graph, tgt

    def predict(self, inputs, network_output: torch.Tensor, mask_truth: bool = False) -> torch.Tensor:
        """``build_outputs`` (simulator.py:178-191); with ``mask_truth`` the rollout's re-imposition
        of the ground truth on the nodes that are not NORMAL / OUTFLOW (lightning_module.py:27-35)."""
        if not (self._can_fuse(inputs) and network_output.is_cuda):
            pred = self.build_outputs(inputs, network_output)
            if mask_truth:
                t = inputs.x[:, self.node_type_index]
                keep = torch.logical_or(t == int(NodeType.NORMAL), t == int(NodeType.OUTFLOW))
                pred = torch.where(keep.unsqueeze(1), pred, inputs.y)
            return pred
        x = inputs.x.contiguous()
        y = inputs.y.contiguous() if (mask_truth and inputs.y is not None) else None
        no = network_output.detach().to(torch.float32).contiguous()
        nz = self._output_normalizer
        pred = torch.empty(x.shape[0], self.output_size, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _capi.lib().mgn_sim_post(x.data_ptr(), x.shape[1], self.output_index_start, self.node_type_index,
                                          (y.data_ptr() if y is not None else None), (y.shape[1] if y is not None else 0),
                                          no.data_ptr(), self.output_size, nz._acc_sum.data_ptr(), nz._acc_sum_squared.data_ptr(),
                                          nz._acc_count.data_ptr(), float(nz._std_epsilon_value), int(bool(mask_truth)), x.shape[0],
                                          pred.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        _capi.check(rc, "mgn_sim_post", prep=True)
        return pred

    def _get_pre_target(self, inputs) -> torch.Tensor:
        return inputs.x[:, self.output_index_start: self.output_index_end]

    def _get_target_normalized(self, inputs, is_training: bool = True) -> torch.Tensor:
        return self._output_normalizer(inputs.y - self._get_pre_target(inputs), is_training)

    def _get_one_hot_type(self, inputs) -> torch.Tensor:
        node_type = inputs.x[:, self.node_type_index]
        return torch.nn.functional.one_hot(torch.squeeze(node_type.long()), NodeType.SIZE)

    def _build_node_features(self, inputs, one_hot_type: torch.Tensor) -> torch.Tensor:
        features = inputs.x[:, self.feature_index_start: self.feature_index_end]
        return torch.cat([features, one_hot_type], dim=1)

    def _build_input_graph(self, inputs, is_training: bool):
        if self._can_fuse(inputs):
            return self._build_input_graph_fused(inputs, is_training)
        target_delta_normalized = self._get_target_normalized(inputs, is_training)
        node_features = self._build_node_features(inputs, self._get_one_hot_type(inputs))
        node_features_normalized = self._node_normalizer(node_features, is_training)
        if self._edge_normalizer is not None:
            edge_attr = self._edge_normalizer(inputs.edge_attr, is_training)
        else:
            edge_attr = inputs.edge_attr
        graph = Graph(x=node_features_normalized, pos=inputs.pos, edge_attr=edge_attr, edge_index=inputs.edge_index)
        topo = getattr(inputs, "mgn_topology", None)
        if topo is not None:
            graph.mgn_topology = topo
        return graph, target_delta_normalized

    def build_outputs(self, inputs, network_output: torch.Tensor) -> torch.Tensor:
        if self._can_fuse(inputs) and network_output.is_cuda and not torch.is_grad_enabled():
            return self.predict(inputs, network_output, mask_truth=False)
        return self._get_pre_target(inputs) + self._output_normalizer.inverse(network_output)

    def forward(self, inputs) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        graph, target_delta_normalized = self._build_input_graph(inputs=inputs, is_training=self.training)
        network_output = self.model(graph)
        if self.training:
            return network_output, target_delta_normalized, None
        return network_output, target_delta_normalized, self.build_outputs(inputs, network_output)

    def freeze_all(self) -> None:
        for p in self.model.parameters():
            p.requires_grad = False
