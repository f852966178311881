"""``Simulator`` boundary wrapper (graphphysics/models/simulator.py:13-217):
normalisers, one-hot node type, delta target, inverse normalisation.  Cheap
elementwise code that brackets the hot path; kept in PyTorch-ROCm."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

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
