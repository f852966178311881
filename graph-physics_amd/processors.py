"""``EncodeProcessDecode`` with the reference's constructor signature, attributes
and ``state_dict`` keys (graphphysics/models/processors.py:57-215), running on the
HIP engine: encoders / decoder are fused 4-layer MLP kernels, the processor loop is
one autograd node (``ops.ProcessorFunction``) that keeps edge latents in the
dst-sorted order for all rounds."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .layers import GraphNetBlock, _block_params, build_mlp


class EncodeProcessDecode(nn.Module):
    def __init__(self, message_passing_num: int, node_input_size: int, edge_input_size: int, output_size: int,
                 hidden_size: int = 128, only_processor: bool = False, use_rope_embeddings: bool = False,
                 use_gated_attention: bool = False, use_gated_mlp: bool = False, rope_pos_dimension: int = 3,
                 rope_base: float = 10000.0, use_temporal_block: bool = False, attention_backend: Optional[str] = None):
        super().__init__()
        # only read by the temporal block: "dgl" = attention over the mesh adjacency, "pyg" (or MGN_ATTENTION_BACKEND=pyg) = what an
        # installation without DGL computes, processors.py:203-209: adj = None -> attention over each node's head axis
        from .transformer import default_attention_backend
        self.attention_backend = attention_backend if attention_backend is not None else default_attention_backend()
        if self.attention_backend not in ("dgl", "pyg"):
            raise ValueError("attention_backend must be 'dgl' or 'pyg'")
        self.only_processor = only_processor
        self.hidden_size = hidden_size
        self.d = output_size
        self.use_temporal_block = use_temporal_block
        self.use_gated_mlp = use_gated_mlp
        self.use_rope = use_rope_embeddings
        self.use_gate = use_gated_attention
        self.rope_axes = rope_pos_dimension
        self.rope_base = rope_base
        if self.use_rope and self.rope_axes not in (2, 3):
            raise ValueError("rope_pos_dimension must be 2 or 3 when use_rope_embeddings=True.")
        if use_temporal_block:  # processors.py:122-126
            from .transformer import TemporalAttention
            self.temporal_block = TemporalAttention(hidden_size=hidden_size)
        else:
            self.temporal_block = None
        if not self.only_processor:
            self.nodes_encoder = build_mlp(node_input_size, hidden_size, hidden_size)
            self.edges_encoder = build_mlp(edge_input_size, hidden_size, hidden_size)
            self.decode_module = build_mlp(hidden_size, hidden_size, output_size, layer_norm=False)
        self.processor_list = nn.ModuleList(
            [GraphNetBlock(hidden_size=hidden_size, use_gated_mlp=use_gated_mlp, use_rope=use_rope_embeddings,
                           rope_axes=rope_pos_dimension, rope_base=rope_base, use_gate=use_gated_attention)
             for _ in range(message_passing_num)])

    def forward(self, graph) -> torch.Tensor:
        edge_index = graph.edge_index
        if self.use_rope and getattr(graph, "pos", None) is None:
            raise ValueError("Graph data must contain `pos` when use_rope_embeddings=True.")
        n = graph.x.shape[0]
        topo = getattr(graph, "mgn_topology", None)
        if topo is None:
            # cached per edge_index; built without a host synchronisation; large meshes are renumbered for locality
            # (ops.set_node_renumbering) -- node rows are permuted here on entry and back on exit
            topo = ops.get_topology(edge_index, n, pos=getattr(graph, "pos", None), renumber=True)
        perm = topo.perm_dst_long
        order, rank = topo.node_order, topo.node_rank
        x_in = graph.x if order is None else graph.x.index_select(0, order)
        if self.only_processor:
            x, e = x_in, graph.edge_attr[perm]
        else:
            x = self.nodes_encoder(x_in)
            # encode edges directly in the engine's dst-sorted order (F_e floats per edge to permute)
            e = self.edges_encoder(graph.edge_attr[perm])
        blocks = list(self.processor_list)
        pos = getattr(graph, "pos", None) if self.use_rope else None
        phi = getattr(graph, "phi", None) if self.use_gate else None
        if order is not None:
            pos = pos.index_select(0, order) if pos is not None else None
            phi = phi.reshape(n, -1).index_select(0, order) if phi is not None else None
        # the temporal block (processors.py:193-209) needs the node latents BEFORE the last round too
        groups = [blocks] if not (self.use_temporal_block and len(blocks) > 1) else [blocks[:-1], blocks[-1:]]
        prev_x = x
        for grp in groups:
            prev_x = x
            if self.use_gated_mlp:
                from .gated import gated_block_forward
                for block in grp:
                    prev_x = x
                    x, e = gated_block_forward(block, x, e, topo, pos, phi)
            elif grp:
                params = []
                for block in grp:
                    params += _block_params(block)
                b0 = grp[0]
                x, e = ops.processor_apply(x, e, topo, len(grp), *params, spec=b0.spec, pos=pos, phi=phi,
                                           rope_inv_freq=b0._rope_inv_freq if self.use_rope else None)
        if self.use_temporal_block and self.temporal_block is not None:
            if self.attention_backend == "pyg":   # row-wise without an adjacency: any numbering
                x = self.temporal_block(prev_x, x, None)
            else:
                from .transformer import get_attn_topology
                if rank is not None:  # the attention topology is in the caller's numbering
                    x, prev_x, rank = x.index_select(0, rank), prev_x.index_select(0, rank), None
                x = self.temporal_block(prev_x, x, get_attn_topology(edge_index, n))
        out = x if self.only_processor else self.decode_module(x)
        if rank is not None:  # back to the caller's numbering (the decoder is row-wise: permute its narrow output)
            out = out.index_select(0, rank)
        # every launch of this pass is queued: the wait for a lazily built topology's flags (stray index -> IndexError,
        # hub tables for later passes) cannot starve the GPU now
        if not topo.resolved:
            topo.resolve()
        return out
