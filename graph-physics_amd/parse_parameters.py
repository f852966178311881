"""JSON ``training_config`` surface for the MeshGraphNet path: same keys as the
reference factories (graphphysics/training/parse_parameters.py:81-190)."""
from __future__ import annotations

from typing import Any, Dict

import torch

from .layers import set_use_silu_activation
from .nodetype import NodeType
from .processors import EncodeProcessDecode
from .simulator import Simulator


def get_model(param: Dict[str, Any], only_processor: bool = False):
    model = param.get("model", {})
    model_type = model.get("type", "")
    node_input_size = param["model"]["node_input_size"] + NodeType.SIZE  # parse_parameters.py:96
    training = param.get("training", {})
    set_use_silu_activation(model.get("use_silu_activation", False))
    if model_type == "epd":
        return EncodeProcessDecode(
            message_passing_num=param["model"]["message_passing_num"],
            node_input_size=node_input_size,
            edge_input_size=param["model"]["edge_input_size"],
            output_size=param["model"]["output_size"],
            hidden_size=param["model"]["hidden_size"],
            only_processor=only_processor,
            use_rope_embeddings=model.get("use_rope_embeddings", False),
            use_gated_attention=model.get("use_gated_attention", False),
            use_gated_mlp=model.get("use_gated_mlp", False),
            rope_pos_dimension=model.get("rope_pos_dimension", 3),
            rope_base=model.get("rope_base", 10000.0),
            use_temporal_block=training.get("use_temporal_block", False),
        )
    if model_type == "transformer":  # parse_parameters.py:129-142
        from .transformer import EncodeTransformDecode
        return EncodeTransformDecode(
            message_passing_num=param["model"]["message_passing_num"],
            node_input_size=node_input_size,
            output_size=param["model"]["output_size"],
            hidden_size=param["model"]["hidden_size"],
            num_heads=param["model"]["num_heads"],
            only_processor=only_processor,
            use_rope_embeddings=model.get("use_rope_embeddings", False),
            use_gated_attention=model.get("use_gated_attention", False),
            rope_pos_dimension=model.get("rope_pos_dimension", 3),
            rope_base=model.get("rope_base", 10000.0),
            use_temporal_block=training.get("use_temporal_block", False),
        )
    if model_type == "transolver":
        raise NotImplementedError("model type 'transolver' is outside the message-passing hot path (SURVEY.md 7b)")
    raise ValueError(f"Model type '{model_type}' not supported.")


def get_simulator(param: Dict[str, Any], model, device: torch.device) -> Simulator:
    return Simulator(
        node_input_size=param["model"]["node_input_size"] + NodeType.SIZE,
        edge_input_size=param["model"]["edge_input_size"],
        output_size=param["model"]["output_size"],
        feature_index_start=param["index"]["feature_index_start"],
        feature_index_end=param["index"]["feature_index_end"],
        output_index_start=param["index"]["output_index_start"],
        output_index_end=param["index"]["output_index_end"],
        node_type_index=param["index"]["node_type_index"],
        model=model,
        device=device,
    )


def matrix_precision_from_config(param: Dict[str, Any]) -> str:
    """``training.enable_vram_optimizations`` is the reference's switch to Lightning
    ``precision="bf16-mixed"`` (train.py:74-78,268-293): bf16 GEMM inputs, fp32 accumulate,
    fp32 RMSNorm / residual stream.  Here that is the engine's "bf16" matrix mode."""
    return "bf16" if param.get("training", {}).get("enable_vram_optimizations", False) else "fp32"


def cylinder_config(message_passing_num: int = 15, hidden_size: int = 128) -> Dict[str, Any]:
    """training_config/cylinder.json with the two benchmark overrides
    (message_passing_num 5->15, hidden_size 32->128; SURVEY.md TL;DR item 1)."""
    return {
        "model": {"type": "epd", "message_passing_num": message_passing_num, "hidden_size": hidden_size,
                  "node_input_size": 2, "output_size": 2, "edge_input_size": 3,
                  "use_silu_activation": False, "use_gated_mlp": False},
        "index": {"feature_index_start": 0, "feature_index_end": 2, "output_index_start": 0,
                  "output_index_end": 2, "node_type_index": 2},
        "training": {"use_spatial_mtp": False, "use_temporal_block": False, "enable_vram_optimizations": False},
    }


def plate_config(message_passing_num: int = 15, hidden_size: int = 128) -> Dict[str, Any]:
    """training_config/plate.json run through the message-passing engine (SURVEY.md TL;DR item 2: ``type: "epd"``,
    ``edge_input_size: 4`` = 3-D Cartesian + distance) under Lightning bf16-mixed (``enable_vram_optimizations``,
    train.py:74-78) -- BASELINE.json configs[2]."""
    return {
        "model": {"type": "epd", "message_passing_num": message_passing_num, "hidden_size": hidden_size,
                  "node_input_size": 6, "output_size": 3, "edge_input_size": 4},
        "index": {"feature_index_start": 0, "feature_index_end": 6, "output_index_start": 0, "output_index_end": 3,
                  "node_type_index": 6},
        "training": {"use_spatial_mtp": False, "use_temporal_block": False, "enable_vram_optimizations": True},
    }
