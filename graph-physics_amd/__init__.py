"""graph_physics_amd: MI355X-native MeshGraphNet message-passing engine behind the
Encoder/Processor/Decoder module API of DonsetPG/graph-physics."""
from .nodetype import NodeType  # noqa: F401
from .mesh import Graph, collate, cylinder_mesh, cylinder_batch, square_mesh, plate_mesh  # noqa: F401
from .layers import GraphNetBlock, Normalizer, RMSNorm, build_mlp  # noqa: F401
from .processors import EncodeProcessDecode  # noqa: F401
from .transformer import Attention, EncodeTransformDecode, TemporalAttention, Transformer  # noqa: F401
from .simulator import Simulator  # noqa: F401
from .parse_parameters import get_model, get_simulator, cylinder_config, plate_config, matrix_precision_from_config  # noqa: F401
from .ops import set_matrix_precision, get_matrix_precision, set_node_renumbering, get_node_renumbering  # noqa: F401
from . import preprocess  # noqa: F401

__version__ = "0.1.0"
