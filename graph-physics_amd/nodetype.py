"""Node-type codes of the mesh datasets (the integer stored in ``x[:, node_type_index]``).

The numeric values are part of the data format -- they index the 9-wide one-hot the
simulator appends to the node features and select the loss / rollout masks -- so they
must equal the reference's (graphphysics/utils/nodetype.py:4-15)."""
import enum

_CODES = {
    "NORMAL": 0,         # interior fluid / solid node: predicted and trained on
    "OBSTACLE": 1,
    "AIRFOIL": 2,
    "HANDLE": 3,
    "INFLOW": 4,
    "OUTFLOW": 5,        # predicted and trained on, like NORMAL
    "WALL_BOUNDARY": 6,
    "SIZE": 9,           # width of the one-hot (codes 7, 8 are reserved by the data format)
}
NodeType = enum.IntEnum("NodeType", _CODES)

#: node types whose values are predicted (everything else is re-imposed from the ground truth)
PREDICTED = (NodeType.NORMAL, NodeType.OUTFLOW)
