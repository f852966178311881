"""Node-type codes used by the simulator's one-hot and by the loss / rollout
masks.  Values mirror the reference enum (graphphysics/utils/nodetype.py:4-15)."""
import enum


class NodeType(enum.IntEnum):
    NORMAL = 0
    OBSTACLE = 1
    AIRFOIL = 2
    HANDLE = 3
    INFLOW = 4
    OUTFLOW = 5
    WALL_BOUNDARY = 6
    SIZE = 9
