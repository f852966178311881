"""Import shim: the product package lives in ``graph-physics_amd/`` (the layout
the build contract names), which is not a valid Python identifier.  This module
makes it importable as ``graph_physics_amd`` by pointing ``__path__`` at that
directory and executing its ``__init__``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "graph-physics_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
