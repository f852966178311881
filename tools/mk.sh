#!/bin/bash
# build the engine library in-tree (same flags as _capi.build()) and record the source hash the loader checks
cd "$(dirname "$0")/.."
make -C graph-physics_amd/csrc 2>&1 | grep -E "error|Error" 
python -c "
from graph_physics_amd import _capi
open(_capi.HASH_PATH,'w').write(_capi.source_hash()+'\n'); print('libmgn_hip.so version', _capi.lib().mgn_version())"
