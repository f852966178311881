#!/bin/bash
# build the engine library in-tree (same flags as _capi.build()) and record the source hash the loader checks.
# A failed build must not leave the previous .so looking current: the hash is written only after make succeeded.
cd "$(dirname "$0")/.."
if ! make -C graph-physics_amd/csrc > /tmp/mgn_make.log 2>&1; then
  grep -E "error|Error" /tmp/mgn_make.log | head -20
  echo "BUILD FAILED (full log: /tmp/mgn_make.log)"
  exit 1
fi
python -c "
from graph_physics_amd import _capi
open(_capi.HASH_PATH,'w').write(_capi.source_hash()+'\n'); print('libmgn_hip.so version', _capi.lib().mgn_version())"
