"""GPU, ONE rank over RCCL (backend "nccl"): the halo-exchange machinery of the partitioned path --
device pack, all_to_all_single on RCCL's stream straight into the ghost rows, interior rows overlapping the
collective, reverse exchange + fixed-order unpack in the backward -- on a SELF-EXCHANGE plan: a set B of
source nodes is duplicated as ghost rows that the single rank "receives" from itself, and the edges leaving
B are re-pointed at the duplicates.  The ghost rows are exact copies, so forward and every gradient must
equal the plain (un-partitioned) engine on the original mesh.  A 1-GPU box cannot host two RCCL ranks; this
is the closest the collective path gets to hardware before the driver's multi-GPU run."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

def test_halo_exchange_over_rccl_self_exchange():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "halo_rccl_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0 or "ok" not in r.stdout:
        print(r.stdout[-3000:])
        print(r.stderr[-6000:])
    assert r.returncode == 0 and "ok" in r.stdout
