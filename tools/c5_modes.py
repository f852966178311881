import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scipy.spatial import Delaunay
import graph_physics_amd as gp
from graph_physics_amd import harness, ops, preprocess as PP, transformer as T
dev = torch.device("cuda:0")
n = 150000
pts = np.random.default_rng(0).random((n, 3)).astype(np.float32)
if "sorted" in sys.argv[1:]:   # the points numbered along a Morton curve (what a node renumbering inside the engine would see)
    q = (pts * 1023).astype(np.int64)
    key = np.zeros(n, dtype=np.int64)
    for b in range(10):
        for ax in range(3):
            key |= ((q[:, ax] >> b) & 1) << (3 * b + ax)
    pts = pts[np.argsort(key, kind="stable")]
ei = PP.faces_to_edges(torch.from_numpy(Delaunay(pts).simplices.T.astype(np.int64)).to(dev), n)
cfg = {"model": {"type": "transformer", "message_passing_num": 10, "hidden_size": 64, "node_input_size": 14, "output_size": 3, "edge_input_size": 0, "num_heads": 4}, "training": {"use_temporal_block": False}}
torch.manual_seed(0)
net = gp.get_model(cfg).to(dev)
g = gp.Graph(x=torch.randn(n, 23, device=dev), edge_index=ei, pos=torch.from_numpy(pts).to(dev))
if os.environ.get("C5_PIN_TOPOLOGY"):   # the caller's numbering (no Morton renumbering inside the engine)
    g.mgn_attn_topology = T.get_attn_topology(ei, n)
tgt = torch.randn(n, 3, device=dev)
opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)
def train():
    loss = ((net(g) - tgt) ** 2).mean(); opt.zero_grad(); loss.backward(); opt.step()
def fwd():
    with torch.no_grad(): net(g)
def timed(fn, k=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
from graph_physics_amd import dense as _D
for rep in range(3):
    for mode in ("fp32", "bf16"):
        ops.set_matrix_precision(mode)
        for wt in ((True, False) if "wt" in sys.argv[1:] else (True,)):
            for fm in ((True, False) if "fused" in sys.argv[1:] else (True,)):
                for s16 in ((True, False) if ("save16" in sys.argv[1:] and mode == "bf16") else (True,)):
                    _D._WT_ON[0], _D._FUSED_MLP[0], _D._SAVE16[0] = wt, fm, s16
                    print(mode, f"forward {timed(fwd):.2f} ms  train {timed(train, 8):.2f} ms" + ("" if wt else "   (W^T materialised)")
                          + ("" if fm else "   (gated-MLP half as separate autograd nodes)") + ("" if s16 else "   (fp32 rows for the saved bf16 values)"), flush=True)
_D._WT_ON[0] = _D._FUSED_MLP[0] = _D._SAVE16[0] = True
ops.set_matrix_precision("fp32")
