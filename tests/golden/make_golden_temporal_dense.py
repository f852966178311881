#!/usr/bin/env python3
"""Mint golden vectors for the temporal block of an installation WITHOUT DGL (build container only): there
``EncodeProcessDecode`` / ``EncodeTransformDecode`` hand ``TemporalAttention`` no adjacency (processors.py:203-209, :376-377) and
``scaled_dot_product_attention`` (layers.py:493-559) attends over the head axis of each node.  No ``dgl`` stand-in here: the
reference is imported as this image has it (HAS_DGL_SPARSE False); loguru / torch_geometric stand-ins as in make_golden.py.

Run:  python tests/golden/make_golden_temporal_dense.py            (needs /root/reference)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    if not os.path.isdir(MG.REF):
        sys.exit("reference checkout not present: goldens can only be minted in the build container")
    os.environ["GRAPH_PHYSICS_ASSUME_NO_DGL"] = "1"   # the reference's own switch: do not prompt for the missing DGL
    MG.install_standins()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import graphphysics.models.layers as RL
    import graphphysics.models.processors as RP
    assert not RL.HAS_DGL_SPARSE and not RP.HAS_DGL_SPARSE
    from torch_geometric.data import Data

    import recipe as R
    from oracle import mgn_oracle as O

    def close(a, b, what, tol=2e-6):
        e = float((a - b).abs().max() / b.abs().max())
        assert e < tol, f"oracle differs from the reference for {what}: {e:.2e}"
        return e

    out = {}
    for name, c in R.TEMPORAL_DENSE_CASES.items():
        H, nh, N, seed = c["hidden"], c["heads"], c["N"], c["seed"]
        mod = RL.TemporalAttention(hidden_size=H, num_heads=nh, use_gate=c.get("gate", True))
        params = R.variant_params(mod.state_dict(), seed)
        mod.load_state_dict(params)
        h_prev, h_pred, cot = (R.randn((N, H), seed + i).requires_grad_(i < 3) for i in (1, 2, 3))
        o = mod(h_prev, h_pred, None)
        err = close(O.temporal_attention(h_prev.detach(), h_pred.detach(), params, "", None, nh, use_gate=c.get("gate", True)), o.detach(), name)
        (o * cot).sum().backward()
        out[name + ".keys"] = np.array("|".join(params.keys()))
        out[name + ".out"] = o.detach()
        out[name + ".d_prev"], out[name + ".d_pred"] = h_prev.grad.clone(), h_pred.grad.clone()
        for k, p_ in mod.state_dict(keep_vars=True).items():
            if p_.grad is not None:
                out[f"{name}.gnorm.{k}"] = p_.grad.norm()
                if p_.dim() == 1:
                    out[f"{name}.g.{k}"] = p_.grad.clone()
        print(f"{name}: oracle vs reference {err:.1e}")

    c = R.EPD_TEMPORAL_NODGL
    H, L, N, seed = c["hidden"], c["L"], c["N"], c["seed"]
    _, ei, ea = R.delaunay_graph(N, seed, dim=2)
    net = RP.EncodeProcessDecode(message_passing_num=L, node_input_size=11, edge_input_size=3, output_size=2, hidden_size=H,
                                 use_temporal_block=True)
    params = R.variant_params(net.state_dict(), seed)
    net.load_state_dict(params)
    x_in, e_in, cot = R.randn((N, 11), seed + 1), R.randn((ea.shape[0], 3), seed + 2), R.randn((N, 2), seed + 3)
    o = net(Data(x=x_in, edge_attr=e_in, edge_index=ei))
    x, e = O.mlp(x_in, params, "nodes_encoder."), O.mlp(e_in, params, "edges_encoder.")
    prev = x
    for i in range(L):
        prev = x
        x, e = O.graph_net_block(x, e, ei, params, f"processor_list.{i}.")
    err = close(O.mlp(O.temporal_attention(prev, x, params, "temporal_block.", None, 4), params, "decode_module."), o.detach(), "epd")
    (o * cot).sum().backward()
    name = "epd_temporal_nodgl"
    out[name + ".keys"] = np.array("|".join(params.keys()))
    out[name + ".out"] = o.detach()
    for k, p_ in net.state_dict(keep_vars=True).items():
        if p_.grad is not None:
            out[f"{name}.gnorm.{k}"] = p_.grad.norm()
    print(f"{name}: oracle vs reference {err:.1e}")
    path = os.path.join(HERE, "temporal_dense.npz")
    np.savez_compressed(path, **{k: (t.detach().numpy() if torch.is_tensor(t) else np.asarray(t)) for k, t in out.items()})
    print(f"wrote temporal_dense.npz ({os.path.getsize(path) / 1024:.0f} kB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
