#!/usr/bin/env python3
"""Mint golden vectors for the sparse-attention Transformer path from the REFERENCE
(build container only).  The reference's DGL branch needs ``dgl.sparse`` (optional dependency, not in its
requirements.txt, not installed here): a throw-away ``dgl`` stand-in on sys.path provides the three
primitives the reference calls -- spmatrix, bsddmm, SparseMatrix.softmax, bspmm -- written on DENSE masked
tensors from DGL's published semantics (a different formulation than the oracle's edge-list one, so the
two check each other).  Everything else is the unmodified reference: Attention, Transformer,
TemporalAttention, EncodeTransformDecode, EncodeProcessDecode(use_temporal_block=True).

Run:  python tests/golden/make_golden_transformer.py            (needs /root/reference)"""
import os
import sys
import tempfile
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def install_dgl_standin():
    d = tempfile.mkdtemp(prefix="gp_dgl_")
    os.makedirs(os.path.join(d, "dgl", "sparse"))
    open(os.path.join(d, "dgl", "__init__.py"), "w").write("")
    with open(os.path.join(d, "dgl", "sparse", "__init__.py"), "w") as f:
        f.write(textwrap.dedent('''
            """dense-tensor stand-in for the dgl.sparse calls of graphphysics/models/layers.py (DGL 2.x semantics)"""
            import torch
            class SparseMatrix:
                def __init__(self, row, col, val, shape):
                    self.row, self.col, self.val, self.shape = row, col, val, shape
                def softmax(self, dim=1):
                    assert dim == 1
                    L, N = self.shape
                    K = self.val.shape[1]
                    dense = torch.full((L, N, K), float("-inf"), dtype=self.val.dtype)
                    dense[self.row, self.col] = self.val
                    sm = torch.softmax(dense, dim=1)
                    return SparseMatrix(self.row, self.col, sm[self.row, self.col], self.shape)
            def spmatrix(indices, val=None, shape=None):
                row, col = indices[0], indices[1]
                assert torch.unique(row * shape[1] + col).numel() == row.numel(), "the dense stand-in needs a duplicate-free edge list"
                return SparseMatrix(row, col, torch.ones(row.numel()) if val is None else val, shape)
            def bsddmm(A, X1, X2):
                full = torch.einsum("lmk,mnk->lnk", X1, X2)              # [L, N, K]
                v = full[A.row, A.col]
                return SparseMatrix(A.row, A.col, v * (A.val.view(-1, 1) if A.val.dim() == 1 else A.val), A.shape)
            def bspmm(A, X):
                L, N = A.shape
                K = A.val.shape[1]
                dense = torch.zeros(L, N, K, dtype=A.val.dtype)
                dense[A.row, A.col] = A.val
                return torch.einsum("lnk,nfk->lfk", dense, X)
        '''))
    sys.path.insert(0, d)


def main():
    if not os.path.isdir(MG.REF):
        sys.exit("reference checkout not present: goldens can only be minted in the build container")
    MG.install_standins()
    install_dgl_standin()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import graphphysics.models.layers as RL
    assert RL.HAS_DGL_SPARSE
    import graphphysics.models.processors as RP
    assert RP.HAS_DGL_SPARSE
    from torch_geometric.data import Data

    import recipe as R
    from oracle import mgn_oracle as O

    def close(a, b, what, tol=2e-6):
        e = float((a - b).abs().max() / b.abs().max())
        assert e < tol, f"oracle differs from the reference for {what}: {e:.2e}"
        return e

    out = {}
    for name, c in R.TRANSFORMER_CASES.items():
        H, nh, L, N, seed = c["hidden"], c["heads"], c["L"], c["N"], c["seed"]
        pos, ei, _ = R.delaunay_graph(N, seed, dim=c.get("pos_dim", 3))
        fin, fout = c.get("f_in", 11), c.get("out", 2)
        x_in, cot = R.randn((N, fin), seed + 1), R.randn((N, fout), seed + 3)
        if c["model"] == "etd":
            net = RP.EncodeTransformDecode(message_passing_num=L, node_input_size=fin, output_size=fout, hidden_size=H, num_heads=nh,
                                           use_rope_embeddings=c.get("rope", False), use_gated_attention=c.get("gate", False),
                                           rope_pos_dimension=c.get("pos_dim", 3), use_temporal_block=c.get("temporal", False))
            params = R.variant_params(net.state_dict(), seed)
            net.load_state_dict(params)
            o = net(Data(x=x_in, edge_index=ei, pos=pos))
            oo = O.etd_forward(x_in, ei, params, L, nh, pos=pos, use_rope=c.get("rope", False), use_gate=c.get("gate", False),
                               pos_dimension=c.get("pos_dim", 3), use_temporal_block=c.get("temporal", False))
        else:  # EncodeProcessDecode with the temporal tail
            _, ei, ea = R.delaunay_graph(N, seed, dim=2)
            net = RP.EncodeProcessDecode(message_passing_num=L, node_input_size=11, edge_input_size=3, output_size=2, hidden_size=H,
                                         use_temporal_block=True)
            params = R.variant_params(net.state_dict(), seed)
            net.load_state_dict(params)
            e_in = R.randn((ea.shape[0], 3), seed + 2)
            o = net(Data(x=x_in, edge_attr=e_in, edge_index=ei))
            # oracle: L-1 rounds, keep x, last round, temporal tail, decoder
            x = O.mlp(x_in, params, "nodes_encoder.")
            e = O.mlp(e_in, params, "edges_encoder.")
            prev = x
            for i in range(L):
                prev = x
                x, e = O.graph_net_block(x, e, ei, params, f"processor_list.{i}.")
            oo = O.mlp(O.temporal_attention(prev, x, params, "temporal_block.", ei, 4), params, "decode_module.")
        err = close(oo, o.detach(), name)
        (o * cot).sum().backward()
        out[name + ".keys"] = np.array("|".join(params.keys()))
        out[name + ".out"] = o.detach()
        for k, p_ in net.state_dict(keep_vars=True).items():
            if p_.grad is not None:
                out[f"{name}.gnorm.{k}"] = p_.grad.norm()
                if p_.dim() == 1 and ("processor_list.0." in k or "temporal" in k):
                    out[f"{name}.g.{k}"] = p_.grad.clone()
        print(f"{name}: oracle vs reference {err:.1e}")
    path = os.path.join(HERE, "transformer.npz")
    np.savez_compressed(path, **{k: (t.detach().numpy() if torch.is_tensor(t) else np.asarray(t)) for k, t in out.items()})
    print(f"wrote transformer.npz ({os.path.getsize(path) / 1024:.0f} kB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
