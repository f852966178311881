#!/usr/bin/env python3
"""Mint golden vectors for the GraphNetBlock VARIANTS from the REFERENCE
implementation (build container only; same stand-ins as make_golden.py):

  silu        model.use_silu_activation      layers.py:132-160
  gate        model.use_gated_attention      layers.py:982-987,1091-1098  (with and without graph.phi)
  rope2/rope3 model.use_rope_embeddings      layers.py:961-981,1020-1026,1104-1149
  gated       model.use_gated_mlp            layers.py:213-278,932-942    (GELU; SiLU with the switch)
  l3nonorm    GraphNetBlock(nb_of_layers=3, layer_norm=False)   layers.py:896-906
  combo       silu + gate + rope in one EncodeProcessDecode

Per variant: one block (H=128, ragged multigraph) with outputs and gradients, and a 2-round
EncodeProcessDecode on a Delaunay mesh.  Weights come from ``recipe.variant_params`` (numpy streams),
so fixtures hold outputs only.  While minting, the oracle's restatement of every variant is checked
bit-for-bit against the reference.

Run:  python tests/golden/make_golden_variants.py            (needs /root/reference)
"""
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
    MG.install_standins()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import graphphysics.models.layers as RL  # noqa: E402
    from graphphysics.models.processors import EncodeProcessDecode as RefEPD  # noqa: E402
    from torch_geometric.data import Data  # stand-in

    import recipe as R
    from oracle import mgn_oracle as O

    def exact(a, b, what):
        d = (a - b).abs().max().item() if a.numel() else 0.0
        assert d == 0.0, f"oracle is not bit-exact vs reference for {what}: max|diff|={d}"

    out = {}
    for name, v in R.VARIANTS.items():
        RL.set_use_silu_activation(v["act"] == "silu")
        try:
            H = 128
            # ---- one block on the ragged multigraph (isolated node, duplicates, self loops)
            N, E, seed = 40, 150, 300 + v["seed"]
            ei = R.random_graph(N, E, seed)
            blk = RL.GraphNetBlock(hidden_size=H, **R.block_kwargs(v))
            params = R.variant_params(blk.state_dict(), seed)
            blk.load_state_dict(params)
            out[name + ".blk.keys"] = np.array("|".join(params.keys()))   # the reference's state_dict order = the weight stream's
            x = R.randn((N, H), seed + 1).requires_grad_(True)
            e = R.randn((E, H), seed + 2).requires_grad_(True)
            pos = R.randn((N, 3), seed + 5, 0.3)
            phi = R.randn((N,), seed + 6) if v.get("phi") else None
            cx, ce = R.randn((N, H), seed + 3), R.randn((E, H), seed + 4)
            x2, e2 = blk(x, ei, e, pos=pos if v["variant"].get("use_rope") else None, phi=phi)
            ((x2 * cx).sum() + (e2 * ce).sum()).backward()
            op = {"processor_list.0." + k: t for k, t in params.items()}
            ox, oe = O.graph_net_block(x.detach(), e.detach(), ei, op, "processor_list.0.", v["act"], variant=v["variant"], pos=pos, phi=phi)
            exact(ox, x2.detach(), name + " block x'")
            exact(oe, e2.detach(), name + " block e'")
            # fixture size: node tensors whole, edge tensors as 32 rows + norm
            out[name + ".blk.x_out"], out[name + ".blk.dx"] = x2.detach(), x.grad
            out[name + ".blk.e_out.rows32"], out[name + ".blk.e_out.norm"] = e2.detach()[:32].clone(), e2.detach().norm()
            out[name + ".blk.de.rows32"], out[name + ".blk.de.norm"] = e.grad[:32].clone(), e.grad.norm()
            for k, p_ in blk.state_dict(keep_vars=True).items():
                g = p_.grad
                if g is None:       # gate_pos without phi
                    continue
                if g.dim() == 2:
                    out[f"{name}.blk.g.{k}.rows4"], out[f"{name}.blk.g.{k}.norm"] = g[:4].clone(), g.norm()
                else:
                    out[f"{name}.blk.g.{k}"] = g.clone()
            # ---- EncodeProcessDecode, 2 rounds, Delaunay mesh
            L, N2, seed2 = 2, 200, 400 + v["seed"]
            pos2, ei2, ea2 = R.delaunay_graph(N2, seed2, dim=v["variant"].get("rope_axes", 3) if v["variant"].get("use_rope") else 2)
            net = RefEPD(message_passing_num=L, node_input_size=11, edge_input_size=ea2.shape[1], output_size=2, hidden_size=H,
                         **R.epd_kwargs(v))
            params2 = R.variant_params(net.state_dict(), seed2)
            net.load_state_dict(params2)
            out[name + ".epd.keys"] = np.array("|".join(params2.keys()))
            x_in, e_in = R.randn((N2, 11), seed2 + 1), R.randn((ea2.shape[0], ea2.shape[1]), seed2 + 2)
            phi2 = R.randn((N2,), seed2 + 6) if v.get("phi") else None
            cot = R.randn((N2, 2), seed2 + 3)
            g = Data(x=x_in, edge_attr=e_in, edge_index=ei2, pos=pos2)
            if phi2 is not None:
                g.phi = phi2
            o = net(g)
            (o * cot).sum().backward()
            oo = O.epd_forward(x_in, e_in, ei2, params2, L, act=v["act"], variant=v["variant"], pos=pos2, phi=phi2)
            exact(oo, o.detach(), name + " EPD output")
            out[name + ".epd.out"] = o.detach()
            for k, p_ in net.state_dict(keep_vars=True).items():
                if p_.grad is not None:
                    out[f"{name}.epd.gnorm.{k}"] = p_.grad.norm()
                    if p_.dim() == 1 and ("processor_list.0." in k or "encoder.0." in k):
                        out[f"{name}.epd.g.{k}"] = p_.grad.clone()
        finally:
            RL.set_use_silu_activation(False)
    path = os.path.join(HERE, "block_variants.npz")
    np.savez_compressed(path, **{k: (t.detach().numpy() if torch.is_tensor(t) else np.asarray(t)) for k, t in out.items()})
    print(f"wrote block_variants.npz ({os.path.getsize(path) / 1024:.0f} kB, {len(out)} arrays); oracle bit-exact vs the reference on every variant")


if __name__ == "__main__":
    main()
