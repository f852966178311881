#!/usr/bin/env python3
"""Mint golden vectors for the RMSNorm constructor variants (partial norm p, offset, non-default eps: layers.py:73-129) from the
REFERENCE module (build container only; same stand-ins as make_golden.py).  The oracle's restatement is checked bit for bit
while minting.  Run:  python tests/golden/make_golden_rmsnorm.py   (needs /root/reference)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

CASES = {"partial": dict(d=64, p=0.25, eps=1e-8, bias=False), "biased": dict(d=48, p=-1.0, eps=1e-8, bias=True),
         "partial_biased_eps": dict(d=128, p=0.5, eps=1e-5, bias=True)}


def main():
    if not os.path.isdir(MG.REF):
        sys.exit("reference checkout not present: goldens can only be minted in the build container")
    MG.install_standins()
    import graphphysics.models.layers as RL  # noqa: E402
    import recipe as R
    from oracle import mgn_oracle as O

    out = {}
    for i, (name, kw) in enumerate(CASES.items()):
        d = kw["d"]
        m = RL.RMSNorm(d, p=kw["p"], eps=kw["eps"], bias=kw["bias"])
        scale = R.randn((d,), 900 + i) * 0.3 + 1.0
        offset = R.randn((d,), 910 + i) * 0.2
        sd = {"scale": scale}
        if kw["bias"]:
            sd["offset"] = offset
        m.load_state_dict(sd)
        x = R.randn((37, d), 920 + i).requires_grad_(True)
        cot = R.randn((37, d), 930 + i)
        y = m(x)
        (y * cot).sum().backward()
        xo = x.detach().clone().requires_grad_(True)
        so = scale.clone().requires_grad_(True)
        yo = O.rms_norm_general(xo, so, d, kw["p"], kw["eps"], offset if kw["bias"] else None)
        (yo * cot).sum().backward()
        assert torch.equal(yo.detach(), y.detach()) and torch.equal(xo.grad, x.grad) and torch.equal(so.grad, m.scale.grad), name
        out[name + ".y"], out[name + ".dx"], out[name + ".dscale"] = y.detach().numpy(), x.grad.numpy(), m.scale.grad.numpy()
        if kw["bias"]:
            out[name + ".doffset"] = m.offset.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "rmsnorm_variants.npz"), **out)
    print("wrote rmsnorm_variants.npz:", sorted(out))


if __name__ == "__main__":
    main()
