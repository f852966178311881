"""GPU: the register-resident-weights generation of the edge update (csrc/mgn_ppr.inc; MGN_PPR: default = launches from 65 536 rows, both modes) through the C ABI (mgn_mlp_fwd):
every output of the launch -- e', the fused aggregation, the saved activations with their sign bits, U, rms -- against an fp64
evaluation of the reference's edge update (layers.py:1044-1060, 163-210, 104-129) on ragged row counts (tile tails, one row,
rows past the last tile) in inference and training mode, and against the x6 static-shape kernel it replaces: saved activations and sign bits BIT-identical (same MFMA terms in the same
order on the same packed weights), u / e' / aggregate to rounding (the row norm adds its squares in another order)."""
import os

import pytest
import torch

import graph_physics_amd as gp
from graph_physics_amd import _capi, ops

pytestmark = pytest.mark.gpu
H = 128


@pytest.fixture(scope="module")
def case():
    dev = torch.device("cuda:0")
    g = gp.cylinder_batch(6, 1885, 0).to(dev)
    topo = ops.Topology(g.edge_index, g.x.shape[0])
    f = dict(dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=gen, **f)  # noqa: E731
    x, e = rn(topo.N, H), rn(topo.E, H)
    W0 = rn(H, 3 * H) * 0.05
    Wh = [rn(H, H) * 0.09 for _ in range(3)]
    bs = [rn(H) * 0.1 for _ in range(4)]
    sc = torch.rand(H, generator=gen, **f) + 0.5
    Pd, Ps = x @ W0[:, H:2 * H].t(), x @ W0[:, 2 * H:].t()
    pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
    units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
    ops.wpack([(W0.data_ptr(), 3 * H, False, units[0])] + [(Wh[l].data_ptr(), H, False, units[l + 1]) for l in range(3)], dev)
    d = torch.float64
    z = e.to(d) @ W0[:, :H].to(d).t() + Pd.to(d)[topo.dst_s.long()] + Ps.to(d)[topo.src_s.long()] + bs[0].to(d)
    hs = []
    for l in range(3):
        h = z.clamp_min(0)
        hs.append(h)
        z = h @ Wh[l].to(d).t() + bs[l + 1].to(d)
    rms = z.norm(dim=1, keepdim=True) / H ** 0.5
    u = z / (rms + 1e-8)
    m = sc.to(d) * u
    return dict(dev=dev, topo=topo, e=e, W0=W0, Wh=Wh, bs=bs, sc=sc, Pd=Pd, Ps=Ps, units=units, pk=pk,
                ref=dict(e_new=e.to(d) + m, m=m, H=hs, U=u, R=rms[:, 0]))


def _run(c, M, save, pp):
    dev, topo = c["dev"], c["topo"]
    f = dict(dtype=torch.float32, device=dev)
    old = os.environ.get("MGN_PP"), os.environ.get("MGN_PPR")
    os.environ["MGN_PP"] = "0"
    os.environ["MGN_PPR"] = "2" if pp else "0"
    try:
        sl = slice(0, M)
        nn = int(topo.dst_s[M - 1]) + 1
        dst = topo.dst_s[sl].contiguous()
        rowptr = torch.searchsorted(dst, torch.arange(nn + 1, device=dev, dtype=torch.int32)).to(torch.int32)
        e_new = torch.full((M, H), float("nan"), **f)
        agg = torch.full((nn, H), float("nan"), **f)
        part = torch.full(((M + 15) // 16, 2, H), float("nan"), **f)
        He = [torch.full((M, H), float("nan"), **f) for _ in range(3)] if save else None
        Ue, Re = (torch.full((M, H), float("nan"), **f), torch.full((M,), float("nan"), **f)) if save else (None, None)
        Me = [torch.zeros(M, 4, dtype=torch.int32, device=dev) for _ in range(3)] if save else None
        ops.mlp_fwd(M, H, [(c["e"][sl], None, H)], [c["W0"]] + c["Wh"], c["bs"], c["sc"], H, c["e"][sl], e_new, None, He, Ue, Re, ldw0=3 * H,
                    adds=[(c["Pd"], dst), (c["Ps"], topo.src_s[sl].contiguous())], wpk=c["units"], saveM=Me, seg=(dst, rowptr, agg, part))
        ops.seg_fix(rowptr, part, agg)
        torch.cuda.synchronize()
    finally:
        for k, v in zip(("MGN_PP", "MGN_PPR"), old):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return dict(e_new=e_new, agg=agg, H=He, U=Ue, R=Re, M=Me)


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("save", [False, True])
@pytest.mark.parametrize("M", [0, -10, -23, 40000, 8193, 257, 129, 128, 33, 32, 31, 17, 16, 1])
def test_ppr_edge_update_vs_fp64_and_x6(case, M, save):
    """M <= 0: all rows of the 6-mesh batch (E ~ 67 500: several tiles per workgroup), minus |M|"""
    topo, ref = case["topo"], case["ref"]
    M = topo.E + M if M <= 0 else M
    got = _run(case, M, save, True)
    base = _run(case, M, save, False)
    nn = int(topo.dst_s[M - 1]) + 1
    agg_ref = torch.zeros(nn, H, dtype=torch.float64, device=case["dev"]).index_add_(0, topo.dst_s[:M].long(), ref["m"][:M])
    tol = 2e-6
    assert _rel(got["e_new"], ref["e_new"][:M]) < tol and not bool(torch.isnan(got["e_new"]).any())
    assert _rel(got["agg"], agg_ref) < tol and not bool(torch.isnan(got["agg"]).any())
    assert _rel(got["e_new"], base["e_new"].double()) < tol and _rel(got["agg"], base["agg"].double()) < tol   # the kernel it replaces
    if save:
        for l in range(3):
            assert _rel(got["H"][l], ref["H"][l][:M]) < tol
            bits = (got["H"][l].view(M, 8, 4, 4) > 0).permute(0, 2, 1, 3).reshape(M, 4, 32).long()   # [row][g][4 ib + r]
            want = (bits << torch.arange(32, device=case["dev"])).sum(-1)
            assert torch.equal(got["M"][l].long() & 0xffffffff, want), f"sign bits of layer {l + 1}"
            assert torch.equal(got["H"][l], base["H"][l]) and torch.equal(got["M"][l], base["M"][l]), f"layer {l + 1} differs from the x6 kernel"
        assert _rel(got["U"], ref["U"][:M]) < tol and _rel(got["R"], ref["R"][:M]) < tol


# ------------------------------------------------------------------ the backward chain (k_edge_bwd_ppr)
@pytest.fixture(scope="module")
def bcase(case):
    """forward saves of the whole batch (x6 kernel) + an fp64 backward FROM THOSE saves (both kernels read the same mask bits)"""
    c = case
    dev, topo = c["dev"], c["topo"]
    f = dict(dtype=torch.float32, device=dev)
    E = topo.E
    fwd = _run(c, E, True, False)
    gen = torch.Generator(device=dev).manual_seed(7)
    dOut = torch.randn(E, H, generator=gen, **f)
    dAgg = torch.randn(topo.N, H, generator=gen, **f)
    pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
    bu = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
    Wh, W0 = c["Wh"], c["W0"]
    ops.wpack([(Wh[2].data_ptr(), H, True, bu[0]), (Wh[1].data_ptr(), H, True, bu[1]), (Wh[0].data_ptr(), H, True, bu[2]), (W0.data_ptr(), 3 * H, True, bu[3])], dev)
    d = torch.float64
    dY = dOut.to(d) + dAgg.to(d)[topo.dst_s.long()]
    U64, R64 = fwd["U"].to(d), fwd["R"].to(d)
    gg = c["sc"].to(d) * dY
    dz3 = gg / (R64[:, None] + 1e-8) - U64 * ((gg * U64).sum(1, keepdim=True) / (H * R64[:, None]))
    dz2 = (dz3 @ Wh[2].to(d)) * (fwd["H"][2] > 0)
    dz1 = (dz2 @ Wh[1].to(d)) * (fwd["H"][1] > 0)
    dz0 = (dz1 @ Wh[0].to(d)) * (fwd["H"][0] > 0)
    dE = dOut.to(d) + dz0 @ W0[:, :H].to(d)
    return dict(fwd=fwd, dOut=dOut, dAgg=dAgg, bu=bu, pk=pk, ref=dict(dZ=[dz0, dz1, dz2, dz3], dE=dE, dYU=dY * U64))


def _run_bwd(c, b, M, ppr):
    dev, topo = c["dev"], c["topo"]
    f = dict(dtype=torch.float32, device=dev)
    old = os.environ.get("MGN_PPR")
    os.environ["MGN_PPR"] = "2" if ppr else "0"
    try:
        sl = slice(0, M)
        dZ = [torch.full((M, H), float("nan"), **f) for _ in range(4)]
        dE = torch.full((M, H), float("nan"), **f)
        dsc = torch.full((H,), float("nan"), **f)
        fw = b["fwd"]
        ops.mlp_bwd(M, H, 4, b["dOut"][sl], b["dAgg"], topo.dst_s[sl].contiguous(), H, fw["U"][sl], fw["R"][sl], c["sc"], [t[sl] for t in fw["H"]],
                    [None] * 4, dZ, [(None, b["dOut"][sl], dE)], [None] * 4, dsc, wpk=b["bu"], Ms=[t[sl] for t in fw["M"]])
        torch.cuda.synchronize()
    finally:
        if old is None:
            os.environ.pop("MGN_PPR", None)
        else:
            os.environ["MGN_PPR"] = old
    return dict(dZ=dZ, dE=dE, dscale=dsc)


@pytest.mark.parametrize("M", [0, -10, -23, 40000, 8193, 257, 129, 128, 33, 32, 31, 17, 16, 1])
def test_ppr_edge_backward_chain_vs_fp64_and_x6(case, bcase, M):
    """every output of the launch -- dZ[3..0], dE, dscale -- against the fp64 backward of the reference's edge update (RMSNorm
    layers.py:104-129, build_mlp :163-210, edge_update :1044-1060) and against the x6 static-shape chain it replaces"""
    topo, ref = case["topo"], bcase["ref"]
    M = topo.E + M if M <= 0 else M
    got = _run_bwd(case, bcase, M, True)
    base = _run_bwd(case, bcase, M, False)
    tol = 2e-6
    for l in range(4):
        assert _rel(got["dZ"][l], ref["dZ"][l][:M]) < tol and not bool(torch.isnan(got["dZ"][l]).any()), f"dZ[{l}]"
        assert _rel(got["dZ"][l], base["dZ"][l].double()) < tol
    assert _rel(got["dE"], ref["dE"][:M]) < tol and not bool(torch.isnan(got["dE"]).any())
    assert _rel(got["dscale"], ref["dYU"][:M].sum(0)) < tol and _rel(got["dscale"], base["dscale"].double()) < 2 * tol
