"""GPU: the BASELINE.json configurations in their STATED form against the oracle, and the
parity gaps the round-1 review named:

  * configs[2]  plate workload: 3-D tetrahedral mesh -> on-device faces_to_edges -> add_world_edges
                -> edge_features -> EncodeProcessDecode(F_e=4, O=3, L=15) in the bf16 matrix mode,
                forward and one training step against the oracle's bf16-mixed semantic;
  * configs[3]  the 1M-node / 6M-edge mesh at full size on one GPU: 15-round inference
                (determinism, finiteness) and a 2-round forward checked against the oracle on
                2-hop neighbourhoods (locality: the size-independent property of L rounds);
  * the aggregation FUSED into the edge kernel against the oracle's ``agg`` on the ragged multigraph;
  * gradients at the benchmark sizes (N=1885 and the batch of 16) by the fp64-oracle criterion,
    with the ReLU masks that differ from the oracle's counted and shown to sit at rounding distance
    from zero;
  * the exact-fp32 MFMA generation (MGN_FP32_MFMA=1) against the oracle on its own.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import recipe as R
import graph_physics_amd as gp
from conftest import REPO, assert_close3, elem_err, rel_err, rms_err
from graph_physics_amd import harness, ops
from graph_physics_amd import preprocess as PP
from oracle import mgn_oracle as O

pytestmark = pytest.mark.gpu
FWD_TOL = 1e-5
GRAD_TOL = 1e-4
BF16_TOL = 3e-2  # stated tolerance of the bf16 matrix mode after 15 rounds ("~1e-2 rel")
BF16_STEP_TOL = ((2e-2, 2e-2), (2e-2, 3e-2))   # (loss, gradient norm) per training step, relative to the bf16-mixed oracle (measured: <= 5.6e-3 / 6.6e-3; round 2 allowed 10 % / 20 %)


# ------------------------------------------------------------------ configs[2]: plate
PLATE_INDEX = {"feature_index_start": 0, "feature_index_end": 6, "output_index_start": 0, "output_index_end": 3,
               "node_type_index": 6}  # training_config/plate.json:23-29


def plate_case(n=1300, seed=61):
    """DeformingPlate-shaped sample: x = [world_pos(3), obstacle displacement(3), node_type],
    y = next world_pos; a block of OBSTACLE nodes hovering next to the NORMAL plate nodes; tetra
    cells from a 3-D Delaunay (plate.json: node_input_size 6, output_size 3; SURVEY C3: F_e = 4)."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    pts = (rng.random((n, 3)) * np.array([1.0, 0.3, 0.3])).astype(np.float32)
    types = np.where(pts[:, 0] < 0.25, 1.0, 0.0).astype(np.float32)      # OBSTACLE | NORMAL
    types[rng.integers(0, n, 40)] = 3.0                                   # HANDLE: never world-linked
    disp = (0.01 * rng.standard_normal((n, 3))).astype(np.float32)
    x = np.concatenate([pts, disp, types[:, None]], axis=1)
    y = (pts + 0.01 * rng.standard_normal((n, 3))).astype(np.float32)
    cells = Delaunay(pts).simplices.T.astype(np.int64)                    # [4, F] tetrahedra
    return torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(pts), torch.from_numpy(cells)


def plate_config(L=15):
    return {"model": {"type": "epd", "message_passing_num": L, "hidden_size": 128, "node_input_size": 6, "output_size": 3,
                      "edge_input_size": 4}, "index": dict(PLATE_INDEX),
            "training": {"enable_vram_optimizations": True}}  # -> Lightning bf16-mixed (train.py:74-78)


def plate_graph_on_device(dev, n=1300, seed=61, radius=0.1):
    """the reference's per-sample transforms, on the device: FaceToEdge -> add_world_edges (:92-140,
    called with the world-position columns and node_type_index) -> Cartesian + Distance (:16-23)"""
    x, y, pos, cells = plate_case(n, seed)
    xd = x.to(dev)
    ei_mesh = PP.faces_to_edges(cells.to(dev), n)
    ei = PP.add_world_edges(xd, ei_mesh, 0, 3, PLATE_INDEX["node_type_index"], radius=radius)
    ea = PP.edge_features(pos.to(dev), ei)
    # the same construction by the oracle (numpy / torch CPU)
    ei_o = O.add_world_edges_oracle(x.numpy(), O.faces_to_edges_oracle(cells.numpy(), n), 0, 3, PLATE_INDEX["node_type_index"], radius)
    assert np.array_equal(ei.cpu().numpy(), ei_o)                         # integer work: bit-exact
    assert ei.shape[1] > ei_mesh.shape[1] + 100                           # the case really has world edges
    ea_o = O.edge_features_oracle(pos, torch.from_numpy(ei_o))
    assert torch.equal(ea.cpu()[:, :3], ea_o[:, :3]) and rel_err(ea[:, 3], ea_o[:, 3]) < 2e-7
    return x, y, pos, torch.from_numpy(ei_o), ea_o, gp.Graph(x=xd, y=y.to(dev), pos=pos.to(dev), edge_index=ei, edge_attr=ea)


def test_plate_world_edges_bf16_forward_vs_mixed_oracle(dev):
    """configs[2] combined: on-device world edges + 4 edge features + 3 outputs + 15 rounds + bf16
    matrix mode.  Bars: fp32 mode <= 1e-5 of the fp32 oracle; bf16 mode within BF16_TOL of the oracle's
    bf16-mixed semantic AND not farther from it than that semantic is from fp32 (x1.5)."""
    L = 15
    x, y, pos, ei, ea, g = plate_graph_on_device(dev)
    N = x.shape[0]
    params = R.make_params(R.epd_param_shapes(L, 128, 6 + 9, 4, 3), 62)
    x_in = R.randn((N, 15), 63)
    ref32 = O.epd_forward(x_in, ea, ei, params, L)
    with O.bf16_mixed():
        ref16 = O.epd_forward(x_in, ea, ei, params, L)
    net = gp.EncodeProcessDecode(L, 15, 4, 3, hidden_size=128).to(dev)
    net.load_state_dict(params)
    graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr, edge_index=g.edge_index, pos=g.pos)
    with torch.no_grad():
        out32 = net(graph)
        ops.set_matrix_precision("bf16")
        try:
            out16 = net(graph)
        finally:
            ops.set_matrix_precision("fp32")
    assert out32.shape == (N, 3)
    assert_close3(out32, ref32, FWD_TOL, "plate fp32")
    semantic_gap = rel_err(ref16, ref32)            # what bf16-mixed itself costs on this net
    e16 = rel_err(out16, ref16)
    assert 1e-5 < rel_err(out16, ref32) < BF16_TOL  # really the bf16 path, inside the stated tolerance
    assert e16 < BF16_TOL and rms_err(out16, ref16) < BF16_TOL, e16
    assert e16 < 1.5 * semantic_gap + 1e-3, (e16, semantic_gap)


def test_plate_bf16_training_step_vs_mixed_oracle(dev):
    """configs[2]: training steps of the plate workload through Simulator + Engine with
    enable_vram_optimizations (the bf16 matrix mode) against O.train_steps(mixed=True): loss and
    gradient norm per step (2 % / 2-3 %: measured 0.6 %); every parameter gradient of the 15-round and of a 2-round net
    against the ORACLE's bf16-mixed gradient (Frobenius-relative), bounded by how far that oracle itself is from fp32."""
    x, y, pos, ei, ea, g = plate_graph_on_device(dev)
    ix = PLATE_INDEX
    for L, check_grads in ((15, True), (2, True)):
        cfg = plate_config(L)
        seed = 70 + L
        params = R.make_params(R.epd_param_shapes(L, 128, 15, 4, 3), seed)
        eng = harness.Engine(cfg, dev, learning_rate=1e-4, num_steps=100, warmup=4)
        try:
            assert ops.get_matrix_precision() == "bf16"
            eng.model.load_state_dict(params)
            logs, grads = [], []
            for _ in range(2):
                logs.append((float(eng.train_step(g)), float(eng.last_grad_norm)))
                if not grads:  # after step(): .grad holds the CLIPPED gradients
                    coef = min(1.0, 1.0 / (logs[0][1] + 1e-6))
                    grads.append({k: p.grad.detach().cpu() / coef for k, p in eng.model.named_parameters()})
        finally:
            ops.set_matrix_precision("fp32")
        ref = {}
        for mixed in (True, False):
            p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
            sim = O.SimulatorOracle(ix, 15, 4, 3)
            go = []
            lg = O.train_steps(p, sim, [(x, y, ea, ei)] * 2, L, 1e-4, 4, 100, mixed=mixed, grads_out=go)
            ref[mixed] = (lg, go[0])
        (lg16, g16), (lg32, g32) = ref[True], ref[False]
        # step 0: the same weights on both sides; step 1 follows an AdamW update (|dw| = lr whatever the
        # gradient's size: bf16 rounding noise on small gradients moves the two trajectories apart)
        for t, (ltol, gtol) in enumerate(BF16_STEP_TOL):
            el, eg = abs(logs[t][0] - lg16[t][0]) / lg16[t][0], abs(logs[t][1] - lg16[t][1]) / lg16[t][1]
            print(f"plate bf16 L={L} step {t}: loss error {el:.2e}, grad-norm error {eg:.2e} vs the mixed oracle "
                  f"(mixed vs fp32 oracle: loss {abs(lg16[t][0] - lg32[t][0]) / lg32[t][0]:.2e}, grad norm {abs(lg16[t][1] - lg32[t][1]) / lg32[t][1]:.2e})")
            assert el < ltol, (L, t, logs, lg16)
            assert eg < gtol, (L, t, logs, lg16)
        if check_grads:
            for k in g16:
                a, b, c = grads[0][k].double(), g16[k].double(), g32[k].double()
                gap = float((b - c).norm() / c.norm())       # bf16-mixed oracle vs fp32 oracle
                err = float((a - b).norm() / b.norm())       # engine vs bf16-mixed oracle
                assert err < max(1.5 * gap, 0.02), (k, err, gap)


def test_bf16_matrix_mode_off_the_packed_path_hidden32_vs_mixed_oracle(dev):
    """VERDICT r3, missing 4: `training.enable_vram_optimizations` (bf16-mixed, train.py:74-78,268-293) applies to any width; the
    shipped training_config/cylinder.json is hidden 32, 5 rounds -- the generic exact-fp32 kernels with operands and layer results
    rounded to bf16 (precision = 1).  Forward against the oracle's bf16-mixed semantic (and not farther from it than that semantic
    is from fp32, x1.5); one training step: loss and gradient norm within 2 % / 3 % of the mixed oracle, every parameter gradient
    bounded by how far the mixed oracle itself is from fp32."""
    L, Hh = 5, 32
    g = gp.cylinder_mesh(600, 3)
    N = g.x.shape[0]
    params = R.make_params(R.epd_param_shapes(L, Hh, 11, 3, 2), 91)
    x_in = R.randn((N, 11), 92)
    ref32 = O.epd_forward(x_in, g.edge_attr, g.edge_index, params, L)
    with O.bf16_mixed():
        ref16 = O.epd_forward(x_in, g.edge_attr, g.edge_index, params, L)
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=Hh).to(dev)
    net.load_state_dict(params)
    graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=g.edge_index.to(dev))
    with torch.no_grad():
        out32 = net(graph)
        ops.set_matrix_precision("bf16")
        try:
            out16 = net(graph)
        finally:
            ops.set_matrix_precision("fp32")
    assert_close3(out32, ref32, FWD_TOL, "hidden 32 fp32")
    gap = rel_err(ref16, ref32)
    e16 = rel_err(out16, ref16)
    # (hidden 32: bf16-mixed itself sits 3 % from fp32 on this net -- the bars scale with that measured gap)
    assert 1e-5 < rel_err(out16, ref32) < max(BF16_TOL, 2.0 * gap)   # really a bf16 evaluation
    assert e16 < max(BF16_TOL, 1.5 * gap) and rms_err(out16, ref16) < max(BF16_TOL, 1.5 * gap), (e16, gap)
    # one training step through Simulator + Engine with the JSON switch
    cfg = gp.cylinder_config(L, Hh)
    cfg["training"]["enable_vram_optimizations"] = True
    eng = harness.Engine(cfg, dev, learning_rate=1e-4, num_steps=100, warmup=4)
    try:
        assert ops.get_matrix_precision() == "bf16"
        eng.model.load_state_dict(params)
        loss = float(eng.train_step(g.to(dev)))
        gn = float(eng.last_grad_norm)
        coef = min(1.0, 1.0 / (gn + 1e-6))
        grads = {k: p.grad.detach().cpu() / coef for k, p in eng.model.named_parameters()}
    finally:
        ops.set_matrix_precision("fp32")
    ref = {}
    for mixed in (True, False):
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        sim = O.SimulatorOracle(cfg["index"], 11, 3, 2)
        go = []
        lg = O.train_steps(p, sim, [(g.x, g.y, g.edge_attr, g.edge_index)], L, 1e-4, 4, 100, mixed=mixed, grads_out=go)
        ref[mixed] = (lg[0], go[0])
    (l16, g16), (l32, g32) = ref[True], ref[False]
    assert abs(loss - l16[0]) / l16[0] < 0.02 and abs(gn - l16[1]) / l16[1] < 0.03, (loss, gn, l16)
    for k in g16:
        a, b, c = grads[k].double(), g16[k].double(), g32[k].double()
        gapk = float((b - c).norm() / c.norm())
        err = float((a - b).norm() / b.norm())
        assert err < max(1.5 * gapk, 0.03), (k, err, gapk)


# ------------------------------------------------------------------ configs[3]: 1M nodes
def test_c4_full_size_one_gpu(dev):
    """configs[3] at full size (1 000 000 nodes / ~6 000 000 directed edges, latent 128) on ONE GPU.
    Forward against the oracle through LOCALITY: the output at a node after L rounds depends on its L-hop
    in-neighbourhood only, so the oracle evaluated on the sub-mesh induced by the L-hop closure of a seed
    set (edge order preserved = same summation order) must reproduce the full-mesh result on the seeds;
    seeds are taken at the start, the middle and the end of the node range (row offsets past 2^31 bytes).
    (i) **[r5]** the config's own 15 rounds on six seeds (rounds 2-4 checked the 15-round run for finiteness and
    determinism only); (ii) 2 rounds on 900 seeds.  Both runs finite and run-to-run bit-identical."""
    N = 1_000_000
    g = gp.square_mesh(N, seed=0)
    ei = g.edge_index
    E = ei.shape[1]
    assert 5_900_000 < E < 6_100_000
    x_in = torch.randn(N, 11, generator=torch.Generator().manual_seed(1))
    e_in = g.edge_attr
    graph = gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=ei.to(dev))
    graph.mgn_topology = ops.Topology(graph.edge_index, N)
    topo = graph.mgn_topology
    assert int(topo.rowptr_dst[-1]) == E and bool((topo.dst_s[1:] >= topo.dst_s[:-1]).all())
    src, dst = ei[0].numpy(), ei[1].numpy()

    def closure_check(L, seeds, param_seed, what):
        """engine on the WHOLE mesh against the oracle on the sub-mesh induced by the L-hop in-closure of ``seeds``"""
        params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), param_seed)
        net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
        net.load_state_dict(params)
        with torch.no_grad():
            full = net(graph)
            again = net(graph)
        assert full.shape == (N, 2) and bool(torch.isfinite(full).all())
        assert torch.equal(full, again)            # fixed summation order, no atomics
        out = full.cpu()
        del net, full, again
        inR = np.zeros(N, dtype=bool)
        inR[seeds] = True
        need_dst = inR.copy()                      # R0
        for hop in range(L):                       # R_{h+1} = R_h + sources of the edges into R_h
            if hop == L - 1:
                need_dst = inR.copy()              # edges into R_{L-1} are the ones the sub-mesh must hold
            inR[src[inR[dst]]] = True
        keep = need_dst[dst]                       # sources are in R_L by construction
        nodes = np.nonzero(inR)[0]
        loc = np.full(N, -1, dtype=np.int64)
        loc[nodes] = np.arange(nodes.size)
        sub_ei = torch.from_numpy(np.stack([loc[src[keep]], loc[dst[keep]]]))
        assert int(sub_ei.min()) >= 0 and nodes.size < 20000, nodes.size
        ref = O.epd_forward(x_in[nodes], e_in[torch.from_numpy(np.nonzero(keep)[0])], sub_ei, params, L)
        assert_close3(out[seeds], ref[loc[seeds]], FWD_TOL, what)

    # (i) [r5] 15 rounds, the config's own depth: finite, bit-identical run to run, and against the ORACLE on the 15-hop
    # closures of six seed nodes (a 15-hop ball of this mesh holds ~700-1100 nodes; generator numbering: the balls are disjoint)
    closure_check(15, np.array([0, 1, N // 2, N // 2 + 1, N - 2, N - 1]), 6, "1M-node mesh, 15 rounds, seeds")
    # (ii) 2 rounds vs the oracle on the 2-hop closures of 900 seeds (row offsets past 2^31 bytes at the end of the range)
    closure_check(2, np.concatenate([np.arange(0, 300), np.arange(N // 2, N // 2 + 300), np.arange(N - 300, N)]), 5,
                  "1M-node mesh, 2 rounds, seeds")

# ------------------------------------------------- fused aggregation vs the oracle
def test_fused_aggregation_vs_oracle_on_ragged_multigraph(dev):
    """The DEFAULT path's aggregation -- segmented DPP scan inside the edge kernel + mgn_seg_fix --
    against ``agg`` (and the messages against ``m``) of the oracle's GraphNetBlock on the ragged
    multigraph (isolated node, self loops, duplicates) and on a hub-heavy one, through the raw C ABI
    with the production launch shape (split first layer, packed units, seg=...)."""
    from graph_physics_amd import _capi

    H = 128
    rng = np.random.default_rng(12)
    N2 = 700
    dst = np.concatenate([rng.integers(0, N2 - 1, 4000), np.full(45, 3), np.full(200, 500), np.full(17, 650)])
    src = rng.integers(0, N2, dst.size)
    src[:50] = dst[:50]
    hub = torch.from_numpy(np.stack([np.concatenate([src, src[:300]]), np.concatenate([dst, dst[:300]])]))
    for gi, (ei, N) in enumerate(((R.random_graph(40, 150, 31), 40), (hub, N2))):
        E = ei.shape[1]
        seed = 31 + gi
        params = R.make_params(R.epd_param_shapes(1, H, 1, 1, 1, only_processor=True), seed)
        x, e = R.randn((N, H), 1), R.randn((E, H), 2)
        _, _, inter = O.graph_net_block(x, e, ei, params, "processor_list.0.", return_intermediates=True)
        topo = ops.Topology(ei.to(dev), N)
        P = {k: v.to(dev) for k, v in params.items()}
        pre = "processor_list.0.edge_block."
        W0 = P[pre + "0.weight"]
        Ws = [W0] + [P[pre + f"{i}.weight"] for i in (2, 4, 6)]
        bs = [P[pre + f"{i}.bias"] for i in (0, 2, 4, 6)]
        xs, es = x.to(dev), e.to(dev)[topo.perm_dst.long()].contiguous()
        Pd, Ps = xs @ W0[:, H:2 * H].t(), xs @ W0[:, 2 * H:].t()   # node projections (plain fp32 here)
        pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
        units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
        ops.wpack([(W0.data_ptr(), 3 * H, False, units[0])] + [(Ws[l].data_ptr(), H, False, units[l]) for l in (1, 2, 3)], dev)
        m, e_new = torch.empty(E, H, device=dev), torch.empty(E, H, device=dev)
        agg = torch.full((N, H), float("nan"), device=dev)
        part = torch.full(((E + 15) // 16, 2, H), float("nan"), device=dev)
        ops.mlp_fwd(E, H, [(es, None, H)], Ws, bs, P[pre + "7.scale"], H, es, e_new, m, ldw0=3 * H,
                    adds=[(Pd, topo.dst_s), (Ps, topo.src_s)], wpk=units, seg=(topo.dst_s, topo.rowptr_dst, agg, part))
        ops.seg_fix(topo.rowptr_dst, part, agg)
        assert not bool(torch.isnan(agg).any())
        assert_close3(m[topo.inv_perm], inter["m"], FWD_TOL, f"messages, graph {gi}")
        assert_close3(agg, inter["agg"], FWD_TOL, f"fused aggregation, graph {gi}")
        iso = np.setdiff1d(np.arange(N), ei[1].numpy())
        assert iso.size > 0 and float(agg[torch.from_numpy(iso).to(dev)].abs().max()) == 0.0  # zeros for isolated nodes


# ------------------------------------------- gradients at the benchmark sizes
def _grad_case(dev, g, L, seed, dtype64=True):
    N, E = g.x.shape[0], g.edge_index.shape[1]
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), seed)
    x_in, e_in, cot = R.randn((N, 11), seed + 1), R.randn((E, 3), seed + 2), R.randn((N, 2), seed + 3)

    def oracle(dtype):
        p = {k: v.clone().to(dtype).requires_grad_(True) for k, v in params.items()}
        inter = [] if dtype == torch.float32 else None
        out = O.epd_forward(x_in.to(dtype), e_in.to(dtype), g.edge_index, p, L, intermediates=inter)
        (out * cot.to(dtype)).sum().backward()
        return out.detach(), {k: v.grad for k, v in p.items()}, inter

    o32, g32, inter = oracle(torch.float32)
    o64, g64, _ = oracle(torch.float64) if dtype64 else (None, None, None)
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
    net.load_state_dict(params)
    graph = gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=g.edge_index.to(dev))
    out = net(graph)
    fn = out.grad_fn
    while fn is not None and type(fn).__name__ != "ProcessorFunctionBackward":  # decoder <- processor
        fn = fn.next_functions[0][0]
    saved, topo = fn.saved_acts, fn.topo
    # ReLU masks of every round against the oracle's pre-activations: differences only where the
    # oracle's pre-activation sits at rounding distance from zero
    flips, total, worst = 0, 0, 0.0
    inv = topo.inv_perm.cpu()
    for i in range(L):
        He, Hn = saved[i]["He"], saved[i]["Hn"]
        for l in range(3):
            for Hs, pre, perm in ((He, inter[i]["edge_pre"], inv), (Hn, inter[i]["node_pre"], None)):
                hip = Hs[l].cpu() > 0
                hip = hip[perm] if perm is not None else hip
                z = pre[l]
                diff = hip != (z > 0)
                flips += int(diff.sum())
                total += z.numel()
                if bool(diff.any()):
                    worst = max(worst, float(z[diff].abs().max() / z.abs().max()))
    (out * cot.to(dev)).sum().backward()
    grads = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    return out.detach().cpu(), grads, (o32, g32), (o64, g64), (flips, total, worst)


def _check_grads(grads, g32, g64, flips, worst):
    """Gradient bar at depth.  Measured on the MI355X (tools/diag_grads.py): with NO differing ReLU mask the
    engine tracks the fp32 oracle to 3e-6 at L=15 while BOTH sit 3e-3 from the fp64 oracle (rounding
    amplified by 30 MLPs: the reference's own fp32 arithmetic is that far from exact); 13 of 75.7 M masks
    differ at N=1885 (every one at |z| < 4e-7 of the layer's scale) and move the engine and the CPU to
    the same distance from fp64.  Hence: no differing mask -> GRAD_TOL against the fp32 oracle; otherwise
    every differing mask must sit at rounding distance from zero, and the engine must be as close to the
    fp64 oracle as the fp32 oracle itself is (x2: both distances are dominated by a handful of flips)."""
    e32 = {k: rel_err(grads[k], g32[k]) for k in grads}
    if flips == 0:
        bad = [(k, v) for k, v in e32.items() if v >= GRAD_TOL]
        assert not bad, bad[:5]
        return
    assert worst < 1e-5, worst
    h64 = max(rel_err(grads[k], g64[k]) for k in grads)
    c64 = max(rel_err(g32[k], g64[k]) for k in grads)
    assert h64 < 2.0 * c64 + 1e-6, (h64, c64, flips)
    assert max(e32.values()) < 4.0 * c64 + GRAD_TOL, (max(e32.values()), c64)


def test_gradients_at_benchmark_mesh_size(dev):
    """N=1885, L=15 (configs[0]/[1] mesh): forward at 1e-5 in all three readings; gradients by the
    flip-aware bar of _check_grads; the differing masks are counted (<= 1e-6 of all activations)."""
    g = gp.cylinder_mesh(1885, 0)
    out, grads, (o32, g32), (o64, g64), (flips, total, worst) = _grad_case(dev, g, 15, 77)
    assert_close3(out, o32, FWD_TOL, "forward N=1885 L=15")
    assert flips <= 1e-6 * total, (flips, total)
    _check_grads(grads, g32, g64, flips, worst)


def test_gradients_small_mesh_deep_net_no_flips_tight(dev):
    """N=400, L=15: a case without a single differing mask -- every gradient within GRAD_TOL (measured 3e-6)
    of the fp32 oracle although both are 3e-3 from fp64."""
    g = gp.cylinder_mesh(400, 1)
    out, grads, (o32, g32), (o64, g64), (flips, total, worst) = _grad_case(dev, g, 15, 79)
    assert_close3(out, o32, FWD_TOL, "forward N=400 L=15")
    _check_grads(grads, g32, g64, flips, worst)
    if flips == 0:
        assert max(rel_err(grads[k], g32[k]) for k in grads) < 2e-5


def test_gradients_at_batch16_size(dev):
    """the batch of 16 meshes (N=30 160, E=180 082: the bench workload), L=15, fp32 + fp64 oracle"""
    g = gp.cylinder_batch(16, 1885, 0)
    out, grads, (o32, g32), (o64, g64), (flips, total, worst) = _grad_case(dev, g, 15, 78)
    assert_close3(out, o32, FWD_TOL, "forward batch-16 L=15")
    assert flips <= 1e-6 * total, (flips, total)
    _check_grads(grads, g32, g64, flips, worst)


def test_gradients_silu_at_benchmark_mesh_size_tight(dev):
    """The same depth and size with SiLU (smooth: no masks to flip): every gradient within 5e-5 of the fp32
    oracle (measured 1e-5) -- the backward arithmetic itself holds the bar at depth."""
    gp.layers.set_use_silu_activation(True)
    try:
        net = gp.EncodeProcessDecode(15, 11, 3, 2, hidden_size=128).to(dev)
    finally:
        gp.layers.set_use_silu_activation(False)
    g = gp.cylinder_mesh(1885, 0)
    N, E = g.x.shape[0], g.edge_index.shape[1]
    params = R.make_params(R.epd_param_shapes(15, 128, 11, 3, 2), 77)
    net.load_state_dict(params)
    x_in, e_in, cot = R.randn((N, 11), 78), R.randn((E, 3), 79), R.randn((N, 2), 80)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.epd_forward(x_in, e_in, g.edge_index, p, 15, act="silu")
    (ref * cot).sum().backward()
    out = net(gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=g.edge_index.to(dev)))
    (out * cot.to(dev)).sum().backward()
    assert_close3(out, ref, FWD_TOL, "SiLU forward N=1885 L=15")
    for k, q in net.named_parameters():
        assert rel_err(q.grad, p[k].grad) < 5e-5, k


def test_training_steps_at_benchmark_mesh_size(dev):
    """3 optimiser steps at N=1885 / L=15 through Simulator + Engine against O.train_steps: loss and
    gradient norm per step, and the weights after the last step.  AdamW moves EVERY weight by ~lr whatever
    the size of its gradient, so an element whose gradient is rounding noise takes a step of random sign
    on either side: weights are compared through the fraction of elements whose update differs, and the
    later steps' losses carry the tolerance that leaves."""
    L, seed, lr = 15, 79, 1e-4
    g = gp.cylinder_mesh(1885, 3)
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), seed)
    eng = harness.Engine(gp.cylinder_config(L, 128), dev, learning_rate=lr, num_steps=100, warmup=4)
    eng.model.load_state_dict(params)
    gd = g.to(dev)
    logs = [(float(eng.train_step(gd)), float(eng.last_grad_norm)) for _ in range(3)]
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.train_steps(p, O.SimulatorOracle(gp.cylinder_config()["index"], 11, 3, 2), [(g.x, g.y, g.edge_attr, g.edge_index)] * 3,
                        L, lr, 4, 100)
    for t, (ltol, gtol) in enumerate(((2e-5, 3e-4), (3e-4, 1e-3), (3e-4, 1e-3))):
        assert abs(logs[t][0] - ref[t][0]) < ltol * ref[t][0], (t, logs, ref)
        assert abs(logs[t][1] - ref[t][1]) < gtol * ref[t][1], (t, logs, ref)
    sd = eng.model.state_dict()
    step = lr * 0.25  # the first (smallest) step of the warm-up
    off = tot = 0
    for k in p:
        d = (sd[k].cpu() - p[k].detach()).abs()
        off += int((d > 0.2 * step).sum())
        tot += d.numel()
        assert float(d.max()) < 8 * lr, k          # never farther apart than the sum of the three steps, both ways
    assert off < 5e-3 * tot, (off, tot)


# ------------------------------------------------ the exact-fp32 MFMA generation
def test_exact_fp32_generation_vs_oracle():
    """MGN_FP32_MFMA=1 selects the exact-fp32 MFMA kernels (k_mlp_*_lds, k_wgrad_lds) for every
    launch: forward 1e-5 / gradients by the suite's criterion against the ORACLE (not against the
    split-bf16 path), training mode, L=3, ragged multigraph + Delaunay mesh."""
    code = r"""
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import recipe as R, graph_physics_amd as gp
from graph_physics_amd import ops
from oracle import mgn_oracle as O
assert not ops.X6_ENABLED
dev = torch.device("cuda:0")
L = 3
for name, N, ei in (("delaunay", 700, R.delaunay_graph(700, 33)[1]), ("multigraph", 300, R.random_graph(300, 2000, 34))):
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 35)
    x_in, e_in, cot = R.randn((N, 11), 1), R.randn((ei.shape[1], 3), 2), R.randn((N, 2), 3)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.epd_forward(x_in, e_in, ei, p, L)
    (ref * cot).sum().backward()
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev); net.load_state_dict(params)
    out = net(gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=ei.to(dev)))
    (out * cot.to(dev)).sum().backward()
    err = float((out.detach().cpu() - ref.detach()).abs().max() / ref.detach().abs().max())
    assert err < 1e-5, (name, err)
    for k, q in net.named_parameters():
        g = p[k].grad
        ge = float((q.grad.cpu() - g).abs().max() / g.abs().max())
        assert ge < 3e-4, (name, k, ge)
print("ok")
"""
    code = code % (REPO, os.path.join(REPO, "tests", "golden"))
    env = dict(os.environ, MGN_FP32_MFMA="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]


# ------------------------------------------------------- activation recompute
def test_activation_recompute_bit_identical_and_smaller(dev, monkeypatch):
    """ops.set_activation_recompute("on"): the forward keeps only each round's inputs and the backward
    re-runs the round in training mode -- same kernels on the same operands: outputs bit-identical, gradients
    equal to rounding (the re-run takes its node projections from two stand-alone launches instead of the
    previous node kernel's post-products), with a fraction of the activation memory."""
    L, N = 6, 6000
    # like with like: the saved-activation backward fuses the dX launch of a round with the node chain of the round before (the default
    # since round 3, other order of additions: 7e-6 on single gradients); a re-run round cannot (its activations do not exist yet)
    monkeypatch.setenv("MGN_FRONT", "0")
    g = gp.cylinder_mesh(N, 2)
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 5)
    x_in, e_in, cot = R.randn((N, 11), 1).to(dev), R.randn((g.edge_index.shape[1], 3), 2).to(dev), R.randn((N, 2), 3).to(dev)
    res = {}
    assert ops.get_activation_recompute() == "auto"
    for mode in ("off", "on", 4):   # 4: the first four of the six rounds re-run, the last two saved
        ops.set_activation_recompute(mode)
        try:
            net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
            net.load_state_dict(params)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(dev)
            base = torch.cuda.memory_allocated(dev)
            out = net(gp.Graph(x=x_in, edge_attr=e_in, edge_index=g.edge_index.to(dev)))
            held = torch.cuda.memory_allocated(dev) - base
            (out * cot).sum().backward()
            res[mode] = (out.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}, held)
            del out, net
        finally:
            ops.set_activation_recompute("auto")
    assert torch.equal(res["on"][0], res["off"][0])          # same forward arithmetic: bit-identical outputs
    worst = max(rel_err(res["on"][1][k], res["off"][1][k]) for k in res["off"][1])
    assert worst < 2e-6, worst                                 # re-run rounds: gradients to fp32 rounding
    assert res["on"][2] < 0.45 * res["off"][2], (res["on"][2], res["off"][2])
    assert torch.equal(res[4][0], res["off"][0])
    assert max(rel_err(res[4][1][k], res["off"][1][k]) for k in res["off"][1]) < 2e-6
    assert res["on"][2] < res[4][2] < 0.75 * res["off"][2], (res["on"][2], res[4][2], res["off"][2])


# ------------------------------------------------------------------ hub nodes (R0)
def test_hub_nodes_take_the_chunked_segment_sums(dev):
    """Arbitrary edge_index (input contract R0): a node with 30 000 in-edges and one with 20 000 out-edges.
    The topology flags them, the processor switches to chunked, parallel segment sums (deterministic, two
    levels) and forward / gradients still match the oracle."""
    N, seed = 600, 9
    rng = np.random.default_rng(seed)
    base = R.delaunay_graph(N, seed)[1].numpy()
    hub_in = np.stack([rng.integers(0, N, 30000), np.full(30000, 7)])       # everything points at node 7
    hub_out = np.stack([np.full(20000, 11), rng.integers(0, N, 20000)])      # node 11 points everywhere
    ei = torch.from_numpy(np.concatenate([base, hub_in, hub_out], axis=1).astype(np.int64))
    E = ei.shape[1]
    topo = ops.Topology(ei.to(dev), N)
    assert topo.hub_dst is not None and topo.hub_src is not None and topo.has_hubs
    crp, ncp = topo.hub_dst
    assert int((crp[1:] - crp[:-1]).max()) <= ops.HUB_CHUNK and int(ncp[-1]) == crp.numel() - 1
    # the chunked sum against the plain one
    m = R.randn((E, 128), 3).to(dev)
    out = torch.empty(N, 128, device=dev)
    # the chunked sums against an fp64 evaluation: at least as close as the one-pass sequential sum (30 000
    # fp32 additions in a row carry ~1e-5 of rounding themselves)
    md = m.double()
    for by, key, rp, pm in (("dst", topo.dst_s, topo.rowptr_dst, None), ("src", topo.src_s, topo.rowptr_src, topo.perm_src)):
        ref64 = torch.zeros(N, 128, dtype=torch.float64, device=dev).index_add_(0, key.long(), md)
        ops.segsum_topo(m, topo, by, out)
        e_chunk, e_seq = rel_err(out, ref64), rel_err(ops.segsum(m, rp, pm), ref64)
        assert e_chunk < 2e-6 and e_chunk <= e_seq + 1e-7, (by, e_chunk, e_seq)
    # end to end
    L = 2
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), seed)
    x_in, e_in, cot = R.randn((N, 11), 1), 0.1 * R.randn((E, 3), 2), R.randn((N, 2), 3)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.epd_forward(x_in, e_in, ei, p, L)
    (ref * cot).sum().backward()
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
    net.load_state_dict(params)
    g = gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=ei.to(dev))
    g.mgn_topology = topo
    o = net(g)
    (o * cot.to(dev)).sum().backward()
    assert_close3(o, ref, FWD_TOL, "hub graph forward")
    for k, q in net.named_parameters():
        assert rel_err(q.grad, p[k].grad) < 1e-3, k   # sums of 30 000 rows in another order + ReLU masks
