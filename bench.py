#!/usr/bin/env python3
"""Headline benchmark: MeshGraphNet (CylinderFlow, 15 rounds, latent 128) training
steps/s on MI355X through the HIP engine, plus rollout node*steps/s.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process starts
``python -m torch.distributed.run --nproc-per-node N bench.py ...`` itself (before anything touches
the GPU) and exits with the child's code; under the driver's own torch.distributed.run launch the
ranks are already there (RANK / LOCAL_RANK / WORLD_SIZE) and ``--gpus`` must equal WORLD_SIZE.

One "step" = one optimiser step (forward + masked L2 + backward + clip 1.0 + AdamW)
on one batch of 16 synthetic CylinderFlow meshes (BASELINE.json configs[1]; data
already resident in HBM).  For N>1 every rank steps its own batch of 16 and the
gradients are all-reduced over RCCL each step (weak scaling; value = N*K/T).

The JSON line also carries
  roofline            HBM roofline of the edge-update kernel (training mode), timed in situ with HIP
                      events on the launch stream; the bf16-MFMA fraction rides along,
  roofline_scatter    HBM roofline of the segment-sum (scatter-add) kernel at the bench size,
  roofline_other_kernels   backward chain, weight gradients, inference-mode edge kernel (rollout),
  topology            CSR build time per batch + steps/s when the topology is rebuilt every step,
  plate_bf16          BASELINE configs[2]: plate meshes with world edges in the bf16 matrix mode (batch 1 and 16),
  c5                  BASELINE configs[4]: the coarse-aneurysm Transformer (10 blocks, hidden 64, 4 heads) on a 3-D mesh, fp32 and
                      bf16, with the sparse-attention kernels' HBM rooflines,
  c4                  BASELINE configs[3]: the 1M-node / 6M-edge mesh -- at N = 1 whole-mesh inference,
                      one rank's share of the 8-way partitioned training step, and the scatter-add
                      roofline past the 256 MiB Infinity Cache; at N > 1 the N-way partitioned
                      training / rollout step with the per-round halo exchange over RCCL,
  cpu_baseline        the torch-CPU oracle timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402  (importing torch does not touch the GPU)

PEAK_MFMA_F32 = 157.3  # TFLOP/s dense fp32 MFMA (MI355X_MICROARCH.md chip table)
PEAK_MFMA_BF16 = 2500.0  # TFLOP/s dense bf16 MFMA (same table)
PEAK_HBM = 8000.0  # GB/s spec (same table; ~6300 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (SURVEY 8d: >= 100)")
    ap.add_argument("--warmup", type=int, default=20, help="untimed warm-up steps (SURVEY 8d: >= 20)")
    ap.add_argument("--batch", type=int, default=16, help="meshes per GPU batch (configs[1] = 16)")
    ap.add_argument("--nodes", type=int, default=1885)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--rollout-steps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="replay the training step as one hipGraph (auto: only in the launch-bound regime, "
                         "E <= 65536 edges per GPU batch; at batch 16 eager launches are already GPU-bound and "
                         "measured 7 %% faster than replay)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                    help="matrix precision of the processor GEMMs: fp32 = BASELINE configs[1] (headline); bf16 = the "
                         "reference's bf16-mixed semantic (configs[2]-style), reported with dtype bf16")
    ap.add_argument("--no-c4", action="store_true", help="skip the 1M-node record (BASELINE configs[3])")
    ap.add_argument("--no-extras", action="store_true", help="skip the batch1 / plate_bf16 / c5 records (a kernel trace of the headline alone)")
    ap.add_argument("--c4-nodes", type=int, default=1_000_000)
    ap.add_argument("--c4-steps", type=int, default=3)
    ap.add_argument("--c5-nodes", type=int, default=150_000, help="nodes of the 3-D mesh of the configs[4] (Transformer) record")
    return ap.parse_args()


def maybe_self_launch(args):
    """``python bench.py --gpus N`` on its own: become the launcher (no GPU call has happened yet)."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}; launch {args.gpus} ranks "
                             f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
        return
    if args.gpus <= 1:
        return
    import socket

    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


class EventTimer:
    """HIP events on torch's current stream (= the stream every engine launch goes to) around the
    launches a predicate selects, inside real steps."""

    def __init__(self):
        self.rec = {}

    def wrap(self, tag, fn, pred):
        def w(*a, **k):
            if not pred(a, k):
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.rec.setdefault(tag, []).append((e0, e1))
            return r
        return w

    def ms(self, tag):
        v = self.rec.get(tag)
        return (sum(a.elapsed_time(b) for a, b in v) / len(v)) if v else None

    def count(self, tag):
        return len(self.rec.get(tag, []))


DEVICE_COPY_GBPS = None  # measured once per run: what a plain device copy (half reads, half writes) sustains


def device_copy_rate(dev):
    """GB/s (bytes read + bytes written) of a 1 GiB device-to-device copy, past the 256 MiB Infinity Cache: the
    practical ceiling for read+write streaming on this box (8 TB/s is the read-only spec peak; the training-mode
    kernels WRITE two thirds of their traffic)."""
    global DEVICE_COPY_GBPS
    if DEVICE_COPY_GBPS is None:
        a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        a.normal_()
        for _ in range(2):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        DEVICE_COPY_GBPS = round(5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del a, b
        torch.cuda.empty_cache()
    return DEVICE_COPY_GBPS


#: what a streaming kernel sustains on this part as a function of the WRITE share of its traffic (tools/rw_mix_probe.hip, 1 GiB per
#: stream, far past the Infinity Cache; profiles/r03_rw_mix_probe.txt, mean of the two grid sizes): a read stream reaches 6.4 TB/s,
#: any mix that carries writes 4.5-5.1 TB/s.  `ceiling_gbps` of a roofline object is this table at the kernel's own read : write ratio.
RW_MIX_GBPS = [(0.0, 6370.0), (0.25, 5124.0), (1.0 / 3.0, 4890.0), (0.5, 4935.0), (2.0 / 3.0, 4998.0), (0.75, 5077.0), (1.0, 4516.0)]


def rw_ceiling_gbps(write_frac):
    w = min(max(float(write_frac), 0.0), 1.0)
    for (w0, g0), (w1, g1) in zip(RW_MIX_GBPS, RW_MIX_GBPS[1:]):
        if w <= w1:
            return g0 + (g1 - g0) * (w - w0) / (w1 - w0)
    return RW_MIX_GBPS[-1][1]


def hbm_obj(kernel, t_ms, nbytes, traffic=None, extra=None, write_bytes=None, traffic_rw=None):
    """``write_bytes``: the algorithmic bytes of ``nbytes`` that are stores (the rest are loads); ``traffic_rw``: counted
    (read, write) bytes per launch of the PMC passes when they exist."""
    ach = nbytes / (t_ms * 1e-3) / 1e9
    d = {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM, "unit": "GB/s",
         "frac": round(ach / PEAK_HBM, 4), "traffic": traffic, "launch_ms": round(t_ms, 5),
         "algorithmic_bytes_per_launch": int(nbytes)}
    if write_bytes is not None:
        ceil = rw_ceiling_gbps(write_bytes / max(nbytes, 1.0))
        d.update(read_bytes=int(nbytes - write_bytes), write_bytes=int(write_bytes), ceiling_gbps=round(ceil, 0),
                 frac_of_ceiling=round(ach / ceil, 4),
                 ceiling_note="streaming rate at this kernel's read : write ratio (profiles/r03_rw_mix_probe.txt)")
    if traffic_rw and traffic_rw[0] is not None:
        d.update(traffic_read_bytes=int(traffic_rw[0]), traffic_write_bytes=int(traffic_rw[1]))
    if DEVICE_COPY_GBPS:
        d["device_copy_gbps"] = DEVICE_COPY_GBPS
        d["frac_of_device_copy"] = round(ach / DEVICE_COPY_GBPS, 4)
        if traffic:
            d["traffic_rate_vs_device_copy"] = round(traffic / (t_ms * 1e-3) / 1e9 / DEVICE_COPY_GBPS, 4)
    if extra:
        d.update(extra)
    return d


def step_floor(N, E, H, L, terms, ms_per_step):
    """What the training step cannot go below on this part, priced both ways: the algorithmic bytes of the four
    E-row kernel families of a round (edge update, edge backward chain, weight gradients, backward scatters -- DESIGN.md section 5;
    node-row and encoder / decoder traffic not counted: a lower bound) at 8 TB/s and at the 6.3 TB/s a streaming kernel reaches, and
    the matrix work (forward 12 E H^2 + 10 N H^2 per round, training = 3 x forward, `terms` bf16 MFMA terms per product) at 2.5 PFLOP/s."""
    row = 4.0 * H
    per_round = (row * (6 * E + 3 * N) + 60.0 * E) + (row * (7 * E + N) + 56.0 * E) + row * (8 * E + 12 * N) + (2 * row * (E + N) + 4.0 * E + 8.0 * (N + 1))
    nbytes = L * per_round
    flops = 3.0 * L * (12.0 * E + 10.0 * N) * H * H * terms
    hbm_ms, hbm63_ms, mfma_ms = 1e3 * nbytes / (PEAK_HBM * 1e9), 1e3 * nbytes / 6.3e12, 1e3 * flops / (PEAK_MFMA_BF16 * 1e12)
    floor = max(hbm_ms, mfma_ms)
    return {"algorithmic_bytes_per_step": int(nbytes), "hbm_floor_ms": round(hbm_ms, 3), "hbm_floor_ms_at_6300_GBps": round(hbm63_ms, 3),
            "mfma_flops_per_step_bf16_terms": int(flops), "mfma_floor_ms": round(mfma_ms, 3), "step_floor_ms": round(floor, 3),
            "frac_of_floor": round(floor / ms_per_step, 4), "frac_of_floor_at_6300_GBps": round(max(hbm63_ms, mfma_ms) / ms_per_step, 4),
            "what": "step_floor_ms / ms_per_step; bytes = the E-row kernels of the 15 rounds only (a lower bound of the step's traffic)"}


def _c4_scatter_traffic(capi, rw=False):
    """counted HBM bytes of k_segsum<8> at the 1M-node size (profiles/pmc_traffic.json: c4_segsum_*; same build-hash gate)"""
    t, _ = load_traffic(capi)
    if rw:
        return (t.get("c4_segsum_read_bytes"), t.get("c4_segsum_write_bytes"))
    return t.get("c4_segsum_bytes")


def load_traffic(capi):
    """profiles/pmc_traffic.json (PMC passes, tools/pmc_traffic_summary.py) -- only when it was measured
    on THIS build of the kernels: the file records the hash of csrc/ it was taken with."""
    try:
        with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as fh:
            t = json.load(fh)
    except Exception:  # noqa: BLE001
        return {}, "profiles/pmc_traffic.json missing"
    if t.get("csrc_hash") != capi.source_hash():
        return {}, "profiles/pmc_traffic.json was measured on a different build of csrc/ (hash mismatch): traffic omitted"
    return t, t.get("source")


def kernel_rooflines(gp, ops, capi, eng, batch, dev, steps=3):
    """In-situ per-kernel timings inside `steps` real training steps (cold caches, the real operands)
    and one inference forward; the rocprofv3 kernel-trace average of the same command (profiles/)
    must agree with `launch_ms`."""
    topo = batch.mgn_topology
    N, E, H = topo.N, topo.E, eng.model.hidden_size
    tm = EventTimer()
    orig = (ops.mlp_fwd, ops.mlp_bwd, ops.wgrad, ops.segsum2)

    def is_edge_fwd(a, k):
        return a[0] == E and a[1] == H and bool(k.get("adds"))

    ops.mlp_fwd = tm.wrap("edge_fwd", orig[0], lambda a, k: is_edge_fwd(a, k) and len(a) > 10 and a[10] is not None)
    ops.mlp_bwd = tm.wrap("edge_bwd", orig[1], lambda a, k: a[0] == E and a[1] == H and a[4] is not None)
    ops.wgrad = tm.wrap("wgrad", orig[2], lambda a, k: len(a[0]) >= 6)
    ops.segsum2 = tm.wrap("segsum", orig[3], lambda a, k: a[0].shape[0] == E)
    orig_segsum = ops.segsum
    ops.segsum = tm.wrap("segsum_src", orig_segsum, lambda a, k: a[0].shape[0] == E and a[2] is not None)
    orig_fused = ops.edge_bwd_fused
    ops.edge_bwd_fused = tm.wrap("edge_bwd_fused", orig_fused, lambda a, k: a[0] == E)
    sync, eng.grad_sync = eng.grad_sync, None  # rank 0 steps alone here: no collective (the timed region is over)
    if sync is not None and hasattr(sync, "close"):
        sync.close()   # ... and no bucket all-reduce out of its backward passes either (a closed wrapper is a plain flat all-reduce)
    try:
        for _ in range(steps):
            eng.train_step(batch)
        torch.cuda.synchronize()
        # inference mode (the rollout's edge kernel: nothing saved)
        ops.mlp_fwd = tm.wrap("edge_inf", orig[0], lambda a, k: is_edge_fwd(a, k) and (len(a) <= 10 or a[10] is None))
        eng.rollout([batch] * 2)
        torch.cuda.synchronize()
    finally:
        ops.mlp_fwd, ops.mlp_bwd, ops.wgrad, ops.segsum2 = orig
        ops.segsum = orig_segsum
        ops.edge_bwd_fused = orig_fused
        eng.grad_sync = sync
    traffic, tnote = load_traffic(capi)
    x6 = ops.X6_ENABLED and H == 128
    units = 8.0 * E * H * H  # 4 GEMM units of 2*E*H*H per edge launch (the x projections are N-row work)
    nterm = 1 if ops.get_matrix_precision() == "bf16" else 6

    def mfma(t, terms):
        ach = terms * units / (t * 1e-3) / 1e12
        peak = PEAK_MFMA_BF16 if x6 else PEAK_MFMA_F32
        return {"mfma_achieved_tflops": round(ach, 1), "mfma_peak_tflops": peak, "mfma_frac": round(ach / peak, 4),
                "mfma_note": (f"{terms} bf16 MFMA term(s) per product, fp32 accumulate" if x6 else "fp32 MFMA"),
                "fp32_equiv_tflops": round(units / (t * 1e-3) / 1e12, 1)}

    # algorithmic bytes: fp32 rows of H floats; masks 3 x 16 B, rms 4 B per row
    row = 4.0 * H
    b_fwd = row * (6 * E + 3 * N) + 52.0 * E + 8.0 * E   # e, e', H1-3, U | Pd, Ps, agg | rms + masks | dst/src idx
    b_inf = row * (2 * E + 3 * N) + 8.0 * E              # e, e' | Pd, Ps, agg | dst/src idx
    b_bwd = row * (7 * E + N) + 52.0 * E + 4.0 * E       # dE', U, dZ0-3 (w), dE (w) | dAgg | rms + masks | dst idx
    b_wg = row * (8 * E + 12 * N)                        # 4 edge jobs (dZ, X) + 6 node jobs, 1 launch per round
    b_seg = 2 * row * (E + N) + 4.0 * E + 8.0 * (N + 1)  # two sums of dZ0: read E rows twice, write 2N rows, perm, 2 rowptr
    # the store share of each figure (the rest are loads): e', H1-3, U, agg, rms + masks | e', agg | dZ0-3, dE | (partials only) | 2N rows
    w_fwd, w_inf, w_bwd, w_wg, w_seg = row * (5 * E + N) + 52.0 * E, row * (E + N), row * 5 * E, 0.0, 2 * row * N

    def rw(key):
        return (traffic.get(key + "_read_bytes"), traffic.get(key + "_write_bytes"))
    t = "x6" if x6 else "lds<1>"
    per_step = lambda tag: tm.count(tag) // steps  # noqa: E731
    # the register-resident-weights generation (csrc/mgn_ppr.inc) takes fp32-grade edge launches from 65 536 rows unless MGN_PPR=0
    # (mgn_kernels.hip: fwd_ppr_ok / bwd_ppr_ok)
    ppr = x6 and nterm == 6 and E >= 65536 and os.environ.get("MGN_PPR", "") != "0"
    ppr_bwd = ppr and os.environ.get("MGN_PPR_BWD", "") != "0"
    roof = hbm_obj((f"k_edge_fwd_ppr<true> [traffic: {tnote}]" if ppr else f"k_mlp_fwd_{t} [traffic: {tnote}]") +
                   " (edge update, training mode: W_e.e + gathered node projections, 3 Linear, RMSNorm, "
                   "residual, saves, fused aggregation)", tm.ms("edge_fwd"), b_fwd, traffic.get("edge_fwd_bytes"),
                   dict(mfma(tm.ms("edge_fwd"), nterm if x6 else 1), launches_per_step=per_step("edge_fwd"), traffic_source=tnote),
                   write_bytes=w_fwd, traffic_rw=rw("edge_fwd"))
    others = []
    if tm.ms("edge_bwd_fused"):
        # fused edge backward: reads dE', U, e, H1-3 (6 E-row tensors) + gathered dAgg; writes dE, dZ0; masks, rms, idx;
        # the per-workgroup partials of dW / db / dscale (256 x 265 KB) written and read once by the reduction
        b_fz = row * (8 * E + N) + 52.0 * E + 4.0 * E + 2.0 * 256 * 4 * (4 * (H * H + H) + H)
        tf = tm.ms("edge_bwd_fused")
        mf = mfma(tf, nterm if x6 else 1)
        for k_ in ("mfma_achieved_tflops", "mfma_frac", "fp32_equiv_tflops"):   # 8 GEMM units per row: 4 chain + 4 weight-gradient
            mf[k_] = round(2 * mf[k_], 4 if k_ == "mfma_frac" else 1)
        others.append(hbm_obj("k_edge_bwd_fused + k_fused_red (edge backward chain AND the four E-row weight gradients of a round in one kernel: "
                              "dZ1..dZ3 never reach memory; one workgroup per CU, dW accumulators in AGPRs)", tf, b_fz,
                              traffic.get("edge_bwd_fused_bytes"), dict(mf, launches_per_step=per_step("edge_bwd_fused"))))
        b_wg = row * (12 * N)                            # the node-row jobs that remain in the weight-gradient launch
    if tm.ms("edge_bwd"):
        others.append(hbm_obj(("k_edge_bwd_ppr" if ppr_bwd else f"k_mlp_bwd_{t}") + " (edge backward chain: RMSNorm bwd, 3 masked dgrad GEMMs, input grad)", tm.ms("edge_bwd"), b_bwd,
                              traffic.get("edge_bwd_bytes"), dict(mfma(tm.ms("edge_bwd"), nterm if x6 else 1), launches_per_step=per_step("edge_bwd")),
                              write_bytes=w_bwd, traffic_rw=rw("edge_bwd")))
    if tm.ms("wgrad"):
        pc = x6 and nterm == 6 and os.environ.get("MGN_WGRAD_PC", "1") != "0"   # mgn_wgrad_p: fp32-row full jobs on the producer / consumer kernel
        others.append(hbm_obj(f"k_wgrad_{('pc' if pc else 'x6') if x6 else 'lds'} + k_wgrad_red (weight gradients of one round: "
                              + ("6..7 node jobs; the E-row jobs run inside the fused edge backward)" if tm.ms("edge_bwd_fused") else "4 edge + 6..7 node jobs)")
                              + " -- a READ stream: its ceiling is the 6.4 TB/s of a streaming read, not the write-carrying mix",
                              tm.ms("wgrad"), b_wg, traffic.get("wgrad_bytes"), {"launches_per_step": per_step("wgrad")},
                              write_bytes=w_wg, traffic_rw=rw("wgrad")))
    if tm.ms("edge_inf"):
        # the ping-pong instance takes fp32-grade inference-mode launches from 65 536 rows unless MGN_PP=0 (mgn_kernels.hip: fwd_pp_ok)
        pp = x6 and nterm == 6 and E >= 65536 and os.environ.get("MGN_PP", "") != "0"
        others.append(hbm_obj(("k_edge_fwd_ppr<false>" if ppr else "k_edge_fwd_pp<false>" if pp else f"k_mlp_fwd_{t}") + " (edge update, inference mode = the rollout's "
                              "dominant kernel: nothing saved, aggregation fused)", tm.ms("edge_inf"), b_inf, None,
                              dict(mfma(tm.ms("edge_inf"), nterm if x6 else 1), launches_per_rollout_step=tm.count("edge_inf") // 2),
                              write_bytes=w_inf))
    if tm.ms("segsum"):
        roof_seg = hbm_obj("k_segsum2<8> (CSR segment sums = the scatter-add of the backward pass onto destination and source nodes; "
                           "the forward aggregation is fused into the edge kernel's epilogue). Working set 217 MB < 256 MiB Infinity "
                           "Cache: see roofline_scatter_c4 for the past-L3 measurement", tm.ms("segsum"), b_seg, traffic.get("segsum_bytes"),
                           {"launches_per_step": per_step("segsum")}, write_bytes=w_seg, traffic_rw=rw("segsum"))
    else:  # the destination-side scatter is fused into the backward chain: only the source-side sum is a launch of its own
        b_src = row * (E + N) + 4.0 * E + 4.0 * (N + 1)
        roof_seg = hbm_obj("k_segsum<8> through perm_src (the scatter-add of the backward pass onto SOURCE nodes: gathered 512-byte rows, "
                           "CSR order; the destination-side sums of both passes are fused into the edge kernels). Working set 108 MB < "
                           "256 MiB Infinity Cache: see roofline_scatter_c4 for the past-L3 measurement", tm.ms("segsum_src"), b_src,
                           traffic.get("segsum_src_bytes"), {"launches_per_step": per_step("segsum_src")}, write_bytes=row * N)
    return roof, roof_seg, others


def cpu_baseline(args, gp):
    """The torch-CPU oracle (bit-exact restatement of the reference, oracle/mgn_oracle.py) timed on
    the host cores per SURVEY.md 8d: thread sweep up to the physical core count AT the batch-16 size
    (1 warm-up + 1 timed step each), then 2 warm-up + 5 timed steps at the best count; a 1-thread
    figure on the batch-1 mesh."""
    sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
    import recipe as R
    from oracle import mgn_oracle as O

    torch.manual_seed(0)
    params = R.make_params(R.epd_param_shapes(args.rounds, args.hidden, 11, 3, 2), 0)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ix = gp.cylinder_config()["index"]
    one = gp.cylinder_mesh(args.nodes, 0)
    sim = O.SimulatorOracle(ix, 11, 3, 2)
    b1 = (one.x, one.y, one.edge_attr, one.edge_index)
    nb = min(args.batch, 16)
    big = gp.cylinder_batch(nb, args.nodes, 0)
    bb = (big.x, big.y, big.edge_attr, big.edge_index)

    def steps(batch, n):
        t0 = time.perf_counter()
        O.train_steps(p, sim, [batch] * n, args.rounds, 1e-4, 10, 100)
        return (time.perf_counter() - t0) / n

    ncpu = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or ncpu
    except Exception:  # noqa: BLE001
        phys = ncpu
    t_start = time.perf_counter()
    torch.set_num_threads(1)
    steps(b1, 1)
    t1 = steps(b1, 2)  # 1 thread, batch-1 mesh
    sweep = {}
    best_t, best_n = None, None
    for n in sorted({min(phys, c) for c in (16, 32, 64, phys)}):
        # bounded: the default bench run has to finish within minutes -- stop once the sweep has used its
        # share, or when more threads already made the step much slower (scaling collapsed)
        if best_t is not None and (time.perf_counter() - t_start > 120 or sweep[max(sweep, key=int)] > 1.1 * best_t):
            break  # (more threads than the best count already cost > 10 %: the larger counts only get worse)
        torch.set_num_threads(n)
        steps(bb, 1)  # warm-up at this thread count
        t = steps(bb, 1)
        sweep[str(n)] = round(t, 3)
        if best_t is None or t < best_t:
            best_t, best_n = t, n
    torch.set_num_threads(best_n)
    steps(bb, 2)
    dt = steps(bb, 5)
    torch.set_num_threads(min(best_n, 16))
    steps(b1, 1)
    tb1 = steps(b1, 3)
    return {"value": round(1.0 / dt * (nb / args.batch), 5), "unit": "steps/s", "cores": best_n,
            "kind": "port", "host_cpus_logical": ncpu, "host_cores_physical": phys,
            "batch16_s_per_step_by_threads": sweep, "batch1_steps_per_s": round(1.0 / tb1, 3),
            "one_thread_batch1_steps_per_s": round(1.0 / t1, 3),
            "batch16_vs_16x_batch1": round(dt / (nb * tb1), 3),
            "sample": f"oracle training steps on the batch of {nb} meshes (N={big.x.shape[0]}, E={big.edge_index.shape[1]}): "
            f"2 warm-up + 5 timed at {best_n} threads = {dt:.2f} s/step (sweep at this size, 1 warm-up + 1 timed each: {sweep}); "
            f"batch-1 mesh {tb1:.3f} s/step at {min(best_n, 16)} threads, {t1:.3f} s/step at 1 thread; "
            f"host has {phys} physical cores / {ncpu} logical CPUs"}


def batch1_record(args, gp, ops, harness, dev):
    """BASELINE configs[0] on the GPU (ONE mesh per step: the reference's own CPU-runnable case, and the latency
    regime -- 173 edge tiles, every launch a single tile per workgroup): training step and rollout step under
    hipGraph replay.  Rank 0 of a single-process run only; not the headline."""
    cfg = gp.cylinder_config(args.rounds, args.hidden)
    eng = harness.Engine(cfg, dev, learning_rate=1e-4, num_steps=10000, warmup=100)
    b = gp.cylinder_batch(1, args.nodes, 0).to(dev)
    b.mgn_topology = ops.Topology(b.edge_index, b.x.shape[0])
    eng.capture_train_step(b, warmup=3)
    for _ in range(20):
        eng.train_step_graphed(None)
    torch.cuda.synchronize()
    k = 200
    t0 = time.perf_counter()
    for _ in range(k):
        eng.train_step_graphed(None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    eng.capture_rollout_step(b)
    frames = [b] * 50
    eng.rollout_graphed(frames[:3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.rollout_graphed(frames)
    torch.cuda.synchronize()
    dr = (time.perf_counter() - t0) / len(frames)
    return {"workload": f"one mesh per step (N={b.x.shape[0]}, E={b.edge_index.shape[1]}); BASELINE.json configs[0] on the GPU",
            "train_steps_per_s": round(1.0 / dt, 1), "train_ms_per_step": round(1e3 * dt, 3),
            "rollout_ms_per_step": round(1e3 * dr, 3), "rollout_node_steps_per_s": round(b.x.shape[0] / dr, 1),
            "launch": "hipGraph replay", "steps": k}


def shipped_cylinder_record(args, gp, ops, harness, dev):
    """training_config/cylinder.json AS SHIPPED (message_passing_num 5, hidden_size 32 -- what a user who drops the engine in unchanged
    runs; BASELINE.md's caveat): batch of 16 meshes, training step and rollout step.  Hidden 32 runs on the generic
    exact-fp32 MFMA kernels (k_mlp_fwd / k_mlp_bwd / k_wgrad at HB = 2), not on the packed split-bf16 path (H = 128 only); a step is
    ~170 launches of 5-80 us, so the record's figures are hipGraph REPLAY (Engine.capture_train_step; Engine.rollout replays by itself
    on a static mesh), with the eager figures beside them."""
    cfg = gp.cylinder_config(5, 32)
    eng = harness.Engine(cfg, dev, learning_rate=1e-4, num_steps=10000, warmup=100)
    b = gp.cylinder_batch(args.batch, args.nodes, 0).to(dev)
    n, e = int(b.x.shape[0]), int(b.edge_index.shape[1])
    b.mgn_topology = ops.Topology(b.edge_index, n)

    def timed_loop(fn, k):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    for _ in range(10):
        eng.train_step(b)
    dt_eager = timed_loop(lambda: eng.train_step(b), 50)
    frames = [b] * 20
    eng.rollout(frames[:3], graph="off")
    dr_eager = timed_loop(lambda: eng.rollout(frames, graph="off"), 2) / len(frames)
    dt, dr, launch = dt_eager, dr_eager, "eager"
    if args.graph != "off":
        try:
            eng.capture_train_step(b, warmup=3)
            dt = timed_loop(lambda: eng.train_step_graphed(None), 200)
            eng.rollout(frames)            # graph="auto": captures a step on the first call
            dr = timed_loop(lambda: eng.rollout(frames), 3) / len(frames)
            launch = "hipGraph replay (training: Engine.capture_train_step; rollout: Engine.rollout, graph='auto')"
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] hipGraph capture of the shipped-config step failed ({type(ex).__name__}: {ex}); eager figures", file=sys.stderr)
    return {"workload": f"training_config/cylinder.json as shipped: 5 MP rounds, latent 32, {args.batch} x {args.nodes}-node meshes per step (N={n}, E={e})",
            "kernels": "generic exact-fp32 MFMA kernels (k_mlp_fwd<2,..> / k_mlp_bwd<2,..> / k_wgrad_row64, v_mfma_f32_16x16x4_f32); the packed split-bf16 path needs hidden 128",
            "train_steps_per_s": round(1.0 / dt, 1), "train_ms_per_step": round(1e3 * dt, 3),
            "rollout_ms_per_step": round(1e3 * dr, 3), "rollout_node_steps_per_s": round(n / dr, 1), "launch": launch,
            "eager_train_ms_per_step": round(1e3 * dt_eager, 3), "eager_rollout_ms_per_step": round(1e3 * dr_eager, 3)}


def plate_bf16_record(args, gp, ops, harness, dev):
    """BASELINE configs[2]: DeformingPlate-shaped meshes (~1.3k nodes, 3-D tetrahedra, world-edge set built on the device),
    the reference's bf16-mixed semantic (training.enable_vram_optimizations -> Lightning bf16-mixed, train.py:74-78):
    processor GEMMs on bf16 operands with fp32 accumulation, RMSNorm / residuals / loss in fp32.  One mesh per step
    (hipGraph replay) and a batch of 16 meshes per step (eager).  Restores the fp32 matrix mode on exit."""
    prev = ops.get_matrix_precision()
    rec = {}
    try:
        cfg = gp.plate_config(args.rounds, args.hidden)
        meshes = [gp.plate_mesh(1300, seed=61 + i, device=dev) for i in range(16)]
        for tag, graphs, use_graph, k in (("batch1", meshes[:1], True, 200), ("batch16", meshes, False, 30)):
            eng = harness.Engine(cfg, dev, learning_rate=1e-4, num_steps=10000, warmup=100)   # sets the bf16 matrix mode
            assert ops.get_matrix_precision() == "bf16"
            b = gp.collate(graphs) if len(graphs) > 1 else graphs[0]
            n, e = int(b.x.shape[0]), int(b.edge_index.shape[1])
            b.mgn_topology = ops.Topology(b.edge_index, n)
            if use_graph:
                eng.capture_train_step(b, warmup=3)
                step = lambda: eng.train_step_graphed(None)  # noqa: E731
            else:
                step = lambda: eng.train_step(b)  # noqa: E731
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / k
            frames = [b] * 20
            roll = lambda fr: eng.rollout(fr, graph="off")  # noqa: E731
            if use_graph:
                eng.capture_rollout_step(b)
                roll = eng.rollout_graphed
            roll(frames[:3])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            roll(frames)
            torch.cuda.synchronize()
            dr = (time.perf_counter() - t0) / len(frames)
            rec[tag] = {"nodes": n, "edges": e, "world_edges": e - sum(int(PP_mesh_edges(g_)) for g_ in graphs),
                        "train_steps_per_s": round(1.0 / dt, 1), "train_ms_per_step": round(1e3 * dt, 3),
                        "rollout_ms_per_step": round(1e3 * dr, 3), "rollout_node_steps_per_s": round(n / dr, 1),
                        "launch": "hipGraph replay" if use_graph else "eager"}
            del eng
        rec["workload"] = ("DeformingPlate-shaped 3-D tetrahedral meshes (1300 nodes each; OBSTACLE/NORMAL world edges within 0.1 added on the "
                           "device), node_input 6 (+9 one-hot), edge_input 4, output 3, 15 MP rounds, latent 128; BASELINE.json configs[2]")
        rec["dtype"] = "bf16 (operands of the processor GEMMs; fp32 accumulate, RMSNorm, residuals, loss -- Lightning bf16-mixed)"
    finally:
        ops.set_matrix_precision(prev)
    return rec


def PP_mesh_edges(g):
    """directed mesh edges of a tetrahedral sample before the world edges were added (for the record's world-edge count)"""
    from graph_physics_amd import preprocess as PP

    return PP.faces_to_edges(g.face, g.x.shape[0]).shape[1]


def c5_record(args, gp, ops, harness, dev):
    """BASELINE configs[4]: training_config/coarse-aneurysm.json's model keys (Transformer processor: 10 blocks, hidden 64, 4 heads,
    node_input_size 14 + 9 one-hot, output_size 3) on a synthetic 3-D tetrahedral mesh large enough to leave the Infinity Cache.
    Inference forward and training step (forward + MSE + backward + clip + AdamW) in fp32 and in the bf16 matrix mode, and the
    sparse-attention kernel on its own against the HBM roofline (it gathers two 4*hidden-byte rows per edge)."""
    import numpy as np
    from scipy.spatial import Delaunay

    from graph_physics_amd import preprocess as PP
    from graph_physics_amd import transformer as T

    n = args.c5_nodes
    rng = np.random.default_rng(0)
    pts = rng.random((n, 3)).astype(np.float32)
    cells = torch.from_numpy(Delaunay(pts).simplices.T.astype(np.int64)).to(dev)
    ei = PP.faces_to_edges(cells, n)
    E = int(ei.shape[1])
    H, nh, L = 64, 4, 10
    cfg = {"model": {"type": "transformer", "message_passing_num": L, "hidden_size": H, "node_input_size": 14, "output_size": 3,
                     "edge_input_size": 0, "num_heads": nh, "use_rope_embeddings": False, "use_gated_attention": False},
           "training": {"use_temporal_block": False}}
    torch.manual_seed(0)
    net = gp.get_model(cfg).to(dev)
    graph = gp.Graph(x=torch.randn(n, 23, device=dev), edge_index=ei, pos=torch.from_numpy(pts).to(dev))
    # (no pinned topology: EncodeTransformDecode.forward builds and caches it, renumbering the randomly numbered nodes along a
    #  Morton curve of graph.pos -- transformer.ATTN_RENUMBER_MIN_NODES)
    tgt = torch.randn(n, 3, device=dev)
    opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)

    def timed(fn, k=6):   # (three steps read 12.5 and 13.4 ms on two boxes of the same build: six)
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    def fwd():
        with torch.no_grad():
            net(graph)

    def train():
        loss = ((net(graph) - tgt) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()

    rec = {"workload": f"3-D Delaunay tetrahedral mesh N={n} E={E} (directed), EncodeTransformDecode {L} blocks, hidden {H}, {nh} heads, "
                       "23 inputs, 3 outputs (training_config/coarse-aneurysm.json:11-22); BASELINE.json configs[4]"}
    prev = ops.get_matrix_precision()
    try:
        for mode in ("fp32", "bf16"):
            ops.set_matrix_precision(mode)
            tf, tt = timed(fwd), timed(train)
            rec[mode] = {"forward_ms": round(1e3 * tf, 2), "forward_node_steps_per_s": round(n / tf, 1), "train_ms_per_step": round(1e3 * tt, 2),
                         "train_steps_per_s": round(1.0 / tt, 2)}
    finally:
        ops.set_matrix_precision(prev)
    rec["node_renumbering"] = ("Morton order of graph.pos inside EncodeTransformDecode.forward (rows permuted on entry, back on exit): "
                               + ("on" if T.want_attn_renumbering(n, graph.pos) else "off"))
    # the sparse-attention kernels on their own, on the topology the model runs on
    topo = T.get_attn_topology(ei, n, pos=graph.pos, renumber=True)
    q, k, v = (torch.randn(n, H, device=dev) for _ in range(3))
    dy = torch.randn(n, H, device=dev)

    def ev(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_f = ev(lambda: T.SparseAttentionFn.apply(q, k, v, topo, nh))
    # COMPULSORY bytes: every matrix once (q, k, v read; y + lse written) + the adjacency -- the two gathered rows per edge come out of
    # L2 / the Infinity Cache (three matrices of 4 H n bytes fit it) and are reported beside it as `gathered_bytes_per_launch`
    b_f = 4 * 4.0 * H * n + 4.0 * nh * n + 4.0 * E + 4.0 * (n + 1)
    g_f = 2 * 4.0 * H * E
    cache_note = {"gathered_bytes_per_launch": int(g_f), "gathered_rate_gbps": round(g_f / (t_f * 1e-3) / 1e9, 1),
                  "resident": f"the gathered k / v rows (two {4 * H}-byte rows per edge) come out of L2 / the Infinity Cache: the matrices are "
                              f"{4 * H * n / 1e6:.0f} MB each; `achieved` prices the COMPULSORY bytes only, so frac <= 1; the kernel is bound by "
                              "vector-instruction issue (DESIGN 4.5), not by either figure"}
    rec["roofline_attention"] = hbm_obj("k_attn_fwd<16> (edge-masked QK^T -> online softmax -> AV over the CSR of the mesh adjacency; no matrix "
                                        "cores)", t_f, b_f, None, cache_note, write_bytes=4.0 * H * n + 4.0 * nh * n)
    # the two backward kernels on their own: the C entry point timed directly (through autograd the interval also held the
    # zero-fills / allocations around them: 0.70 ms in situ against 0.54 ms of kernel time under rocprofv3 in round 3)
    from graph_physics_amd import _capi as capi_
    y, lse = T.SparseAttentionFn.apply(q, k, v, topo, nh)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    ws = torch.empty(max(2 * topo.E * nh, 1), dtype=torch.float32, device=dev)

    def bwd():
        with torch.cuda.device(dev):
            rc = capi_.lib().mgn_sparse_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), y.data_ptr(), lse.data_ptr(), dy.data_ptr(),
                                                 topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.cptr.data_ptr(), topo.cperm.data_ptr(),
                                                 topo.crow.data_ptr(), n, topo.E, H, nh, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                                 ws.data_ptr(), ws.numel() * 4, ops._stream(dev))
        capi_.check(rc, "mgn_sparse_attn_bwd", attn=True)

    t_b = ev(bwd)
    # compulsory: q, k, v, y, dy, lse read + dq, dk, dv written once, the per-edge weights / score gradients written by the row pass and
    # read by the column pass, the two adjacencies; gathered rows (5 per edge) reported beside it
    b_b = 4.0 * H * 8 * n + 4.0 * nh * n + 4.0 * nh * 4 * E + 2 * (4.0 * E + 4.0 * (n + 1)) + 8.0 * E
    g_b = 4.0 * H * 5 * E
    note_b = dict(cache_note, gathered_bytes_per_launch=int(g_b), gathered_rate_gbps=round(g_b / (t_b * 1e-3) / 1e9, 1))
    rec["roofline_attention_backward"] = hbm_obj("k_attn_bwd_row + k_attn_bwd_col (two passes: by row dq + per-edge attn / dscore, by column dk, dv); "
                                                 "the two launches timed directly (mgn_sparse_attn_bwd)", t_b, b_b, None, note_b,
                                                 write_bytes=4.0 * H * 3 * n + 4.0 * nh * 2 * E)
    return rec


def broadcast_mesh(gp, P, n, world, rank, dev):
    """rank 0: Delaunay mesh of n uniform points + the world-way node partition; every other rank receives pos, edge_index, edge_attr
    and the partition vector by broadcast (device tensors over RCCL, host tensors over gloo).  Returns (graph, part, seconds on rank 0)."""
    import numpy as np
    import torch.distributed as dist

    on_dev = dist.get_backend() == "nccl"
    t0 = time.perf_counter()
    if rank == 0:
        g = gp.square_mesh(n, seed=0)
        part = np.ascontiguousarray(P.partition_nodes(g.pos.numpy(), g.edge_index, world)).astype(np.int64)
        hdr = torch.tensor([g.pos.shape[0], g.pos.shape[1], g.edge_index.shape[1], g.edge_attr.shape[1]], dtype=torch.int64)
    else:
        g, part, hdr = None, None, torch.zeros(4, dtype=torch.int64)
    t_build = time.perf_counter() - t0

    def bc(t):
        if on_dev:
            t = t.to(dev)
        dist.broadcast(t, src=0)
        return t.cpu()

    hdr = bc(hdr)
    npos, dpos, E, fe = (int(v) for v in hdr)
    if rank == 0:
        pos, ei, ea, pt = g.pos.contiguous(), g.edge_index.contiguous(), g.edge_attr.contiguous(), torch.from_numpy(part)
    else:
        pos = torch.empty(npos, dpos, dtype=torch.float32)
        ei = torch.empty(2, E, dtype=torch.int64)
        ea = torch.empty(E, fe, dtype=torch.float32)
        pt = torch.empty(npos, dtype=torch.int64)
    pos, ei, ea, pt = bc(pos.float()), bc(ei), bc(ea.float()), bc(pt)
    if rank != 0:
        g = gp.Graph(pos=pos, edge_index=ei, edge_attr=ea)
    return g, pt.numpy(), t_build


def c5_dp_record(args, gp, D, ops, harness, rank, world, dev):
    """BASELINE configs[4] as the driver's N > 1 launch reaches it ("coarse-aneurysm Transformer, 4 x MI355X, bf16"): data-parallel
    replicas of EncodeTransformDecode in the bf16 matrix mode, every rank its own 3-D mesh (weak scaling), parameters broadcast from
    rank 0, gradients averaged by one flat all-reduce per step.  The reference has no multi-GPU form of its own (train.py:278,
    devices = 1).  Every rank enters; rank 0's dict is reported."""
    import numpy as np
    import torch.distributed as dist
    from scipy.spatial import Delaunay

    from graph_physics_amd import preprocess as PP
    from graph_physics_amd import transformer as T

    n = args.c5_nodes
    pts = np.random.default_rng(100 + rank).random((n, 3)).astype(np.float32)
    cells = torch.from_numpy(Delaunay(pts).simplices.T.astype(np.int64)).to(dev)
    ei = PP.faces_to_edges(cells, n)
    E = int(ei.shape[1])
    H, nh, L = 64, 4, 10
    cfg = {"model": {"type": "transformer", "message_passing_num": L, "hidden_size": H, "node_input_size": 14, "output_size": 3,
                     "edge_input_size": 0, "num_heads": nh, "use_rope_embeddings": False, "use_gated_attention": False},
           "training": {"use_temporal_block": False}}
    torch.manual_seed(0)
    net = gp.get_model(cfg).to(dev)
    D.broadcast_parameters(net)
    graph = gp.Graph(x=torch.randn(n, 23, device=dev), edge_index=ei, pos=torch.from_numpy(pts).to(dev))
    tgt = torch.randn(n, 3, device=dev)
    opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)
    sync = D.GradAllReduce()

    def train():
        loss = ((net(graph) - tgt) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        sync(net.parameters())
        opt.step()

    def timed(fn, k=3):
        fn()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        dist.barrier()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt) / k

    prev = ops.get_matrix_precision()
    try:
        ops.set_matrix_precision("bf16")
        tt = timed(train)
        ta = timed(lambda: sync(net.parameters()))
    finally:
        ops.set_matrix_precision(prev)
    return {"workload": f"{world} replicas, each a 3-D Delaunay tetrahedral mesh of N={n} (E={E} on rank 0), EncodeTransformDecode {L} blocks, hidden {H}, "
                        f"{nh} heads (training_config/coarse-aneurysm.json:11-22), bf16 matrix mode; BASELINE.json configs[4]",
            "parallelism": f"dp{world}: replicas, flat gradient all-reduce ({dist.get_backend()})", "dtype": "bf16 (matrix operands; fp32 accumulate)",
            "train_ms_per_step": round(1e3 * tt, 2), "node_train_steps_per_s": round(world * n / tt, 1),
            "allreduce_ms_per_step": round(1e3 * ta, 3), "scaling": "weak"}


def c4_record(args, gp, D, ops, harness, rank, world, dev):
    """BASELINE configs[3]: synthetic Delaunay mesh (1M nodes / 6M directed edges), latent 128, 15 rounds."""
    from graph_physics_amd import partition as P
    import torch.distributed as dist

    n = args.c4_nodes
    part_shared = None
    if world > 1:
        # ONE rank triangulates and partitions; the others receive the arrays (no reliance on eight processes producing
        # bit-identical scipy / numpy results, and 7/8 of the host time saved)
        g, part_shared, t_build = broadcast_mesh(gp, P, n, world, rank, dev)
    else:
        g = gp.square_mesh(n, seed=0)  # nodes in GENERATOR order: no locality
    E = int(g.edge_index.shape[1])
    rec = {"workload": f"Delaunay mesh of {n} uniform points in the unit square, seed 0, numbered in generator order (no locality), "
                       f"E={E}, {args.rounds} MP rounds, latent {args.hidden}, fp32; BASELINE.json configs[3].  The ENGINE renumbers the "
                       "nodes along a Morton curve of graph.pos on entry and un-does it on exit (ops.set_node_renumbering, default auto: "
                       "on from 200k nodes); its cost is inside topology_build_ms; *_raw_numbering = the same run with renumbering off"}
    torch.manual_seed(0)
    net = gp.EncodeProcessDecode(args.rounds, 11, 3, 2, hidden_size=args.hidden).to(dev)
    x_in = torch.randn(n, 11, generator=torch.Generator().manual_seed(1))
    tgt = torch.randn(n, 2, generator=torch.Generator().manual_seed(2))
    nt = torch.zeros(n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, k, warm=1):
        for _ in range(warm):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt) / k

    def partitioned(nparts, prank, exchange):
        t0 = time.perf_counter()
        part = part_shared if (exchange and part_shared is not None) else P.partition_nodes(g.pos.numpy(), g.edge_index, nparts)
        plan = P.build_rank_plan(g.edge_index, part, prank, nparts, pos=g.pos.numpy())  # local numbering along a Morton curve
        t_part = time.perf_counter() - t0
        if not exchange:
            plan.world = 1  # no process group: ghost rows are zero-filled (compute of one rank's share only)
        pm = D.PartitionedEPD(net, plan)
        xo, eo = x_in[plan.owned].to(dev), g.edge_attr[plan.edge_ids].to(dev)
        to, no = tgt[plan.owned].to(dev), nt[plan.owned].to(dev)
        opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)
        sync = D.GradAllReduce(average=False) if exchange else (lambda ps: None)

        def train_step():
            out = pm(xo, eo)
            loss = D.partitioned_loss(out, to, no) if exchange else harness.l2_loss(out, to, no)
            opt.zero_grad()
            loss.backward()
            sync(net.parameters())
            opt.step()

        def infer_step():
            with torch.no_grad():
                pm(xo, eo)

        def host_time(fn, k=3):
            """wall time the HOST needs to enqueue one step (Python glue, ctypes launches, c10d enqueue) with the device's queue
            never empty and never waited for: no synchronisation inside the timed region.  The partitioned step must stay well
            below the device time of a rank's share, or the 8-GPU step is host-bound whatever the kernels do."""
            fn()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for _ in range(k):
                fn()
            dt = (time.perf_counter() - t0) / k
            torch.cuda.synchronize()
            return dt

        torch.cuda.reset_peak_memory_stats(dev)
        t_train = timed(train_step, args.c4_steps)
        mem = torch.cuda.max_memory_allocated(dev) / 2**30
        t_inf = timed(infer_step, args.c4_steps)
        r = {"owned_nodes": plan.n_own, "ghost_rows": plan.n_ghost, "local_edges": int(plan.edge_ids.numel()),
             "edge_cut": round(P.edge_cut(g.edge_index, part), 5), "partition_s": round(t_part, 2),
             "train_ms_per_step": round(1e3 * t_train, 2), "rollout_ms_per_step": round(1e3 * t_inf, 2),
             "peak_mem_gib": round(mem, 1)}
        if world == 1 or dist.get_backend() == "nccl":   # (over gloo the collectives block the host: the figure would mean nothing)
            r["host_ms_per_step"] = round(1e3 * host_time(train_step), 2)
            r["host_rollout_ms_per_step"] = round(1e3 * host_time(infer_step), 2)
            r["host_what"] = ("host wall time to ENQUEUE one step (no synchronisation in the timed region): Python glue + ctypes launches"
                              + (" + c10d enqueue of the 2 x rounds halo exchanges and the gradient all-reduce" if exchange else "")
                              + "; must stay below train_ms_per_step for the step to be device-bound")
        if exchange:
            # the step's communication on its own (nothing to overlap with: an upper bound of what it costs inside the step):
            # the 2 x rounds halo exchanges of a step, and the gradient all-reduce
            halo = D.HaloState(plan, dev)
            buf = torch.zeros(plan.n_own + plan.n_ghost, args.hidden, device=dev)
            Ss = torch.zeros(plan.n_own + plan.n_ghost, args.hidden, device=dev)

            def halo_step():
                for _ in range(args.rounds):
                    halo.finish_forward(halo.start_forward(buf))
                for _ in range(args.rounds):
                    halo.finish_backward(halo.start_backward(Ss), Ss)

            r["halo_ms_per_step"] = round(1e3 * timed(halo_step, 3), 3)
            r["halo_what"] = (f"{2 * args.rounds} neighbour exchanges (pack, all_to_all_single, unpack-add) run back to back, nothing overlapped; "
                              f"{int(plan.send_idx.numel())} rows of {4 * args.hidden} B sent per exchange by this rank")
            r["allreduce_ms_per_step"] = round(1e3 * timed(lambda: sync(net.parameters()), 3), 3)
            r["allreduce_what"] = "flat all-reduce of every parameter gradient (11.5 MB), on its own"
            # one forward exchange on its own (event-timed on the device), for the top-level `collective` object
            r["halo_one_exchange_ms"] = round(1e3 * timed(lambda: halo.finish_forward(halo.start_forward(buf)), 5), 4)
            r["halo_one_exchange_bytes_sent"] = int(plan.send_idx.numel()) * 4 * args.hidden
            r["halo_one_exchange_bytes_received"] = int(plan.n_ghost) * 4 * args.hidden
            r["allreduce_bytes"] = int(sum(p_.numel() for p_ in net.parameters())) * 4
        return r

    if world > 1:
        D.broadcast_parameters(net)
        r = partitioned(world, rank, True)
        rec.update(r)
        rec["mesh_build_and_partition_s_rank0"] = round(t_build, 2)
        rec["parallelism"] = f"{world}-way node partition (dst-owner edges), one-hop halo exchange of ghost latents per round " \
                             f"({dist.get_backend()}), gradient all-reduce"
        rec["node_train_steps_per_s"] = round(n / (r["train_ms_per_step"] * 1e-3), 1)
        rec["rollout_node_steps_per_s"] = round(n / (r["rollout_ms_per_step"] * 1e-3), 1)
        return rec
    # ---- one GPU: (a) whole-mesh inference, (b) the scatter-add past the Infinity Cache, (c) one rank's share of 8
    graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=g.edge_index.to(dev), pos=g.pos.to(dev))

    def build_topo(renumber):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t = ops.Topology(graph.edge_index, n, renumber="morton" if renumber else None, pos=graph.pos if renumber else None)
        torch.cuda.synchronize()
        return t, 1e3 * (time.perf_counter() - t0)

    build_topo(True)  # warm-up (first rocPRIM sort of the process)
    topo_raw, ms_raw = build_topo(False)
    topo, ms_ren = build_topo(True)
    rec["topology_build_ms"] = round(ms_ren, 2)
    rec["topology_build_ms_raw_numbering"] = round(ms_raw, 2)
    rec["topology_build_what"] = ("Morton keys + radix sort of the nodes + relabelled edge list + CSR by destination and by source (one host "
                                  "synchronisation: the eager build; the model's own path builds it without one)")

    def fwd():
        with torch.no_grad():
            net(graph)

    graph.mgn_topology = topo_raw
    t_inf_raw = timed(fwd, args.c4_steps)
    rec["rollout_ms_per_step_raw_numbering"] = round(1e3 * t_inf_raw, 2)
    graph.mgn_topology = topo
    torch.cuda.reset_peak_memory_stats(dev)
    t_inf = timed(fwd, args.c4_steps)
    rec["rollout_ms_per_step"] = round(1e3 * t_inf, 2)
    rec["rollout_node_steps_per_s"] = round(n / t_inf, 1)
    rec["inference_peak_mem_gib"] = round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)
    del graph
    # (b) scatter-add at the C4 size: 6M message rows of 512 B = 3.07 GB read per sum, far past the 256 MiB L3
    H = args.hidden
    f = dict(dtype=torch.float32, device=dev)
    m = torch.randn(E, H, **f)
    agg, agg2 = torch.empty(n, H, **f), torch.empty(n, H, **f)

    def ev_time(fn, k=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k

    t_seg = ev_time(lambda: ops.segsum(m, topo.rowptr_dst, None, agg))
    b_seg = 4.0 * H * (E + n) + 4.0 * (n + 1)  # 4EH + 4NH + 4(N+1)
    from graph_physics_amd import _capi as _capi_mod
    rec["roofline_scatter"] = hbm_obj("k_segsum<8> AS A STAND-ALONE LAUNCH (forward scatter-add agg[i] = sum of the messages of node i's incoming "
                                      "edges, CSR order, atomics-free; 3.6 GB per launch, past the 256 MiB Infinity Cache; inside a step the forward "
                                      "aggregation is the edge kernel's fused epilogue and the backward scatters are roofline_scatter_backward)", t_seg, b_seg,
                                      _c4_scatter_traffic(_capi_mod), write_bytes=4.0 * H * n, traffic_rw=_c4_scatter_traffic(_capi_mod, rw=True))
    t_seg2 = ev_time(lambda: ops.segsum2(m, topo.rowptr_dst, None, agg, topo.rowptr_src, topo.perm_src, agg2))
    b_seg2 = 2 * 4.0 * H * (E + n) + 4.0 * E + 8.0 * (n + 1)
    rec["roofline_scatter_backward"] = hbm_obj("k_segsum2<8> (both backward scatters of dZ0 in one launch: onto destinations in CSR order and "
                                               "onto sources through perm_src = gathered 512-byte rows; engine-renumbered nodes)", t_seg2, b_seg2,
                                               write_bytes=2 * 4.0 * H * n)
    t_seg2r = ev_time(lambda: ops.segsum2(m, topo_raw.rowptr_dst, None, agg, topo_raw.rowptr_src, topo_raw.perm_src, agg2))
    rec["roofline_scatter_backward_raw_numbering"] = hbm_obj("k_segsum2<8>, nodes in generator order: the source-side rows are random 512-byte "
                                                             "gathers over 3 GB", t_seg2r, b_seg2)
    del m, agg, agg2, topo, topo_raw
    torch.cuda.empty_cache()
    # (b2) the whole-mesh TRAINING step on one GPU: the saves of 15 rounds (~280 GB) do not fit, so the
    # processor recomputes each round's activations inside the backward pass (ops.set_activation_recompute,
    # "auto" switches it on here) -- the N = 1 point of the strong-scaling curve
    graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=g.edge_index.to(dev), pos=g.pos.to(dev))
    tg, ntg = tgt.to(dev), nt.to(dev)
    opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)

    def whole_train_step(model=net, optim=opt):
        loss = harness.l2_loss(model(graph), tg, ntg)  # topology through ops.get_topology: lazy build, Morton renumbering (auto)
        optim.zero_grad()
        loss.backward()
        optim.step()

    torch.cuda.reset_peak_memory_stats(dev)
    n_rec = ops.recompute_rounds(E, n, H, 4, args.rounds, 0, dev)   # what "auto" decides with the memory free right now
    t_tr = timed(whole_train_step, max(2, args.c4_steps - 1))
    rec["train_ms_per_step"] = round(1e3 * t_tr, 2)
    rec["node_train_steps_per_s"] = round(n / t_tr, 1)
    rec["train_peak_mem_gib"] = round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)
    ops.set_node_renumbering("off")
    try:
        rec["train_ms_per_step_raw_numbering"] = round(1e3 * timed(whole_train_step, 2), 2)
    finally:
        ops.set_node_renumbering("auto")
    rec["train_activation_recompute"] = "%s: %d of %d rounds recomputed in the backward pass, the others saved (all saves would be %.0f GB)" % (
        ops.get_activation_recompute(), n_rec, args.rounds, ops.saved_activation_bytes(E, n, H, 4, args.rounds, 0) / 1e9)
    # What the same step costs WITHOUT the recompute (the N > 1 runs keep their saves): measured on a 5-round
    # model, whose saves fit, with recompute off and on; the ratio carries over (every round costs the same).
    # A 1 -> 8 speed-up read against train_ms_per_step contains the recompute switching off; read against
    # scaling_baseline_ms it does not.
    try:
        l5 = min(5, args.rounds)
        net5 = gp.EncodeProcessDecode(l5, 11, 3, 2, hidden_size=args.hidden).to(dev)
        opt5 = harness.FusedClipAdamW(net5.parameters(), 1e-4, max_norm=1.0)
        old_mode = ops.get_activation_recompute()
        t5 = {}
        for mode in ("off", "on"):
            ops.set_activation_recompute(mode)
            t5[mode] = timed(lambda: whole_train_step(net5, opt5), 2)
        ops.set_activation_recompute(old_mode)
        extra = (t5["on"] / t5["off"] - 1.0) * n_rec / max(args.rounds, 1)   # share of the step that is re-run forward work
        rec["scaling_baseline_ms"] = round(1e3 * t_tr / (1.0 + extra), 2)
        rec["scaling_baseline_note"] = (f"train_ms_per_step / (1 + {n_rec}/{args.rounds} x (recompute / no-recompute - 1)), the ratio measured on a "
                                        f"{l5}-round model of the same mesh ({1e3 * t5['on']:.1f} / {1e3 * t5['off']:.1f} ms): the single-GPU step "
                                        "as if all its saves fitted")
        del net5, opt5
    except Exception as ex:  # noqa: BLE001
        ops.set_activation_recompute("auto")
        rec["scaling_baseline_note"] = f"not measured ({type(ex).__name__}: {ex})"
    del graph, tg, ntg, opt
    torch.cuda.empty_cache()
    # (c) rank 0 of the 8-way partition, forward + backward + optimiser, ghost rows zero-filled
    share = partitioned(8, 0, False)
    rec["rank_share_of_8"] = dict(share, note="per-GPU compute of the 8-way partitioned step (no exchange on one GPU)",
                                  est_8gpu_node_train_steps_per_s_before_comms=round(n / (share["train_ms_per_step"] * 1e-3), 1))
    rec["note"] = ("N=1: train_ms_per_step is the whole mesh on one GPU with PARTIAL activation recompute (as many rounds saved as fit); the N>1 runs of this command carry "
                   "the partitioned step (saves kept when they fit: recompute switches itself off)")
    return rec


def main():
    args = parse()
    maybe_self_launch(args)
    import graph_physics_amd as gp
    from graph_physics_amd import _capi as capi
    from graph_physics_amd import distributed as D
    from graph_physics_amd import harness, ops

    rank, world, local = D.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = gp.cylinder_config(args.rounds, args.hidden)
    cfg["training"]["enable_vram_optimizations"] = (args.precision == "bf16")
    torch.manual_seed(0)
    eng = harness.Engine(cfg, dev, learning_rate=1e-4, num_steps=10000, warmup=100)
    if world > 1:
        D.broadcast_parameters(eng.sim)
        eng.grad_sync = D.OverlappedGradAllReduce(params=list(eng.sim.parameters()))   # 4 MB buckets reduced while the backward pass runs
    batch = gp.cylinder_batch(args.batch, args.nodes, seed0=rank * args.batch).to(dev)
    N, E = batch.x.shape[0], batch.edge_index.shape[1]
    # topology prep (CSR by dst + by src): timed on its own, outside the headline's timed region --
    # the topology of a fixed batch is cached; `topology` below reports what a rebuild per step costs
    ops.Topology(batch.edge_index, N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        batch.mgn_topology = ops.Topology(batch.edge_index, N)
    torch.cuda.synchronize()
    topo_ms = 1e3 * (time.perf_counter() - t0) / 5

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(step, k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # single GPU: the whole step (fwd, loss, bwd, clip, AdamW) is captured once in a hipGraph and
    # replayed; multi GPU keeps eager launches (the RCCL all-reduce sits between bwd and clip)
    use_graph = (world == 1) and (args.graph == "on" or (args.graph == "auto" and E <= 65536))
    graph_note = "eager"
    if use_graph:
        try:
            eng.capture_train_step(batch, warmup=max(2, min(args.warmup, 3)))
            step = lambda: eng.train_step_graphed(None)  # noqa: E731
            graph_note = "hipGraph replay"
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] hipGraph capture failed ({type(ex).__name__}: {ex}); falling back to eager launches", file=sys.stderr)
            use_graph = False
    if not use_graph:
        step = lambda: eng.train_step(batch)  # noqa: E731
    for _ in range(args.warmup):
        step()
    dt = timed(step, args.steps)
    steps_per_s = world * args.steps / dt

    # the reference's shuffled loader hands over a NEW 16-mesh union every step: same step with the
    # topology rebuilt inside it (mgn_topology_build_async: queued without a host synchronisation)
    def step_rebuild():
        batch.mgn_topology = ops.Topology(batch.edge_index, N, lazy=True)  # queued: no host synchronisation (flags read after the forward)
        eng.train_step(batch)

    k_rb = max(3, args.steps // 3)
    step_rebuild()
    dt_rb = timed(step_rebuild, k_rb)

    # rollout (second half of the metric): one mesh batch advanced autoregressively
    frames = [batch] * args.rollout_steps
    rollout, rollout_note = (lambda fr: eng.rollout(fr, graph="off")), "eager"
    # single process only: with a live RCCL process group its watchdog thread polls events, which is
    # not allowed while another thread captures; at batch 16 replay and eager rollout time the same
    if args.graph != "off" and world == 1:
        try:
            eng.capture_rollout_step(batch)
            rollout, rollout_note = eng.rollout_graphed, "hipGraph replay"
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] hipGraph capture of the rollout step failed ({type(ex).__name__}: {ex}); eager launches", file=sys.stderr)
    rollout(frames[:3])
    dt_r = timed(lambda: rollout(frames), 1)
    rollout_nps = world * N * args.rollout_steps / dt_r

    out = None
    if rank == 0:
        out = {
            "metric": "training steps/sec, CylinderFlow 15-round MGN (batch 16 meshes per step)",
            "value": round(steps_per_s, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "matrix_path": ("bf16 operands (one term), fp32 accumulate / RMSNorm / residuals" if args.precision == "bf16" else
                            "bf16x3 split operands, 6-term products on v_mfma_f32_16x16x32_bf16, fp32 accumulate "
                            "(fp32-grade accuracy: forward parity 1e-5 vs the CPU oracle)" if ops.X6_ENABLED else "fp32 MFMA"),
            "config": {"workload": f"BASELINE.json configs[1], {'fp32' if args.precision == 'fp32' else 'bf16'}: {args.batch} x {args.nodes}-node CylinderFlow-like meshes "
                       f"per GPU (N={N}, E={E}), {args.rounds} rounds, latent {args.hidden}, random init", "global_batch_meshes": args.batch * world, "launch": graph_note,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "rollout_node_steps_per_s": round(rollout_nps, 1),
            "rollout_ms_per_step": round(1e3 * dt_r / args.rollout_steps, 3), "rollout_launch": rollout_note,
            "topology": {"topology_build_ms": round(topo_ms, 3),
                         "what": "ops.Topology of the batch (eager build incl. its one host synchronisation): CSR by destination + by source, "
                                 "outside the headline's timed region (a fixed batch re-uses it); rebuild_every_step queues the build "
                                 "without a host synchronisation (lazy flags, read once the forward pass is queued)",
                         "steps_per_s_rebuild_every_step": round(world * k_rb / dt_rb, 3),
                         "ms_per_step_rebuild_every_step": round(1e3 * dt_rb / k_rb, 3)},
        }
        out["step_floor"] = step_floor(N, E, args.hidden, args.rounds, 1 if args.precision == "bf16" else 6, 1e3 * dt / args.steps)
        if not args.no_kernel_timing:
            device_copy_rate(dev)
            roof, roof_seg, others = kernel_rooflines(gp, ops, capi, eng, batch, dev)
            out["roofline"], out["roofline_scatter"], out["roofline_other_kernels"] = roof, roof_seg, others
    # free the configs[1] state before the 1M-node record
    if world > 1:
        eng.grad_sync.close()   # the partitioned model below sums its own gradients: stop listening to backward passes
    del eng, batch, frames, rollout, step
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_kernel_timing and not args.no_extras:
        try:
            out["batch1"] = batch1_record(args, gp, ops, harness, dev)
        except Exception as ex:  # noqa: BLE001  (an extra record must not cost the headline)
            out["batch1"] = {"error": f"{type(ex).__name__}: {ex}"}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_kernel_timing and not args.no_extras and args.precision == "fp32":
        try:
            out["shipped_cylinder_json"] = shipped_cylinder_record(args, gp, ops, harness, dev)
        except Exception as ex:  # noqa: BLE001
            out["shipped_cylinder_json"] = {"error": f"{type(ex).__name__}: {ex}"}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_kernel_timing and not args.no_extras and args.precision == "fp32":
        try:
            out["plate_bf16"] = plate_bf16_record(args, gp, ops, harness, dev)
        except Exception as ex:  # noqa: BLE001
            out["plate_bf16"] = {"error": f"{type(ex).__name__}: {ex}"}
            ops.set_matrix_precision("fp32")
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_kernel_timing and not args.no_c4 and not args.no_extras and args.precision == "fp32":
        try:
            out["c5"] = c5_record(args, gp, ops, harness, dev)
        except Exception as ex:  # noqa: BLE001
            out["c5"] = {"error": f"{type(ex).__name__}: {ex}"}
            ops.set_matrix_precision("fp32")
        torch.cuda.empty_cache()
    if not args.no_c4:
        # N > 1: the partitioned record runs collectives no single-GPU box has exercised over RCCL.  The HEADLINE is measured by
        # now and must reach stdout whatever happens here: a rank that raises may leave its peers inside a collective, so every rank
        # arms a watchdog that -- after a deadline no healthy run comes near -- has rank 0 print the line without the record and
        # ends the process (exit code 0 on every rank: nothing is left that could synchronise them).
        wd = None
        if world > 1:
            import threading
            wd_done = threading.Event()

            def bail(why):
                # a plain process exit (never a re-exec: this process has touched the GPU).  Rank 0 prints the headline it has measured
                # and leaves with 0; every other rank leaves NON-ZERO -- the launcher's return code then shows that the partitioned
                # record failed -- and ten seconds later, so that the launcher's clean-up of its peers cannot cut off rank 0's line
                if rank == 0:
                    out["c4"] = {"error": why}
                    print(json.dumps(out), flush=True)
                    os._exit(0)
                time.sleep(10.0)
                os._exit(3)

            def watchdog():
                if not wd_done.wait(float(os.environ.get("MGN_BENCH_C4_DEADLINE_S", "480"))):
                    bail("the partitioned 1M-node record did not finish before the deadline (a rank stuck in a collective?)")

            wd = threading.Thread(target=watchdog, daemon=True)
            wd.start()
        c5dp = None
        try:
            if world > 1 and not args.no_extras and args.precision == "fp32":
                c5dp = c5_dp_record(args, gp, D, ops, harness, rank, world, dev)
                torch.cuda.empty_cache()
            c4 = c4_record(args, gp, D, ops, harness, rank, world, dev)
        except Exception as ex:  # noqa: BLE001  (the headline must survive a failure of the extra record)
            c4 = {"error": f"{type(ex).__name__}: {ex}"}
            if world > 1:   # the peers may be waiting for this rank inside a collective: no barrier can be trusted any more
                print(f"[bench] rank {rank}: c4 record failed: {c4['error']}", file=sys.stderr, flush=True)
                if rank == 0:
                    bail(c4["error"])
                wd_done.wait()   # never set: this rank idles until its own watchdog ends it (rank 0 prints meanwhile)
        if wd is not None:
            wd_done.set()
        if rank == 0:
            out["c4"] = c4
            if world > 1 and isinstance(c4, dict):
                import torch.distributed as dist_
                try:
                    rccl_v = ".".join(str(v_) for v_ in torch.cuda.nccl.version())
                except Exception:  # noqa: BLE001
                    rccl_v = None
                out["collective"] = {
                    "backend": dist_.get_backend(), "world_size_seen_by_process_group": dist_.get_world_size(), "rccl_version": rccl_v,
                    "halo_exchange": {"what": "ONE forward exchange of ghost-node latents of the partitioned 1M-node mesh (pack, all_to_all_single, "
                                              "unpack), rank 0, on its own, HIP events",
                                      "bytes_sent_rank0": c4.get("halo_one_exchange_bytes_sent"), "bytes_received_rank0": c4.get("halo_one_exchange_bytes_received"),
                                      "ms": c4.get("halo_one_exchange_ms"), "exchanges_per_training_step": 2 * args.rounds},
                    "gradient_all_reduce": {"what": "flat all-reduce of every parameter gradient, on its own, HIP events",
                                            "bytes": c4.get("allreduce_bytes"), "ms": c4.get("allreduce_ms_per_step")},
                    "headline_gradient_sync": "bucketed all-reduce inside the backward pass (OverlappedGradAllReduce: three collectives of 3-4 MB per step on "
                                              "a side stream)"}
            # the north-star scatter-add figure (>= 40 % of the HBM roofline past the Infinity Cache) as a top-level key: a reader of
            # the parsed line need not open the nested record
            if isinstance(c4, dict) and "roofline_scatter" in c4:
                out["roofline_scatter_c4"] = c4["roofline_scatter"]
                if "roofline_scatter_backward" in c4:   # ... and the launch a training step really runs at this size (k_segsum2)
                    out["roofline_scatter_c4_in_step"] = c4["roofline_scatter_backward"]
            if c5dp is not None:
                out["c5"] = c5dp
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, gp)
            cb = out["cpu_baseline"]
            cpu_best_mesh_steps = max(cb["value"] * args.batch, cb["batch1_steps_per_s"])  # the CPU's best regime (batch 1: its caches hold the mesh)
            out["speedup_vs_cpu_baseline"] = round(out["value"] * args.batch / cpu_best_mesh_steps, 1)
            out["speedup_vs_cpu_same_batch"] = round(out["value"] / cb["value"], 1)
            out["speedup_note"] = ("speedup_vs_cpu_baseline: mesh-steps/s of the GPU at batch 16 against the CPU oracle's BEST regime (mesh-steps/s "
                                   "at batch 1 or 16, whichever is higher) -- the conservative ratio; speedup_vs_cpu_same_batch: the same 16-mesh "
                                   "batch on both sides; batch1.speedup_vs_cpu_same_mesh: one mesh per step on both sides")
            if "train_steps_per_s" in out.get("batch1", {}):
                out["batch1"]["speedup_vs_cpu_same_mesh"] = round(out["batch1"]["train_steps_per_s"] / out["cpu_baseline"]["batch1_steps_per_s"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
