"""Training-step and rollout loops that reproduce the semantics of the reference's
LightningModule for the MeshGraphNet path without Lightning (not installed on the
GPU box):

  * training step  (training/lightning_module.py:270-320): Simulator forward,
    masked L2 on NORMAL|OUTFLOW nodes (utils/loss.py:37-75, default masks
    lightning_module.py:48), backward, clip-grad-norm 1.0 (train.py:288),
    AdamW(lr, betas=(0.9,0.95), weight_decay=1e-4) + cosine warm-up
    (lightning_module.py:494-511, utils/scheduler.py:51-67);
  * rollout  (lightning_module.py:375-409): autoregressive feedback of the last
    prediction, ground truth re-imposed on the non NORMAL/OUTFLOW nodes.

These are what ``bench.py`` times ("training steps/sec", "rollout nodes*steps/sec").
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence

import torch

from .mesh import Graph
from .nodetype import NodeType
from .parse_parameters import get_model, get_simulator, matrix_precision_from_config
from . import ops as _ops


def lr_factor(last_epoch: int, warmup: int, max_iters: int, min_lr_factor: float = 0.001) -> float:
    epoch = last_epoch + 1
    f = 0.5 * (1 + math.cos(math.pi * epoch / max_iters))
    if epoch <= warmup:
        f *= epoch * 1.0 / warmup
    return max(f, min_lr_factor)


class _MaskedMseFn(torch.autograd.Function):
    """the loss below as two engine launches forward and one backward (``mgn_masked_mse_fwd`` / ``_bwd``) instead of torch's ~20
    elementwise / reduction launches -- 1-2 % of the batch-16 step, 4 % of the one-mesh step"""

    @staticmethod
    def forward(ctx, out, tgt, node_type, types):
        import ctypes as C
        from . import _capi
        out = out.float() if out.dtype != torch.float32 else out
        tgt = tgt.float() if tgt.dtype != torch.float32 else tgt
        node_type = node_type.float() if node_type.dtype != torch.float32 else node_type
        if out.stride(-1) != 1:
            out = out.contiguous()
        if tgt.stride(-1) != 1:
            tgt = tgt.contiguous()
        N, O = int(out.shape[0]), int(out.shape[1])
        if tgt.shape != out.shape or int(node_type.shape[0]) != N:
            raise ValueError("l2_loss: output, target and node_type must describe the same rows")
        dev = out.device
        scratch = torch.empty(514, dtype=torch.float32, device=dev)   # 512 partials | unused | 1 / (rows x O) for the backward
        loss, inv = torch.empty((), dtype=torch.float32, device=dev), scratch[513]
        tarr = (C.c_float * len(types))(*[float(t) for t in types])
        ldty = int(node_type.stride(0)) if N > 0 else 1
        with torch.cuda.device(dev):
            rc = _capi.lib().mgn_masked_mse_fwd(out.data_ptr(), int(out.stride(0)), tgt.data_ptr(), int(tgt.stride(0)), node_type.data_ptr(), ldty,
                                                N, O, tarr, len(types), scratch.data_ptr(), loss.data_ptr(), inv.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(rc, "mgn_masked_mse_fwd", prep=True)
        ctx.save_for_backward(out, tgt, node_type, scratch)
        ctx.types = tuple(float(t) for t in types)
        return loss

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        from . import _capi
        out, tgt, node_type, scratch = ctx.saved_tensors
        N, O = int(out.shape[0]), int(out.shape[1])
        dev = out.device
        g = g.float().reshape(1).contiguous()
        d_out = torch.empty(N, O, dtype=torch.float32, device=dev)
        tarr = (C.c_float * len(ctx.types))(*ctx.types)
        ldty = int(node_type.stride(0)) if N > 0 else 1
        with torch.cuda.device(dev):
            rc = _capi.lib().mgn_masked_mse_bwd(out.data_ptr(), int(out.stride(0)), tgt.data_ptr(), int(tgt.stride(0)), node_type.data_ptr(), ldty,
                                                N, O, tarr, len(ctx.types), scratch[513].data_ptr(), g.data_ptr(), d_out.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(rc, "mgn_masked_mse_bwd", prep=True)
        return d_out, None, None, None


def l2_loss(network_output: torch.Tensor, target: torch.Tensor, node_type: torch.Tensor,
            masks: Sequence[int] = (NodeType.NORMAL, NodeType.OUTFLOW)) -> torch.Tensor:
    """mean squared error over the rows whose node type is in ``masks`` (graphphysics/training/loss.py:70-75 with the masks of
    lightning_module.py:27-35); on the device one fused engine call, on the CPU (host-logic tests) the plain torch formula"""
    import os as _os
    if network_output.is_cuda and network_output.dim() == 2 and node_type.dim() == 1 and 1 <= len(masks) <= 4 and \
            _os.environ.get("MGN_TORCH_LOSS") is None:
        return _MaskedMseFn.apply(network_output, target, node_type, tuple(int(t) for t in masks))
    mask = node_type == int(masks[0])
    for t in masks[1:]:
        mask = torch.logical_or(mask, node_type == int(t))
    # mean over the selected rows' elements, without a data-dependent shape (no host sync)
    w = mask.to(network_output.dtype).unsqueeze(1)
    err = (network_output - target) ** 2 * w
    return err.sum() / (w.sum() * network_output.shape[1])


def build_mask(node_type: torch.Tensor) -> torch.Tensor:
    keep = torch.logical_or(node_type == int(NodeType.NORMAL), node_type == int(NodeType.OUTFLOW))
    return torch.logical_not(keep)


class FusedClipAdamW:
    """clip_grad_norm_(max_norm) + AdamW(lr, betas, eps, weight_decay) of the reference's training
    step (train.py:288, lightning_module.py:494-511) as ONE engine call (``mgn_clip_adamw``: two
    launches per 96 tensors instead of ~25 multi-tensor launches).  ``lr`` and the step counter are
    device scalars, so the same object works under hipGraph replay.  After ``step()`` the ``.grad``
    tensors hold the clipped gradients, as after ``clip_grad_norm_``."""

    def __init__(self, params, lr: float, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 1e-4,
                 max_norm: float = 1.0):
        from . import _capi
        self._capi = _capi
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        self.device = dev
        self.betas, self.eps, self.weight_decay, self.max_norm = betas, eps, weight_decay, max_norm
        self.exp_avg = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.step_t = torch.zeros((), dtype=torch.float32, device=dev)
        self.lr_t = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        self.norm_t = torch.zeros((), dtype=torch.float32, device=dev)
        self.param_groups = [{"lr": float(lr)}]
        self._ws = None
        self._table, self._table_key = None, None   # device table of the tensors' fixed fields (mgn_clip_adamw_table)

    def set_lr(self, lr: float):
        self.param_groups[0]["lr"] = float(lr)
        self.lr_t.fill_(float(lr))

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        C = self._capi
        live = [(p, m, v) for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq) if p.grad is not None]
        arr = (C.OptTensor * len(live))()
        for i, (p, m, v) in enumerate(live):
            if not p.grad.is_contiguous():  # e.g. the un-padded slice of an encoder's first-layer gradient
                p.grad = p.grad.contiguous()
            if not p.data.is_contiguous():
                raise RuntimeError("FusedClipAdamW needs contiguous parameters")
            arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].n = p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        L = C.lib()
        need = L.mgn_clip_adamw_workspace_bytes(len(live), arr)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        # two launches whatever the number of tensors: parameter / moment pointers and lengths sit in a device table built once per
        # parameter set (a blocking copy: the first, eager step builds it -- before any capture), the gradient pointers travel as
        # kernel arguments (MGN_OPT_NO_TABLE: the 96-tensors-per-launch form, for A/B; same bits)
        import os as _os
        key = tuple((arr[i].p, arr[i].m, arr[i].v, arr[i].n) for i in range(len(live)))
        tb = L.mgn_clip_adamw_table_bytes(len(live), arr) if _os.environ.get("MGN_OPT_NO_TABLE") is None else 0
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            if tb > 0:
                if self._table_key != key:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("FusedClipAdamW: the parameter table must be built by an eager step before the capture")
                    self._table = torch.empty(tb, dtype=torch.uint8, device=self.device)
                    C.check(L.mgn_clip_adamw_table(len(live), arr, self._table.data_ptr(), tb), "mgn_clip_adamw_table", prep=True)
                    self._table_key = key
                rc = L.mgn_clip_adamw_t(len(live), arr, self._table.data_ptr(), float(self.max_norm), self.lr_t.data_ptr(),
                                        self.step_t.data_ptr(), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                        float(self.weight_decay), self.norm_t.data_ptr(), self._ws.data_ptr(), self._ws.numel(), stream)
            else:
                rc = L.mgn_clip_adamw(len(live), arr, float(self.max_norm), self.lr_t.data_ptr(), self.step_t.data_ptr(),
                                      float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                                      self.norm_t.data_ptr(), self._ws.data_ptr(), self._ws.numel(), stream)
        C.check(rc, "mgn_clip_adamw", prep=True)


class Engine:
    """Model + simulator + optimiser for one device."""

    def __init__(self, param: Dict[str, Any], device: torch.device, learning_rate: float = 1e-4,
                 num_steps: int = 1000, warmup: int = 100, grad_clip: float = 1.0):
        self.param, self.device = param, device
        _ops.set_matrix_precision(matrix_precision_from_config(param))  # bf16-mixed <=> enable_vram_optimizations
        self.model = get_model(param)
        self.sim = get_simulator(param, self.model, device)
        self.learning_rate, self.num_steps, self.warmup, self.grad_clip = learning_rate, num_steps, warmup, grad_clip
        self.opt = self._make_optimizer(learning_rate, capturable=False)
        self.step_count = 0
        self.grad_sync = None  # set by distributed.DataParallel: callable(params) all-reducing .grad
        self._graph = None     # hipGraph of one training step (capture_train_step)

    def _make_optimizer(self, lr, capturable: bool):
        """AdamW(lr, betas=(0.9,0.95), weight_decay=1e-4) as the reference configures it
        (lightning_module.py:494-511); the fused multi-tensor implementation when the device
        supports it (one kernel instead of ~15 foreach launches over 296 tensors)."""
        kw = dict(lr=lr, weight_decay=0.0001, betas=(0.9, 0.95))
        params = list(self.sim.parameters())
        import os as _os
        if params and params[0].is_cuda and all(p.dtype == torch.float32 for p in params) and _os.environ.get("MGN_TORCH_ADAMW") is None:
            if getattr(self, "opt", None) is not None and isinstance(self.opt, FusedClipAdamW):
                return self.opt  # already device-resident state: nothing to rebuild for graph capture
            return FusedClipAdamW(params, float(lr) if not torch.is_tensor(lr) else float(lr.item()), betas=kw["betas"],
                                  weight_decay=kw["weight_decay"], max_norm=self.grad_clip)
        if params and params[0].is_cuda:
            try:
                return torch.optim.AdamW(params, fused=True, capturable=capturable, **kw)
            except (RuntimeError, ValueError, TypeError):
                pass
        return torch.optim.AdamW(params, capturable=capturable, **kw) if capturable else torch.optim.AdamW(params, **kw)

    def train_step(self, batch: Graph) -> torch.Tensor:
        self.sim.train()
        lr_now = self.learning_rate * lr_factor(self.step_count, self.warmup, self.num_steps)
        fused = isinstance(self.opt, FusedClipAdamW)
        if fused:
            self.opt.set_lr(lr_now)
        else:
            for g in self.opt.param_groups:
                g["lr"] = lr_now
        node_type = batch.x[:, self.sim.node_type_index]
        net_out, target, _ = self.sim(batch)
        loss = l2_loss(net_out, target, node_type)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync(self.sim.parameters())
        if fused:  # clip + AdamW in one engine call
            self.opt.step()
            self.last_grad_norm = self.opt.norm_t
        else:
            self.last_grad_norm = torch.nn.utils.clip_grad_norm_(self.sim.parameters(), self.grad_clip)
            self.opt.step()
        self.step_count += 1
        return loss.detach()

    # ---------------------------------------------------------------- hipGraph replay
    def capture_train_step(self, batch: Graph, warmup: int = 3):
        """Capture one whole training step (forward, loss, backward, clip, AdamW) in a hipGraph.

        A step is ~450 dependent launches; driven from Python that is ~10 ms of host time --
        launch-bound for a 2k-node mesh.  The engine's kernels are launched on torch's current
        stream, allocate nothing and never synchronise, so the step captures as is; replay costs
        ~1 us per node.  The mesh topology and tensor shapes are frozen in the graph: capture
        once per mesh/batch shape, feed new data through ``train_step_graphed`` (copied into the
        static input tensors).  The learning rate lives in a device tensor updated before replay.
        """
        dev = self.device
        assert self.grad_sync is None, "graph capture of the multi-GPU step is not supported yet"
        # >= 1 eager step first: the optimiser creates its state lazily, and state created under
        # capture would be re-zeroed by every replay
        warmup = max(1, warmup)
        self.sim.train()
        from .layers import Normalizer
        for mod in self.sim.modules():  # host mirrors of the accumulation counters: no .item() under capture
            if isinstance(mod, Normalizer) and mod._host_num_acc is None:
                mod._host_num_acc = int(mod._num_accumulations.item())
        if isinstance(self.opt, FusedClipAdamW):
            self._lr_t = self.opt.lr_t  # lr and step already live on the device
        else:
            self._lr_t = torch.tensor(self.learning_rate, dtype=torch.float32, device=dev)
            self.opt = self._make_optimizer(self._lr_t, capturable=True)
        self._static = batch.clone()
        if getattr(batch, "mgn_topology", None) is not None:
            self._static.mgn_topology = batch.mgn_topology
        else:
            from . import ops
            self._static.mgn_topology = ops.Topology(self._static.edge_index, self._static.x.shape[0])

        def body():
            node_type = self._static.x[:, self.sim.node_type_index]
            net_out, target, _ = self.sim(self._static)
            loss = l2_loss(net_out, target, node_type)
            self.opt.zero_grad(set_to_none=True)
            loss.backward()
            if isinstance(self.opt, FusedClipAdamW):
                self.opt.step()
                gn = self.opt.norm_t
            else:
                gn = torch.nn.utils.clip_grad_norm_(self.sim.parameters(), self.grad_clip)
                self.opt.step()
            return loss.detach(), gn

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._lr_t.fill_(self.learning_rate * lr_factor(self.step_count, self.warmup, self.num_steps))
                body()
                self.step_count += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._g_loss, self._g_gn = body()
        self._graph = g
        return g

    def train_step_graphed(self, batch: Optional[Graph] = None) -> torch.Tensor:
        """Replay the captured step; ``batch`` (same shapes / topology) is copied into the static inputs."""
        if batch is not None and batch is not self._static:
            self._static.x.copy_(batch.x, non_blocking=True)
            self._static.y.copy_(batch.y, non_blocking=True)
            self._static.edge_attr.copy_(batch.edge_attr, non_blocking=True)
        self._lr_t.fill_(self.learning_rate * lr_factor(self.step_count, self.warmup, self.num_steps))
        self._graph.replay()
        self.step_count += 1
        self.last_grad_norm = self._g_gn
        return self._g_loss

    @torch.no_grad()
    def predict_step(self, batch: Graph, last_prediction: Optional[torch.Tensor]) -> torch.Tensor:
        self.sim.eval()
        batch = batch.clone()
        i0, i1 = self.sim.output_index_start, self.sim.output_index_end
        if last_prediction is not None:
            batch.x[:, i0:i1] = last_prediction
        graph, _ = self.sim._build_input_graph(batch, False)
        return self.sim.predict(batch, self.sim.model(graph), mask_truth=True)

    # ---------------------------------------------------------------- hipGraph rollout
    @torch.no_grad()
    def capture_rollout_step(self, frame: Graph, warmup: int = 2):
        """Capture one autoregressive rollout step (lightning_module.py:375-409: feed the last
        prediction back, Simulator forward in eval mode, re-impose the ground truth on the non
        NORMAL/OUTFLOW nodes) in a hipGraph.  A step is ~60 launches of 5-250 us; driven from
        Python the gaps between them cost ~25 % of the step.  Topology and shapes are frozen;
        ``rollout_graphed`` copies each frame's x / y / edge_attr into the static tensors."""
        dev = self.device
        self.sim.eval()
        st = frame.clone()
        if getattr(frame, "mgn_topology", None) is not None:
            st.mgn_topology = frame.mgn_topology
        else:
            from . import ops
            st.mgn_topology = ops.Topology(st.edge_index, st.x.shape[0])
        i0, i1 = self.sim.output_index_start, self.sim.output_index_end
        last = st.x[:, i0:i1].clone()

        def body():
            b = Graph(x=st.x.clone(), y=st.y, pos=st.pos, edge_attr=st.edge_attr, edge_index=st.edge_index)
            b.mgn_topology = st.mgn_topology
            b.x[:, i0:i1] = last
            graph, _ = self.sim._build_input_graph(b, False)
            pred = self.sim.predict(b, self.sim.model(graph), mask_truth=True)
            last.copy_(pred)
            return pred

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._r_pred = body()
        self._r_graph, self._r_static, self._r_last = g, st, last
        self._r_seen = {}
        return g

    @torch.no_grad()
    def rollout_graphed(self, frames: Sequence[Graph]) -> List[torch.Tensor]:
        """``rollout`` through the captured step (same shapes / topology for every frame)."""
        st, i0, i1 = self._r_static, self.sim.output_index_start, self.sim.output_index_end
        out = []
        # a field whose source is the very tensor OBJECT copied last, at the same version, is not copied again: the mesh's edge
        # features -- and, for a repeated frame, everything -- stay in place (3 launches of ~5 us each per step otherwise).  The
        # object is kept referenced: an address alone may be recycled by the allocator for another frame's data.
        seen = getattr(self, "_r_seen", None)
        if seen is None:
            seen = self._r_seen = {}

        def put(name, dst, src, may_skip):
            # (the skip is for the MESH's fields only -- edge features, positions: tensors nobody refills in place; x / y are copied
            # every frame: the engine's own kernels write through raw pointers without bumping _version)
            last = seen.get(name)
            if src is dst or (may_skip and last is not None and last[0] is src and last[1] == src._version):
                return
            dst.copy_(src, non_blocking=True)
            seen[name] = (src, src._version)

        for k, fr in enumerate(frames):
            if fr is not st:
                put("x", st.x, fr.x, False)
                put("y", st.y, fr.y, False)
                put("edge_attr", st.edge_attr, fr.edge_attr, True)
                # a deforming mesh: positions are per frame (relative RoPE reads them inside the captured step)
                if st.pos is not None and getattr(fr, "pos", None) is not None:
                    put("pos", st.pos, fr.pos, True)
            if k == 0:
                self._r_last.copy_(fr.x[:, i0:i1])
            self._r_graph.replay()
            out.append(self._r_pred.clone())
        self._r_seen = {}   # (no frame tensor stays referenced by the engine after the call)
        return out

    #: ``rollout(..., graph="auto")`` replays a captured step when the trajectory has at least this many frames on ONE mesh of at
    #: most AUTO_GRAPH_MAX_EDGES edges (small meshes are launch-bound driven from Python: the shipped cylinder.json, 5 rounds of
    #: latent 32 on a 16-mesh batch, runs 0.88 ms per step eager and 0.45 ms replayed; large meshes gain nothing and a capture holds a
    #: private copy of the step's activations)
    AUTO_GRAPH_MIN_FRAMES = 4
    AUTO_GRAPH_MAX_EDGES = 1_000_000

    def _rollout_graph_key(self, frames: Sequence[Graph]):
        f0 = frames[0]
        ei = f0.edge_index
        if not (torch.is_tensor(ei) and ei.is_cuda):
            return None
        for fr in frames:
            if fr.edge_index is not ei and (fr.edge_index.data_ptr() != ei.data_ptr() or fr.edge_index.shape != ei.shape):
                return None
            if fr.x.shape != f0.x.shape or fr.edge_attr.shape != f0.edge_attr.shape or fr.y.shape != f0.y.shape:
                return None
        from . import ops
        has_pos = getattr(f0, "pos", None) is not None
        for fr in frames:   # positions present for all frames or for none (the captured step is built for one of the two)
            if (getattr(fr, "pos", None) is not None) != has_pos:
                return None
        return (ei.data_ptr(), ei._version, tuple(ei.shape), tuple(f0.x.shape), tuple(f0.edge_attr.shape), tuple(f0.y.shape), has_pos,
                ops.get_matrix_precision())

    @torch.no_grad()
    def rollout(self, frames: Sequence[Graph], graph: str = "auto") -> List[torch.Tensor]:
        """Autoregressive rollout over ``frames`` (lightning_module.py:375-409).  ``graph``: "off" -- every step driven from Python;
        "on" -- one step captured in a hipGraph and replayed (the frames must share one mesh and one shape); "auto" (default) --
        replay when they do, the trajectory is long enough to pay for the capture and the mesh is small enough to be
        launch-bound; same results either way (tests/test_hip_parity.py: graphed == eager)."""
        if graph not in ("auto", "on", "off"):
            raise ValueError("graph must be 'auto', 'on' or 'off'")
        import torch.distributed as _dist
        # (a live RCCL process group's watchdog thread polls events, which is not allowed while another thread captures)
        rccl = _dist.is_available() and _dist.is_initialized() and _dist.get_backend() == "nccl"
        if graph != "off" and len(frames) > 0 and self.grad_sync is None and not (rccl and graph == "auto"):
            key = self._rollout_graph_key(frames)
            small = key is not None and len(frames) >= self.AUTO_GRAPH_MIN_FRAMES and key[2][1] <= self.AUTO_GRAPH_MAX_EDGES
            if graph == "on" and key is None:
                raise ValueError("rollout(graph='on') needs frames of one shape on one edge_index tensor")
            if key is not None and (graph == "on" or small):
                # the keyed edge_index is kept REFERENCED and compared by identity: an address alone may be recycled by the allocator
                # for another trajectory's topology of the same size
                if (getattr(self, "_r_key", None) != key or getattr(self, "_r_graph", None) is None
                        or getattr(self, "_r_ei", None) is not frames[0].edge_index):
                    self.capture_rollout_step(frames[0])
                    self._r_key = key
                    self._r_ei = frames[0].edge_index
                return self.rollout_graphed(frames)
        last, out = None, []
        for fr in frames:
            last = self.predict_step(fr, last)
            out.append(last)
        return out
