"""Multi-GPU plumbing (one process per GPU, ``torch.distributed``; backend "nccl"
is RCCL on ROCm, "gloo" in the CPU tests).

Small meshes (CylinderFlow) are data-parallel replicas: every rank steps its own
batch and the only exchange is ONE flat all-reduce of the 2.87 M-parameter
gradient (11.5 MB fp32) per step -- a single bucket, because over point-to-point
xGMI a ring all-reduce is per-link latency bound at this size and splitting it
only adds hops.  The partitioned large-mesh path lives in ``partition.py``.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, world, local_rank); initialises the default group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MGN_DIST_BACKEND=gloo: rehearsal of the multi-process path on a box with fewer GPUs than ranks
            backend = os.environ.get("MGN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if os.environ.get("MGN_SHARE_GPU") and torch.cuda.is_available():
        local = local % torch.cuda.device_count()  # rehearsal only: several ranks on one device (not with RCCL)
    return rank, world, local


class GradAllReduce:
    """callable(params): average ``.grad`` over the group through one flat buffer."""

    def __init__(self, group=None, average: bool = True):
        self.group = group
        self.average = average  # False: sum (partitioned mesh, the loss is already global)
        self._flat: Optional[torch.Tensor] = None

    def __call__(self, params: Iterable[torch.nn.Parameter]):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        ps = [p for p in params if p.grad is not None]
        n = sum(p.grad.numel() for p in ps)
        if self._flat is None or self._flat.numel() != n or self._flat.device != ps[0].grad.device:
            self._flat = torch.empty(n, dtype=torch.float32, device=ps[0].grad.device)
        torch._foreach_copy_(list(torch.split(self._flat, [p.grad.numel() for p in ps])),
                             [p.grad.reshape(-1) for p in ps])
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            self._flat.div_(dist.get_world_size(self.group))
        # one multi-tensor launch back (296 single copies cost ~0.6 ms of the step on the GPU)
        sizes = [p.grad.numel() for p in ps]
        torch._foreach_copy_([p.grad for p in ps],
                             [v.view_as(p.grad) for v, p in zip(torch.split(self._flat, sizes), ps)])


class OverlappedGradAllReduce(GradAllReduce):
    """Data-parallel gradient averaging that starts DURING the backward pass: ``ops.ProcessorFunction.backward`` reports every
    round's weight gradients as soon as their launches are queued (last round first); they are copied into a flat bucket and
    every ``bucket_bytes`` one asynchronous all-reduce goes out (RCCL: on its own stream, behind the bucket copy) while the
    earlier rounds are still being differentiated.  ``__call__(params)`` after ``backward()`` waits for the buckets, writes the
    averages into ``.grad`` and reduces what never passed the hook (encoders, decoder, RMSNorm scales) in one more flat
    all-reduce.  Same sums as :class:`GradAllReduce` element by element (bit-identical at world 2; beyond, the collective's order of
    additions depends on an element's place in its buffer: equal to rounding); the bucket sequence depends on tensor sizes only, so
    every rank issues the same collectives in the same order.  The listener is process-global (``ops.set_grad_ready_hook``): a
    backward pass that must NOT take part -- a rank stepping alone, another model -- needs ``close()`` first.  xGMI is point-to-point and a
    ring all-reduce of 9 MB is latency-bound: 4 MB buckets = three collectives per step for the 15-round model.
    Assumes one backward pass per step into empty gradients (``zero_grad(set_to_none=True)``, as ``harness.Engine`` does): the
    bucketed values REPLACE ``.grad``, they are not added to an accumulated one -- a parameter reported twice before ``__call__``
    (gradient accumulation, two losses, a step that aborted without ``reset()``) raises instead of keeping the last pass only.
    ``params``: the parameters of the wrapped model; gradients of any other tensor that passes the process-global hook (another
    model's backward pass in the same process) are ignored.  Without it every reported gradient is taken."""

    def __init__(self, group=None, bucket_bytes: int = 4 << 20, params: Optional[Iterable[torch.nn.Parameter]] = None):
        super().__init__(group, True)
        self.bucket_bytes = bucket_bytes
        self._cur, self._cur_bytes, self._inflight = [], 0, []
        self._seen = set()     # data_ptrs reported since the last __call__ / reset
        self._mine = None if params is None else {p.data_ptr() for p in params}
        from . import ops
        ops.set_grad_ready_hook(self._on_ready)

    def reset(self):
        """drop what a step that did not reach ``__call__`` has left behind (pending collectives are waited for first: every
        rank issued them, none may be abandoned half-way)"""
        for work, _flat, _ents in self._inflight:
            work.wait()
        self._cur, self._cur_bytes, self._inflight = [], 0, []
        self._seen = set()

    def close(self):
        """stop listening (before another model -- e.g. the partitioned one -- runs its backward pass)"""
        from . import ops
        ops.set_grad_ready_hook(None)
        self.reset()

    def _active(self) -> bool:
        return dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _on_ready(self, pairs):
        if not self._active():
            return
        for p, g in pairs:
            ptr = p.data_ptr()
            if self._mine is not None and ptr not in self._mine:
                continue   # not a parameter of the wrapped model
            if ptr in self._seen:
                raise RuntimeError("OverlappedGradAllReduce: a parameter's gradient was reported twice before the step's "
                                   "all-reduce (gradient accumulation / several backward passes per step are not supported: "
                                   "use GradAllReduce, or call reset() after an aborted step)")
            self._seen.add(ptr)
            self._cur.append((ptr, g))
            self._cur_bytes += 4 * g.numel()
        if self._cur_bytes >= self.bucket_bytes:
            self._flush()

    def _flush(self):
        if not self._cur:
            return
        flat = torch.cat([g.reshape(-1) for _, g in self._cur])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((work, flat, [(ptr, g.numel()) for ptr, g in self._cur]))
        self._cur, self._cur_bytes = [], 0

    def __call__(self, params: Iterable[torch.nn.Parameter]):
        if not self._active():
            return
        self._flush()
        world = dist.get_world_size(self.group)
        done = {}
        for work, flat, ents in self._inflight:
            work.wait()
            flat.div_(world)
            off = 0
            for ptr, n in ents:
                done[ptr] = flat[off:off + n]
                off += n
        self._inflight = []
        self._seen = set()
        ps = [p for p in params if p.grad is not None]
        stray = set(done) - {p.data_ptr() for p in ps}
        if stray:
            raise RuntimeError(f"OverlappedGradAllReduce: {len(stray)} bucketed gradient(s) belong to no parameter of this step "
                               "(another model's backward pass fed the hook: pass params= at construction)")
        early = [p for p in ps if p.data_ptr() in done]
        if early:
            torch._foreach_copy_([p.grad for p in early], [done[p.data_ptr()].view_as(p.grad) for p in early])
        rest = [p for p in ps if p.data_ptr() not in done]
        if rest:
            super().__call__(rest)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Make every replica start from rank ``src``'s weights and normaliser buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


# ----------------------------------------------------------------------------------
# Partitioned large-mesh path: one-hop halo exchange of ghost-node latents per round.
# ----------------------------------------------------------------------------------
class HaloState:
    """Device side of a :class:`partition.RankPlan` for the HIP engine (ops.ProcessorFunction drives it).

    Forward, once per round: the rows of ``Ps`` (next round's source projection x W_s^T -- the only
    remote data an edge row needs) that peers use as ghost sources are packed by ``mgn_gather_rows``
    and travel by ONE ``all_to_all_single`` straight into the ghost rows of the same ``[n_own +
    n_ghost, H]`` buffer the edge kernel gathers from (no concatenation); the collective is started
    right after the node kernel and completed only before the BOUNDARY edge rows -- the interior rows
    (destinations without a ghost in-neighbour) run in between.  Backward: the ghost rows of the
    source-side scatter ``Ss`` go back the same way (contiguous: no packing) while the E-row weight
    gradients run, and ``mgn_halo_unpack_add`` sums them into their owners in a fixed order -- no
    atomics anywhere, so partitioned gradients are bit-reproducible.

    Over RCCL (backend "nccl") ``async_op=True`` puts the collective on RCCL's own stream: it waits for
    the packed rows, the compute stream waits for it only at ``finish_*``.  The exchange is a
    neighbour all-to-all of 0.4-0.7 MB per rank and round on the 1M-node mesh: latency-bound over
    point-to-point xGMI, hence one collective per round rather than a ring."""

    def __init__(self, plan, device, group=None):
        self.plan, self.group, self.device = plan, group, device
        self.n_own, self.n_ghost = plan.n_own, plan.n_ghost
        self.n_interior, self.n_interior_edges = plan.n_interior, plan.n_interior_edges
        i32 = dict(dtype=torch.int32, device=device)
        self.send_idx = plan.send_idx.to(**i32)
        self.send_nodes = plan.send_nodes.to(**i32)
        self.send_rowptr = plan.send_rowptr.to(**i32)
        self.send_perm = plan.send_perm.to(**i32)
        self.send_counts, self.recv_counts = list(plan.send_counts), list(plan.recv_counts)
        # a collective is needed whenever rows are exchanged and a process group exists (a world-1 group is legal:
        # the self-exchange plan of tests/test_halo_rccl_single.py runs the whole path over RCCL on one GPU)
        # (all_to_all_single is a collective of the WHOLE group: a rank without boundary nodes -- a mesh component of
        # its own -- still has to enter it, with empty splits, or its peers wait for ever)
        self.active = dist.is_initialized() and dist.get_world_size(group) == plan.world and \
            (plan.world > 1 or (getattr(plan, "self_exchange", False) and (self.n_ghost > 0 or int(plan.send_idx.numel()) > 0)))
        self._rowptr_bnd = None
        self._back = None  # receive buffer of the backward exchange (one per state: 30 exchanges a step re-use it)

    def rowptr_bnd(self, topo):
        """rowptr_dst shifted to the first boundary edge row: the boundary launch numbers its rows from 0"""
        if self._rowptr_bnd is None:
            self._rowptr_bnd = (topo.rowptr_dst - self.n_interior_edges).contiguous()
        return self._rowptr_bnd

    # ---- forward: owners' rows -> ghost rows of the same buffer
    def start_forward(self, buf: torch.Tensor):
        from . import ops
        if not self.active:  # a single process holding a multi-rank plan (timing rehearsal): ghosts read as zeros
            if self.n_ghost > 0:
                buf[self.n_own:].zero_()
            return None
        send = ops.gather_rows(buf, self.send_idx)
        work = dist.all_to_all_single(buf[self.n_own:], send, output_split_sizes=self.recv_counts,
                                      input_split_sizes=self.send_counts, group=self.group, async_op=True)
        return (work, send)  # `send` stays referenced until the collective has consumed it

    def finish_forward(self, pending):
        if pending is not None:
            pending[0].wait()

    # ---- backward: ghost-row gradients -> summed into their owners
    def start_backward(self, Ss: torch.Tensor):
        if not self.active:
            return None
        # allocated once: the unpack of round i has consumed it (same stream) before round i-1's exchange is
        # started after it, and the collective itself is ordered behind the compute stream
        if self._back is None or self._back.shape[1] != Ss.shape[1] or self._back.dtype != Ss.dtype:
            self._back = torch.empty(self.send_idx.numel(), Ss.shape[1], dtype=Ss.dtype, device=Ss.device)
        back = self._back
        work = dist.all_to_all_single(back, Ss[self.n_own:], output_split_sizes=self.send_counts,
                                      input_split_sizes=self.recv_counts, group=self.group, async_op=True)
        return (work, back)

    def finish_backward(self, pending, Ss: torch.Tensor):
        from . import ops
        if pending is None:
            return
        pending[0].wait()
        ops.halo_unpack_add(pending[1], self.send_nodes, self.send_rowptr, self.send_perm, Ss)


class HaloExchange(torch.autograd.Function):
    """Generic (backend-agnostic) exchange of latent rows: ghost rows <- owners' rows (forward);
    ghost-row gradients summed back into the owners' rows (backward, in the fixed order of the plan's
    send list grouped by node).  Used by compute backends that need the ghost LATENTS themselves (the
    CPU oracle backend of the tests, widths the packed kernels do not take); the HIP engine exchanges
    projected rows inside ``ops.ProcessorFunction`` instead (:class:`HaloState`)."""

    @staticmethod
    def forward(ctx, x_own, plan, group):
        ctx.plan, ctx.group, ctx.n_own = plan, group, x_own.shape[0]
        idx = plan.send_idx.to(x_own.device)
        ctx.local = plan.world == 1 or not dist.is_initialized()
        send = x_own.detach().index_select(0, idx).contiguous()
        recv = torch.empty(plan.n_ghost, x_own.shape[1], dtype=x_own.dtype, device=x_own.device)
        if ctx.local:  # single process: nothing to exchange (a world-1 plan has no ghosts)
            return recv.zero_()
        dist.all_to_all_single(recv, send, output_split_sizes=plan.recv_counts, input_split_sizes=plan.send_counts, group=group)
        return recv

    @staticmethod
    def backward(ctx, d_ghost):
        plan = ctx.plan
        d_ghost = d_ghost.contiguous()
        d_own = torch.zeros(ctx.n_own, d_ghost.shape[1], dtype=d_ghost.dtype, device=d_ghost.device)
        if ctx.local:
            return d_own, None, None
        back = torch.empty(plan.send_idx.numel(), d_ghost.shape[1], dtype=d_ghost.dtype, device=d_ghost.device)
        dist.all_to_all_single(back, d_ghost, output_split_sizes=plan.send_counts, input_split_sizes=plan.recv_counts, group=ctx.group)
        # fixed order (no index_add_ atomics): rows grouped by node, ascending position inside a group
        dev = d_ghost.device
        grouped = back.index_select(0, plan.send_perm.to(dev))
        rp = plan.send_rowptr
        nodes = plan.send_nodes.to(dev)
        maxc = int((rp[1:] - rp[:-1]).max()) if nodes.numel() else 0
        start = rp[:-1].to(dev)
        cnt = (rp[1:] - rp[:-1]).to(dev)
        for k in range(maxc):  # k-th contribution of every node that has one: a plain indexed add per k
            sel = cnt > k
            d_own[nodes[sel]] += grouped[start[sel] + k]
        return d_own, None, None


class HipBackend:
    """Compute backend of the partitioned model: the HIP engine (default, GPU only)."""

    fused = True  # all rounds + the halo exchange inside ONE autograd node (ops.ProcessorFunction)

    def prepare(self, edge_index, n_local):
        from . import ops
        return ops.Topology(edge_index, n_local)

    def mlp(self, module, x):
        return module(x)

    def order_edges(self, edge_attr, ctx):
        return edge_attr[ctx.perm_dst_long]

    def block(self, block, x, e, ctx, pos=None, phi=None):
        from . import ops
        from .layers import _block_params
        if block.use_gated_mlp:   # gated-MLP blocks run on the dense kernels (gated.py), on the same dst-sorted edge rows
            from .gated import gated_block_forward
            return gated_block_forward(block, x, e, ctx, pos if block.use_rope else None, phi if block.use_gate else None)
        return ops.processor_apply(x, e, ctx, 1, *_block_params(block), spec=block.spec,
                                   pos=pos if block.use_rope else None, phi=phi if block.use_gate else None,
                                   rope_inv_freq=block._rope_inv_freq if block.use_rope else None)


class PartitionedEPD(torch.nn.Module):
    """EncodeProcessDecode over a node-partitioned mesh.  Each rank holds its owned nodes,
    the ghost sources of its edges, and the edges whose destination it owns; results on the
    owned nodes equal the un-partitioned forward / backward (the parity oracle of section 8e).

    forward(x_in_own[n_own,F_n], edge_attr_loc[E_loc,F_e]) -> out_own[n_own,O]
    """

    def __init__(self, model, plan, group=None, backend=None, cache_positions: bool = False, temporal_plan=None):
        super().__init__()
        self.model, self.plan, self.group = model, plan, group
        self.backend = backend if backend is not None else HipBackend()
        self.temporal_plan = temporal_plan
        self._ttopo, self._t_of, self._t_inv = None, None, None
        self._ctx = None
        self._halo = None
        # [n_own + n_ghost, D]: owned positions + the ghosts'.  The ghosts' rows are exchanged on EVERY forward (one small
        # neighbour exchange beside the one-per-block exchanges of the latents) unless ``cache_positions=True`` -- the caller's
        # promise that the mesh does not deform and the same sample is evaluated: then they travel once (``invalidate_positions()``
        # drops them).  The decision must be the same on every rank (the exchange is a collective), so it is a constructor
        # argument and never inferred from the tensor a rank happens to pass.  Owned rows always come from the current pos_own.
        self.cache_positions = bool(cache_positions)
        self._pos_full = None
        self._pos_ghost = None
        # the temporal block (processors.py:183-184, :203-209) attends along ROWS of spmatrix(indices=edge_index): row i = the edges
        # whose SOURCE is i, columns = their destinations -- the other end of the edges this rank holds (it holds the edges whose
        # DESTINATION it owns).  So it needs a second plan, built on the flipped edge list with the same partition vector:
        #     temporal_plan = partition.build_rank_plan(edge_index.flip(0), part, rank, world)
        # (its "destination" is the attention row, its ghosts the columns this rank does not own; on a mesh whose edge list holds both
        # directions of every edge the two plans have the same ghosts).  Without DGL the reference hands the block no adjacency and
        # it is row-wise (attention_backend == "pyg"): no plan needed.
        self._temporal = bool(getattr(model, "use_temporal_block", False)) and getattr(model, "temporal_block", None) is not None
        if self._temporal and getattr(model, "attention_backend", "dgl") == "dgl":
            if temporal_plan is None:
                raise ValueError("PartitionedEPD: a model with use_temporal_block needs temporal_plan = "
                                 "build_rank_plan(edge_index.flip(0), part, rank, world)")
            o1, o2 = plan.owned.cpu(), temporal_plan.owned.cpu()
            s1, s2 = torch.argsort(o1), torch.argsort(o2)
            if o1.numel() != o2.numel() or not torch.equal(o1[s1], o2[s2]):
                raise ValueError("PartitionedEPD: temporal_plan must be built with the same partition vector as plan")
            # both plans list the owned nodes interior-first, and "interior" depends on the direction: position of each of
            # temporal_plan's owned nodes in plan's numbering (None: the same order, every mesh with both directions of each edge)
            t_of = torch.empty_like(s1)
            t_of[s2] = s1
            self._t_of = None if torch.equal(t_of, torch.arange(t_of.numel())) else t_of
            self._t_inv = None if self._t_of is None else torch.argsort(t_of)

    def invalidate_positions(self) -> None:
        """drop the cached ghost positions (``cache_positions=True``): the next forward exchanges them again -- call it on
        EVERY rank (the exchange is a collective)"""
        self._pos_full = None
        self._pos_ghost = None

    def forward(self, x_in_own: torch.Tensor, edge_attr_loc: torch.Tensor, phi_own: Optional[torch.Tensor] = None,
                pos_own: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``phi_own``: the owned rows of ``graph.phi`` (only read by blocks with the sigmoid gate, layers.py:1091-1098).
        ``pos_own``: the owned rows of ``graph.pos`` -- required with ``use_rope_embeddings`` (layers.py:1020-1026 rotates
        x_src by pos[src] - pos[dst]: a rank needs the positions of its ghost sources; they travel ONCE per plan, by the same
        neighbour exchange as the latents, on every call unless the module was built with ``cache_positions=True``)."""
        plan, be, m = self.plan, self.backend, self.model
        dev = x_in_own.device
        if phi_own is not None and phi_own.reshape(-1).shape[0] != plan.n_own:
            raise ValueError("phi_own must hold one value per owned node")
        use_rope = bool(getattr(m, "use_rope", False))
        if use_rope:
            if pos_own is None:
                raise ValueError("Graph data must contain `pos` when use_rope_embeddings=True.")   # processors.py:188-191
            if pos_own.shape[0] != plan.n_own:
                raise ValueError("pos_own must hold one row per owned node")
            p_own = pos_own.detach().to(dev, torch.float32).contiguous()
            # cache_positions: only the GHOST rows are cached; every forward builds a fresh tensor (a tensor written in place would trip
            # autograd's version check -- or silently show newer positions to a RoPE node that kept the raw pointer -- when two
            # forwards precede one backward: gradient accumulation)
            if (not self.cache_positions or self._pos_ghost is None or self._pos_ghost.device != dev
                    or self._pos_ghost.shape[1] != p_own.shape[1]):
                self._pos_ghost = HaloExchange.apply(p_own, plan, self.group)
            self._pos_full = torch.cat([p_own, self._pos_ghost], dim=0)
        if self._ctx is None:
            self._ctx = be.prepare(plan.edge_index.to(dev), plan.n_own + plan.n_ghost)
        x_own = be.mlp(m.nodes_encoder, x_in_own)
        e = be.mlp(m.edges_encoder, be.order_edges(edge_attr_loc, self._ctx))
        blocks = list(m.processor_list)
        fused = getattr(be, "fused", False) and blocks and m.hidden_size == 128 and not m.use_rope and not m.use_gated_mlp
        prev_own = x_own
        if fused:
            from . import ops
            from .layers import _block_params
            if self._halo is None:
                self._halo = HaloState(plan, dev, self.group)
            # the temporal block needs the node latents BEFORE the last round too (processors.py:193-209): two autograd nodes then
            groups = [blocks] if not (self._temporal and len(blocks) > 1) else [blocks[:-1], blocks[-1:]]
            for grp in groups:
                prev_own = x_own
                params = []
                for blk in grp:
                    params += _block_params(blk)
                x_own, e = ops.processor_apply(x_own, e, self._ctx, len(grp), *params, spec=grp[0].spec, halo=self._halo,
                                               phi=phi_own)
        else:
            # per-block path (RoPE, gated-MLP blocks, widths off the packed kernels): the ghost LATENTS are exchanged before
            # every block (HaloExchange: differentiable, fixed-order backward), the block runs on owned + ghost rows and the
            # ghost rows of its output -- nodes without their in-edges here -- are dropped.  phi: the gate reads it per
            # node, so the ghosts' values never matter (zeros).
            phi_full = None
            if phi_own is not None:
                phi_full = torch.cat([phi_own.reshape(-1).to(dev, torch.float32), torch.zeros(plan.n_ghost, device=dev)])
            for blk in blocks:
                prev_own = x_own
                x_gh = HaloExchange.apply(x_own, plan, self.group)
                x_full = torch.cat([x_own, x_gh], dim=0)
                x_full, e = be.block(blk, x_full, e, self._ctx, pos=self._pos_full if use_rope else None, phi=phi_full)
                x_own = x_full[: plan.n_own]
        if self._temporal:
            x_own = self._temporal_tail(prev_own, x_own, dev)
        return be.mlp(m.decode_module, x_own)

    def _temporal_tail(self, prev_own: torch.Tensor, last_own: torch.Tensor, dev) -> torch.Tensor:
        """processors.py:203-209: x = temporal_block(prev_x, last_x, adj) on the owned rows.  Row i attends over the edges whose
        source is i (``temporal_plan``: the flipped edge list); k comes from prev_x, q / v from last_x, so the ghosts of BOTH
        travel -- as one exchange of [prev | last] rows; the ghosts' own output rows are dropped."""
        m, be, tb = self.model, self.backend, self.model.temporal_block
        if getattr(m, "attention_backend", "dgl") != "dgl":      # no DGL: no adjacency, the block is row-wise
            return tb(prev_own, last_own, None) if isinstance(be, HipBackend) else be.temporal_block(tb, prev_own, last_own, None)
        tp = self.temporal_plan
        if self._t_of is not None:
            idx = self._t_of.to(dev)
            prev_own, last_own = prev_own.index_select(0, idx), last_own.index_select(0, idx)
        h = prev_own.shape[1]
        both = torch.cat([prev_own, last_own], dim=1)
        both = torch.cat([both, HaloExchange.apply(both, tp, self.group)], dim=0)
        prev_full, last_full = both[:, :h].contiguous(), both[:, h:].contiguous()
        ei = tp.edge_index.flip(0).contiguous().to(dev)          # (row = owned source, column)
        if isinstance(be, HipBackend):
            from .transformer import get_attn_topology
            if self._ttopo is None:
                self._ttopo = get_attn_topology(ei, tp.n_own + tp.n_ghost)
            y = tb(prev_full, last_full, self._ttopo)
        else:
            y = be.temporal_block(tb, prev_full, last_full, ei)
        y = y[: tp.n_own]
        return y if self._t_of is None else y.index_select(0, self._t_inv.to(dev))


class PartitionedETD(torch.nn.Module):
    """EncodeTransformDecode (processors.py:218-384, the sparse-attention branch) over a node-partitioned mesh [r5].

    A Transformer block's attention row i = sum over the edges with ``edge_index[0] == i`` of softmax_i(q_i . k_j) v_j,
    j = ``edge_index[1]`` (layers.py:493-559: rows of ``dglsp.spmatrix(indices=edge_index)``), so a rank must hold every edge
    whose ROW it owns and the latents of the columns it does not own: build the plan on the FLIPPED edge list,

        plan = partition.build_rank_plan(edge_index.flip(0), part, rank, world)

    (its "destination" is then the attention row).  Per block the ghosts' latents are exchanged (``HaloExchange``:
    differentiable, fixed-order backward), the block runs on owned + ghost rows -- the ghosts' k / v come out of their
    exchanged latents, their own outputs (rows without their edges here) are dropped.  RoPE: the ghosts' positions travel as in
    ``PartitionedEPD`` (every call unless ``cache_positions``).  Results on the owned rows equal the un-partitioned forward /
    backward (tests: gloo world 4 against the oracle on the CPU, the HIP engine with all ranks on one device).
    The temporal block (processors.py:376-377) runs the same way after the last block: one more exchange for the ghosts of ``last_x``.

    forward(x_in_own[n_own, F_n], pos_own=None) -> out_own[n_own, O]"""

    def __init__(self, model, plan, group=None, backend=None, cache_positions: bool = False):
        super().__init__()
        self.model, self.plan, self.group = model, plan, group
        self.backend = backend    # None: the HIP engine (the model's own Transformer modules)
        self.cache_positions = bool(cache_positions)
        self._pos_full = None
        self._pos_ghost = None
        self._topo = None
        if getattr(model, "attention_backend", "dgl") != "dgl":
            raise NotImplementedError("PartitionedETD: only the sparse-attention branch (attention_backend='dgl') is partitioned")

    def invalidate_positions(self) -> None:
        self._pos_full = None
        self._pos_ghost = None

    def local_edge_index(self) -> torch.Tensor:
        """the rank's edges as (row, column) of the attention in local numbering: rows are owned nodes"""
        return self.plan.edge_index.flip(0).contiguous()

    def forward(self, x_in_own: torch.Tensor, pos_own: Optional[torch.Tensor] = None) -> torch.Tensor:
        plan, m = self.plan, self.model
        dev = x_in_own.device
        use_rope = bool(getattr(m, "use_rope_embeddings", False))
        pos_full = None
        if use_rope:
            if pos_own is None:
                raise ValueError("use_rope_embeddings=True requires 'pos' attribute in the input graph.")   # processors.py:340-343
            if pos_own.shape[0] != plan.n_own:
                raise ValueError("pos_own must hold one row per owned node")
            p_own = pos_own.detach().to(dev, torch.float32).contiguous()
            # cache_positions: only the GHOST rows are cached; every forward builds a fresh tensor (a tensor written in place would trip
            # autograd's version check -- or silently show newer positions to a RoPE node that kept the raw pointer -- when two
            # forwards precede one backward: gradient accumulation)
            if (not self.cache_positions or self._pos_ghost is None or self._pos_ghost.device != dev
                    or self._pos_ghost.shape[1] != p_own.shape[1]):
                self._pos_ghost = HaloExchange.apply(p_own, plan, self.group)
            self._pos_full = torch.cat([p_own, self._pos_ghost], dim=0)
            pos_full = self._pos_full
        ei = self.local_edge_index().to(dev)
        n_loc = plan.n_own + plan.n_ghost
        be = self.backend
        if be is None:
            from .transformer import get_attn_topology
            if self._topo is None:
                self._topo = get_attn_topology(ei, n_loc)
            x_own = x_in_own if m.only_processor else m.nodes_encoder(x_in_own)
        else:
            x_own = x_in_own if m.only_processor else be.mlp(m.nodes_encoder, x_in_own)
        x_full = None
        for blk in m.processor_list:
            x_full = torch.cat([x_own, HaloExchange.apply(x_own, plan, self.group)], dim=0)
            y = blk(x_full, self._topo, pos=pos_full) if be is None else be.transformer_block(blk, x_full, ei, pos_full)
            x_own = y[: plan.n_own]
        if getattr(m, "use_temporal_block", False) and m.temporal_block is not None:
            # processors.py:376-377: temporal_block(prev_x, last_x, adj) -- row i reads q_i from last_x and k_j (of prev_x), v_j (of
            # last_x) over the same adjacency as the blocks.  prev_x with its ghost rows is the last block's input (already
            # exchanged); the ghosts of last_x travel in ONE more neighbour exchange; the ghosts' own output rows are dropped.
            last_full = torch.cat([x_own, HaloExchange.apply(x_own, plan, self.group)], dim=0)
            prev_full = x_full if x_full is not None else last_full          # no block: prev_x = last_x = the encoder's output
            tb = m.temporal_block
            y = tb(prev_full, last_full, self._topo) if be is None else be.temporal_block(tb, prev_full, last_full, ei)
            x_own = y[: plan.n_own]
        if m.only_processor:
            return x_own
        return m.decode_module(x_own) if be is None else be.mlp(m.decode_module, x_own)


def partitioned_loss(net_out_own, target_own, node_type_own, group=None):
    """Masked L2 over ALL ranks' NORMAL|OUTFLOW nodes (same value on every rank); its local
    gradient, SUMMED over ranks (GradAllReduce(average=False)), is the global gradient."""
    from .nodetype import NodeType
    mask = (node_type_own == int(NodeType.NORMAL)) | (node_type_own == int(NodeType.OUTFLOW))
    w = mask.to(net_out_own.dtype).unsqueeze(1)
    num = ((net_out_own - target_own) ** 2 * w).sum()
    cnt = (w.sum() * net_out_own.shape[1]).detach()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(cnt, group=group)
    return num / cnt
