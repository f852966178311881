"""Host side of the engine: thin wrappers over the C ABI (include/mgn_hip.h) and
the ``torch.autograd.Function``s that make ``loss.backward()``, gradient clipping
and AdamW work unchanged on top of the HIP kernels.

PyTorch is plumbing here: it owns device memory and the stream; every FLOP of the
message-passing path runs in ``csrc/mgn_kernels.hip``.  There is no CPU path: a
CPU tensor or a missing library raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _capi

EPS = 1e-8  # RMSNorm epsilon of the reference (layers.py:80)
#: matrix path of the H = 128 kernels: split-bf16 (bf16x3 operands, 6 MFMA terms, fp32-grade
#: accuracy at 2.67x the fp32 MFMA rate) unless MGN_FP32_MFMA is set (exact-fp32 MFMA kernels)
import os as _os
X6_ENABLED = _os.environ.get("MGN_FP32_MFMA") is None

#: "fp32" (default: bf16x3 operands, 6 product terms, fp32-grade) or "bf16" (operands rounded to
#: bf16, one term, fp32 accumulate; the residual streams, RMSNorm and the node-row saves stay fp32, the saved edge activations and
#: the edge chain's dZ rows travel as two-byte rows [r4]) -- the processor's GEMMs only.
#: "bf16" mirrors the reference under Lightning ``precision="bf16-mixed"`` (train.py:74-78,268-293).
_matrix_precision = "fp32"


def set_matrix_precision(p: str) -> None:
    global _matrix_precision
    if p not in ("fp32", "bf16"):
        raise ValueError("matrix precision must be 'fp32' or 'bf16'")
    if p == "bf16" and not X6_ENABLED:
        raise RuntimeError("bf16 matrix mode runs on the split-bf16 kernels; unset MGN_FP32_MFMA")
    _matrix_precision = p


def get_matrix_precision() -> str:
    return _matrix_precision
SUPPORTED_H = (16, 32, 64, 128)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev: torch.device) -> int:
    """the hipStream_t of torch's current stream on ``dev`` (what every launch of the engine is queued on).  The raw query is one C call;
    building a ``torch.cuda.Stream`` object for it costs ~5 us, 60-90 times per training step"""
    if _raw_stream is not None:
        idx = dev.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(dev).cuda_stream


def _require_device(*ts: torch.Tensor):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "graph_physics_amd runs on MI355X only: got a CPU tensor. Move the model and the "
                "graph to the GPU (there is no CPU fallback).")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def pad16(n: int) -> int:
    return (n + 15) & ~15


# --------------------------------------------------------------------- topology
#: node renumbering of EncodeProcessDecode.forward: "off", "on" (Morton order of graph.pos when the graph has
#: positions, otherwise reverse Cuthill-McKee of the edge list, computed on the host), or "auto" (default): on
#: from RENUMBER_MIN_NODES nodes -- below that every node row the edge kernels gather is L2-resident whatever the
#: numbering.  The reference has no counterpart (it gathers x[col], x[row] in the dataset's numbering,
#: layers.py:1017-1018); results are un-permuted on exit.  The forward aggregation still sums a node's incoming
#: messages in ascending ORIGINAL edge id (the CSR build is a stable sort of edge ids by destination, whatever the
#: destination is called), so it equals the un-renumbered result up to the association of partial sums at tile
#: boundaries; the source-side scatter of the backward pass walks the dst-sorted rows, whose order does change.
_renumber_mode = _os.environ.get("MGN_RENUMBER", "auto")
if _renumber_mode not in ("off", "on", "auto"):
    raise ValueError(f"MGN_RENUMBER={_renumber_mode!r}: expected 'off', 'on' or 'auto'")
RENUMBER_MIN_NODES = 200_000


def set_node_renumbering(mode: str) -> None:
    global _renumber_mode
    if mode not in ("off", "on", "auto"):
        raise ValueError("node renumbering mode must be 'off', 'on' or 'auto'")
    _renumber_mode = mode


def get_node_renumbering() -> str:
    return _renumber_mode


def want_renumbering(num_nodes: int) -> bool:
    return _renumber_mode == "on" or (_renumber_mode == "auto" and num_nodes >= RENUMBER_MIN_NODES)


def morton_order(pos: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(order[N] int32: old id at new position i, rank[N] int32: new position of old id) along a Morton curve of
    the first 2 or 3 columns of ``pos`` -- on the device, no host synchronisation (mgn_morton_order)."""
    _require_device(pos)
    pos = _f32c(pos)
    if pos.dim() != 2 or pos.shape[1] < 2:
        raise ValueError("pos must be [N, >= 2]")
    N, D = int(pos.shape[0]), min(int(pos.shape[1]), 3)
    dev = pos.device
    L = _capi.lib()
    order = torch.empty(N, dtype=torch.int32, device=dev)
    rank = torch.empty(N, dtype=torch.int32, device=dev)
    ws = torch.empty(max(L.mgn_morton_order_workspace_bytes(N), 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.mgn_morton_order(_ptr(pos), int(pos.shape[1]), D, N, _ptr(order), _ptr(rank), _ptr(ws), ws.numel(), _stream(dev))
    _capi.check(rc, "mgn_morton_order", prep=True)
    return order, rank


def rcm_order(edge_index: torch.Tensor, num_nodes: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(order, rank) by reverse Cuthill-McKee of the symmetrised edge list: the fallback when a graph carries no
    positions.  One-time topology preparation on the HOST (scipy.sparse.csgraph): one device->host copy of the
    edge list, ~1 s per million nodes; cached with the topology."""
    import numpy as np
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee

    ei = edge_index.detach().to("cpu", torch.int64).numpy()
    ok = (ei >= 0).all(axis=0) & (ei < num_nodes).all(axis=0)   # stray indices are reported by the topology build
    ei = ei[:, ok]
    a = coo_matrix((np.ones(ei.shape[1], dtype=np.int8), (ei[0], ei[1])), shape=(num_nodes, num_nodes)).tocsr()
    order = np.asarray(reverse_cuthill_mckee(a, symmetric_mode=False), dtype=np.int64)
    rank = np.empty(num_nodes, dtype=np.int64)
    rank[order] = np.arange(num_nodes, dtype=np.int64)
    dev = edge_index.device
    return torch.from_numpy(order).to(dev, torch.int32), torch.from_numpy(rank).to(dev, torch.int32)


class Topology:
    """dst-sorted (CSR) edge order of one ``edge_index`` plus the src-grouped view
    the backward scatter needs.  Built on the device by ``mgn_topology_build[_async]``; cached
    per ``edge_index`` tensor by :func:`get_topology` (mesh topology is static
    along a trajectory).

    ``lazy=True``: the build is queued WITHOUT a host synchronisation (a shuffled loader hands the engine a new
    edge_index every step).  The error flag and the degree maxima travel to pinned host memory behind an event;
    :meth:`resolve` (called by ``EncodeProcessDecode.forward`` once its launches are queued, and by any later
    query) waits for that event only -- by then the GPU has the whole forward pass in its queue, so the wait
    never starves it.  Until resolved the topology is used OPTIMISTICALLY as hub-free: always correct (the
    arrays are safe even with stray indices, a long segment is merely summed serially), and deterministic --
    the first forward pass of a fresh topology takes the hub-free launches, every later use the resolved ones.

    ``renumber`` ("morton" with ``pos``, "rcm", or None): the CSRs are built over RENUMBERED node ids;
    ``node_order`` / ``node_rank`` (int64) map new -> old / old -> new.  Callers gather their node rows with
    ``node_order`` on entry and with ``node_rank`` on exit (processors.EncodeProcessDecode.forward does)."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, lazy: bool = False, renumber: Optional[str] = None,
                 pos: Optional[torch.Tensor] = None):
        _require_device(edge_index)
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must have shape [2, E]")
        dev = edge_index.device
        ei = edge_index.to(torch.int64).contiguous()
        E, N = int(ei.shape[1]), int(num_nodes)
        self.N, self.E, self.device = N, E, dev
        self.node_order = self.node_rank = None
        if renumber is not None and N > 0:
            if renumber == "morton":
                if pos is None or pos.shape[0] != N:
                    raise ValueError("Morton renumbering needs one position per node")
                order, rank = morton_order(pos)
            elif renumber == "rcm":
                order, rank = rcm_order(ei, N)
            else:
                raise ValueError("renumber must be 'morton', 'rcm' or None")
            self.node_order, self.node_rank = order.long(), rank.long()
            # relabel the edge list (stray indices stay stray: the build reports them)
            okm = (ei >= 0) & (ei < N)
            ei = torch.where(okm, self.node_rank[ei.clamp(0, max(N - 1, 0))], ei).contiguous()
        L = _capi.lib()
        i32 = dict(dtype=torch.int32, device=dev)
        self.rowptr_dst, self.rowptr_src = torch.empty(N + 1, **i32), torch.empty(N + 1, **i32)
        self.perm_dst, self.perm_src = torch.empty(E, **i32), torch.empty(E, **i32)
        self.src_s, self.dst_s = torch.empty(E, **i32), torch.empty(E, **i32)
        ws = torch.empty(max(L.mgn_topology_workspace_bytes(E, N), 16), dtype=torch.uint8, device=dev)
        self._inv = None
        self._pending = None
        self._error = None   # sticky: a topology whose build reported stray indices raises on EVERY resolve / use
        self._uses = 0
        self.hub_dst = self.hub_src = None
        if lazy:
            flags = torch.zeros(4, **i32)
            with torch.cuda.device(dev):
                rc = L.mgn_topology_build_async(ei[0].data_ptr(), ei[1].data_ptr(), E, N, _ptr(self.rowptr_dst), _ptr(self.perm_dst),
                                                _ptr(self.src_s), _ptr(self.dst_s), _ptr(self.rowptr_src), _ptr(self.perm_src),
                                                _ptr(flags), _ptr(ws), ws.numel(), _stream(dev))
                _capi.check(rc, "mgn_topology_build_async")
                host = torch.empty(4, dtype=torch.int32, pin_memory=True)
                host.copy_(flags, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
            self._pending = (ev, host, flags, ws)   # flags / ws stay referenced until the copy has run
            self.max_in_degree = self.max_out_degree = None
            return
        mx = (C.c_int32 * 2)()
        with torch.cuda.device(dev):
            rc = L.mgn_topology_build(ei[0].data_ptr(), ei[1].data_ptr(), E, N, _ptr(self.rowptr_dst), _ptr(self.perm_dst), _ptr(self.src_s),
                                      _ptr(self.dst_s), _ptr(self.rowptr_src), _ptr(self.perm_src), mx, _ptr(ws), ws.numel(), _stream(dev))
        if rc == 3:
            raise IndexError(f"edge_index has entries outside [0, {N})")
        _capi.check(rc, "mgn_topology_build")
        self._set_degrees(int(mx[0]), int(mx[1]))

    def _set_degrees(self, din: int, dout: int):
        self.max_in_degree, self.max_out_degree = din, dout
        # Hub nodes.  A segment is summed by ONE lane group walking it in order -- ideal for meshes (degree
        # ~6), unbounded for the arbitrary edge_index the input contract allows (a node with 100 000 in-edges
        # would serialise 100 000 row loads).  Segments longer than HUB_CHUNK are therefore cut into chunks
        # summed in parallel, and a second small segment sum adds the chunk rows of each node (fixed order:
        # still deterministic, no atomics).
        self.hub_dst = _chunk_csr(self.rowptr_dst) if din > HUB_CHUNK else None
        self.hub_src = _chunk_csr(self.rowptr_src) if dout > HUB_CHUNK else None

    @property
    def resolved(self) -> bool:
        return self._pending is None

    def resolve(self) -> "Topology":
        """wait for the build's flags (lazy builds): raises IndexError on a stray index, sets the hub tables"""
        if self._pending is not None:
            ev, host, _flags, _ws = self._pending
            ev.synchronize()
            self._pending = None
            err, din, dout = int(host[0]), int(host[1]), int(host[2])
            if err:
                self.max_in_degree = self.max_out_degree = 0
                self._error = IndexError(f"edge_index has entries outside [0, {self.N})")
            else:
                self._set_degrees(din, dout)
        if self._error is not None:   # also on a cache hit of get_topology: never compute silently on clamped indices
            raise self._error
        return self

    def begin_use(self) -> bool:
        """hub state for ONE pass over this topology (see the class docstring): a fresh lazy build is taken
        as hub-free by its first user, later users wait for the flags"""
        if self._error is not None or (self._pending is not None and self._uses > 0):
            self.resolve()
        self._uses += 1
        return self.has_hubs

    @property
    def has_hubs(self) -> bool:
        return self.hub_dst is not None or self.hub_src is not None

    @property
    def perm_dst_long(self) -> torch.Tensor:
        """``perm_dst`` as int64 (what a torch row gather wants), converted once per topology instead of once per forward pass"""
        if getattr(self, "_perm_long", None) is None:
            self._perm_long = self.perm_dst.long()
        return self._perm_long

    @property
    def inv_perm(self) -> torch.Tensor:
        """position in the sorted order of each original edge id"""
        if self._inv is None:
            inv = torch.empty(self.E, dtype=torch.int64, device=self.device)
            inv[self.perm_dst_long] = torch.arange(self.E, device=self.device)
            self._inv = inv
        return self._inv


HUB_CHUNK = 1024  # longest segment one lane group sums on its own


def _chunk_csr(rowptr: torch.Tensor):
    """(chunk_rowptr[C+1], node_chunkptr[N+1]): every segment cut into chunks of at most HUB_CHUNK rows;
    chunk c of node i covers rows chunk_rowptr[c] .. chunk_rowptr[c+1], the chunks of node i are
    node_chunkptr[i] .. node_chunkptr[i+1].  One-time topology prep (torch index arithmetic on the device)."""
    rp = rowptr.long()
    deg = rp[1:] - rp[:-1]
    nch = (deg + HUB_CHUNK - 1) // HUB_CHUNK
    node_chunkptr = torch.zeros(rp.numel(), dtype=torch.int64, device=rp.device)
    node_chunkptr[1:] = torch.cumsum(nch, 0)
    C = int(node_chunkptr[-1])
    owner = torch.repeat_interleave(torch.arange(deg.numel(), device=rp.device), nch)
    k = torch.arange(C, device=rp.device) - node_chunkptr[owner]
    start = rp[owner] + k * HUB_CHUNK
    chunk_rowptr = torch.empty(C + 1, dtype=torch.int64, device=rp.device)
    chunk_rowptr[:C] = start
    chunk_rowptr[C] = rp[-1]
    # chunk c ends where the next one starts, except the last chunk of a node: its segment end
    end = torch.minimum(start + HUB_CHUNK, rp[owner + 1])
    assert bool((end[:-1] == start[1:]).all()) if C > 1 else True
    return chunk_rowptr.to(torch.int32), node_chunkptr.to(torch.int32)


def segsum_topo(src: torch.Tensor, topo: "Topology", by: str, out: torch.Tensor, n_rows: Optional[int] = None) -> torch.Tensor:
    """segment sum of the dst-sorted edge rows ``src`` onto nodes, ``by`` = "dst" (rows in segment order) or
    "src" (through perm_src); hub-safe (see Topology)."""
    rowptr, perm, hub = (topo.rowptr_dst, None, topo.hub_dst) if by == "dst" else (topo.rowptr_src, topo.perm_src, topo.hub_src)
    n = out.shape[0] if n_rows is None else n_rows
    if hub is None:
        return segsum(src, rowptr[:n + 1], perm, out)
    chunk_rowptr, node_chunkptr = hub
    part = segsum(src, chunk_rowptr, perm)                      # one row per chunk
    return segsum(part, node_chunkptr[:n + 1], None, out)       # chunks of a node, in order


_topo_cache: dict = {}


def get_topology(edge_index: torch.Tensor, num_nodes: int, pos: Optional[torch.Tensor] = None, renumber: bool = False) -> Topology:
    """cached Topology of an ``edge_index`` tensor (lazy build: no host synchronisation).  ``renumber``: let the
    engine renumber the nodes for locality when :func:`want_renumbering` says so (Morton order of ``pos`` when
    given, reverse Cuthill-McKee otherwise)."""
    ren = None
    if renumber and want_renumbering(int(num_nodes)):
        has_pos = pos is not None and pos.dim() == 2 and pos.shape[1] >= 2 and pos.shape[0] == num_nodes and pos.is_cuda
        # without device positions the only order available is reverse Cuthill-McKee ON THE HOST (a device-to-host copy of
        # the edge list + ~1 s per million nodes, per new edge_index tensor): only on explicit request ("on"), never in "auto"
        ren = "morton" if has_pos else ("rcm" if _renumber_mode == "on" else None)
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes), str(edge_index.device), ren,
           (pos.data_ptr(), pos._version) if ren == "morton" else None)
    hit = _topo_cache.get(key)
    if hit is not None and hit[0]() is edge_index:
        if hit[1]._error is not None:   # a bad topology is not served from the cache: the caller sees the error again
            raise hit[1]._error
        return hit[1]
    import weakref

    topo = Topology(edge_index, num_nodes, lazy=True, renumber=ren, pos=pos)
    if len(_topo_cache) > 64:
        _topo_cache.clear()
    try:
        _topo_cache[key] = (weakref.ref(edge_index), topo)
    except TypeError:
        pass
    return topo


def csr_build(key: torch.Tensor, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rowptr[n+1] int32, perm[E] int32): stable grouping of edge ids by key."""
    _require_device(key)
    L = _capi.lib()
    key = key.to(torch.int64).contiguous()
    E = key.numel()
    dev = key.device
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(E, dtype=torch.int32, device=dev)
    nbytes = L.mgn_csr_workspace_bytes(E, n)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.mgn_csr_build(_ptr(key), E, n, _ptr(rowptr), _ptr(perm), _ptr(ws), ws.numel(), _stream(dev))
    if rc == 3:
        raise IndexError(f"edge_index has entries outside [0, {n})")
    _capi.check(rc, "mgn_csr_build")
    return rowptr, perm


def segsum(src: torch.Tensor, rowptr: torch.Tensor, perm: Optional[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = sum of src rows of segment i (sequential in k order)."""
    _require_device(src)
    n = rowptr.numel() - 1
    H = src.shape[1]
    if out is None:
        out = torch.empty(n, H, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        rc = _capi.lib().mgn_segsum(_ptr(src), _ptr(rowptr), _ptr(perm), _ptr(out), n, H, _stream(src.device))
    _capi.check(rc, "mgn_segsum")
    return out


def seg_fix(rowptr: torch.Tensor, part: torch.Tensor, out: torch.Tensor):
    """second stage of the segment sum fused into ``mlp_fwd(seg=...)``"""
    with torch.cuda.device(out.device):
        rc = _capi.lib().mgn_seg_fix(_ptr(rowptr), _ptr(part), _ptr(out), rowptr.numel() - 1, _stream(out.device))
    _capi.check(rc, "mgn_seg_fix")


def segsum2(src: torch.Tensor, rowptr0, perm0, out0, rowptr1, perm1, out1):
    """two segment sums of the same source rows in one launch (H = 128)"""
    n = rowptr0.numel() - 1
    fn = _capi.lib().mgn_segsum2_b16 if src.dtype == torch.bfloat16 else _capi.lib().mgn_segsum2   # (two-byte rows: dZ[0] of precision 3)
    with torch.cuda.device(src.device):
        rc = fn(_ptr(src), _ptr(rowptr0), _ptr(perm0), _ptr(out0), _ptr(rowptr1), _ptr(perm1), _ptr(out1),
                                     n, src.shape[1], _stream(src.device))
    _capi.check(rc, "mgn_segsum2")


# ----------------------------------------------------------------- raw launches
def mlp_fwd(M: int, H: int, phases: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], int]],
            Ws: Sequence[torch.Tensor], bs: Sequence[Optional[torch.Tensor]], scale: Optional[torch.Tensor],
            out_w: int, resid: Optional[torch.Tensor], out: torch.Tensor, y_out: Optional[torch.Tensor] = None,
            saveH: Optional[Sequence[torch.Tensor]] = None, saveU: Optional[torch.Tensor] = None,
            saveR: Optional[torch.Tensor] = None, ldw0: int = 0,
            adds: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]] = (),
            posts: Sequence[Tuple[int, torch.Tensor]] = (), post_ldw: int = 0, wpk: Sequence[int] = (),
            saveM: Optional[Sequence[torch.Tensor]] = None, precision: int = 0, out_relu: bool = False,
            seg=None, act: int = 0, saveZ: Optional[Sequence[torch.Tensor]] = None):
    """``seg`` = (key[M] int32 sorted, rowptr[n+1] int32, out[n,H], part[ceil(M/16),2,H]): fused
    segment sum of y (finish with :func:`seg_fix`).
    ``adds``: (rows[*,H], idx or None) gathered into the layer-0 pre-activation;
    ``posts``: (device address of a [H,H] weight block with leading dim ``post_ldw``, out[M,H]);
    ``wpk``: device addresses of the launch's GEMM units packed by :func:`wpack` (phases of
    layer 0, layers 1.., post-products) -- selects the split-bf16 kernels."""
    a = _capi.MlpFwdArgs()
    a.M, a.H, a.NL, a.nphase = M, H, len(Ws), len(phases)
    for p, (src, idx, kw) in enumerate(phases):
        a.src[p], a.idx[p], a.kw[p] = _ptr(src), _ptr(idx), kw
    for l, (W, b) in enumerate(zip(Ws, bs)):
        a.W[l], a.b[l] = _ptr(W), _ptr(b)
    a.scale, a.eps, a.out_w = _ptr(scale), EPS, out_w
    a.resid, a.out, a.y_out = _ptr(resid), _ptr(out), _ptr(y_out)
    if saveH is not None:
        for l, h in enumerate(saveH):
            a.saveH[l] = _ptr(h)
    a.saveU, a.saveR = _ptr(saveU), _ptr(saveR)
    a.ldw0, a.n_add, a.n_post, a.post_ldw = ldw0, len(adds), len(posts), post_ldw
    for q, (t, ix) in enumerate(adds):
        a.add_src[q], a.add_idx[q] = _ptr(t), _ptr(ix)
    for q, (wptr, o) in enumerate(posts):
        a.post_W[q], a.post_out[q] = wptr, _ptr(o)
    for u, addr in enumerate(wpk):
        a.wpk[u] = addr
    if saveM is not None:
        for l, t in enumerate(saveM):
            a.saveM[l] = _ptr(t)
    a.precision = precision
    a.out_relu = int(out_relu)
    if seg is not None:
        a.seg_key, a.seg_rowptr, a.seg_out, a.seg_part = (_ptr(t) for t in seg)
    a.act = act
    if saveZ is not None:
        for l, t in enumerate(saveZ):
            a.saveZ[l] = _ptr(t)
    dev = out.device
    with torch.cuda.device(dev):
        rc = _capi.lib().mgn_mlp_fwd(C.byref(a), _stream(dev))
    _capi.check(rc, "mgn_mlp_fwd")


def mlp_bwd(M: int, H: int, NL: int, dOut: torch.Tensor, dOut2, idx2, out_w: int, U, R, scale,
            Hs: Sequence[torch.Tensor], WT: Sequence[Optional[torch.Tensor]], dZ: Sequence[Optional[torch.Tensor]],
            din: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]],
            db: Sequence[Optional[torch.Tensor]], dscale: Optional[torch.Tensor], wpk: Sequence[int] = (),
            Ms: Optional[Sequence[torch.Tensor]] = None, precision: int = 0, front=None, defer: Optional[list] = None,
            act: int = 0, Zs: Optional[Sequence[torch.Tensor]] = None, seg=None):
    """``seg`` = (key[M] int32 sorted, rowptr, out[n,H], part[ceil(M/16),2,H]): fused segment sum of dZ[0]
    (finish with :func:`seg_fix`).
    ``front`` = (rows[<=3] each [M,H], resid[M,H] or None, out[M,H] or None): the fused front
    stage of the packed kernel, dY = resid + sum_p wpk[p] . rows[p] (then ``dOut`` is ignored)."""
    L = _capi.lib()
    a = _capi.MlpBwdArgs()
    a.M, a.H, a.NL = M, H, NL
    a.dOut, a.dOut2, a.idx2, a.out_w = _ptr(dOut), _ptr(dOut2), _ptr(idx2), out_w
    a.U, a.R, a.scale, a.eps = _ptr(U), _ptr(R), _ptr(scale), EPS
    for l, h in enumerate(Hs or ()):
        a.Hs[l] = _ptr(h)
    for l in range(NL):
        a.WT[l] = _ptr(WT[l])
        a.dZ[l] = _ptr(dZ[l])
        a.db[l] = _ptr(db[l])
    a.n_din = len(din)
    for q, (wt0, res, dst) in enumerate(din):
        a.WT0[q], a.din_resid[q], a.dIn[q] = _ptr(wt0), _ptr(res), _ptr(dst)
    a.dscale = _ptr(dscale)
    for u, addr in enumerate(wpk):
        a.wpk[u] = addr
    if Ms is not None:
        for l, t in enumerate(Ms):
            a.Ms[l] = _ptr(t)
    a.precision = precision
    a.act = act
    if Zs is not None:
        for l, t in enumerate(Zs):
            a.Zs[l] = _ptr(t)
    if seg is not None:
        a.seg_key, a.seg_rowptr, a.seg_out, a.seg_part = (_ptr(t) for t in seg)
    if front is not None:
        rows, fres, fout = front
        a.n_front = len(rows)
        for p_, t in enumerate(rows):
            a.front_src[p_] = _ptr(t)
        a.front_resid, a.front_out = _ptr(fres), _ptr(fout)
    dev = dOut.device
    nbytes = L.mgn_mlp_bwd_workspace_bytes(M, H, NL)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    a.red_ws, a.red_ws_bytes = _ptr(ws), ws.numel()
    if defer is not None:  # column sums finished later, many launches at once (colred_batch)
        a.defer_reduce = 1
        defer.append((ws, M, H, NL, out_w, len(din), list(db), dscale if scale is not None else None))
    with torch.cuda.device(dev):
        rc = L.mgn_mlp_bwd(C.byref(a), _stream(dev))
    _capi.check(rc, "mgn_mlp_bwd")


_fused_ws: dict = {}


def edge_bwd_fused(E: int, dOut, dAgg, idx, U, R, scale, X: Sequence[torch.Tensor], Ms: Sequence[torch.Tensor], wpk: Sequence[int],
                   dIn, dZ0, dW: Sequence[Tuple[torch.Tensor, int, int]], db: Sequence[Optional[torch.Tensor]], dscale, precision: int = 0):
    """mgn_edge_bwd_fused: the edge backward chain of a round AND its four E-row weight gradients in one kernel
    (include/mgn_hip.h).  ``dW``: (tensor, element offset, leading dimension) per layer."""
    L = _capi.lib()
    a = _capi.EdgeBwdFusedArgs()
    a.M = E
    a.dOut, a.dAgg, a.idx, a.U, a.R, a.scale, a.eps = _ptr(dOut), _ptr(dAgg), _ptr(idx), _ptr(U), _ptr(R), _ptr(scale), EPS
    for l in range(4):
        a.X[l] = _ptr(X[l])
        a.wpk[l] = wpk[l]
        t, off, ld = dW[l]
        a.dW[l], a.ldw[l] = t.data_ptr() + 4 * off, ld
        a.db[l] = _ptr(db[l])
    for l in range(3):
        a.Ms[l] = _ptr(Ms[l])
    a.dIn, a.dZ0, a.dscale = _ptr(dIn), _ptr(dZ0), _ptr(dscale)
    dev = dOut.device
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _fused_ws.get(key)   # 68 MB of per-workgroup partials, re-used by every round (launches of one stream are ordered)
    if ws is None:
        ws = torch.empty(L.mgn_edge_bwd_fused_workspace_bytes(), dtype=torch.uint8, device=dev)
        _fused_ws[key] = ws
    a.ws, a.ws_bytes, a.precision = _ptr(ws), ws.numel(), precision
    with torch.cuda.device(dev):
        rc = L.mgn_edge_bwd_fused(C.byref(a), _stream(dev))
    _capi.check(rc, "mgn_edge_bwd_fused")


def colred_batch(deferred: list, dev):
    """finish the column sums (bias / scale gradients) of ``mlp_bwd(..., defer=deferred)`` launches"""
    if not deferred:
        return
    arr = (_capi.ColredJob * len(deferred))()
    for i, (ws, M, H, NL, out_w, n_din, db, dscale) in enumerate(deferred):
        arr[i].red_ws, arr[i].M, arr[i].H, arr[i].NL, arr[i].out_w, arr[i].n_din = _ptr(ws), M, H, NL, out_w, n_din
        for l in range(NL):
            arr[i].db[l] = _ptr(db[l])
        arr[i].dscale = _ptr(dscale)
    with torch.cuda.device(dev):
        rc = _capi.lib().mgn_colred_batch(len(deferred), arr, _stream(dev))
    _capi.check(rc, "mgn_colred_batch")
    deferred.clear()


def wgrad(jobs, dev, precision: int = 0):
    """jobs: (A, lda, nja, B, ldb, nkb, kw, dW_tensor, dW_offset_elems, ldw[, db]); M = A.shape[0].
    ``db`` (optional tensor [16*nja]) receives the column sums of A = the bias gradient."""
    L = _capi.lib()
    for i0 in range(0, len(jobs), _capi.MAX_WGRAD_JOBS):
        chunk = jobs[i0:i0 + _capi.MAX_WGRAD_JOBS]
        arr = (_capi.WgradJob * len(chunk))()
        for j, job in enumerate(chunk):
            A, lda, nja, B, ldb, nkb, kw, dW, off, ldw = job[:10]
            arr[j].db = _ptr(job[10]) if len(job) > 10 else None
            arr[j].A, arr[j].B = _ptr(A), _ptr(B)
            arr[j].dW = dW.data_ptr() + 4 * off
            arr[j].M = A.shape[0]
            arr[j].lda, arr[j].ldb, arr[j].ldw = lda, ldb, ldw
            arr[j].nja, arr[j].nkb, arr[j].kw = nja, nkb, kw
        nbytes = L.mgn_wgrad_workspace_bytes(len(chunk), arr)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = L.mgn_wgrad_p(len(chunk), arr, _ptr(ws), ws.numel(), precision, _stream(dev))
        _capi.check(rc, "mgn_wgrad")


def wpack(blocks: Sequence[Tuple[int, int, bool, int]], dev):
    """blocks: (src_address, ld_src, transpose, dst_address) of 128 x 128 fp32 blocks ->
    96 KB bf16x3 MFMA images (include/mgn_hip.h, mgn_wpack)."""
    if not blocks:
        return
    arr = (_capi.WpackBlock * len(blocks))()
    for i, (src, ld, tr, dst) in enumerate(blocks):
        arr[i].src, arr[i].ld_src, arr[i].transpose, arr[i].dst = src, ld, int(bool(tr)), dst
    with torch.cuda.device(dev):
        rc = _capi.lib().mgn_wpack(len(blocks), arr, _stream(dev))
    _capi.check(rc, "mgn_wpack")


def transpose_blocks(blocks: Sequence[Tuple[int, int, int, int]], H: int, dev):
    """blocks: (src_address, ld_src, dst_address, ld_dst) of H x H blocks: dst[k,j] = src[j,k]."""
    if not blocks:
        return
    arr = (_capi.TBlock * len(blocks))()
    for i, (src, lds, dst, ldd) in enumerate(blocks):
        arr[i].src, arr[i].ld_src, arr[i].dst, arr[i].ld_dst = src, lds, dst, ldd
    with torch.cuda.device(dev):
        rc = _capi.lib().mgn_transpose_blocks(len(blocks), arr, H, _stream(dev))
    _capi.check(rc, "mgn_transpose_blocks")


# ------------------------------------------------------------ small row kernels
def gather_rows(src: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i,:] = src[idx[i],:] (idx int32): packs the rows a halo peer needs."""
    n, H = int(idx.numel()), int(src.shape[1])
    if out is None:
        out = torch.empty(n, H, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        rc = _capi.lib().mgn_gather_rows(_ptr(src), _ptr(idx), n, H, _ptr(out), _stream(src.device))
    _capi.check(rc, "mgn_gather_rows", prep=True)
    return out


def halo_unpack_add(rows: torch.Tensor, nodes: torch.Tensor, rowptr: torch.Tensor, perm: torch.Tensor, dst: torch.Tensor):
    """dst[nodes[j]] += sum of rows[perm[rowptr[j]:rowptr[j+1]]] in ascending order (atomics-free)."""
    with torch.cuda.device(dst.device):
        rc = _capi.lib().mgn_halo_unpack_add(_ptr(rows), _ptr(nodes), _ptr(rowptr), _ptr(perm), int(nodes.numel()), int(dst.shape[1]),
                                             _ptr(dst), _stream(dst.device))
    _capi.check(rc, "mgn_halo_unpack_add", prep=True)


def gate_fwd(G, phi, gate_pos, agg, gate_out, agg_out):
    with torch.cuda.device(agg.device):
        rc = _capi.lib().mgn_gate_fwd(_ptr(G), _ptr(phi), _ptr(gate_pos), _ptr(agg), int(agg.shape[0]), int(agg.shape[1]),
                                      _ptr(gate_out), _ptr(agg_out), _stream(agg.device))
    _capi.check(rc, "mgn_gate_fwd", prep=True)


def gate_bwd(dAggG, agg, gate, dAgg, dG):
    with torch.cuda.device(agg.device):
        rc = _capi.lib().mgn_gate_bwd(_ptr(dAggG), _ptr(agg), _ptr(gate), int(agg.shape[0]), int(agg.shape[1]), _ptr(dAgg), _ptr(dG),
                                      _stream(agg.device))
    _capi.check(rc, "mgn_gate_bwd", prep=True)


def rope_gather(x, pos, inv_freq, topo: "Topology", axes: int, out):
    """out[k] = RoPE(x[src[k]], pos[src[k]] - pos[dst[k]]) for the dst-sorted edges (layers.py:1020-1026,1104-1149)."""
    with torch.cuda.device(x.device):
        rc = _capi.lib().mgn_rope_gather(_ptr(x), _ptr(pos), int(pos.shape[1]), _ptr(inv_freq), int(inv_freq.numel()), axes,
                                         _ptr(topo.src_s), _ptr(topo.dst_s), topo.E, int(x.shape[1]), _ptr(out), _stream(x.device))
    _capi.check(rc, "mgn_rope_gather", prep=True)


def rope_scatter(T, pos, inv_freq, topo: "Topology", axes: int, resid, out):
    """out[j] = resid[j] + sum_{k: src[k]=j} RoPE^T(T[k]) (the gradient of rope_gather wrt x), in the
    fixed order of the src-grouped CSR."""
    n = int(out.shape[0])
    with torch.cuda.device(T.device):
        rc = _capi.lib().mgn_rope_scatter(_ptr(T), _ptr(pos), int(pos.shape[1]), _ptr(inv_freq), int(inv_freq.numel()), axes,
                                          _ptr(topo.src_s), _ptr(topo.dst_s), _ptr(topo.rowptr_src), _ptr(topo.perm_src), n,
                                          int(T.shape[1]), _ptr(resid), _ptr(out), _stream(T.device))
    _capi.check(rc, "mgn_rope_scatter", prep=True)


# ``ctx.needs_input_grad`` mirrors ``tensor.requires_grad`` even under ``torch.no_grad()``, and
# grad mode is always off INSIDE ``Function.forward`` -- so whether activations must be saved for
# a backward pass has to be read BEFORE ``apply``.  Without this an inference forward ran the
# training-mode kernels (4 extra 512-byte stores per row and layer) and kept every round's
# activations alive (270 GB on the 1M-node mesh).
import threading as _threading
from dataclasses import dataclass

_call = _threading.local()


def _saving() -> bool:
    return getattr(_call, "grad", True)


ACT_IDS = {"relu": 0, "silu": 1, "gelu": 2}   # gelu: stand-alone build_mlp(act="gelu") only (generic kernels)


def mlp_apply(x, has_norm, *params, act: str = "relu"):
    """MlpFunction.apply with the caller's grad mode recorded."""
    _call.grad = torch.is_grad_enabled()
    try:
        return MlpFunction.apply(x, has_norm, ACT_IDS[act], *params)
    finally:
        _call.grad = True


@dataclass(frozen=True)
class BlockSpec:
    """Variant of GraphNetBlock the processor runs (layers.py:890-987): Linear layers per MLP, trailing
    RMSNorm, activation, sigmoid gate on the aggregate, relative RoPE on the source features."""
    nb_layers: int = 4
    layer_norm: bool = True
    act: str = "relu"
    gate: bool = False
    rope: bool = False
    rope_axes: int = 3

    @property
    def mlp_params(self) -> int:
        return 2 * self.nb_layers + (1 if self.layer_norm else 0)

    @property
    def per_block(self) -> int:
        return 2 * self.mlp_params + (3 if self.gate else 0)

    @property
    def act_id(self) -> int:
        return ACT_IDS[self.act]


DEFAULT_SPEC = BlockSpec()
PARAMS_PER_BLOCK = DEFAULT_SPEC.per_block  # 18: edge W0,b0..W3,b3,scale ; node W0,b0..W3,b3,scale


_side_streams: dict = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    key = (dev.type, dev.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


def processor_apply(x, e, topo, L, *params, spec: BlockSpec = DEFAULT_SPEC, halo=None, pos=None, phi=None, rope_inv_freq=None):
    """ProcessorFunction.apply with the caller's grad mode recorded."""
    _call.grad = torch.is_grad_enabled()
    try:
        return ProcessorFunction.apply(x, e, topo, L, spec, halo, pos, phi, rope_inv_freq, *params)
    finally:
        _call.grad = True


#: Called by ProcessorFunction.backward right after a round's weight-gradient launches are queued, with that round's parameter
#: tensors and their gradients (the tensors backward() will return for them): a data-parallel wrapper starts the all-reduce of a
#: bucket of rounds while the earlier rounds are still being differentiated (distributed.OverlappedGradAllReduce).  None: no call.
_grad_ready_hook = None

def _bf16_rounded_many(ts):
    """bf16-rounded fp32 copies of a list of tensors (bf16 matrix mode off the packed path) in two multi-tensor launches.  Never
    cached: the fused optimiser writes parameters through raw pointers (no version bump) and a captured training step replays
    without Python, so a copy kept across calls would silently go stale."""
    if not ts:
        return []
    src = [t.detach() for t in ts]
    lo = [torch.empty_like(t, dtype=torch.bfloat16) for t in src]
    out = [torch.empty_like(t) for t in src]
    torch._foreach_copy_(lo, src)
    torch._foreach_copy_(out, lo)
    return out


def set_grad_ready_hook(fn) -> None:
    global _grad_ready_hook
    _grad_ready_hook = fn


#: activation recompute of the processor: "off" (save every round's activations: 2.5 KB per edge + 3 KB per
#: node and round), "on", or "auto" (default): on when the saves would not fit comfortably -- more than
#: MGN_RECOMPUTE_FRACTION (0.65) of the free device memory.  The 1M-node / 6M-edge mesh needs ~270 GB of
#: saves for 15 rounds: with recompute it trains on ONE MI355X (~50 GB), at the price of one extra
#: training-mode forward per round inside the backward pass.
def _parse_recompute_env(v: str):
    """MGN_RECOMPUTE: "off" | "on" | "auto" | a number of rounds; anything else is an error (a typo must not fall through
    to the heuristic silently)"""
    v = v.strip()
    if v.isdigit():
        return int(v)
    if v not in ("off", "on", "auto"):
        raise ValueError(f"MGN_RECOMPUTE={v!r}: expected 'off', 'on', 'auto' or a number of rounds")
    return v


_recompute_mode = _parse_recompute_env(_os.environ.get("MGN_RECOMPUTE", "auto"))


def set_activation_recompute(mode) -> None:
    """"off" | "on" | "auto", or an int: recompute exactly that many (the first) rounds and save the others"""
    global _recompute_mode
    if not (mode in ("off", "on", "auto") or (isinstance(mode, int) and not isinstance(mode, bool) and mode >= 0)):
        raise ValueError("activation recompute mode must be 'off', 'on', 'auto' or a number of rounds")
    _recompute_mode = mode


def get_activation_recompute() -> str:
    return _recompute_mode


def saved_activation_bytes(E: int, Nn: int, H: int, NL: int, L: int, act: int, save16: bool = False) -> int:
    """what ProcessorFunction keeps for the backward pass without recompute (``save16``: the edge rows' H1.. are two-byte rows,
    the bf16 matrix mode's default on the packed path)"""
    per_row = 4 * H * ((NL - 1) * (2 if act == 1 else 1) + 2) + 16 * (NL - 1) + 4   # H1.., [Z1..], U, input | masks | rms
    per_edge = per_row - (2 * H * (NL - 1) if save16 else 0)
    return L * (E * per_edge + Nn * (per_row + 4 * H))                                 # + agg per node


def recompute_rounds(E, Nn, H, NL, L, act, dev, save16: bool = False) -> int:
    """how many of the L rounds keep only their inputs and are re-run in the backward pass (the FIRST ones: they are
    differentiated last, when the saved rounds have been released).  "auto": as many rounds are saved as fit in
    MGN_RECOMPUTE_FRACTION (default 0.65; round 5: 0.5 left 65 GiB of the 268 unused on the 1M-node mesh -- tools/c4_recompute_sweep.py:
    438 / 431 / 424 / 417 ms per step at 7 / 6 / 5 / 4 recomputed rounds, peak 203 / 217 / 231 / 245 GiB) of the free device memory -- 0 recomputed on the bench batch, 5 of 15 on the
    1M-node mesh on one GPU (all 15 before round 3: one extra forward per step where 40 % of one is enough)."""
    if _recompute_mode == "on":
        return L
    if _recompute_mode == "off":
        return 0
    if isinstance(_recompute_mode, int):
        return min(_recompute_mode, L)
    try:
        free, _total = torch.cuda.mem_get_info(dev)
        # what the caching allocator holds without using it is free for this purpose (else the answer would change from the second
        # step on, when the first step's blocks sit in the cache)
        free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    except Exception:  # noqa: BLE001
        return 0
    frac = float(_os.environ.get("MGN_RECOMPUTE_FRACTION", "0.65"))
    per_round = saved_activation_bytes(E, Nn, H, NL, 1, act, save16)
    fit = int(frac * free // max(per_round, 1))
    return max(0, L - fit)


# ------------------------------------------------------------ generic MLP (R2)
class MlpFunction(torch.autograd.Function):
    """build_mlp forward/backward on the engine (encoders, decoder, stand-alone MLPs).

    apply(x, has_norm, act_id, W0, b0, ..., W_{NL-1}, b_{NL-1} [, scale]) -> y[M, out]
    First-layer columns / last-layer rows are zero-padded to multiples of 16 here
    (plumbing) so that the kernels only see aligned weights.  H = 128 with a full-width output runs
    on the packed split-bf16 kernels in both directions (an encoder's narrow first layer as a
    stand-alone generic launch); everything else on the generic kernels.
    """

    @staticmethod
    def forward(ctx, x, has_norm, act, *params):
        _require_device(x, *params)
        x = _f32c(x)
        NL = (len(params) - (1 if has_norm else 0)) // 2
        Ws = [_f32c(params[2 * l]) for l in range(NL)]
        bs = [_f32c(params[2 * l + 1]) for l in range(NL)]
        scale = _f32c(params[2 * NL]) if has_norm else None
        if NL < 2:
            raise AssertionError("The MLP must have at least 2 layers (input and output).")
        M, kin = x.shape
        H = Ws[0].shape[0]
        out_w = Ws[-1].shape[0]
        if H not in SUPPORTED_H or kin > H or out_w > H:
            raise NotImplementedError(f"MLP widths (in={kin}, hidden={H}, out={out_w}) not supported by the MI355X engine")
        dev = x.device
        kp, op = pad16(kin), pad16(out_w)
        W0 = Ws[0]
        if kp != kin:
            W0 = torch.nn.functional.pad(W0, (0, kp - kin))
        Wl, bl = Ws[-1], bs[-1]
        if op != out_w:
            Wl = torch.nn.functional.pad(Wl, (0, 0, 0, op - out_w))
            bl = torch.nn.functional.pad(bl, (0, op - out_w))
        Wk = [W0] + Ws[1:-1] + [Wl]
        bk = bs[:-1] + [bl]
        # bf16 matrix mode on the generic kernels (widths off the packed path, e.g. hidden 32): operands rounded here (weights,
        # biases) and in the kernel (rows, layer results) -- precision = 1, see mgn_hip.h.  The packed H = 128 modes of this
        # function stay fp32-grade (encoders / decoder: ~4 % of the step's matrix work).
        prec = 1 if (_matrix_precision == "bf16" and not (H == 128 and out_w == H and X6_ENABLED and act != 2)) else 0
        if prec:
            Wk = [w.to(torch.bfloat16).to(torch.float32) for w in Wk]
            bk = [b.to(torch.bfloat16).to(torch.float32) for b in bk]
        need = any(ctx.needs_input_grad) and _saving()
        f = dict(dtype=torch.float32, device=dev)
        y = torch.empty(M, out_w, **f)
        x6 = H == 128 and out_w == H and X6_ENABLED and M > 0 and act != 2   # GELU: generic kernels
        mode = "generic"
        if x6 and kin < H and 3 <= NL <= 4:
            mode = "enc"    # narrow first layer stand-alone, layers 1.. on the packed kernels
        elif x6 and kin == H and NL <= 4:
            mode = "full"
        packed = mode != "generic"
        saveH = [torch.empty(M, H, **f) for _ in range(NL - 1)] if need else None
        U = torch.empty(M, H, **f) if (need and has_norm) else None
        R = torch.empty(M, **f) if (need and has_norm) else None
        Ms = [torch.empty(M, 4, dtype=torch.int32, device=dev) for _ in range(NL - 1)] if (need and packed and act == 0) else None
        Zs = [torch.empty(M, H, **f) for _ in range(NL - 1)] if (need and act != 0) else None
        if mode == "enc":
            # h1 = act(W0 x + b0): the activation the backward needs anyway
            h1 = saveH[0] if need else torch.empty(M, H, **f)
            mlp_fwd(M, H, [(x, None, kin)], [Wk[0]], [bk[0]], None, H, None, h1, Zs[0] if Zs else None, out_relu=True, act=act,
                    saveM=[Ms[0]] if Ms else None)
            pk = torch.empty((NL - 1) * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
            units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(NL - 1)]
            wpack([(Wk[l + 1].data_ptr(), H, False, units[l]) for l in range(NL - 1)], dev)
            mlp_fwd(M, H, [(h1, None, H)], Wk[1:], bk[1:], scale, out_w, None, y, None,
                    saveH[1:] if need else None, U, R, wpk=units, saveM=Ms[1:] if Ms else None,
                    saveZ=Zs[1:] if Zs else None, act=act)
        elif mode == "full":
            pk = torch.empty(NL * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
            units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(NL)]
            wpack([(Wk[l].data_ptr(), H, False, units[l]) for l in range(NL)], dev)
            mlp_fwd(M, H, [(x, None, H)], Wk, bk, scale, out_w, None, y, None, saveH, U, R, wpk=units, saveM=Ms, saveZ=Zs, act=act)
        else:
            mlp_fwd(M, H, [(x, None, kin)], Wk, bk, scale, out_w, None, y, None, saveH, U, R, act=act, saveZ=Zs, precision=prec)
        ctx.meta = (NL, H, kin, out_w, has_norm, kp != kin, op != out_w, act, mode)
        ctx.prec, ctx.Wk_rounded = prec, (Wk if prec else None)
        # inputs / parameters through save_for_backward (autograd's version counter then catches an
        # in-place update between forward and backward); padded weight copies and the activations
        # the kernels wrote are ours and live until the graph is freed
        ctx.save_for_backward(x, *Ws, *([scale] if has_norm else []))
        ctx.pads = (W0 if kp != kin else None, Wl if op != out_w else None)
        ctx.saved_acts = (saveH, U, R, Ms, Zs)
        return y

    @staticmethod
    def backward(ctx, dy):
        NL, H, kin, out_w, has_norm, pad0, padl, act, mode = ctx.meta
        if ctx.saved_acts is None or ctx.saved_acts[0] is None:
            raise RuntimeError("MlpFunction: no saved activations -- backward ran a second time without retain_graph "
                               "support, or the forward ran under no_grad")
        t = ctx.saved_tensors
        x, Ws = t[0], list(t[1:1 + NL])
        scale = t[1 + NL] if has_norm else None
        Wk = [ctx.pads[0] if pad0 else Ws[0]] + Ws[1:-1] + [ctx.pads[1] if padl else Ws[-1]]
        prec = getattr(ctx, "prec", 0)
        if prec:
            Wk = ctx.Wk_rounded
        saveH, U, R, Ms, Zs = ctx.saved_acts
        dy = _f32c(dy)
        M, dev = x.shape[0], x.device
        kp, op = pad16(kin), pad16(out_w)
        f = dict(dtype=torch.float32, device=dev)
        widths = [H] * (NL - 1) + [op]
        zero = M == 0  # the launches return early: nothing would be written
        mk = torch.zeros if zero else torch.empty
        dZ = [mk(M, w, **f) for w in widths]
        db = [mk(w, **f) for w in widths]
        dscale = mk(H, **f) if has_norm else None
        dx = None
        want_dx = ctx.needs_input_grad[0]
        if want_dx and kin != H:
            raise NotImplementedError("input gradient of a ragged-width MLP input is not needed by the path")
        if want_dx:
            dx = mk(M, H, **f)
        if mode in ("enc", "full") and M > 0:
            # packed split-bf16 chain: W^T units of layers NL-1 .. 1 (then layer 0's for the input gradient)
            tr = [Wk[l] for l in range(NL - 1, 0, -1)] + ([Wk[0]] if want_dx else [])
            pk = torch.empty(len(tr) * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
            units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(len(tr))]
            wpack([(w.data_ptr(), H, True, units[u]) for u, w in enumerate(tr)], dev)
            din = [(None, None, dx)] if want_dx else []
            mlp_bwd(M, H, NL, dy, None, None, out_w, U, R, scale, saveH, [None] * NL, dZ, din, [None] * NL, dscale,
                    wpk=units, Ms=Ms, Zs=Zs, act=act)
        else:
            # W^T operands of the generic chain: the H x H ones in ONE batched transpose launch (a torch transpose + copy each before)
            sq = [l for l in range(1, NL) if tuple(Wk[l].shape) == (H, H) and Wk[l].is_contiguous()]
            wt_buf = torch.empty(len(sq), H, H, **f) if sq else None
            if sq:
                transpose_blocks([(Wk[l].data_ptr(), H, wt_buf[k].data_ptr(), H) for k, l in enumerate(sq)], H, dev)
            WT = [None] + [wt_buf[sq.index(l)] if l in sq else Wk[l].t().contiguous() for l in range(1, NL)]
            din = [(Wk[0].t().contiguous(), None, dx)] if want_dx else []
            # bias gradients are a by-product of the weight-gradient kernel (it reads dZ anyway)
            mlp_bwd(M, H, NL, dy, None, None, out_w, U, R, scale, saveH, WT, dZ, din, [None] * NL, dscale, act=act, Zs=Zs, precision=prec)
        # (bf16 matrix mode: the reference's autocast Linear multiplies the bf16 copy of its input in the weight gradient too; the saved
        #  layer results are rounded by the kernels already)
        ins = [x.to(torch.bfloat16).to(torch.float32) if prec else x] + list(saveH)
        in_w = [kin] + [H] * (NL - 1)
        # (the first layer's gradient in its exact [H, kin] shape: the reduction writes columns < ldw only, so ldw = kin needs no
        #  padded buffer -- and no strided slice that the optimiser would have to copy into a contiguous gradient every step)
        dWs = [mk(widths[l], in_w[l], **f) for l in range(NL)]
        jobs = []
        for l in range(NL):
            jobs.append((dZ[l], widths[l], widths[l] // 16, ins[l], in_w[l], pad16(in_w[l]) // 16, in_w[l], dWs[l], 0, in_w[l], db[l]))
        if M > 0:
            wgrad(jobs, dev)
        grads = []
        for l in range(NL):
            dW, dbl = dWs[l], db[l]
            if l == NL - 1 and op != out_w:
                dW, dbl = dW[:out_w], dbl[:out_w]
            grads += [dW, dbl]
        if has_norm:
            grads.append(dscale)
        ctx.saved_acts = None
        return (dx, None, None, *grads)


# ------------------------------------------------- processor: L GraphNetBlocks (R3-R5)
def _split_block(q, spec: BlockSpec):
    """block parameters (state_dict order) -> (We, be, se, Wn, bn, sn, gate params or None)"""
    NL, k = spec.nb_layers, spec.mlp_params
    We, be = [q[2 * l] for l in range(NL)], [q[2 * l + 1] for l in range(NL)]
    se = q[2 * NL] if spec.layer_norm else None
    Wn, bn = [q[k + 2 * l] for l in range(NL)], [q[k + 2 * l + 1] for l in range(NL)]
    sn = q[k + 2 * NL] if spec.layer_norm else None
    gp_ = tuple(q[2 * k: 2 * k + 3]) if spec.gate else None
    return We, be, se, Wn, bn, sn, gp_


class ProcessorFunction(torch.autograd.Function):
    """L rounds of gather -> edge MLP -> segment-sum -> node MLP -> residuals.

    apply(x[Nn,H], e_sorted[E,H], topo, L, spec, halo, pos, phi, rope_inv_freq, *params) -> (x_out, e_out_sorted)
    ``e`` is in the topology's dst-sorted order.  params: ``spec.per_block`` tensors per block in
    state_dict order (edge_block.{0,2,..}.{weight,bias}[, .scale], node_block..., [gate_proj.weight,
    gate_proj.bias, gate_pos]).  ``halo`` (distributed.HaloState): the node-partitioned mesh -- ``x`` holds
    the OWNED rows, the topology counts owned + ghost nodes, and once per round the projected latents of
    the ghost sources are exchanged (overlapped with the edge rows that need no remote data).
    """

    @staticmethod
    def forward(ctx, x, e, topo: Topology, L: int, spec: BlockSpec, halo, pos, phi, rope_inv_freq, *params):
        _require_device(x, e, *params)
        x, e = _f32c(x), _f32c(e)
        Nn, H = x.shape            # node rows computed here (all of them, or the owned ones)
        N = topo.N                 # node index space of the topology (owned + ghosts under a halo)
        E = e.shape[0]
        NL, act, PB = spec.nb_layers, spec.act_id, spec.per_block
        if H not in SUPPORTED_H:
            raise NotImplementedError(f"hidden_size={H} not supported by the MI355X engine (16/32/64/128)")
        if E != topo.E or (halo is None and Nn != N) or (halo is not None and (Nn != halo.n_own or N != halo.n_own + halo.n_ghost)):
            raise ValueError("x / edge_attr do not match the topology")
        if len(params) != L * PB:
            raise ValueError(f"expected {L} x {PB} block parameters, got {len(params)}")
        if NL < 2 or NL > _capi.MAX_LAYERS:
            raise AssertionError("The MLP must have at least 2 layers (input and output)." if NL < 2 else f"at most {_capi.MAX_LAYERS} layers per MLP")
        dev = x.device
        P = [_f32c(p) for p in params]
        P_params = P   # (P may be re-bound to kernel operands below: bf16 matrix mode off the packed path)
        need = any(ctx.needs_input_grad) and _saving()
        f = dict(dtype=torch.float32, device=dev)
        _require_device(pos if spec.rope else None, phi, rope_inv_freq if spec.rope else None)
        if spec.rope:
            if pos is None:
                raise ValueError("Node positions `pos` must be provided when use_rope=True.")
            pos = _f32c(pos)
            if pos.dim() != 2 or pos.shape[1] < spec.rope_axes:
                raise ValueError("pos has fewer columns than rope_axes")
            if pos.shape[0] < N:  # the reference's x[col] / pos[col] indexing would raise here
                raise IndexError(f"pos has {pos.shape[0]} rows for {N} nodes")
            if rope_inv_freq is None or rope_inv_freq.numel() != H // (2 * spec.rope_axes):
                raise ValueError("rope_inv_freq must hold hidden_size // (2 * rope_axes) frequencies")
            rope_inv_freq = _f32c(rope_inv_freq)
        if phi is not None:
            phi = _f32c(phi).reshape(-1)
            if phi.numel() != Nn:  # the reference's broadcast of phi [N, 1] against the gate [N, H] would raise
                raise ValueError(f"phi holds {phi.numel()} values for {Nn} nodes")
        # packed split-bf16 kernels: H = 128, at most 4 layers per MLP
        x6 = (H == 128) and X6_ENABLED and NL <= 4 and L > 0 and E > 0
        # algebraic split of the first edge layer (W0 = [W_e | W_d | W_s]):
        #   W0.[e, x_dst, x_src] = W_e.e + (x W_d^T)[dst] + (x W_s^T)[src]
        # the two node-level projections of round i+1 are post-products of round i's node kernel (x'
        # still in registers); round 0's come from two small launches.  RoPE rotates x_src per edge,
        # so there the three slabs stay three gathered phases.
        split = (H == 128) and NL <= 4 and not spec.rope and (x6 or act == 0) and L > 0 and E > 0
        prec = 1 if _matrix_precision == "bf16" else 0
        if prec and not x6 and L > 0 and E > 0:
            # [r4] bf16 matrix mode off the packed path (any supported width, e.g. the shipped cylinder.json's hidden 32): the
            # generic kernels round the row operands and every layer's result to bf16 (precision = 1); the weights and biases of
            # the Linear layers go in rounded (the RMSNorm scales, gate positions stay fp32: autocast does not touch them)
            # P_params stays the list of PARAMETERS (what autograd version-checks and what the grad-ready hook reports); the rounded
            # operands live in their own list (two multi-tensor cast launches per forward pass)
            k_ = spec.mlp_params
            P = list(P)
            idx_ = [i_ * PB + j_ for i_ in range(L)
                    for j_ in list(range(2 * NL)) + list(range(k_, k_ + 2 * NL)) + ([2 * k_, 2 * k_ + 1] if spec.gate else [])]
            for i_, r_ in zip(idx_, _bf16_rounded_many([P_params[i_] for i_ in idx_])):
                P[i_] = r_
        if halo is not None and not (x6 and split):
            raise NotImplementedError("the partitioned path runs on the packed H = 128 kernels (no RoPE)")
        # (the fused aggregation's second stage walks a node's tile partials serially: hub topologies take the
        # stand-alone, chunked segment sum instead)
        hubs = topo.begin_use()   # (a fresh lazily-built topology is hub-free for its first pass: Topology docstring)
        fuse_agg = x6 and _os.environ.get("MGN_NO_FUSED_AGG") is None and not hubs
        relu_bits = x6 and act == 0
        # [r4] bf16 matrix mode on the packed path: the saved edge activations H1..H_{NL-1} are bf16 tensors in the reference (autocast
        # Linear -> ReLU) and only the weight gradients read them back, so they are STORED as bf16 rows (precision 2 of mgn_mlp_fwd,
        # ldb = -128 of mgn_wgrad): half the bytes of those saves on both sides, bit-identical gradients.  MGN_SAVE16=0: fp32 saves.
        # [r5] also on a partitioned mesh (the interior / boundary launches write row ranges of the same two-byte tensors;
        # tests/test_partitioned_hip_multirank.py::test_partitioned_bf16_two_byte_saves_world4_vs_mixed_oracle)
        save16 = (relu_bits and prec == 1 and (split or spec.rope) and _os.environ.get("MGN_SAVE16", "1") != "0"
                  and _os.environ.get("MGN_FUSED_BWD", "0") != "1")
        # ---- packed units of all rounds, one launch.  Per round:
        #   edge  [We0|e (, We0|x_dst, We0|x_src with RoPE), We1 .. We_{NL-1}]
        #   node  [Wn0|x, Wn0|agg, Wn1 .. Wn_{NL-1}, NEXT round's We0|x_dst, We0|x_src (split)]
        #   gate  [W_gate]
        ne = NL + (2 if spec.rope else 0)
        un0, up0 = ne, ne + NL + 1
        ug = up0 + 2
        NU = ug + (1 if spec.gate else 0)
        if x6:
            pk = torch.empty((L * NU + 2) * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
            pk0 = pk.data_ptr()

            def unit(i, u):
                return pk0 + (i * NU + u) * _capi.WPACK_BYTES

            blocks = []
            for i in range(L):
                We, _, _, Wn, _, _, gpar = _split_block(P[PB * i: PB * (i + 1)], spec)
                We0, Wn0 = We[0].data_ptr(), Wn[0].data_ptr()
                srcs = [(We0, 3 * H)] + ([(We0 + 4 * H, 3 * H), (We0 + 8 * H, 3 * H)] if spec.rope else [])
                srcs += [(We[l].data_ptr(), H) for l in range(1, NL)]
                srcs += [(Wn0, 2 * H), (Wn0 + 4 * H, 2 * H)] + [(Wn[l].data_ptr(), H) for l in range(1, NL)]
                for u, (sa, ld) in enumerate(srcs):
                    blocks.append((sa, ld, False, unit(i, u)))
                if split and i + 1 < L:
                    Wnx = P[PB * (i + 1)].data_ptr()
                    blocks += [(Wnx + 4 * H, 3 * H, False, unit(i, up0)), (Wnx + 8 * H, 3 * H, False, unit(i, up0 + 1))]
                if spec.gate:
                    blocks.append((gpar[0].data_ptr(), H, False, unit(i, ug)))
            if split:
                W00 = P[0].data_ptr()
                blocks += [(W00 + 4 * H, 3 * H, False, unit(L, 0)), (W00 + 8 * H, 3 * H, False, unit(L, 1))]
            wpack(blocks, dev)
        def project(i, xi):
            """round i's node projections Pd = x W_d^T, Ps = x W_s^T from the node latents (two one-unit launches)"""
            W0 = P[PB * i]
            Pd_, Ps_ = torch.empty(Nn, H, **f), torch.empty(N, H, **f)
            for slab, dst_t in ((1, Pd_), (2, Ps_)):
                u = (unit(L, slab - 1) if i == 0 else unit(i - 1, up0 + slab - 1)) if x6 else None
                # (packed path: the kernel multiplies by the packed unit, the fp32 slab is never dereferenced -- no copy of it)
                mlp_fwd(Nn, H, [(xi, None, H)], [None if x6 else W0[:, slab * H:(slab + 1) * H].contiguous()], [None], None, H, None, dst_t,
                        wpk=[u] if x6 else (), precision=prec)
            return Pd_, Ps_

        def run_round(i, x, e, Pd, Ps, save, make_posts, pending):
            """one GraphNetBlock round on the engine; ``save``: training-mode launches, activations returned in S"""
            We, be, se, Wn, bn, sn, gpar = _split_block(P[PB * i: PB * (i + 1)], spec)
            e_new = torch.empty(E, H, **f)
            x_new = torch.empty(Nn, H, **f)
            agg = torch.empty(Nn, H, **f)
            m = None if fuse_agg else torch.empty(E, H, **f)
            He = Hn = Me = Mn = Ze = Zn = None
            Ue = Re = Un = Rn = None
            if save:
                He = [torch.empty(E, H, dtype=torch.bfloat16 if save16 else torch.float32, device=dev) for _ in range(NL - 1)]
                Hn = [torch.empty(Nn, H, **f) for _ in range(NL - 1)]
                if spec.layer_norm:
                    Ue, Re = torch.empty(E, H, **f), torch.empty(E, **f)
                    Un, Rn = torch.empty(Nn, H, **f), torch.empty(Nn, **f)
                if relu_bits:  # ReLU masks as bits (16 B per row and layer): what the packed backward chain reads
                    Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(NL - 1)]
                    Mn = [torch.empty(Nn, 4, dtype=torch.int32, device=dev) for _ in range(NL - 1)]
                if act == 1:   # SiLU: the pre-activations
                    Ze = [torch.empty(E, H, **f) for _ in range(NL - 1)]
                    Zn = [torch.empty(Nn, H, **f) for _ in range(NL - 1)]
            xj = None
            if spec.rope:
                xj = torch.empty(E, H, **f)
                rope_gather(x, pos, rope_inv_freq, topo, spec.rope_axes, xj)

            # R3: m = edge_block(cat[e, x[dst], x[src]]);  e' = e + m     (layers.py:1017-1028,1039)
            def edge_rows(r0, r1, rowptr, part_rows):
                """rows [r0, r1) of the dst-sorted edge arrays (one launch); with the aggregation fused"""
                sl = slice(r0, r1)
                M = r1 - r0
                sv = lambda ts: [t[sl] for t in ts] if ts is not None else None  # noqa: E731
                seg = None
                if fuse_agg:
                    part = torch.empty((M + 15) // 16, 2, H, **f)
                    seg = (topo.dst_s[sl], rowptr, agg, part)
                y = m[sl] if m is not None else None
                common = dict(saveM=sv(Me), saveZ=sv(Ze), act=act, precision=2 if (save16 and He is not None) else prec, seg=seg)
                if spec.rope:
                    mlp_fwd(M, H, [(e[sl], None, H), (x, topo.dst_s[sl], H), (xj[sl], None, H)], We, be, se, H, e[sl], e_new[sl], y,
                            sv(He), Ue[sl] if Ue is not None else None, Re[sl] if Re is not None else None,
                            wpk=[unit(i, u) for u in range(ne)] if x6 else (), **common)
                elif split:
                    mlp_fwd(M, H, [(e[sl], None, H)], We, be, se, H, e[sl], e_new[sl], y, sv(He),
                            Ue[sl] if Ue is not None else None, Re[sl] if Re is not None else None, ldw0=3 * H,
                            adds=[(Pd, topo.dst_s[sl]), (Ps, topo.src_s[sl])],
                            wpk=[unit(i, u) for u in range(ne)] if x6 else (), **common)
                else:
                    mlp_fwd(M, H, [(e[sl], None, H), (x, topo.dst_s[sl], H), (x, topo.src_s[sl], H)], We, be, se, H, e[sl], e_new[sl], y,
                            sv(He), Ue[sl] if Ue is not None else None, Re[sl] if Re is not None else None, act=act, saveZ=sv(Ze),
                            precision=prec)
                if fuse_agg:
                    n0, n1 = part_rows
                    seg_fix(rowptr[n0:n1 + 1], seg[3], agg[n0:n1])   # the kernel summed inside its wave tiles

            if E > 0:
                if halo is None:
                    edge_rows(0, E, topo.rowptr_dst, (0, Nn))
                else:
                    # rows whose destination is an interior node need no remote data: they run while the
                    # exchange of this round's ghost projections is in flight
                    Ei, ni = halo.n_interior_edges, halo.n_interior
                    if Ei > 0:
                        edge_rows(0, Ei, topo.rowptr_dst, (0, ni))
                    halo.finish_forward(pending)
                    pending = None
                    if E > Ei:
                        edge_rows(Ei, E, halo.rowptr_bnd(topo), (ni, Nn))
                    elif fuse_agg and Nn > ni:
                        agg[ni:].zero_()
            # R4: agg = segment-sum of m over dst                          (layers.py:1031-1037)
            if not fuse_agg:
                segsum_topo(m, topo, "dst", agg, Nn)
            # gate on the aggregate (layers.py:1091-1098): agg * sigmoid(gate_proj(x) + phi * gate_pos)
            agg_in, gate_t = agg, None
            if spec.gate:
                Wg, bg, gpos = gpar
                G = torch.empty(Nn, H, **f)
                mlp_fwd(Nn, H, [(x, None, H)], [Wg], [bg], None, H, None, G, wpk=[unit(i, ug)] if x6 else (), precision=prec)
                agg_in = torch.empty(Nn, H, **f)
                gate_t = torch.empty(Nn, H, **f) if save else None
                gate_fwd(G, phi, gpos if phi is not None else None, agg, gate_t, agg_in)
            # R5: x' = x + node_block(cat[x, agg])                         (layers.py:1100-1101,1040)
            posts, Pd_n, Ps_n = (), None, None
            if make_posts:
                W0n = P[PB * (i + 1)]
                Pd_n, Ps_n = torch.empty(Nn, H, **f), torch.empty(N, H, **f)
                posts = [(W0n.data_ptr() + 4 * H, Pd_n), (W0n.data_ptr() + 8 * H, Ps_n)]
            wn = ()
            if x6:
                wn = [unit(i, un0 + u) for u in range(NL + 1)] + ([unit(i, up0), unit(i, up0 + 1)] if posts else [])
            if Nn > 0:
                mlp_fwd(Nn, H, [(x, None, H), (agg_in, None, H)], Wn, bn, sn, H, x, x_new, None, Hn, Un, Rn,
                        posts=posts, post_ldw=3 * H, wpk=wn, saveM=Mn, saveZ=Zn, act=act, precision=prec)
            if halo is not None and Ps_n is not None:
                pending = halo.start_forward(Ps_n)
            S = None
            if save:
                S = dict(x=x, e=e, agg=agg, agg_in=agg_in, gate=gate_t, xj=xj, He=He, Ue=Ue, Re=Re, Hn=Hn, Un=Un, Rn=Rn,
                         Me=Me, Mn=Mn, Ze=Ze, Zn=Zn)
            return x_new, e_new, Pd_n, Ps_n, S, pending

        # activation recompute (see set_activation_recompute): the forward keeps only every round's INPUTS
        # and runs the inference-mode launches; the backward re-runs a round in training mode right before
        # differentiating it.  2.5 KB per edge and round shrink to 0.5 KB.
        n_rec = recompute_rounds(E, Nn, H, NL, L, act, dev, save16) if (need and halo is None) else 0
        recompute = n_rec > 0
        Pd = Ps = None
        pending = None  # halo exchange in flight
        if split:
            Pd, Ps = project(0, x)
            if halo is not None:
                pending = halo.start_forward(Ps)
        saved = []
        for i in range(L):
            x_in, e_in = x, e
            x, e, Pd, Ps, S, pending = run_round(i, x, e, Pd, Ps, need and i >= n_rec, split and i + 1 < L, pending)
            if need:
                saved.append(S if i >= n_rec else dict(x=x_in, e=e_in))
        # the closures address the packed weights by raw pointer: keep the buffer alive with them
        ctx.rerun = (run_round, project, pk if x6 else None) if recompute else None
        ctx.topo, ctx.L, ctx.saved_acts, ctx.prec, ctx.spec, ctx.halo = topo, L, (saved if need else None), prec, spec, halo
        ctx.aux = (pos, phi, rope_inv_freq, x6, split)
        ctx.save16 = save16
        ctx.P_ops = P if P is not P_params else None   # rounded kernel operands of the bf16 mode off the packed path
        ctx.save_for_backward(*P_params)  # version-checked by autograd (an optimiser step in between is an error)
        return x, e

    @staticmethod
    def backward(ctx, dx, de):
        topo, L, saved, prec, spec, halo = ctx.topo, ctx.L, ctx.saved_acts, ctx.prec, ctx.spec, ctx.halo
        pos, phi, rope_inv_freq, x6, split = ctx.aux
        if saved is None:
            raise RuntimeError("ProcessorFunction: the saved activations were released by an earlier backward pass "
                               "(retain_graph is not supported: ~2.5 KB per edge and round are freed eagerly), "
                               "or the forward ran under no_grad")
        P_params = list(ctx.saved_tensors)
        P = ctx.P_ops if ctx.P_ops is not None else P_params   # kernel operands (rounded copies in the bf16 mode off the packed path)
        NL, act, PB = spec.nb_layers, spec.act_id, spec.per_block
        N, E = topo.N, topo.E
        if not topo.resolved:  # lazily built topology: the forward pass is queued, the flags have long arrived
            topo.resolve()
        if L == 0:
            return (dx, de, None, None, None, None, None, None, None)
        dev = P[0].device
        H = P[1].numel()
        Nn = saved[0]["x"].shape[0]
        f = dict(dtype=torch.float32, device=dev)
        dx = _f32c(dx) if dx is not None else torch.zeros(Nn, H, **f)
        de = _f32c(de) if de is not None else torch.zeros(E, H, **f)
        empty = (E == 0 or Nn == 0)  # some launches return early at M == 0 and would leave their outputs unwritten
        mk = torch.zeros if empty else torch.empty
        dZn = [mk(Nn, H, **f) for _ in range(NL)]
        d16 = bool(getattr(ctx, "save16", False)) and x6
        # (two-byte rows of dZe[1..]: written by the chain with precision 2, read by the weight gradients with lda = -128)
        # dZe[0] too where its only other reader is the two-segment-sum launch (default blocks, no hubs, no fused destination scatter)
        z16 = (d16 and not spec.rope and H == 128 and not topo.has_hubs and _os.environ.get("MGN_FUSED_SD") is None
               and _os.environ.get("MGN_SAVE16", "1") != "2")
        dZe = [mk(E, H, dtype=torch.bfloat16, device=dev) if (d16 and (l > 0 or z16)) else mk(E, H, **f) for l in range(NL)]
        dAgg, Sd, Ss = mk(Nn, H, **f), mk(N, H, **f), mk(N, H, **f)
        # Weight gradients on a SIDE STREAM (MGN_WGRAD_STREAM=1): dW of round i depends on nothing the rest of
        # the backward pass waits for, so its launch can fill the bubbles of the main stream (launch gaps, the
        # tail of the persistent chain kernels, the one-tile node launches).  Its operands (dZ, Sd, Ss) are then
        # double-buffered across rounds and the streams are joined by events.
        side = None
        # (not under activation recompute: a round's recomputed activations are dropped when the loop rebinds S,
        # while its weight-gradient launch may still read them on the side stream)
        if _os.environ.get("MGN_WGRAD_STREAM") is not None and halo is None and not empty and not spec.gate and ctx.rerun is None:
            side = _side_stream(dev)
            wsets = [(dZn, dZe, Sd, Ss), ([mk(Nn, H, **f) for _ in range(NL)], [torch.empty_like(t) for t in dZe], mk(N, H, **f), mk(N, H, **f))]
            wdone = [None, None]
            main = torch.cuda.current_stream(dev)
        dx_buf, de_buf = [mk(Nn, H, **f), mk(Nn, H, **f)], [mk(E, H, **f), mk(E, H, **f)]
        # [r5] Weight gradients of SEVERAL rounds in one launch (up to MAX_WGRAD_JOBS jobs = 4 rounds of the default block): a
        # round's dW depends on nothing the rest of the backward pass waits for, so its jobs can wait until a launch is full -- one
        # k_wgrad_pc + one k_wgrad_red per four rounds instead of per round (a one-mesh step: 39 -> 11 weight-gradient launches
        # of ~14 us each).  The operands of a waiting round (dZ rows, the two scatters, the round's saved activations) stay alive
        # until the launch and are read back from HBM instead of the Infinity Cache, so only steps whose rounds are small do it
        # (MGN_WGRAD_BATCH_MB, default 256 MB per waiting round; 0: one launch per round.  Measured, tools/ab_wbatch.sh: one mesh
        # per step 3.03 -> 2.88 ms; the 16-mesh batch -- 953 MB per round -- 12.60-12.64 against 12.53-12.70 ms: the launches it saves
        # there are paid back by dZ rows that no longer come out of the cache the chain kernel just wrote them through); not with a side
        # stream, the partitioned mesh or the fused edge backward.  A grad-ready listener hears of a waiting round when its launch
        # has been queued (the same launches with or without a listener: a rank's gradients do not depend on how they are reduced).
        per_round_mb = 4.0 * H * ((2 * NL + 1) * E + (NL + 7) * Nn) / 2**20
        wb_rounds = max(1, _capi.MAX_WGRAD_JOBS // (2 * NL + 3 + (1 if spec.gate else 0)))
        defer_w = (x6 and side is None and halo is None and not empty and wb_rounds > 1
                   and _os.environ.get("MGN_FUSED_BWD", "0") != "1"
                   and per_round_mb <= float(_os.environ.get("MGN_WGRAD_BATCH_MB", "256")))
        w_pend, w_rounds = [], 0
        hook_pend = []   # grad-ready reports of rounds whose weight-gradient launch has not been queued yet
        dzn_round = {}

        def dzn_of(r):
            """dZn rows of round r: a set of its own per round while weight-gradient jobs wait, the two alternating sets otherwise"""
            if not defer_w:
                return dZn_sets[r & 1]
            if r not in dzn_round:
                dzn_round[r] = [mk(Nn, H, **f) for _ in range(NL)]
            return dzn_round[r]

        grads: List[Optional[torch.Tensor]] = [None] * (PB * L)
        nb = H // 16
        HH = H * H
        k_ = spec.mlp_params
        # ---- transposed weights of every round.  Packed: per round
        #   [Wn_{NL-1}^T .. Wn_1^T, Wn0|agg^T] [We_{NL-1}^T .. We_1^T, We0|e^T] [Wn0|x^T, We0|x_dst^T, We0|x_src^T] [W_gate^T]
        NU = 2 * NL + 3 + (1 if spec.gate else 0)
        ukn, uke, ukx, ukg = 0, NL, 2 * NL, 2 * NL + 3
        if x6:
            pk = torch.empty(L * NU * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
            pk0 = pk.data_ptr()

            def unit(i, u):
                return pk0 + (i * NU + u) * _capi.WPACK_BYTES

            blocks = []
            for i in range(L):
                We, _, _, Wn, _, _, gpar = _split_block(P[PB * i: PB * (i + 1)], spec)
                We0, Wn0 = We[0].data_ptr(), Wn[0].data_ptr()
                srcs = [(Wn[l].data_ptr(), H) for l in range(NL - 1, 0, -1)] + [(Wn0 + 4 * H, 2 * H)]
                srcs += [(We[l].data_ptr(), H) for l in range(NL - 1, 0, -1)] + [(We0, 3 * H)]
                srcs += [(Wn0, 2 * H), (We0 + 4 * H, 3 * H), (We0 + 8 * H, 3 * H)]
                if spec.gate:
                    srcs.append((gpar[0].data_ptr(), H))
                for u, (sa, ld) in enumerate(srcs):
                    blocks.append((sa, ld, True, unit(i, u)))
            wpack(blocks, dev)
            wt = None
        else:
            # W^T operands of every round in one buffer, filled by one batched transpose launch: per round
            #   [WTn_1..NL-1 | WTe_1..NL-1 | WT0n_agg | WT0e_e | Wcat (H x 3H) | WgT]
            nblk = 2 * (NL - 1) + 2 + 3 + (1 if spec.gate else 0)
            wt = torch.empty(L, nblk, H, H, **f)
            tb = []
            for i in range(L):
                We, _, _, Wn, _, _, gpar = _split_block(P[PB * i: PB * (i + 1)], spec)
                base = wt.data_ptr() + 4 * (i * nblk * HH)
                We0, Wn0 = We[0].data_ptr(), Wn[0].data_ptr()
                for k, l in enumerate(range(1, NL)):
                    tb.append((Wn[l].data_ptr(), H, base + 4 * (k * HH), H))                    # WTn[l]
                    tb.append((We[l].data_ptr(), H, base + 4 * ((NL - 1 + k) * HH), H))         # WTe[l]
                o = 2 * (NL - 1)
                tb.append((Wn0 + 4 * H, 2 * H, base + 4 * (o * HH), H))                         # (W0n[:, H:])^T
                tb.append((We0, 3 * H, base + 4 * ((o + 1) * HH), H))                           # (W0e[:, :H])^T
                cat = base + 4 * ((o + 2) * HH)                                                 # Wcat [H, 3H]
                tb.append((Wn0, 2 * H, cat, 3 * H))                                             # (W0n[:, :H])^T
                tb.append((We0 + 4 * H, 3 * H, cat + 4 * H, 3 * H))                             # (W0e[:, H:2H])^T
                tb.append((We0 + 8 * H, 3 * H, cat + 8 * H, 3 * H))                             # (W0e[:, 2H:])^T
                if spec.gate:
                    tb.append((gpar[0].data_ptr(), H, base + 4 * ((o + 5) * HH), H))
            transpose_blocks(tb, H, dev)
        gs = [[(torch.zeros_like if empty else torch.empty_like)(t) for t in P[PB * i: PB * (i + 1)]] for i in range(L)]
        # packed path: the dX launch of round i and the node chain of round i-1 work on the same
        # rows -> one launch (front stage of mgn_mlp_bwd); dZn double buffered across rounds
        # (a launch costs as many tile times as it has GEMM units -- 88 us fused against 57 + 30 us at 30k node rows -- but it is one
        # launch tail less per round: +0.7-0.9 % on the bench step in two alternating A/B pairs (74.86 / 74.77 against 74.36 / 74.06
        # steps/s), so it is ON by default since round 3; MGN_FRONT=0 keeps the two launches; tests/test_hip_parity.py compares both.
        # Not under activation recompute -- the previous round's activations do not exist yet -- nor on a partitioned mesh.)
        fuse = (x6 and _os.environ.get("MGN_FRONT", "1") != "0" and spec == DEFAULT_SPEC and halo is None and ctx.rerun is None)
        dZn_sets = [dZn, [torch.empty(Nn, H, **f) for _ in range(NL)] if fuse and L > 1 else dZn]
        node_done = False
        # scale-gradient partials of all chain launches, reduced by ONE launch at the end (MGN_NO_DEFER: per launch)
        deferred = [] if _os.environ.get("MGN_NO_DEFER") is None else None

        for i in reversed(range(L)):
            q = P[PB * i: PB * (i + 1)]
            We, be, se, Wn, bn, sn, gpar = _split_block(q, spec)
            S = saved[i]
            saved[i] = None  # a round's activations are released as soon as it is differentiated
            if ctx.rerun is not None and "He" not in S:  # activation recompute: this round's training-mode forward, now
                run_round, project = ctx.rerun[:2]
                Pd_i, Ps_i = project(i, S["x"]) if split else (None, None)
                S = run_round(i, S["x"], S["e"], Pd_i, Ps_i, True, False, None)[4]
                saved[i] = None
            x, e, agg, agg_in = S["x"], S["e"], S["agg"], S["agg_in"]
            He, Ue, Re, Hn, Un, Rn = S["He"], S["Ue"], S["Re"], S["Hn"], S["Un"], S["Rn"]
            g = gs[i]
            gWe = [g[2 * l] for l in range(NL)]
            gbe = [g[2 * l + 1] for l in range(NL)]
            gWn = [g[k_ + 2 * l] for l in range(NL)]
            gbn = [g[k_ + 2 * l + 1] for l in range(NL)]
            gse = g[2 * NL] if spec.layer_norm else None
            gsn = g[k_ + 2 * NL] if spec.layer_norm else None
            dZn = dzn_of(i)
            if defer_w:   # this round's operand rows must outlive the round: fresh ones (the caching allocator recycles them)
                dZe = [torch.empty_like(t) for t in dZe]
                Sd, Ss = mk(N, H, **f), mk(N, H, **f)
            if side is not None:
                dZn, dZe, Sd, Ss = wsets[i & 1]
                if wdone[i & 1] is not None:  # the weight gradients of round i + 2 still read this set
                    main.wait_event(wdone[i & 1])
            if x6:
                # never dereferenced on the packed path
                WTn = WTe = [None] * NL
                WT0n_agg = WT0e_e = Wcat = WgT = None
                kn = [unit(i, ukn + u) for u in range(NL)]
                ke = [unit(i, uke + u) for u in range(NL)]
                kx = [unit(i, ukx + u) for u in range(3)]
                kg = [unit(i, ukg)] if spec.gate else []
            else:
                w = wt[i]
                o = 2 * (NL - 1)
                WTn = [None] + [w[k] for k in range(NL - 1)]
                WTe = [None] + [w[NL - 1 + k] for k in range(NL - 1)]
                WT0n_agg, WT0e_e = w[o], w[o + 1]
                Wcat = w[o + 2:o + 5].reshape(H, 3 * H)
                WgT = w[o + 5] if spec.gate else None
                kn = ke = kx = kg = ()
            # node MLP chain: dX' -> dZn[NL-1..0], dAgg = W0n[:,H:]^T dZn0  (already done by the
            # previous iteration's fused launch except for the last round)
            if not node_done and Nn > 0:
                mlp_bwd(Nn, H, NL, dx, None, None, H, Un, Rn, sn, Hn, WTn, dZn, [(WT0n_agg, None, dAgg)],
                        [None] * NL, gsn, wpk=kn, Ms=S["Mn"], Zs=S["Zn"], act=act, precision=prec, defer=deferred)
            dG = None
            if spec.gate:  # d(agg * gate): dAgg <- dAggG * gate, dG = dAggG * agg * gate (1 - gate)
                dG = torch.empty(Nn, H, **f)
                gate_bwd(dAgg, agg, S["gate"], dAgg, dG)
            # edge MLP chain: dM = dE' + dAgg[dst] -> dZe[NL-1..0], dE = dE' + W0e[:, :H]^T dZe0
            de_new = de_buf[0] if de.data_ptr() != de_buf[0].data_ptr() else de_buf[1]
            # ... with the destination-side scatter of dZe[0] fused into the chain (rows are dst-sorted: the
            # forward's segmented scan), packed fp32-grade ReLU path only
            # (measured neutral on the bench workload -- the chain kernel gains the 10 us the stand-alone sum
            # loses -- so it is opt-in: MGN_FUSED_SD=1; tests/test_hip_parity.py covers it)
            fuse_sd = x6 and act == 0 and prec == 0 and E > 0 and _os.environ.get("MGN_FUSED_SD") is not None and not topo.has_hubs
            # the edge chain AND its four E-row weight gradients in one kernel (mgn_edge_bwd_fused: dZ1..dZ3 never reach
            # memory; 763 MB per launch instead of 1594 MB).  Built, parity-green -- and measured SLOWER than the split
            # launches at the bench size (545-560 us against 390-405 us: one wave per SIMD serialises what two waves overlap,
            # DESIGN.md section 4.6), so it is opt-in: MGN_FUSED_BWD=1
            fused_bwd = (x6 and act == 0 and NL == 4 and spec.layer_norm and not spec.rope and E > 0 and Nn > 0 and not fuse_sd
                         and side is None and _os.environ.get("MGN_FUSED_BWD", "0") == "1")
            if fused_bwd:
                edge_bwd_fused(E, de, dAgg, topo.dst_s, Ue, Re, se, [e] + list(He), S["Me"], ke, de_new, dZe[0],
                               [(gWe[0], 0, 3 * H), (gWe[1], 0, H), (gWe[2], 0, H), (gWe[3], 0, H)], gbe, gse, precision=prec)
            elif E > 0:
                seg = None
                if fuse_sd:
                    part_b = torch.empty((E + 15) // 16, 2, H, **f)
                    seg = (topo.dst_s, topo.rowptr_dst, Sd, part_b)
                mlp_bwd(E, H, NL, de, dAgg, topo.dst_s, H, Ue, Re, se, He, WTe, dZe, [(WT0e_e, de, de_new)],
                        [None] * NL, gse, wpk=ke, Ms=S["Me"], Zs=S["Ze"], act=act, precision=(3 if z16 else 2) if d16 else prec, defer=deferred, seg=seg)
                if fuse_sd:
                    seg_fix(topo.rowptr_dst, part_b, Sd)
            else:
                de_new = de
            # scatter of the first-layer pre-activations' grads onto dst / src nodes
            dx_res = dx
            if spec.rope:
                # the source slab saw ROTATED rows: T = W_s^T dz0 per edge, rotated back and summed over
                # the edges of each source node, straight into the residual of the dX launch
                if not fuse_sd:
                    segsum_topo(dZe[0], topo, "dst", Sd)
                T = torch.empty(E, H, **f)
                WsT = None if x6 else Wcat[:, 2 * H:].contiguous()
                mlp_fwd(E, H, [(dZe[0], None, H)], [WsT], [None], None, H, None, T, wpk=[kx[2]] if x6 else (), precision=prec)
                dx_res = torch.empty(Nn, H, **f)
                rope_scatter(T, pos, rope_inv_freq, topo, spec.rope_axes, dx, dx_res)
            elif fuse_sd:
                segsum_topo(dZe[0], topo, "src", Ss)
            elif H == 128 and not topo.has_hubs:
                segsum2(dZe[0], topo.rowptr_dst, None, Sd, topo.rowptr_src, topo.perm_src, Ss)
            else:
                segsum_topo(dZe[0], topo, "dst", Sd)
                segsum_topo(dZe[0], topo, "src", Ss)
            back = None
            if halo is not None:  # ghost rows of Ss go back to their owners while the E-row weight gradients run
                back = halo.start_backward(Ss)
            # weight gradients: dW = dZ^T X
            # (A = dZ, B = layer input, dW slab[, db = bias gradient as a by-product])
            ejobs = [(dZe[0], -H if dZe[0].dtype == torch.bfloat16 else H, nb, e, H, nb, H, gWe[0], 0, 3 * H, gbe[0])]
            if spec.rope:
                ejobs.append((dZe[0], H, nb, S["xj"], H, nb, H, gWe[0], 2 * H, 3 * H))
            njobs = [
                (dZn[0], H, nb, x, H, nb, H, gWn[0], 0, 2 * H, gbn[0]),
                (dZn[0], H, nb, agg_in, H, nb, H, gWn[0], H, 2 * H),
                (Sd[:Nn], H, nb, x, H, nb, H, gWe[0], H, 3 * H),
            ]
            if not spec.rope:
                njobs.append((Ss[:Nn], H, nb, x, H, nb, H, gWe[0], 2 * H, 3 * H))
            for l in range(1, NL):
                njobs.append((dZn[l], H, nb, Hn[l - 1], H, nb, H, gWn[l], 0, H, gbn[l]))
                ejobs.append((dZe[l], -H if dZe[l].dtype == torch.bfloat16 else H, nb, He[l - 1], -H if He[l - 1].dtype == torch.bfloat16 else H, nb, H,
                              gWe[l], 0, H, gbe[l]))
            if spec.gate:
                gWg, gbg, gpos = g[2 * k_], g[2 * k_ + 1], g[2 * k_ + 2]
                njobs.append((dG, H, nb, x, H, nb, H, gWg, 0, H, gbg))
            if fused_bwd:
                ejobs = []
            if halo is not None:
                if E > 0 and ejobs:
                    wgrad(ejobs, dev, prec)
                halo.finish_backward(back, Ss)
                if Nn > 0:
                    wgrad(njobs, dev, prec)
            elif E > 0 and Nn > 0:
                # one launch for the whole round: node first-layer jobs, edge first-layer jobs, the two
                # scattered slabs, then the deeper layers alternating node / edge (same plan as round 1)
                n_sc = 1 if spec.rope else 2
                deep = [j for pair in zip(njobs[2 + n_sc:2 + n_sc + NL - 1], ejobs[len(ejobs) - (NL - 1):]) for j in pair]
                alljobs = njobs[:2] + ejobs[:len(ejobs) - (NL - 1)] + njobs[2:2 + n_sc] + deep + (njobs[-1:] if spec.gate else [])
                if fused_bwd:  # the E-row jobs ran inside the fused kernel
                    alljobs = njobs
                if side is not None:
                    ready = torch.cuda.Event()
                    ready.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(ready)
                        wgrad(alljobs, dev, prec)
                        wdone[i & 1] = torch.cuda.Event()
                        wdone[i & 1].record(side)
                elif defer_w:
                    w_pend += alljobs
                    w_rounds += 1
                    if w_rounds == wb_rounds or i == 0:
                        wgrad(w_pend, dev, prec)
                        w_pend, w_rounds = [], 0
                        for rep in hook_pend:
                            _grad_ready_hook(rep)
                        hook_pend = []
                        for r in [r for r in dzn_round if r >= i]:
                            del dzn_round[r]
                else:
                    wgrad(alljobs, dev, prec)
            if spec.gate:
                if phi is not None:  # d gate_pos = sum_n phi[n] * dG[n, :]  (a [H, 1] weight-gradient job)
                    tmp = torch.empty(H, 16, **f)
                    wgrad([(dG, H, nb, phi.reshape(-1, 1), 1, 1, 1, tmp, 0, 16)], dev)
                    gpos.copy_(tmp[:, 0])
                else:
                    gpos.zero_()
            # dX = dX' + W0n[:, :H]^T dZn0 + W0e[:, H:2H]^T Sd + W0e[:, 2H:]^T Ss  (+ W_gate^T dG)
            dx_new = dx_buf[0] if dx.data_ptr() != dx_buf[0].data_ptr() else dx_buf[1]
            if fuse and i > 0:  # ... fused with the node chain of round i-1 (same rows)
                Sp = saved[i - 1]
                _, _, _, _, _, snp, _ = _split_block(P[PB * (i - 1): PB * i], spec)
                kn_p = [unit(i - 1, ukn + u) for u in range(NL)]
                mlp_bwd(Nn, H, NL, dx, None, None, H, Sp["Un"], Sp["Rn"], snp, Sp["Hn"], WTn, dzn_of(i - 1),
                        [(WT0n_agg, None, dAgg)], [None] * NL, gs[i - 1][k_ + 2 * NL], wpk=kx + kn_p, Ms=Sp["Mn"], precision=prec,
                        front=([dZn[0], Sd, Ss], dx, dx_new), defer=deferred)
                node_done = True
            elif Nn > 0:
                if spec.rope:
                    Wc2 = None if x6 else Wcat[:, :2 * H].contiguous()
                    mlp_fwd(Nn, H, [(dZn[0], None, H), (Sd[:Nn], None, H)], [Wc2], [None], None, H, dx_res, dx_new,
                            wpk=kx[:2] if x6 else (), precision=prec)
                else:
                    mlp_fwd(Nn, H, [(dZn[0], None, H), (Sd[:Nn], None, H), (Ss[:Nn], None, H)], [Wcat], [None], None, H, dx, dx_new,
                            wpk=kx, precision=prec)
                if spec.gate:
                    dx_g = torch.empty(Nn, H, **f)
                    mlp_fwd(Nn, H, [(dG, None, H)], [WgT], [None], None, H, dx_new, dx_g, wpk=kg, precision=prec)
                    dx_new = dx_g
                node_done = False
            else:
                dx_new = dx
            grads[PB * i: PB * (i + 1)] = g
            if _grad_ready_hook is not None and halo is None and side is None:
                # this round's weight / bias gradients are final once their launches are queued (the two RMSNorm scale gradients come
                # out of the deferred reduction at the very end): a data-parallel wrapper may start reducing them now
                late = {2 * NL, k_ + 2 * NL} if (spec.layer_norm and deferred is not None) else set()
                rep = [(P_params[PB * i + t], g[t]) for t in range(len(g)) if t not in late]
                if defer_w and w_rounds > 0:     # this round's jobs are still waiting for their launch
                    hook_pend.append(rep)
                else:
                    _grad_ready_hook(rep)
            dx, de = dx_new, de_new
        if side is not None:
            for ev in wdone:
                if ev is not None:
                    main.wait_event(ev)
        colred_batch(deferred, dev)
        ctx.saved_acts = None
        return (dx, de, None, None, None, None, None, None, None, *grads)
