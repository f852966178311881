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
#: bf16, one term, fp32 accumulate; everything stored stays fp32) -- the processor's GEMMs only.
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


def _stream(dev: torch.device) -> int:
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
class Topology:
    """dst-sorted (CSR) edge order of one ``edge_index`` plus the src-grouped view
    the backward scatter needs.  Built on the device by ``mgn_csr_build``; cached
    per ``edge_index`` tensor by :func:`get_topology` (mesh topology is static
    along a trajectory)."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int):
        _require_device(edge_index)
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must have shape [2, E]")
        dev = edge_index.device
        ei = edge_index.to(torch.int64).contiguous()
        E, N = int(ei.shape[1]), int(num_nodes)
        self.N, self.E, self.device = N, E, dev
        self.rowptr_dst, self.perm_dst = csr_build(ei[1], N)
        p = self.perm_dst.long()
        self.src_s = ei[0][p].to(torch.int32).contiguous()
        self.dst_s = ei[1][p].to(torch.int32).contiguous()
        self.rowptr_src, self.perm_src = csr_build(self.src_s.long(), N)
        self._inv = None

    @property
    def inv_perm(self) -> torch.Tensor:
        """position in the sorted order of each original edge id"""
        if self._inv is None:
            inv = torch.empty(self.E, dtype=torch.int64, device=self.device)
            inv[self.perm_dst.long()] = torch.arange(self.E, device=self.device)
            self._inv = inv
        return self._inv


_topo_cache: dict = {}


def get_topology(edge_index: torch.Tensor, num_nodes: int) -> Topology:
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes), str(edge_index.device))
    hit = _topo_cache.get(key)
    if hit is not None and hit[0]() is edge_index:
        return hit[1]
    import weakref

    topo = Topology(edge_index, num_nodes)
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
    with torch.cuda.device(src.device):
        rc = _capi.lib().mgn_segsum2(_ptr(src), _ptr(rowptr0), _ptr(perm0), _ptr(out0), _ptr(rowptr1), _ptr(perm1), _ptr(out1),
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
            seg=None):
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
    dev = out.device
    with torch.cuda.device(dev):
        rc = _capi.lib().mgn_mlp_fwd(C.byref(a), _stream(dev))
    _capi.check(rc, "mgn_mlp_fwd")


def mlp_bwd(M: int, H: int, NL: int, dOut: torch.Tensor, dOut2, idx2, out_w: int, U, R, scale,
            Hs: Sequence[torch.Tensor], WT: Sequence[Optional[torch.Tensor]], dZ: Sequence[Optional[torch.Tensor]],
            din: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]],
            db: Sequence[Optional[torch.Tensor]], dscale: Optional[torch.Tensor], wpk: Sequence[int] = (),
            Ms: Optional[Sequence[torch.Tensor]] = None, precision: int = 0, front=None, defer: Optional[list] = None):
    """``front`` = (rows[<=3] each [M,H], resid[M,H] or None, out[M,H] or None): the fused front
    stage of the packed kernel, dY = resid + sum_p wpk[p] . rows[p] (then ``dOut`` is ignored)."""
    L = _capi.lib()
    a = _capi.MlpBwdArgs()
    a.M, a.H, a.NL = M, H, NL
    a.dOut, a.dOut2, a.idx2, a.out_w = _ptr(dOut), _ptr(dOut2), _ptr(idx2), out_w
    a.U, a.R, a.scale, a.eps = _ptr(U), _ptr(R), _ptr(scale), EPS
    for l, h in enumerate(Hs):
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


# ``ctx.needs_input_grad`` mirrors ``tensor.requires_grad`` even under ``torch.no_grad()``, and
# grad mode is always off INSIDE ``Function.forward`` -- so whether activations must be saved for
# a backward pass has to be read BEFORE ``apply``.  Without this an inference forward ran the
# training-mode kernels (4 extra 512-byte stores per row and layer) and kept every round's
# activations alive (270 GB on the 1M-node mesh).
import threading as _threading

_call = _threading.local()


def _saving() -> bool:
    return getattr(_call, "grad", True)


def mlp_apply(*args):
    """MlpFunction.apply with the caller's grad mode recorded."""
    _call.grad = torch.is_grad_enabled()
    try:
        return MlpFunction.apply(*args)
    finally:
        _call.grad = True


def processor_apply(*args):
    """ProcessorFunction.apply with the caller's grad mode recorded."""
    _call.grad = torch.is_grad_enabled()
    try:
        return ProcessorFunction.apply(*args)
    finally:
        _call.grad = True


# ------------------------------------------------------------ generic MLP (R2)
class MlpFunction(torch.autograd.Function):
    """build_mlp forward/backward on the engine (encoders, decoder, stand-alone MLPs).

    apply(x, has_norm, W0, b0, ..., W_{NL-1}, b_{NL-1} [, scale]) -> y[M, out]
    First-layer columns / last-layer rows are zero-padded to multiples of 16 here
    (plumbing) so that the kernels only see aligned weights.
    """

    @staticmethod
    def forward(ctx, x, has_norm, *params):
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
        need = any(ctx.needs_input_grad) and _saving()
        y = torch.empty(M, out_w, dtype=torch.float32, device=dev)
        saveH = [torch.empty(M, H, dtype=torch.float32, device=dev) for _ in range(NL - 1)] if need else None
        U = torch.empty(M, H, dtype=torch.float32, device=dev) if (need and has_norm) else None
        R = torch.empty(M, dtype=torch.float32, device=dev) if (need and has_norm) else None
        if H == 128 and kin < H and out_w == H and NL >= 3 and NL - 1 <= 4 and X6_ENABLED and M > 0:
            # encoder: narrow first layer stand-alone (h1 = relu(W0 x + b0), the activation the
            # backward needs anyway), the full-width layers on the packed split-bf16 path
            h1 = saveH[0] if need else torch.empty(M, H, dtype=torch.float32, device=dev)
            mlp_fwd(M, H, [(x, None, kin)], [Wk[0]], [bk[0]], None, H, None, h1, out_relu=True)
            pk = torch.empty((NL - 1) * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
            units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(NL - 1)]
            wpack([(Wk[l + 1].data_ptr(), H, False, units[l]) for l in range(NL - 1)], dev)
            mlp_fwd(M, H, [(h1, None, H)], Wk[1:], bk[1:], scale, out_w, None, y, None,
                    saveH[1:] if need else None, U, R, wpk=units)
        else:
            mlp_fwd(M, H, [(x, None, kin)], Wk, bk, scale, out_w, None, y, None, saveH, U, R)
        ctx.meta = (NL, H, kin, out_w, has_norm, kp != kin, op != out_w)
        # inputs / parameters through save_for_backward (autograd's version counter then catches an
        # in-place update between forward and backward); padded weight copies and the activations
        # the kernels wrote are ours and live until the graph is freed
        ctx.save_for_backward(x, *Ws, *([scale] if has_norm else []))
        ctx.pads = (W0 if kp != kin else None, Wl if op != out_w else None)
        ctx.saved_acts = (saveH, U, R)
        return y

    @staticmethod
    def backward(ctx, dy):
        NL, H, kin, out_w, has_norm, pad0, padl = ctx.meta
        if ctx.saved_acts is None or ctx.saved_acts[0] is None:
            raise RuntimeError("MlpFunction: no saved activations -- backward ran a second time without retain_graph "
                               "support, or the forward ran under no_grad")
        t = ctx.saved_tensors
        x, Ws = t[0], list(t[1:1 + NL])
        scale = t[1 + NL] if has_norm else None
        Wk = [ctx.pads[0] if pad0 else Ws[0]] + Ws[1:-1] + [ctx.pads[1] if padl else Ws[-1]]
        saveH, U, R = ctx.saved_acts
        dy = _f32c(dy)
        M, dev = x.shape[0], x.device
        kp, op = pad16(kin), pad16(out_w)
        widths = [H] * (NL - 1) + [op]
        dZ = [torch.empty(M, w, dtype=torch.float32, device=dev) for w in widths]
        db = [torch.empty(w, dtype=torch.float32, device=dev) for w in widths]
        dscale = torch.empty(H, dtype=torch.float32, device=dev) if has_norm else None
        WT = [None] + [Wk[l].t().contiguous() for l in range(1, NL)]
        din = []
        dx = None
        if ctx.needs_input_grad[0]:
            if kin != H:
                raise NotImplementedError("input gradient of a ragged-width MLP input is not needed by the path")
            dx = torch.empty(M, H, dtype=torch.float32, device=dev)
            din = [(Wk[0].t().contiguous(), None, dx)]
        # bias gradients are a by-product of the weight-gradient kernel (it reads dZ anyway)
        mlp_bwd(M, H, NL, dy, None, None, out_w, U, R, scale, saveH, WT, dZ, din, [None] * NL, dscale)
        ins = [x] + list(saveH)
        in_w = [kin] + [H] * (NL - 1)
        dWs = [torch.empty(widths[l], pad16(in_w[l]), dtype=torch.float32, device=dev) for l in range(NL)]
        jobs = []
        for l in range(NL):
            jobs.append((dZ[l], widths[l], widths[l] // 16, ins[l], in_w[l], pad16(in_w[l]) // 16, in_w[l], dWs[l], 0, pad16(in_w[l]), db[l]))
        wgrad(jobs, dev)
        grads = []
        for l in range(NL):
            dW, dbl = dWs[l], db[l]
            if l == 0 and kp != kin:
                dW = dW[:, :kin]
            if l == NL - 1 and op != out_w:
                dW, dbl = dW[:out_w], dbl[:out_w]
            grads += [dW, dbl]
        if has_norm:
            grads.append(dscale)
        return (dx, None, *grads)


# ------------------------------------------------- processor: L GraphNetBlocks (R3-R5)
PARAMS_PER_BLOCK = 18  # edge W0,b0..W3,b3,scale ; node W0,b0..W3,b3,scale


class ProcessorFunction(torch.autograd.Function):
    """L rounds of gather -> edge MLP -> segment-sum -> node MLP -> residuals.

    apply(x[N,H], e_sorted[E,H], topo, L, *params) -> (x_out, e_out_sorted)
    ``e`` is in the topology's dst-sorted order.  params: 18 tensors per block in
    state_dict order (edge_block.{0,2,4,6}.{weight,bias}, edge_block.7.scale,
    node_block...).
    """

    @staticmethod
    def forward(ctx, x, e, topo: Topology, L: int, *params):
        _require_device(x, e, *params)
        x, e = _f32c(x), _f32c(e)
        N, H = x.shape
        E = e.shape[0]
        if H not in SUPPORTED_H:
            raise NotImplementedError(f"hidden_size={H} not supported by the MI355X engine (16/32/64/128)")
        if N != topo.N or E != topo.E:
            raise ValueError("x / edge_attr do not match the topology")
        dev = x.device
        P = [_f32c(p) for p in params]
        need = any(ctx.needs_input_grad) and _saving()
        f = dict(dtype=torch.float32, device=dev)
        saved = []
        fuse_agg = (H == 128) and X6_ENABLED and E > 0 and _os.environ.get("MGN_NO_FUSED_AGG") is None
        m = None if fuse_agg else torch.empty(E, H, **f)
        part = torch.empty((E + 15) // 16, 2, H, **f) if fuse_agg else None
        # H = 128: algebraic split of the first edge layer (W0 = [W_e | W_d | W_s]):
        #   W0.[e, x_dst, x_src] = W_e.e + (x W_d^T)[dst] + (x W_s^T)[src]
        # the two node-level projections of round i+1 are post-products of round i's node
        # kernel (x' still in registers); round 0's come from two small launches.
        split = (H == 128)
        Pd = Ps = None
        # split-bf16 matrix path: all GEMM units of all rounds packed by one launch; per round
        # [We0|e, We1, We2, We3, Wn0|x, Wn0|agg, Wn1, Wn2, Wn3, We0|x_dst, We0|x_src]
        x6 = split and L > 0 and E > 0 and X6_ENABLED
        prec = 1 if _matrix_precision == "bf16" else 0
        if prec and not x6 and L > 0 and E > 0:
            raise NotImplementedError("bf16 matrix mode needs hidden_size=128 (the packed split-bf16 kernels)")
        prec = prec if x6 else 0
        NU = 11
        if x6:
            # per round [We0|e, We1, We2, We3 | Wn0|x, Wn0|agg, Wn1, Wn2, Wn3, NEXT round's We0|x_dst, We0|x_src]:
            # the units of every launch lie back to back (edge 0..3, node 4..8 + its two post-products 9..10);
            # round 0's own projections sit in two extra slots at the end
            pk = torch.empty((L * NU + 2) * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
            pk0 = pk.data_ptr()
            blocks = []
            for i in range(L):
                q = P[PARAMS_PER_BLOCK * i: PARAMS_PER_BLOCK * (i + 1)]
                We0, Wn0 = q[0].data_ptr(), q[9].data_ptr()
                srcs = [(We0, 3 * H), (q[2].data_ptr(), H), (q[4].data_ptr(), H), (q[6].data_ptr(), H),
                        (Wn0, 2 * H), (Wn0 + 4 * H, 2 * H), (q[11].data_ptr(), H), (q[13].data_ptr(), H), (q[15].data_ptr(), H)]
                if i + 1 < L:
                    Wnx = P[PARAMS_PER_BLOCK * (i + 1)].data_ptr()
                    srcs += [(Wnx + 4 * H, 3 * H), (Wnx + 8 * H, 3 * H)]
                for u, (sa, ld) in enumerate(srcs):
                    blocks.append((sa, ld, False, pk0 + (i * NU + u) * _capi.WPACK_BYTES))
            W00 = P[0].data_ptr()
            blocks.append((W00 + 4 * H, 3 * H, False, pk0 + (L * NU) * _capi.WPACK_BYTES))
            blocks.append((W00 + 8 * H, 3 * H, False, pk0 + (L * NU + 1) * _capi.WPACK_BYTES))
            wpack(blocks, dev)

            def unit(i, u):
                return pk0 + (i * NU + u) * _capi.WPACK_BYTES
        if split and L > 0 and E > 0:
            W0 = P[0]
            Pd, Ps = torch.empty(N, H, **f), torch.empty(N, H, **f)
            for slab, dst_t in ((1, Pd), (2, Ps)):
                mlp_fwd(N, H, [(x, None, H)], [W0[:, slab * H:(slab + 1) * H].contiguous()], [None], None, H, None, dst_t,
                        wpk=[unit(L, slab - 1)] if x6 else (), precision=prec)
        for i in range(L):
            q = P[PARAMS_PER_BLOCK * i: PARAMS_PER_BLOCK * (i + 1)]
            We, be, se = [q[0], q[2], q[4], q[6]], [q[1], q[3], q[5], q[7]], q[8]
            Wn, bn, sn = [q[9], q[11], q[13], q[15]], [q[10], q[12], q[14], q[16]], q[17]
            e_new = torch.empty(E, H, **f)
            x_new = torch.empty(N, H, **f)
            agg = torch.empty(N, H, **f)
            if need:
                He = [torch.empty(E, H, **f) for _ in range(3)]
                Ue, Re = torch.empty(E, H, **f), torch.empty(E, **f)
                Hn = [torch.empty(N, H, **f) for _ in range(3)]
                Un, Rn = torch.empty(N, H, **f), torch.empty(N, **f)
                # ReLU masks as bits (16 B per row and layer): what the backward chain reads
                Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)] if x6 else None
                Mn = [torch.empty(N, 4, dtype=torch.int32, device=dev) for _ in range(3)] if x6 else None
            else:
                He = Hn = None
                Ue = Re = Un = Rn = None
                Me = Mn = None
            # R3: m = edge_block(cat[e, x[dst], x[src]]);  e' = e + m     (layers.py:1017-1028,1039)
            if split and E > 0:
                mlp_fwd(E, H, [(e, None, H)], We, be, se, H, e, e_new, m, He, Ue, Re, ldw0=3 * H,
                        adds=[(Pd, topo.dst_s), (Ps, topo.src_s)],
                        wpk=[unit(i, u) for u in range(4)] if x6 else (), saveM=Me, precision=prec,
                        seg=(topo.dst_s, topo.rowptr_dst, agg, part) if (fuse_agg and x6) else None)
            else:
                mlp_fwd(E, H, [(e, None, H), (x, topo.dst_s, H), (x, topo.src_s, H)], We, be, se, H, e, e_new, m, He, Ue, Re)
            # R4: agg = segment-sum of m over dst                          (layers.py:1031-1037)
            if fuse_agg and x6 and split:
                seg_fix(topo.rowptr_dst, part, agg)   # the edge kernel summed inside its wave tiles
            else:
                segsum(m, topo.rowptr_dst, None, agg)
            # R5: x' = x + node_block(cat[x, agg])                         (layers.py:1100-1101,1040)
            posts, Pd_n, Ps_n = (), None, None
            if split and E > 0 and i + 1 < L:
                W0n = P[PARAMS_PER_BLOCK * (i + 1)]
                Pd_n, Ps_n = torch.empty(N, H, **f), torch.empty(N, H, **f)
                posts = [(W0n.data_ptr() + 4 * H, Pd_n), (W0n.data_ptr() + 8 * H, Ps_n)]
            wn = ()
            if x6:
                wn = [unit(i, u) for u in range(4, 9)] + ([unit(i, 9), unit(i, 10)] if posts else [])
            mlp_fwd(N, H, [(x, None, H), (agg, None, H)], Wn, bn, sn, H, x, x_new, None, Hn, Un, Rn,
                    posts=posts, post_ldw=3 * H, wpk=wn, saveM=Mn, precision=prec)
            if need:
                saved.append((x, e, agg, He, Ue, Re, Hn, Un, Rn, Me, Mn))
            x, e = x_new, e_new
            Pd, Ps = Pd_n, Ps_n
        ctx.topo, ctx.L, ctx.saved_acts, ctx.prec = topo, L, (saved if need else None), prec
        ctx.save_for_backward(*P)  # version-checked by autograd (an optimiser step in between is an error)
        return x, e

    @staticmethod
    def backward(ctx, dx, de):
        topo, L, saved, prec = ctx.topo, ctx.L, ctx.saved_acts, ctx.prec
        if saved is None:
            raise RuntimeError("ProcessorFunction: the saved activations were released by an earlier backward pass "
                               "(retain_graph is not supported: ~2.5 KB per edge and round are freed eagerly), "
                               "or the forward ran under no_grad")
        P = list(ctx.saved_tensors)
        dev = P[0].device
        N, E = topo.N, topo.E
        H = P[1].numel()
        f = dict(dtype=torch.float32, device=dev)
        dx = _f32c(dx) if dx is not None else torch.zeros(N, H, **f)
        de = _f32c(de) if de is not None else torch.zeros(E, H, **f)
        dZn = [torch.empty(N, H, **f) for _ in range(4)]
        dZe = [torch.empty(E, H, **f) for _ in range(4)]
        dAgg, Sd, Ss = torch.empty(N, H, **f), torch.empty(N, H, **f), torch.empty(N, H, **f)
        dx_buf, de_buf = [torch.empty(N, H, **f), torch.empty(N, H, **f)], [torch.empty(E, H, **f), torch.empty(E, H, **f)]
        grads: List[Optional[torch.Tensor]] = [None] * (PARAMS_PER_BLOCK * L)
        nb = H // 16
        HH = H * H
        x6 = (H == 128) and X6_ENABLED and L > 0 and saved[0][9] is not None
        NU = 11
        if x6:
            # split-bf16 path: the transposed GEMM units of every round packed by one launch; per
            # round [Wn3^T, Wn2^T, Wn1^T, Wn0|agg^T, We3^T, We2^T, We1^T, We0|e^T, Wn0|x^T, We0|x_dst^T, We0|x_src^T]
            pk = torch.empty(L * NU * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
            pk0 = pk.data_ptr()
            blocks = []
            for i in range(L):
                q = P[PARAMS_PER_BLOCK * i: PARAMS_PER_BLOCK * (i + 1)]
                We0, Wn0 = q[0].data_ptr(), q[9].data_ptr()
                srcs = [(q[15].data_ptr(), H), (q[13].data_ptr(), H), (q[11].data_ptr(), H), (Wn0 + 4 * H, 2 * H),
                        (q[6].data_ptr(), H), (q[4].data_ptr(), H), (q[2].data_ptr(), H), (We0, 3 * H),
                        (Wn0, 2 * H), (We0 + 4 * H, 3 * H), (We0 + 8 * H, 3 * H)]
                for u, (sa, ld) in enumerate(srcs):
                    blocks.append((sa, ld, True, pk0 + (i * NU + u) * _capi.WPACK_BYTES))
            wpack(blocks, dev)

            def unit(i, u):
                return pk0 + (i * NU + u) * _capi.WPACK_BYTES
            wt = None
        else:
            # W^T operands of every round in one buffer, filled by one batched transpose launch:
            # per round 11 H x H blocks  [WTn1..3 | WTe1..3 | WT0n_agg | WT0e_e | Wcat (H x 3H)]
            wt = torch.empty(L, 11, H, H, **f)
            tb = []
            for i in range(L):
                q = P[PARAMS_PER_BLOCK * i: PARAMS_PER_BLOCK * (i + 1)]
                base = wt.data_ptr() + 4 * (i * 11 * HH)
                We0, Wn0 = q[0].data_ptr(), q[9].data_ptr()
                for k, l in enumerate((1, 2, 3)):
                    tb.append((q[9 + 2 * l].data_ptr(), H, base + 4 * (k * HH), H))          # WTn[l]
                    tb.append((q[2 * l].data_ptr(), H, base + 4 * ((3 + k) * HH), H))        # WTe[l]
                tb.append((Wn0 + 4 * H, 2 * H, base + 4 * (6 * HH), H))                      # (W0n[:, H:])^T
                tb.append((We0, 3 * H, base + 4 * (7 * HH), H))                              # (W0e[:, :H])^T
                cat = base + 4 * (8 * HH)                                                    # Wcat [H, 3H]
                tb.append((Wn0, 2 * H, cat, 3 * H))                                          # (W0n[:, :H])^T
                tb.append((We0 + 4 * H, 3 * H, cat + 4 * H, 3 * H))                          # (W0e[:, H:2H])^T
                tb.append((We0 + 8 * H, 3 * H, cat + 8 * H, 3 * H))                          # (W0e[:, 2H:])^T
            transpose_blocks(tb, H, dev)
        # E == 0 / N == 0: some launches return early at M == 0 and would leave their outputs unwritten
        _alloc = torch.zeros_like if (E == 0 or N == 0) else torch.empty_like
        gs = [[_alloc(t) for t in P[PARAMS_PER_BLOCK * i: PARAMS_PER_BLOCK * (i + 1)]] for i in range(L)]
        # packed path: the dX launch of round i and the node chain of round i-1 work on the same
        # rows -> one launch (front stage of mgn_mlp_bwd); dZn double buffered across rounds
        # (measured neutral at N = 30k rows -- 88 us fused vs 57 + 30 us: a launch costs as many tile
        # times as it has GEMM units -- so it is opt-in: MGN_FRONT=1; tests/test_hip_parity.py covers it)
        fuse = x6 and _os.environ.get("MGN_FRONT") is not None
        dZn_sets = [dZn, [torch.empty(N, H, **f) for _ in range(4)] if fuse and L > 1 else dZn]
        node_done = False
        # scale-gradient partials of all chain launches, reduced by ONE launch at the end (MGN_NO_DEFER: per launch)
        deferred = [] if _os.environ.get("MGN_NO_DEFER") is None else None

        def units(i):
            return [unit(i, u) for u in range(4)], [unit(i, u) for u in range(4, 8)], [unit(i, u) for u in range(8, 11)]

        for i in reversed(range(L)):
            q = P[PARAMS_PER_BLOCK * i: PARAMS_PER_BLOCK * (i + 1)]
            We, se = [q[0], q[2], q[4], q[6]], q[8]
            Wn, sn = [q[9], q[11], q[13], q[15]], q[17]
            x, e, agg, He, Ue, Re, Hn, Un, Rn, Me, Mn = saved[i]
            g = gs[i]
            dZn = dZn_sets[i & 1]
            if x6:
                # never dereferenced on the packed path: any valid [H,H] / [H,3H] tensors do
                WTn = WTe = [None, q[2], q[2], q[2]]
                WT0n_agg = WT0e_e = q[2]
                Wcat = q[0]
                kn, ke, kx = units(i)
            else:
                w = wt[i]
                WTn = [None, w[0], w[1], w[2]]
                WTe = [None, w[3], w[4], w[5]]
                WT0n_agg, WT0e_e = w[6], w[7]
                Wcat = w[8:11].reshape(H, 3 * H)
                kn = ke = kx = ()
            # node MLP chain: dX' -> dZn[3..0], dAgg = W0n[:,H:]^T dZn0  (already done by the
            # previous iteration's fused launch except for the last round)
            if not node_done:
                mlp_bwd(N, H, 4, dx, None, None, H, Un, Rn, sn, Hn, WTn, dZn, [(WT0n_agg, None, dAgg)],
                        [None] * 4, g[17], wpk=kn, Ms=Mn, precision=prec, defer=deferred)
            # edge MLP chain: dM = dE' + dAgg[dst] -> dZe[3..0], dE = dE' + W0e[:, :H]^T dZe0
            de_new = de_buf[0] if de.data_ptr() != de_buf[0].data_ptr() else de_buf[1]
            mlp_bwd(E, H, 4, de, dAgg, topo.dst_s, H, Ue, Re, se, He, WTe, dZe, [(WT0e_e, de, de_new)],
                    [None] * 4, g[8], wpk=ke, Ms=Me, precision=prec, defer=deferred)
            # scatter of the first-layer pre-activations' grads onto dst / src nodes
            if H == 128:
                segsum2(dZe[0], topo.rowptr_dst, None, Sd, topo.rowptr_src, topo.perm_src, Ss)
            else:
                segsum(dZe[0], topo.rowptr_dst, None, Sd)
                segsum(dZe[0], topo.rowptr_src, topo.perm_src, Ss)
            # weight gradients: dW = dZ^T X
            # (A = dZ, B = layer input, dW slab[, db = bias gradient as a by-product])
            jobs = [
                (dZn[0], H, nb, x, H, nb, H, g[9], 0, 2 * H, g[10]),
                (dZn[0], H, nb, agg, H, nb, H, g[9], H, 2 * H),
                (dZe[0], H, nb, e, H, nb, H, g[0], 0, 3 * H, g[1]),
                (Sd, H, nb, x, H, nb, H, g[0], H, 3 * H),
                (Ss, H, nb, x, H, nb, H, g[0], 2 * H, 3 * H),
            ]
            for l in (1, 2, 3):
                jobs.append((dZn[l], H, nb, Hn[l - 1], H, nb, H, g[9 + 2 * l], 0, H, g[10 + 2 * l]))
                jobs.append((dZe[l], H, nb, He[l - 1], H, nb, H, g[2 * l], 0, H, g[1 + 2 * l]))
            wgrad(jobs, dev, prec)
            # dX = dX' + W0n[:, :H]^T dZn0 + W0e[:, H:2H]^T Sd + W0e[:, 2H:]^T Ss
            dx_new = dx_buf[0] if dx.data_ptr() != dx_buf[0].data_ptr() else dx_buf[1]
            if fuse and i > 0:  # ... fused with the node chain of round i-1 (same rows)
                qp = P[PARAMS_PER_BLOCK * (i - 1): PARAMS_PER_BLOCK * i]
                _, _, _, _, _, _, Hn_p, Un_p, Rn_p, _, Mn_p = saved[i - 1]
                kn_p, _, _ = units(i - 1)
                mlp_bwd(N, H, 4, dx, None, None, H, Un_p, Rn_p, qp[17], Hn_p, WTn, dZn_sets[(i - 1) & 1],
                        [(WT0n_agg, None, dAgg)], [None] * 4, gs[i - 1][17], wpk=kx + kn_p, Ms=Mn_p, precision=prec,
                        front=([dZn[0], Sd, Ss], dx, dx_new), defer=deferred)
                node_done = True
            else:
                mlp_fwd(N, H, [(dZn[0], None, H), (Sd, None, H), (Ss, None, H)], [Wcat], [None], None, H, dx, dx_new, wpk=kx, precision=prec)
                node_done = False
            grads[PARAMS_PER_BLOCK * i: PARAMS_PER_BLOCK * (i + 1)] = g
            dx, de = dx_new, de_new
        colred_batch(deferred, dev)
        ctx.saved_acts = None
        return (dx, de, None, None, *grads)
