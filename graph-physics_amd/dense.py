"""Dense row work of the sparse-attention Transformer processor on the engine (csrc/mgn_dense.hip): every ``nn.Linear`` of a
Transformer block (graphphysics/models/layers.py:564-697,700-819), of its gated MLP (:213-278) and of ``TemporalAttention``
(:822-887) is ONE fused launch -- RMSNorm as a prologue, activation / gated product / bias / residual as the epilogue, a
concatenated input as two phases -- wrapped in a ``torch.autograd.Function`` whose backward is built from the same launch
(dX = dZ W), ``mgn_act_gate_bwd``, ``mgn_rownorm_bwd`` and the engine's weight-gradient kernel.  No ``F.linear``, no
``torch.cat``; CUDA tensors only."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _capi, ops

ACT_NONE = -1
ACT_IDS = {None: ACT_NONE, "none": ACT_NONE, "relu": 0, "silu": 1, "gelu": 2}


def _prec() -> int:
    return 1 if ops.get_matrix_precision() == "bf16" else 0


def linear_launch(x, W, b=None, x2=None, W2=None, b2=None, norm_scale=None, act: int = ACT_NONE, resid=None, out=None, inv_out=None,
                  n_out=None, saveZ1=None, saveZ2=None, precision: int = 0, x3=None, idx=(None, None, None), M: Optional[int] = None,
                  w_transposed: bool = False, gate_bwd=None, norm_outer=None, z16: bool = False, out16: bool = False):
    """one ``mgn_linear_fwd`` launch (include/mgn_hip.h); x / x2 / x3 / resid may be row-strided views (stride(1) == 1);
    ``idx[p]`` (int32 [M]) gathers the rows of phase p; ``M`` = output rows (default: rows of x); ``w_transposed``: W (and W2) are
    [K, N] -- the launch multiplies by their transpose (see :func:`input_gradient`)."""
    M = int(x.shape[0]) if M is None else int(M)
    K1 = int(x.shape[1])
    K2 = int(x2.shape[1]) if x2 is not None else 0
    K3 = int(x3.shape[1]) if x3 is not None else 0
    N = int(W.shape[1] if w_transposed else W.shape[0])
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16 if out16 else torch.float32, device=x.device)
    a = _capi.LinearArgs()
    a.M, a.x, a.ldx, a.K1 = M, x.data_ptr(), int(x.stride(0)), K1
    a.x2, a.ldx2, a.K2 = (x2.data_ptr() if x2 is not None else None), (int(x2.stride(0)) if x2 is not None else 0), K2
    a.x3, a.ldx3, a.K3 = (x3.data_ptr() if x3 is not None else None), (int(x3.stride(0)) if x3 is not None else 0), K3
    a.idx, a.idx2, a.idx3 = ops._ptr(idx[0]), ops._ptr(idx[1]), ops._ptr(idx[2])
    a.norm_scale, a.eps = ops._ptr(norm_scale), ops.EPS
    a.inv_out, a.n_out = ops._ptr(inv_out), ops._ptr(n_out)
    a.W, a.ldw, a.b = W.data_ptr(), int(W.stride(0)), ops._ptr(b)
    a.W2, a.b2 = ops._ptr(W2), ops._ptr(b2)
    a.act, a.N = act, N
    a.resid, a.ldr = ops._ptr(resid), (int(resid.stride(0)) if resid is not None else 0)
    a.out, a.ldo = out.data_ptr(), int(out.stride(0))
    a.saveZ1, a.saveZ2, a.precision = ops._ptr(saveZ1), ops._ptr(saveZ2), precision
    a.w_transposed = 1 if w_transposed else 0
    a.z16 = 1 if z16 else 0
    a.x16 = 1 if x.dtype == torch.bfloat16 else 0          # two-byte input rows (bf16 mode; written by a launch with out16)
    a.out16 = 1 if out16 else 0
    if norm_outer is not None:   # (scale, inv_out or None): a second RMSNorm in front of the norm prologue
        a.norm_scale_outer, a.inv_outer_out = norm_outer[0].data_ptr(), ops._ptr(norm_outer[1])
    if gate_bwd is not None:   # (Z1, Z2, out2): the gated product's backward as the epilogue -- out = dZ1, out2 = dZ2
        a.gb_z1, a.gb_z2, a.out2 = gate_bwd[0].data_ptr(), gate_bwd[1].data_ptr(), gate_bwd[2].data_ptr()
    with torch.cuda.device(x.device):
        rc = _capi.lib().mgn_linear_fwd(C.byref(a), ops._stream(x.device))
    _capi.check(rc, "mgn_linear_fwd", dense=True)
    return out


#: MGN_DENSE_WT=0: always materialise W^T (for A/B; tools/c5_modes.py flips the list entry in one process)
_WT_ON = [os.environ.get("MGN_DENSE_WT", "1") != "0"]


def input_gradient(dZ, W, resid=None, precision: int = 0):
    """dX = dZ W [+ resid] for the ``nn.Linear`` weight W [N, K] on the same launch.  At row counts where the launch stages its weights
    through LDS the staging reads W transposed (no copy); elsewhere W^T is materialised first."""
    M, N = int(dZ.shape[0]), int(dZ.shape[1])
    K = int(W.shape[1])
    if _WT_ON[0] and W.stride(1) == 1 and (W.stride(0) & 3) == 0 and _capi.lib().mgn_linear_accepts_transposed(M, N, K, 0, precision):
        return linear_launch(dZ, W, resid=resid, precision=precision, w_transposed=True)
    return linear_launch(dZ, W.t().contiguous(), resid=resid, precision=precision)


def _rows(t: torch.Tensor) -> torch.Tensor:
    """fp32 rows with unit column stride and 16-byte aligned row starts (a column slab of a wider matrix is fine)"""
    t = t.float() if t.dtype != torch.float32 else t
    if t.dim() != 2 or t.stride(1) != 1 or (t.stride(0) & 3) or (t.data_ptr() & 15):
        t = t.contiguous()
    return t


class DenseFn(torch.autograd.Function):
    """out = [resid +] epi(n W^T + b),  n = RMSNorm(cat[x, x2, x3]; norm_scale) or the plain concatenation;
    epi(z) = act(z) [* (n W2^T + b2)]                       (see include/mgn_hip.h, mgn_linear_fwd)
    ``gather`` = (topology, ("dst" | "src" | None) per phase): a gathered phase reads x_p[topo.dst_s / src_s] per edge row and
    its input gradient is the CSR segment sum over the same grouping (atomics-free, hub-safe)."""

    @staticmethod
    def forward(ctx, x, x2, x3, W, b, W2, b2, norm_scale, resid, act: int, precision: int, gather, pass_x: bool = False):
        ops._require_device(x, x2, x3, W, b, W2, b2, norm_scale, resid)
        xs = [_rows(t) if t is not None else None for t in (x, x2, x3)]
        resid = _rows(resid) if resid is not None else None
        W = ops._f32c(W)
        W2 = ops._f32c(W2) if W2 is not None else None
        topo, by = gather if gather is not None else (None, (None, None, None))
        idx = tuple((topo.dst_s if g_ == "dst" else topo.src_s) if g_ is not None else None for g_ in by)
        M = topo.E if any(g_ is not None for g_ in by) else xs[0].shape[0]
        K = sum(t.shape[1] for t in xs if t is not None)
        N = W.shape[0]
        dev = xs[0].device
        need = any(ctx.needs_input_grad) and ops._saving()
        f = dict(dtype=torch.float32, device=dev)
        if any(g_ is not None for g_ in by) and norm_scale is None and need:
            raise NotImplementedError("gathered phases without the norm prologue have no weight-gradient operand")
        Z1 = torch.empty(M, N, **f) if (need and (act != ACT_NONE or W2 is not None)) else None
        Z2 = torch.empty(M, N, **f) if (need and W2 is not None) else None
        inv = torch.empty(M, **f) if (need and norm_scale is not None) else None
        n_out = torch.empty(M, K, **f) if (need and norm_scale is not None) else None
        out = linear_launch(xs[0], W, b, xs[1], W2, b2, norm_scale, act, resid, None, inv, n_out, Z1, Z2, precision, xs[2], idx, M)
        ctx.save_for_backward(xs[0], xs[1], xs[2], W, W2, norm_scale)
        ctx.aux = (act, precision, Z1, Z2, inv, n_out, b is not None, b2 is not None, resid is not None, topo, by, idx, M)
        ctx.pass_x = bool(pass_x)
        if pass_x:
            # (out, x): the SAME rows handed on (autograd makes the second output an alias of x).  A caller that feeds them into a later
            # residual gets that residual's gradient delivered HERE, where the norm backward adds it in its own pass (acc) -- instead
            # of autograd summing the two uses of x in a launch of its own.  Needs the norm prologue on an ungathered single phase.
            if norm_scale is None or x2 is not None or any(g_ is not None for g_ in by):
                raise ValueError("DenseFn: pass_x needs the norm prologue on one ungathered input phase")
            return out, x
        return out

    @staticmethod
    def backward(ctx, dy, d_pass=None):
        if ctx.aux is None:
            raise RuntimeError("DenseFn: backward ran twice (the saved activations are released eagerly)")
        d_pass = ops._f32c(d_pass) if (ctx.pass_x and d_pass is not None) else None
        x, x2, x3, W, W2, norm_scale = ctx.saved_tensors
        act, prec, Z1, Z2, inv, n_out, has_b, has_b2, has_res, topo, by, idx, M = ctx.aux
        if Z1 is None and (act != ACT_NONE or W2 is not None):
            raise RuntimeError("DenseFn: no saved activations (the forward ran under no_grad)")
        xs = [x, x2, x3]
        dy = ops._f32c(dy)
        N = dy.shape[1]
        Ks = [t.shape[1] if t is not None else 0 for t in xs]
        K = sum(Ks)
        dev = dy.device
        f = dict(dtype=torch.float32, device=dev)
        L = _capi.lib()
        # ---- through the epilogue
        if act != ACT_NONE or W2 is not None:
            dZ1 = torch.empty(M, N, **f)
            dZ2 = torch.empty(M, N, **f) if W2 is not None else None
            with torch.cuda.device(dev):
                rc = L.mgn_act_gate_bwd(dy.data_ptr(), Z1.data_ptr(), ops._ptr(Z2), M, N, act, prec, dZ1.data_ptr(), ops._ptr(dZ2), ops._stream(dev))
            _capi.check(rc, "mgn_act_gate_bwd", dense=True)
        else:
            dZ1, dZ2 = dy, None
        # ---- input gradient: dn = dZ1 W (+ dZ2 W2) -- the same launch with the transposed weight
        want = [xs[p] is not None and ctx.needs_input_grad[p] for p in range(3)]
        dxs = [None, None, None]
        dscale = None
        if M > 0 and (any(want) or norm_scale is not None):
            if N > 384:
                raise NotImplementedError("input gradient of a Linear wider than 384 outputs")
            dn = input_gradient(dZ1, W, precision=prec)
            if W2 is not None:
                dn = input_gradient(dZ2, W2, resid=dn, precision=prec)
            if norm_scale is not None:
                nph = sum(1 for t in xs if t is not None)
                rows = [torch.empty(M, Ks[p], **f) for p in range(nph)]       # per (edge) row
                arr = (_capi.RownormPhase * nph)()
                for p in range(nph):
                    arr[p].x, arr[p].ldx, arr[p].K, arr[p].idx = xs[p].data_ptr(), int(xs[p].stride(0)), Ks[p], ops._ptr(idx[p])
                    arr[p].dx, arr[p].lddx = rows[p].data_ptr(), Ks[p]
                if d_pass is not None:   # the gradient of the rows handed on (pass_x): added inside the norm backward
                    arr[0].acc = d_pass.data_ptr()
                dscale = torch.empty(K, **f)
                ws = torch.empty(max(L.mgn_rownorm_bwd_workspace_bytes(K), 16), dtype=torch.uint8, device=dev)
                with torch.cuda.device(dev):
                    rc = L.mgn_rownorm_bwd(dn.data_ptr(), arr, nph, inv.data_ptr(), norm_scale.data_ptr(), ops.EPS, M, dscale.data_ptr(),
                                           ws.data_ptr(), ws.numel(), ops._stream(dev))
                _capi.check(rc, "mgn_rownorm_bwd", dense=True)
            else:
                rows, k0 = [], 0
                for p in range(3):
                    if xs[p] is not None:
                        rows.append(dn[:, k0:k0 + Ks[p]])
                        k0 += Ks[p]
            for p in range(len(rows)):
                if not want[p]:
                    continue
                if by[p] is not None:   # gathered phase: sum the edge rows over their segments (the gather's transpose)
                    src_rows = rows[p] if rows[p].is_contiguous() else rows[p].contiguous()
                    out_n = torch.empty(xs[p].shape[0], Ks[p], **f)
                    dxs[p] = ops.segsum_topo(src_rows, topo, by[p], out_n)
                else:
                    dxs[p] = rows[p]
        elif M == 0:
            dxs = [torch.zeros(t.shape[0], t.shape[1], **f) if t is not None else None for t in xs]
            dscale = torch.zeros(K, **f) if norm_scale is not None else None
        # ---- weight / bias gradients: dW = dZ^T n on the engine's weight-gradient kernel, 128-column slabs
        if norm_scale is not None:
            srcs = [(n_out, 0, K)]
        else:
            srcs, k0 = [], 0
            for p in range(3):
                if xs[p] is not None:
                    srcs.append((xs[p], k0, Ks[p]))
                    k0 += Ks[p]

        def wgrad_of(dZ, want_b):
            dW = torch.empty(N, K, **f) if M > 0 else torch.zeros(N, K, **f)
            db = (torch.empty(N, **f) if M > 0 else torch.zeros(N, **f)) if want_b else None
            jobs = []
            # the generic weight-gradient kernel is instantiated for the widest block count of a launch: with 64-wide inputs
            # 64-row slabs keep it on the 4-block instance (measured 94 us against 183 us for dW[192, 64] at 150 000 rows)
            nslab = 64 if (N <= 64 or all(kw_ <= 64 for _, _, kw_ in srcs)) else 128   # (a 64-wide side: 64 x 64 blocks on both)
            for j0 in range(0, N, nslab):
                nj = min(nslab, N - j0)
                A = dZ[:, j0:j0 + nj]
                first = True
                for src, koff, kw_all in srcs:
                    for k0_ in range(0, kw_all, nslab):
                        nk = min(nslab, kw_all - k0_)
                        B = src[:, k0_:k0_ + nk]
                        job = (A, int(dZ.stride(0)), nj // 16, B, int(src.stride(0)), nk // 16, nk, dW, j0 * K + koff + k0_, K)
                        if first and db is not None:
                            job = job + (db[j0:j0 + nj],)
                        first = False
                        jobs.append(job)
            return dW, db, jobs

        dW, db, jobs = wgrad_of(dZ1, has_b)
        dW2 = db2 = None
        if W2 is not None:   # both matrices of a gated product in ONE weight-gradient launch (and one reduction)
            dW2, db2, jobs2 = wgrad_of(dZ2, has_b2)
            jobs = jobs + jobs2
        if M > 0:
            ops.wgrad(jobs, dev, prec)
        ctx.aux = None
        if d_pass is not None and M == 0:
            dxs[0] = d_pass
        return (dxs[0] if want[0] else None, dxs[1] if want[1] else None, dxs[2] if want[2] else None, dW, db, dW2, db2,
                dscale, dy if has_res else None, None, None, None, None)


def dense(x, W, b=None, x2=None, W2=None, b2=None, norm_scale=None, act: Optional[str] = None, resid=None, x3=None, gather=None,
          pass_x: bool = False):
    """fused Linear on the engine (module docstring); ``act``: None / "relu" / "silu" / "gelu"; ``gather`` = (topology,
    (by_1, by_2, by_3)) with by_p in ("dst", "src", None).  ``pass_x``: returns (out, x') with x' the input rows again -- use x' for a
    later residual and its gradient is added inside this node's norm backward (see DenseFn.forward)."""
    ops._call.grad = torch.is_grad_enabled()
    try:
        return DenseFn.apply(x, x2, x3, W, b, W2, b2, norm_scale, resid, ACT_IDS[act], _prec(), gather, pass_x)
    finally:
        ops._call.grad = True


def _wgrad_jobs(dZ, n, dW, db):
    """64-column slab jobs of dW [N, K] = dZ^T n (+ db), as DenseFn.backward cuts them"""
    N, K = dW.shape
    slab = 64 if (N <= 64 or K <= 64) else 128
    jobs = []
    for j0 in range(0, N, slab):
        nj = min(slab, N - j0)
        first = True
        for k0 in range(0, K, slab):
            nk = min(slab, K - k0)
            # (a bf16 tensor = two-byte rows: the C side takes them as a negative leading dimension, precision 1 only)
            lda = -int(dZ.stride(0)) if dZ.dtype == torch.bfloat16 else int(dZ.stride(0))
            ldb = -int(n.stride(0)) if n.dtype == torch.bfloat16 else int(n.stride(0))
            job = (dZ[:, j0:j0 + nj], lda, nj // 16, n[:, k0:k0 + nk], ldb, nk // 16, nk, dW, j0 * K + k0, K)
            if first and db is not None:
                job = job + (db[j0:j0 + nj],)
            first = False
            jobs.append(job)
    return jobs


class GatedMlpResidualFn(torch.autograd.Function):
    """x + W3 (act(W1 n + b1) * (W2 n + b2)) + b3,  n = RMSNorm(RMSNorm(x; s_outer); s_inner) -- the second half of a Transformer block
    (layers.py:700-819: x + gated_mlp(norm2(x)), build_gated_mlp starting with a norm of its own, :256-278) as ONE autograd node on the
    launches of this module, so that its backward can keep what separate nodes hand each other through HBM:
      * dP = dY W3 never exists: the launch that forms it applies the gated product's backward in its epilogue and writes dZ1 / dZ2
        (``gate_bwd`` of :func:`linear_launch`; mgn_act_gate_bwd read and wrote five [M, 3K] matrices for it);
      * both norms are the prologue of the gated launch (norm2's output is never stored) and one backward pass (mgn_rownorm2_bwd);
      * the residual's gradient is the accumulator input of that pass (no autograd add).
    Taken by :class:`transformer.Transformer` where the LDS-staged launches apply (mgn_linear_accepts_transposed: 65 536 rows or more);
    elsewhere the block stays on :func:`dense` / :func:`rms_norm`."""

    @staticmethod
    def usable(x, W1, W3, precision: int) -> bool:
        M, K = int(x.shape[0]), int(x.shape[1])
        N = int(W1.shape[0])
        L = _capi.lib()
        return bool(x.is_cuda and _FUSED_MLP[0] and K <= 192 and L.mgn_linear_accepts_transposed(M, N, K, 0, precision)
                    and L.mgn_linear_accepts_transposed(M, K, N, 1, precision))

    @staticmethod
    def forward(ctx, x, s_outer, s_inner, W1, b1, W2, b2, W3, b3, act: int, precision: int):
        ops._require_device(x, s_outer, s_inner, W1, b1, W2, b2, W3, b3)
        x = _rows(x)
        W1, W2, W3 = ops._f32c(W1), ops._f32c(W2), ops._f32c(W3)
        M, K = x.shape
        N = W1.shape[0]
        dev = x.device
        f = dict(dtype=torch.float32, device=dev)
        need = any(ctx.needs_input_grad) and ops._saving()
        # bf16 mode: the pre-activations are bf16 numbers -- kept as two-byte rows (exact; MGN_DENSE_SAVE16=0: fp32 rows).  Only where every
        # weight-gradient job of the backward lands on the row-vector kernel (mgn_wgrad_p takes two-byte rows in 64-wide slabs or as
        # packed full 128 x 128 jobs with ld = -128, not the -384-strided 128-column slabs _wgrad_jobs cuts at hidden 128)
        z16 = precision == 1 and _SAVE16[0] and min(int(W1.shape[0]), int(x.shape[1])) <= 64
        zt = dict(dtype=torch.bfloat16 if z16 else torch.float32, device=dev)
        Z1 = torch.empty(M, N, **zt) if need else None
        Z2 = torch.empty(M, N, **zt) if need else None
        inv_o = torch.empty(M, **f) if need else None
        inv_i = torch.empty(M, **f) if need else None
        n = torch.empty(M, K, **f) if need else None
        # both norms are the prologue of the gated launch (the outer one's output is never stored)
        # (bf16 mode: the gated product itself is a bf16 tensor too -- two-byte rows between the two launches and for dW3)
        p_ = linear_launch(x, W1, b1, W2=W2, b2=b2, norm_scale=s_inner, act=act, inv_out=inv_i, n_out=n, saveZ1=Z1, saveZ2=Z2, precision=precision,
                           norm_outer=(s_outer, inv_o), z16=z16 and need, out16=z16 and need)   # (inference keeps fp32 rows: measured faster there)
        out = linear_launch(p_, W3, b3, resid=x, precision=precision)
        if need:
            ctx.save_for_backward(x, s_outer, s_inner, W1, W2, W3)
            ctx.aux = (act, precision, inv_o, inv_i, n, Z1, Z2, p_, b1 is not None, b2 is not None, b3 is not None)
        else:
            ctx.aux = None
        return out

    @staticmethod
    def backward(ctx, dy):
        if ctx.aux is None:
            raise RuntimeError("GatedMlpResidualFn: no saved activations (the forward ran under no_grad, or backward ran twice)")
        x, s_outer, s_inner, W1, W2, W3 = ctx.saved_tensors
        act, prec, inv_o, inv_i, n, Z1, Z2, p_, has_b1, has_b2, has_b3 = ctx.aux
        ctx.aux = None
        dy = ops._f32c(dy)
        M, K = x.shape
        N = W1.shape[0]
        dev = dy.device
        f = dict(dtype=torch.float32, device=dev)
        L = _capi.lib()
        # W3: weight gradient from (dy, p_); its input gradient goes straight through the gated product's backward
        dW3, db3 = torch.empty(K, N, **f), (torch.empty(K, **f) if has_b3 else None)
        ops.wgrad(_wgrad_jobs(dy, p_, dW3, db3), dev, prec)
        s16 = Z1.dtype == torch.bfloat16     # bf16 mode: dZ1 / dZ2 are bf16 numbers -- two-byte rows for the two dX launches and dW1 / dW2
        zt = dict(dtype=torch.bfloat16 if s16 else torch.float32, device=dev)
        dZ1, dZ2 = torch.empty(M, N, **zt), torch.empty(M, N, **zt)
        linear_launch(dy, W3, out=dZ1, precision=prec, w_transposed=True, act=act, gate_bwd=(Z1, Z2, dZ2), z16=s16, out16=s16)
        dn = input_gradient(dZ1, W1, precision=prec)
        dn = input_gradient(dZ2, W2, resid=dn, precision=prec)
        dW1, db1 = torch.empty(N, K, **f), (torch.empty(N, **f) if has_b1 else None)
        dW2, db2 = torch.empty(N, K, **f), (torch.empty(N, **f) if has_b2 else None)
        ops.wgrad(_wgrad_jobs(dZ1, n, dW1, db1) + _wgrad_jobs(dZ2, n, dW2, db2), dev, prec)

        # both norms backward in one pass; dx = dy (the residual) + the norm path
        dx, ds_io = torch.empty(M, K, **f), torch.empty(2 * K, **f)
        ws = torch.empty(max(L.mgn_rownorm_bwd_workspace_bytes(2 * K), 16), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = L.mgn_rownorm2_bwd(dn.data_ptr(), x.data_ptr(), int(x.stride(0)), K, inv_o.data_ptr(), s_outer.data_ptr(), inv_i.data_ptr(),
                                    s_inner.data_ptr(), ops.EPS, M, dy.data_ptr(), dx.data_ptr(), ds_io.data_ptr(), ws.data_ptr(), ws.numel(),
                                    ops._stream(dev))
        _capi.check(rc, "mgn_rownorm2_bwd", dense=True)
        ds_inner, ds_outer = ds_io[:K], ds_io[K:]
        return dx, ds_outer, ds_inner, dW1, db1, dW2, db2, dW3, db3, None, None


#: MGN_DENSE_SAVE16=0: the bf16 mode's saved pre-activations of the gated half stay fp32 rows (A/B)
_SAVE16 = [os.environ.get("MGN_DENSE_SAVE16", "1") != "0"]
#: MGN_FUSED_MLP=0: Transformer blocks keep the gated-MLP half on separate autograd nodes (A/B)
_FUSED_MLP = [os.environ.get("MGN_FUSED_MLP", "1") != "0"]


class SigmoidGateFn(torch.autograd.Function):
    """y * sigmoid(G) on ``mgn_gate_fwd`` / ``mgn_gate_bwd`` (the gated attention of layers.py:688-693)"""

    @staticmethod
    def forward(ctx, y, G):
        y, G = ops._f32c(y), ops._f32c(G)
        gate, out = torch.empty_like(y), torch.empty_like(y)
        ops.gate_fwd(G, None, None, y, gate, out)
        ctx.save_for_backward(y, gate)
        return out

    @staticmethod
    def backward(ctx, d):
        y, gate = ctx.saved_tensors
        d = ops._f32c(d)
        dy, dG = torch.empty_like(y), torch.empty_like(y)
        ops.gate_bwd(d, y, gate, dy, dG)
        return dy, dG


class RMSNormFn(torch.autograd.Function):
    """stand-alone RMSNorm on ``mgn_rownorm_fwd`` / ``mgn_rownorm_bwd`` (layers.py:73-129; widths up to 384)"""

    @staticmethod
    def forward(ctx, x, scale):
        ops._require_device(x, scale)
        shape = x.shape
        x2d = _rows(x.reshape(-1, shape[-1]))
        M, K = x2d.shape
        y = torch.empty(M, K, dtype=torch.float32, device=x.device)
        inv = torch.empty(M, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _capi.lib().mgn_rownorm_fwd(x2d.data_ptr(), int(x2d.stride(0)), K, scale.data_ptr(), ops.EPS, M, y.data_ptr(), inv.data_ptr(),
                                             ops._stream(x.device))
        _capi.check(rc, "mgn_rownorm_fwd", dense=True)
        ctx.save_for_backward(x2d, scale, inv)
        ctx.shape = shape
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        x, scale, inv = ctx.saved_tensors
        M, K = x.shape
        dy = ops._f32c(dy.reshape(M, K))
        dev = dy.device
        dx = torch.empty(M, K, dtype=torch.float32, device=dev)
        dscale = torch.empty(K, dtype=torch.float32, device=dev) if M > 0 else torch.zeros(K, dtype=torch.float32, device=dev)
        if M > 0:
            L = _capi.lib()
            ws = torch.empty(max(L.mgn_rownorm_bwd_workspace_bytes(K), 16), dtype=torch.uint8, device=dev)
            arr = (_capi.RownormPhase * 1)()
            arr[0].x, arr[0].ldx, arr[0].K, arr[0].idx, arr[0].dx, arr[0].lddx = x.data_ptr(), int(x.stride(0)), K, None, dx.data_ptr(), K
            with torch.cuda.device(dev):
                rc = L.mgn_rownorm_bwd(dy.data_ptr(), arr, 1, inv.data_ptr(), scale.data_ptr(), ops.EPS, M, dscale.data_ptr(), ws.data_ptr(),
                                       ws.numel(), ops._stream(dev))
            _capi.check(rc, "mgn_rownorm_bwd", dense=True)
        return dx.reshape(ctx.shape), dscale


def rms_norm(x, scale):
    return RMSNormFn.apply(x, scale)
