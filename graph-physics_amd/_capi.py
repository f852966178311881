"""ctypes binding of the C-ABI library ``csrc/libmgn_hip.so`` (include/mgn_hip.h).

The product path has no CPU fallback: if the library cannot be loaded (or built
with hipcc when missing), every op raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_REPO = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("MGN_LIB") or os.path.join(_CSRC, "libmgn_hip.so")
SOURCES = [os.path.join(_CSRC, "mgn_kernels.hip"), os.path.join(_CSRC, "mgn_prep.hip"), os.path.join(_CSRC, "mgn_attn.hip"),
           os.path.join(_CSRC, "mgn_dense.hip")]
DEPS = [os.path.join(_CSRC, "mgn_x6.inc"), os.path.join(_CSRC, "mgn_fused.inc"), os.path.join(_CSRC, "mgn_pp.inc"), os.path.join(_CSRC, "mgn_ppr.inc")]  # included by the source
HEADER = os.path.join(_REPO, "include", "mgn_hip.h")
# No packed-fp32 VALU (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) in device code: beside a SIMD partner that streams MFMAs
# one of them takes ~70 cycles instead of ~6 (tools/coissue_probe.hip; plain VALU: 8), and hipcc forms them everywhere --
# the SLP vectorizer out of scalar code, the legalizer out of float4 arithmetic (csrc/Makefile carries the same flags).
DEVICE_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]

MAX_LAYERS = 8
MAX_PHASES = 3
MAX_WGRAD_JOBS = 48

_f32p = C.c_void_p  # raw device pointers travel as integers
_i32p = C.c_void_p


class MlpFwdArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int64),
        ("H", C.c_int), ("NL", C.c_int), ("nphase", C.c_int),
        ("src", _f32p * MAX_PHASES),
        ("idx", _i32p * MAX_PHASES),
        ("kw", C.c_int * MAX_PHASES),
        ("W", _f32p * MAX_LAYERS),
        ("b", _f32p * MAX_LAYERS),
        ("scale", _f32p),
        ("eps", C.c_float),
        ("out_w", C.c_int),
        ("resid", _f32p),
        ("out", _f32p),
        ("y_out", _f32p),
        ("saveH", _f32p * MAX_LAYERS),
        ("saveU", _f32p),
        ("saveR", _f32p),
        ("ldw0", C.c_int), ("n_add", C.c_int),
        ("add_src", _f32p * 2), ("add_idx", _i32p * 2),
        ("n_post", C.c_int), ("post_ldw", C.c_int),
        ("post_W", _f32p * 2), ("post_out", _f32p * 2),
        ("wpk", C.c_void_p * 8),
        ("saveM", C.c_void_p * MAX_LAYERS),
        ("precision", C.c_int),
        ("out_relu", C.c_int),
        ("seg_key", C.c_void_p), ("seg_rowptr", C.c_void_p), ("seg_out", _f32p), ("seg_part", _f32p),
        ("act", C.c_int),
        ("saveZ", _f32p * MAX_LAYERS),
    ]


class MlpBwdArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int64),
        ("H", C.c_int), ("NL", C.c_int),
        ("dOut", _f32p),
        ("dOut2", _f32p),
        ("idx2", _i32p),
        ("out_w", C.c_int),
        ("U", _f32p), ("R", _f32p), ("scale", _f32p), ("eps", C.c_float),
        ("Hs", _f32p * MAX_LAYERS),
        ("WT", _f32p * MAX_LAYERS),
        ("dZ", _f32p * MAX_LAYERS),
        ("n_din", C.c_int),
        ("WT0", _f32p * MAX_PHASES),
        ("din_resid", _f32p * MAX_PHASES),
        ("dIn", _f32p * MAX_PHASES),
        ("db", _f32p * MAX_LAYERS),
        ("dscale", _f32p),
        ("red_ws", C.c_void_p), ("red_ws_bytes", C.c_size_t),
        ("wpk", C.c_void_p * 8),
        ("Ms", C.c_void_p * MAX_LAYERS),
        ("precision", C.c_int),
        ("n_front", C.c_int),
        ("front_src", _f32p * MAX_PHASES),
        ("front_resid", _f32p),
        ("front_out", _f32p),
        ("defer_reduce", C.c_int),
        ("act", C.c_int),
        ("Zs", _f32p * MAX_LAYERS),
        ("seg_key", C.c_void_p), ("seg_rowptr", C.c_void_p), ("seg_out", _f32p), ("seg_part", _f32p),
    ]


class ColredJob(C.Structure):
    _fields_ = [("red_ws", C.c_void_p), ("M", C.c_int64), ("H", C.c_int), ("NL", C.c_int), ("out_w", C.c_int), ("n_din", C.c_int),
                ("db", _f32p * MAX_LAYERS), ("dscale", _f32p)]


class WgradJob(C.Structure):
    _fields_ = [
        ("A", _f32p), ("B", _f32p), ("dW", _f32p),
        ("M", C.c_int64),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldw", C.c_int),
        ("nja", C.c_int), ("nkb", C.c_int), ("kw", C.c_int),
        ("db", _f32p),
    ]


class WpackBlock(C.Structure):
    _fields_ = [("src", _f32p), ("dst", C.c_void_p), ("ld_src", C.c_int), ("transpose", C.c_int)]


WPACK_BYTES = 98304


class SimDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int64), ("E", C.c_int64),
        ("x", _f32p), ("x_w", C.c_int),
        ("y", _f32p), ("y_w", C.c_int),
        ("edge_attr", _f32p), ("edge_w", C.c_int),
        ("feat_start", C.c_int), ("feat_end", C.c_int), ("out_start", C.c_int), ("out_w", C.c_int), ("type_idx", C.c_int),
        ("acc_sum", _f32p * 3), ("acc_sumsq", _f32p * 3), ("acc_count", _f32p * 3), ("num_acc", _f32p * 3),
        ("accumulate", C.c_int * 3),
        ("std_eps", C.c_float),
        ("node_out", _f32p), ("target_out", _f32p), ("edge_out", _f32p),
        ("norm_w", C.c_int * 3),
        ("max_accumulations", C.c_float),
        ("type_err", C.c_void_p),
    ]


class EdgeBwdFusedArgs(C.Structure):
    """mgn_edge_bwd_fused_args (include/mgn_hip.h)"""
    _fields_ = [
        ("M", C.c_int64), ("dOut", _f32p), ("dAgg", _f32p), ("idx", C.c_void_p), ("U", _f32p), ("R", _f32p), ("scale", _f32p),
        ("eps", C.c_float), ("X", _f32p * 4), ("Ms", C.c_void_p * 3), ("wpk", C.c_void_p * 4), ("dIn", _f32p), ("dZ0", _f32p),
        ("dW", _f32p * 4), ("ldw", C.c_int * 4), ("db", _f32p * 4), ("dscale", _f32p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("precision", C.c_int),
    ]


class LinearArgs(C.Structure):
    """mgn_linear_args (include/mgn_hip.h)"""
    _fields_ = [
        ("M", C.c_int64), ("x", _f32p), ("ldx", C.c_int), ("K1", C.c_int), ("x2", _f32p), ("ldx2", C.c_int), ("K2", C.c_int),
        ("x3", _f32p), ("ldx3", C.c_int), ("K3", C.c_int), ("idx", C.c_void_p), ("idx2", C.c_void_p), ("idx3", C.c_void_p),
        ("norm_scale", _f32p), ("eps", C.c_float), ("inv_out", _f32p), ("n_out", _f32p), ("W", _f32p), ("ldw", C.c_int), ("b", _f32p),
        ("W2", _f32p), ("b2", _f32p), ("act", C.c_int), ("N", C.c_int), ("resid", _f32p), ("ldr", C.c_int), ("out", _f32p),
        ("ldo", C.c_int), ("saveZ1", _f32p), ("saveZ2", _f32p), ("precision", C.c_int), ("w_transposed", C.c_int),
        ("gb_z1", _f32p), ("gb_z2", _f32p), ("out2", _f32p), ("norm_scale_outer", _f32p), ("inv_outer_out", _f32p), ("z16", C.c_int), ("x16", C.c_int), ("out16", C.c_int),
    ]


class RownormPhase(C.Structure):
    """mgn_rownorm_phase (include/mgn_hip.h)"""
    _fields_ = [("x", _f32p), ("ldx", C.c_int), ("K", C.c_int), ("idx", C.c_void_p), ("dx", _f32p), ("lddx", C.c_int), ("acc", _f32p)]


class OptTensor(C.Structure):
    _fields_ = [("p", _f32p), ("g", _f32p), ("m", _f32p), ("v", _f32p), ("n", C.c_int64)]


class TBlock(C.Structure):
    _fields_ = [("src", _f32p), ("dst", _f32p), ("ld_src", C.c_int), ("ld_dst", C.c_int)]


#: every symbol include/mgn_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mgn_version": (C.c_int, []),
    "mgn_last_error": (C.c_char_p, []),
    "mgn_csr_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "mgn_csr_build": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_topology_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "mgn_topology_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6 + [C.POINTER(C.c_int32), C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "mgn_topology_build_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6 + [C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p]),
    "mgn_segsum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "mgn_seg_fix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mgn_segsum2": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int, C.c_void_p]),
    "mgn_segsum2_b16": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int, C.c_void_p]),
    "mgn_mlp_fwd": (C.c_int, [C.POINTER(MlpFwdArgs), C.c_void_p]),
    "mgn_mlp_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "mgn_mlp_bwd": (C.c_int, [C.POINTER(MlpBwdArgs), C.c_void_p]),
    "mgn_colred_batch": (C.c_int, [C.c_int, C.POINTER(ColredJob), C.c_void_p]),
    "mgn_wgrad_workspace_bytes": (C.c_size_t, [C.c_int, C.POINTER(WgradJob)]),
    "mgn_wgrad": (C.c_int, [C.c_int, C.POINTER(WgradJob), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_wgrad_p": (C.c_int, [C.c_int, C.POINTER(WgradJob), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "mgn_edge_bwd_fused_workspace_bytes": (C.c_size_t, []),
    "mgn_edge_bwd_fused": (C.c_int, [C.POINTER(EdgeBwdFusedArgs), C.c_void_p]),
    "mgn_debug_occupancy": (C.c_int, [C.POINTER(C.c_int)]),
    "mgn_transpose_blocks": (C.c_int, [C.c_int, C.POINTER(TBlock), C.c_int, C.c_void_p]),
    "mgn_wpack": (C.c_int, [C.c_int, C.POINTER(WpackBlock), C.c_void_p]),
    "mgn_faces_to_edges_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "mgn_faces_to_edges": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_world_edges_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "mgn_add_world_edges": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_edge_features": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mgn_sim_workspace_bytes": (C.c_size_t, []),
    "mgn_sim_pre": (C.c_int, [C.POINTER(SimDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_sim_post": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "mgn_clip_adamw_workspace_bytes": (C.c_size_t, [C.c_int, C.POINTER(OptTensor)]),
    "mgn_clip_adamw": (C.c_int, [C.c_int, C.POINTER(OptTensor), C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_masked_mse_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgn_masked_mse_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgn_clip_adamw_table_bytes": (C.c_size_t, [C.c_int, C.POINTER(OptTensor)]),
    "mgn_clip_adamw_table": (C.c_int, [C.c_int, C.POINTER(OptTensor), C.c_void_p, C.c_size_t]),
    "mgn_clip_adamw_t": (C.c_int, [C.c_int, C.POINTER(OptTensor), C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mgn_halo_unpack_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mgn_gate_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgn_gate_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgn_rope_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_int, C.c_void_p, C.c_void_p]),
    "mgn_rope_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgn_add_noise": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float),
                                C.c_int, C.c_uint64, C.c_uint32, C.c_void_p]),
    "mgn_morton_order_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "mgn_morton_order": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_prep_last_error": (C.c_char_p, []),
    "mgn_sparse_attn_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgn_sparse_attn_bwd": (C.c_int, [C.c_void_p] * 11 + [C.c_int64, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]),
    "mgn_sparse_attn_fwd_b16": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 4),
    "mgn_sparse_attn_bwd_b16": (C.c_int, [C.c_void_p] * 11 + [C.c_int64, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]),
    "mgn_sparse_attn_fwd_s": (C.c_int, [C.c_void_p, C.c_int64] * 3 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 4),
    "mgn_sparse_attn_bwd_s": (C.c_int, [C.c_void_p, C.c_int64] * 3 + [C.c_int] + [C.c_void_p] * 8 + [C.c_int64, C.c_int64, C.c_int, C.c_int]
                              + [C.c_void_p, C.c_int64] * 3 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_sparse_attn_weights": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgn_head_axis_attn_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 3),
    "mgn_head_axis_attn_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 4),
    "mgn_attn_last_error": (C.c_char_p, []),
    "mgn_linear_fwd": (C.c_int, [C.POINTER(LinearArgs), C.c_void_p]),
    "mgn_linear_accepts_transposed": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mgn_rownorm2_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_act_gate_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgn_rownorm_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgn_rownorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int]),
    "mgn_rownorm_bwd": (C.c_int, [C.c_void_p, C.POINTER(RownormPhase), C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "mgn_dense_last_error": (C.c_char_p, []),
}

_lib = None
_lock = threading.Lock()

#: ABI version this binding was written against (mgn_version() of the library must match: the
#: ctypes structs above mirror exactly that header)
EXPECTED_VERSION = 136
HASH_PATH = os.path.join(_CSRC, "libmgn_hip.srchash")
LOCK_PATH = os.path.join(_CSRC, ".build.lock")


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if p and os.path.exists(p):
            return p
    from shutil import which

    return which("hipcc")


def source_hash() -> str:
    """sha256 over the HIP sources, their includes and the header (file contents, not mtimes: a
    snapshot copy of the tree -- the GPU box -- must not look stale)."""
    import hashlib

    h = hashlib.sha256()
    for f in SOURCES + DEPS + [HEADER]:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read() + b"\0")
    h.update(" ".join(DEVICE_FLAGS).encode())  # a library built with other code-generation flags is stale too
    return h.hexdigest()


def needs_build() -> bool:
    if os.environ.get("MGN_LIB"):  # A/B experiments: an explicitly named build is taken as is
        return False
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(HASH_PATH) as fh:
            return fh.read().strip() != source_hash()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into csrc/libmgn_hip.so (in-tree).  Safe under
    ``torch.distributed.run``: the build is serialised by an flock, every process compiles into
    its own temporary file, and a process that waited for the lock re-checks before compiling."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build csrc/libmgn_hip.so")
    import fcntl

    with open(LOCK_PATH, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another rank built it while we waited
                return LIB_PATH
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + DEVICE_FLAGS + [
                   "-I", os.path.join(_REPO, "include"), "-o", tmp] + SOURCES
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, LIB_PATH)
            with open(f"{HASH_PATH}.{os.getpid()}.tmp", "w") as fh:
                fh.write(source_hash() + "\n")
            os.replace(f"{HASH_PATH}.{os.getpid()}.tmp", HASH_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


def lib():
    """The loaded C-ABI library.  Raises RuntimeError if it is missing / stale and cannot be
    rebuilt, or if its ABI version is not the one this binding mirrors."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if needs_build():
                try:
                    build()
                except Exception as e:  # noqa: BLE001
                    raise RuntimeError(
                        f"MI355X engine library {LIB_PATH} is missing or older than its sources and could not be "
                        f"rebuilt ({e}); run `python -c 'import __graft_entry__ as g; g.build()'`") from e
            try:
                L = C.CDLL(LIB_PATH)
            except OSError as e:
                raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
            v = L.mgn_version()
            if v != EXPECTED_VERSION:
                raise RuntimeError(f"{LIB_PATH} has ABI version {v}, this binding needs {EXPECTED_VERSION}: rebuild it "
                                   "(`python -c 'import __graft_entry__ as g; g.build()'`)")
            _lib = L
    return _lib


def check(rc: int, what: str, prep: bool = False, attn: bool = False, dense: bool = False):
    if rc != 0:
        fn = lib().mgn_dense_last_error if dense else (lib().mgn_attn_last_error if attn else (lib().mgn_prep_last_error if prep else lib().mgn_last_error))
        msg = fn().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
