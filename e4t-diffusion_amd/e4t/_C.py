"""ctypes binding of libe4t_hip.so (the C ABI declared in include/e4t_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved, importing a
kernel raises.  The library is built in-tree by ``__graft_entry__.build()`` (csrc/build.sh).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# E4T_LIB: another build of the same library (tools/build_variant.sh: compile-time A/B of kernel variants); default = the in-tree build
LIB_PATH = os.environ.get("E4T_LIB") or os.path.join(_HERE, "libe4t_hip.so")

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", vp), ("A2", vp), ("B", vp), ("C", vp), ("bias", vp), ("residual", vp), ("rowbias", vp),
        ("workspace", vp), ("workspace_bytes", sz),
        ("M", i32), ("N", i32), ("K", i32), ("K1", i32),
        ("lda", i32), ("lda2", i32), ("ldb", i32), ("ldc", i32), ("ldr", i32),
        ("rows_per_batch", i32), ("flags", i32), ("tile", i32), ("splitk", i32), ("batch", i32),
        ("strideA", i64), ("strideB", i64), ("strideC", i64), ("strideBias", i64),
        ("alpha", f32), ("ldrb", i32), ("colstats", vp),
        ("panel_rows", i32), ("panel_stride", i32), ("panel_off", i32),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("X", vp), ("W", vp), ("Y", vp), ("bias", vp), ("residual", vp), ("rowbias", vp),
        ("workspace", vp), ("workspace_bytes", sz),
        ("B", i32), ("Hin", i32), ("Win", i32), ("Cin", i32), ("Hout", i32), ("Wout", i32), ("Cout", i32),
        ("mode", i32), ("flags", i32), ("tile", i32), ("splitk", i32), ("ldrb", i32), ("colstats", vp),
    ]


class GemmPlan(C.Structure):
    """e4t_gemm_plan_t: what the launcher will run for a descriptor (tile code, tile dims, split-K, workspace it wants)"""
    _fields_ = [("tile", i32), ("tile_m", i32), ("tile_n", i32), ("splitk", i32), ("workspace_bytes", sz), ("tail_rows", i32), ("stages", i32)]


class WODesc(C.Structure):
    _fields_ = (
        [(n, vp) for n in ("v", "w1", "b1", "w2", "b2", "wc", "bc", "wr", "br", "W", "vecs", "partial", "weff", "weffT", "dweff")]
        + [(n, vp) for n in ("g_v", "g_w1", "g_b1", "g_w2", "g_b2", "g_wc", "g_bc", "g_wr", "g_br", "g_W")]
        + [(n, i32) for n in ("row", "col", "ld_weff", "ld_weffT", "ld_dweff", "mode")]
    )


# symbol -> (restype, argtypes).  Every symbol declared in include/e4t_hip.h must appear here
# (tests/test_abi.py checks the header against this table and against the built library).
SIGNATURES = {
    "e4t_version": (i32, []),
    "e4t_build_flags": (i32, []),
    "e4t_last_error": (C.c_char_p, []),
    "e4t_device_info": (i32, [C.c_char_p, i32, C.POINTER(i32)]),
    "e4t_set_launch_log": (i32, [C.c_char_p]),
    "e4t_gemm_nt": (i32, [C.POINTER(GemmDesc), vp]),
    "e4t_gemm_tn": (i32, [C.POINTER(GemmDesc), vp]),
    "e4t_conv3x3": (i32, [C.POINTER(ConvDesc), vp]),
    "e4t_gemm_plan": (i32, [C.POINTER(GemmDesc), C.POINTER(GemmPlan)]),
    "e4t_gemm_tn_plan": (i32, [C.POINTER(GemmDesc), C.POINTER(GemmPlan)]),
    "e4t_conv3x3_plan": (i32, [C.POINTER(ConvDesc), C.POINTER(GemmPlan)]),
    "e4t_attention_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, f32, i32, vp]),
    "e4t_attention_bwd": (i32, [vp] * 10 + [i32] * 9 + [i64] * 4 + [f32, i32, vp]),
    "e4t_attention_bwd_workspace_floats": (sz, [i32] * 5),
    "e4t_attention_bwd_ws": (i32, [vp] * 7 + [sz] + [vp] * 3 + [i32] * 9 + [i64] * 4 + [f32, i32, vp]),
    "e4t_groupnorm_num_chunks": (i32, [i32, i32]),
    "e4t_groupnorm_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "e4t_groupnorm_stats": (i32, [vp, i32, vp, i32, i32, i32, i32, f32, vp, vp, sz, vp]),
    "e4t_groupnorm_apply": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "e4t_groupnorm_fwd": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, sz, vp]),
    "e4t_groupnorm_fwd_cs": (i32, [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, sz, vp]),
    "e4t_groupnorm_bwd": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "e4t_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "e4t_layernorm_fwd_f32": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "e4t_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "e4t_colreduce_splits": (i32, [i32]),
    "e4t_colsum": (i32, [vp, i32, i32, i32, vp, i32, vp, sz, vp]),
    "e4t_layernorm_param_grad": (i32, [vp, vp, vp, i32, i32, vp, vp, i32, vp, sz, vp]),
    "e4t_wo_vecs_floats": (sz, [i32, i32]),
    "e4t_wo_partial_floats": (sz, [i32, i32]),
    "e4t_wo_forward": (i32, [vp, i32, i32, i32, vp]),
    "e4t_wo_backward": (i32, [vp, i32, i32, i32, i32, vp]),
    "e4t_weight_prepare": (i32, [vp, i32, i32, i32, vp]),
    "e4t_conv_weight_prepare": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "e4t_geglu_fwd": (i32, [vp, vp, i64, i32, vp]),
    "e4t_geglu_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "e4t_unary": (i32, [vp, vp, vp, i64, i32, vp]),
    "e4t_add": (i32, [vp, vp, vp, i64, vp]),
    "e4t_transpose": (i32, [vp, vp, i32, i32, i32, i32, i32, i64, i64, vp]),
    "e4t_sumpool2": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "e4t_spatial_mean": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "e4t_spatial_mean_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "e4t_timestep_embedding": (i32, [vp, vp, i32, i32, vp]),
    "e4t_clip_preprocess": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "e4t_guided_step": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "e4t_image_prep": (i32, [vp, vp, vp, i32, i32, vp]),
    "e4t_softmax_rows": (i32, [vp, i64, i32, i32, vp]),
    "e4t_im2col3_rgb": (i32, [vp, vp, i32, i32, i32, vp]),
    "e4t_im2col_T": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "e4t_im2col": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "e4t_adamw": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
    "e4t_adamw_hyper": (i32, [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, vp]),
    "e4t_adamw_rank": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, f32, f32, f32, f32, f32, i32, f32, vp, vp]),
    "e4t_sumsq_partial": (i32, [vp, i64, vp, i32, vp]),
    "e4t_probe_mfma_layout": (i32, [vp, vp, vp]),
    "e4t_comm_unique_id": (i32, [vp]),
    "e4t_comm_init": (i32, [C.POINTER(vp), vp, i32, i32]),
    "e4t_comm_allreduce": (i32, [vp, vp, i64, i32, i32, vp]),
    "e4t_comm_allgather": (i32, [vp, vp, vp, i64, i32, vp]),
    "e4t_comm_wait": (i32, [vp, vp]),
    "e4t_comm_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)]),
    "e4t_comm_destroy": (i32, [vp]),
}

# flag / enum mirrors of the header
OUT_F32, RES_F32, ACT_GELU, ACCUM, REDUCE_BATCH = 1, 2, 4, 8, 16
CONV_S1, CONV_S2, CONV_UP2, CONV_S2T, CONV_S2A = 1, 2, 3, 4, 5
WO_STORE_F32, WO_OFFSETS_ONLY = 1, 2
COMM_F32, COMM_BF16 = 0, 1
COMM_SUM, COMM_AVG, COMM_MIN, COMM_MAX = 0, 1, 2, 3
OP_SILU, OP_SILU_BWD, OP_GELU, OP_GELU_BWD, OP_LRELU, OP_LRELU_BWD, OP_QGELU, OP_QGELU_BWD = range(8)

_lib = None


class E4TError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise E4TError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU / PyTorch fallback for the E4T hot path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud failure
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str) -> int:
    """negative = error (raises); 0 / positive = success (a few entry points return a positive status, e.g. 'colstats written')"""
    if code >= 0:
        return code
    msg = load().e4t_last_error()
    raise E4TError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")
