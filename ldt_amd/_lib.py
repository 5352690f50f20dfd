"""ctypes binding of libldt_hip.so (C-ABI: include/ldt_hip.h).

The product path has NO fallback: if the HIP library is missing or an entry point
fails, this raises — it never routes through PyTorch eager or the CPU oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LDT_HIP_LIB", os.path.join(_HERE, "libldt_hip.so"))   # override: debug builds only
ABI_VERSION = 22
MAX_BLOCKS = 64

EPI_F32, EPI_BF16, EPI_GELU_BF16, EPI_RELU_BF16, EPI_RESID_F32 = range(5)
ACT_NONE, ACT_SILU, ACT_RELU, ACT_GELU = range(4)
# enum ldt_block_act: tools/utils.py:104-124 get_activation (unknown names fall to ReLU there; rrelu in eval mode)
BLOCK_ACTS = {"gelu": 1, "silu": 2, "swish": 2, "relu": 3, "leakyrelu": 4, "leakyrelu0.2": 5, "rrelu": 6, "hardswish": 7, "selu": 8}


def block_act_id(name):
    """None -> 0 (Identity); a name -> its enum ldt_block_act value (anything unknown is ReLU, as upstream)."""
    return 0 if name is None else BLOCK_ACTS.get(str(name).lower(), 3)
PROF_CLASSES = ("other", "gemm_io", "ln_modulate", "gemm_qkv", "attention", "gemm_o", "gemm_gelu", "gemm_dn")

_vp, _i32, _i64, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64


class ScorePlan(C.Structure):
    """Mirror of `ldt_score_plan` (include/ldt_hip.h)."""
    _fields_ = (
        [(n, _i32) for n in ("hidden", "heads", "blocks", "z_dim", "z_pad", "mlp_hidden", "tokens", "batch")]
        + [("w_in", _vp), ("b_in", _vp)]
        + [("w_qkv", _vp * MAX_BLOCKS), ("b_qkv", _vp * MAX_BLOCKS),
           ("w_o", _vp * MAX_BLOCKS), ("b_o", _vp * MAX_BLOCKS),
           ("w_up", _vp * MAX_BLOCKS), ("b_up", _vp * MAX_BLOCKS),
           ("w_dn", _vp * MAX_BLOCKS), ("b_dn", _vp * MAX_BLOCKS)]
        + [("w_out", _vp), ("b_out", _vp)]
        + [("w_q", _vp * MAX_BLOCKS), ("b_q", _vp * MAX_BLOCKS), ("kv_cond", _vp * MAX_BLOCKS), ("cond_tokens", _i32), ("_pad0", _i32)]
        + [("mod", _vp), ("mod_step_stride", _i64), ("mod_sample_stride", _i64)]
        + [(n, _vp) for n in ("xin", "X", "Hb", "QKV", "Ob", "U")]
        + [("fold", _vp), ("fold_step_stride", _i64), ("stats", _vp), ("gemm_wgs", _i32), ("fold_monitor_every", _i32), ("fold_monitor", _vp)]
    )


class CondArgs(C.Structure):
    """Mirror of `ldt_cond_args` (include/ldt_hip.h)."""
    _fields_ = ([(n, _vp) for n in ("temb", "extra", "w_ada", "b_ada", "c_buf", "mod_buf")] + [("t_dim", _i32), ("n_mod", _i32)] +
                [("w_ada_bf16", _vp), ("c_buf_bf16", _vp)])


# name -> argtypes; every symbol include/ldt_hip.h declares (tests/test_abi.py checks the two lists agree)
SIGNATURES = {
    "ldt_abi_version": [],
    "ldt_last_error": [],
    "ldt_cast_pad_bf16": [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp],
    "ldt_gemm_bf16": [_i32, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i64,
                      _i32, _i32, _i32, _vp],
    "ldt_gemm_resid_lnstats": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _i64,
                               _i32, _i32, _i32, _i32, _vp],
    "ldt_gemm_lnfold": [_i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "ldt_score_lnfold_route": [_i32, _i32, _i32, _i32],
    "ldt_layernorm_modulate": [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i64, _i32, _vp],
    "ldt_attention_fwd": [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ldt_attention_route": [_i32, _i32, _i32, _i32, _i32],
    "ldt_attention_oproj_resid": [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp],
    "ldt_sgemm": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "ldt_sinusoid": [_vp, _vp, _vp, _i32, _i32, _vp],
    "ldt_sampler_step": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _u64, _i32, _i32, _vp],
    "ldt_philox_normal": [_vp, _i64, _i64, _i32, _u64, _vp],
    "ldt_batch_norm_sum": [_vp, _i32, _i64, _vp, _vp, _vp],
    "ldt_langevin_coef": [_vp, _i32, C.c_float, C.c_float, _vp, _vp],
    "ldt_pndm_transfer": [_vp, _vp, C.c_float, C.c_float, C.c_float, _vp, _i64, _vp],
    "ldt_lincomb4": [_vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _i64, _vp],
    "ldt_vpsde_score": [_vp, _vp, C.c_float, C.c_float, C.c_float, _vp, _i32, _i64, _vp],
    "ldt_sde_score": [_vp, _vp, _i32, C.c_float, C.c_float, C.c_float, _vp, _i32, _i64, _vp],
    "ldt_add_f32": [_vp, _vp, _vp, _i64, _vp],
    "ldt_block_activation": [_vp, _i64, _i64, _i32, _i32, _vp],
    "ldt_group_stats": [_vp, _i64, _i32, _i32, _i32, _i32, C.c_float, _vp, _vp],
    "ldt_norm_apply": [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "ldt_widen_bf16": [_vp, _vp, _i64, _vp],
    "ldt_fold_mean_ratio": [_vp, _i32, _i64, _i32, _vp, _vp],
    "ldt_fps": [_vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "ldt_norm_points": [_vp, _i32, _i32, _vp, _vp],
    "ldt_mixture_seed": [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _vp],
    "ldt_knn": [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "ldt_group_normalize": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp],
    "ldt_gather_rows": [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "ldt_maxpool": [_vp, _i32, _i64, _i64, _i32, _i32, _vp, _vp],
    "ldt_actnorm": [_vp, _vp, _vp, _i64, _i64, _vp],
    "ldt_reparam": [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, C.c_float, C.c_float, _vp],
    "ldt_chamfer": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "ldt_grouper_mlp": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "ldt_ln_mlp_resid": [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "ldt_ln_mlp_resid_next": [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64,
                              _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _i64, _vp],
    "ldt_ln_linear": [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _i64, _vp],
    "ldt_chamfer_pairwise": [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "ldt_emd_approx": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "ldt_score_forward": [C.POINTER(ScorePlan), _vp, _vp, _vp, _vp],
    "ldt_score_forward_profile": [C.POINTER(ScorePlan), _vp, _vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), _vp],
    "ldt_dbg_gemm_group_m": [_i32],
    "ldt_dbg_gemm_epi": [_i32],
    "ldt_dbg_gemm_wreg": [_i32],
    "ldt_sample_loop": [C.POINTER(ScorePlan), _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i64, _u64, _vp, _i32, C.POINTER(CondArgs), _vp, _i32, _vp],
}

_lib = None


class LdtHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LdtHipError(
            "HIP extension %s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or ldt_amd/csrc/build.sh).  There is no CPU fallback." % LIB_PATH)
    h = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(h, name)           # AttributeError => stale .so
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "ldt_last_error" else C.c_int
    if h.ldt_abi_version() != ABI_VERSION:
        raise LdtHipError("libldt_hip.so ABI %d != binding ABI %d: rebuild" % (h.ldt_abi_version(), ABI_VERSION))
    _lib = h
    return h


def check(rc, what=""):
    if rc != 0:
        msg = lib().ldt_last_error()
        raise LdtHipError("%s failed (status %d): %s" % (what or "libldt_hip", rc, (msg or b"").decode()))
