"""ctypes binding of libm3ae_hip.so (include/m3ae_hip.h).  Loading fails loudly: there is no CPU fallback."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# M3AE_DIAGNOSTIC_LIB=1: load the timing-only / traced build of m3ae_amd/build.py (M3AE_EXTRA_HIPCC_FLAGS -> lib_diag/); tools only
DIAGNOSTIC = os.environ.get("M3AE_DIAGNOSTIC_LIB", "") == "1"
LIB_PATH = os.path.join(_PKG, "lib_diag" if DIAGNOSTIC else "lib", "libm3ae_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_QUICKGELU, ACT_TANH, ACT_RELU, ACT_MULAUX = 0, 1, 2, 3, 4, 5

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", i64), ("N", i64), ("K", i64), ("batch1", i64), ("batch2", i64),
        ("A", vp), ("a_sm", i64), ("a_sk", i64), ("a_sb1", i64), ("a_sb2", i64),
        ("B", vp), ("b_sk", i64), ("b_sn", i64), ("b_sb1", i64), ("b_sb2", i64),
        ("C", vp), ("c_sm", i64), ("c_sn", i64), ("c_sb1", i64), ("c_sb2", i64),
        ("dtype_a", i32), ("dtype_b", i32), ("dtype_c", i32),
        ("alpha", f32), ("accumulate", i32), ("bias", vp), ("act", i32), ("preact", vp), ("residual", vp),
        ("dact_aux", vp), ("dact", i32), ("force_generic", i32), ("a_rowsum", vp),
        ("dropout_p", f32), ("dropout_seed", C.c_uint64), ("preact_grad", i32), ("launch_flags", i32), ("dropout_salt", vp),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("B", i64), ("H", i64), ("Lq", i64), ("Lk", i64), ("Dh", i64),
        ("q", vp), ("q_sb", i64), ("q_sl", i64),
        ("k", vp), ("k_sb", i64), ("k_sl", i64),
        ("v", vp), ("v_sb", i64), ("v_sl", i64),
        ("o", vp), ("o_sb", i64), ("o_sl", i64),
        ("key_mask", vp), ("pos_bias", vp), ("scale", f32), ("causal", i32),
        ("lse", vp), ("lse_stride", i64), ("dtype", i32), ("workspace", vp), ("workspace_bytes", i64),
        ("d_o", vp), ("dq", vp), ("dk", vp), ("dv", vp), ("delta", vp), ("d_pos_bias", vp),
        ("dropout_p", f32), ("dropout_seed", C.c_uint64), ("dropout_salt", vp), ("launch_flags", i32),
    ]


class XattnDesc(C.Structure):
    _fields_ = [
        ("dir", i32), ("B", i64), ("Lq", i64), ("Lk", i64), ("D", i64), ("H", i64),
        ("x", vp), ("y", vp), ("key_mask", vp),
        ("wq", vp), ("wq_t", vp), ("wkv", vp), ("wkv_t", vp), ("wo", vp), ("wo_t", vp),
        ("bq", vp), ("bkv", vp), ("bo", vp), ("ln_g", vp), ("ln_b", vp),
        ("ln_eps", f32), ("dropout_p", f32), ("seed_attn", C.c_uint64), ("seed_hidden", C.c_uint64),
        ("proj", vp), ("prime", vp), ("colbias", vp), ("probs", vp), ("probs_drop", vp), ("rowsum", vp),
        ("zctx", vp), ("ctx", vp), ("s", vp), ("out", vp), ("mean", vp), ("rstd", vp),
        ("d_out", vp), ("dx", vp), ("dy", vp),
        ("g_wq", vp), ("g_wkv", vp), ("g_wo", vp), ("g_bq", vp), ("g_bkv", vp), ("g_bo", vp), ("g_ln_g", vp), ("g_ln_b", vp),
        ("ws_ds", vp), ("ws_dsd", vp), ("ws_dscores", vp), ("ws_dprime", vp), ("ws_dproj", vp), ("ws_dz", vp), ("ws_dctx", vp),
        ("ws_vec", vp), ("ws_ln", vp), ("launch_flags", i32), ("dropout_salt", vp),
    ]


GEMM_NO_PERSISTENT = 1                            # m3ae_gemm_desc.launch_flags
ATTN_LEGACY_KERNELS = 1                           # m3ae_attn_desc.launch_flags
XATTN_NO_PERSISTENT, XATTN_LEGACY_CHAIN = 1, 2    # m3ae_xattn_desc.launch_flags
ABI_VERSION = 3


_SIGS = {
    "m3ae_abi_version": (C.c_int, []),
    "m3ae_desc_sizes": (None, [C.POINTER(i64)]),
    "m3ae_last_gemm_path": (C.c_char_p, []),
    "m3ae_gemm": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "m3ae_attn_workspace_bytes": (i64, [C.POINTER(AttnDesc), C.c_int]),
    "m3ae_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), vp]),
    "m3ae_attn_bwd": (C.c_int, [C.POINTER(AttnDesc), vp]),
    "m3ae_xattn_supported": (C.c_int, [C.POINTER(XattnDesc)]),
    "m3ae_xattn_bwd_supported": (C.c_int, [C.POINTER(XattnDesc)]),
    "m3ae_xattn_probs_ld": (i64, [C.POINTER(XattnDesc)]),
    "m3ae_xattn_fwd": (C.c_int, [C.POINTER(XattnDesc), vp]),
    "m3ae_xattn_bwd": (C.c_int, [C.POINTER(XattnDesc), vp]),
    "m3ae_layernorm_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, f32, C.c_int, C.c_int, C.c_int, vp]),
    "m3ae_layernorm_bwd_blocks": (i64, [i64]),
    "m3ae_layernorm_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, vp]),
    "m3ae_layernorm_bwd_drop": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, f32, C.c_uint64, vp, vp, vp, vp, i64, i64, C.c_int, vp]),
    "m3ae_dropout": (C.c_int, [vp, vp, vp, i64, i64, f32, C.c_uint64, vp, C.c_int, vp]),
    "m3ae_colsum": (C.c_int, [vp, vp, i64, i64, i64, C.c_int, C.c_int, vp]),
    "m3ae_roberta_embed_fwd": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, i64, C.c_int, vp]),
    "m3ae_roberta_embed_bwd": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, i64, C.c_int, vp]),
    "m3ae_patchify": (C.c_int, [vp, vp, i64, i64, i64, C.c_int, vp]),
    "m3ae_image_normalize_u8": (C.c_int, [vp, vp, i64, i64, i64, C.POINTER(C.c_float), C.POINTER(C.c_float), vp]),
    "m3ae_vit_tokens_fwd": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, C.c_int, vp]),
    "m3ae_vit_tokens_bwd": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, C.c_int, vp]),
    "m3ae_bce_logits": (C.c_int, [vp, vp, vp, vp, i64, i64, f32, C.c_int, vp]),
    "m3ae_xent": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, f32, C.c_int, vp]),
    "m3ae_adamw": (C.c_int, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, f32, vp, vp]),
    "m3ae_cast_transpose": (C.c_int, [vp, vp, vp, i64, i64, vp]),
    "m3ae_cast_transpose_batched": (C.c_int, [vp, C.c_int, i64, vp]),
    "m3ae_transpose_bf16_batched": (C.c_int, [vp, C.c_int, i64, vp]),
    "m3ae_cast": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp]),
    "m3ae_zero": (C.c_int, [vp, i64, vp]),
    "m3ae_add": (C.c_int, [vp, vp, vp, i64, C.c_int, vp]),
    "m3ae_act_fwd": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp]),
    "m3ae_act_bwd": (C.c_int, [vp, vp, vp, i64, C.c_int, C.c_int, vp]),
    "m3ae_gather_rows": (C.c_int, [vp, vp, vp, i64, i64, C.c_int, vp]),
    "m3ae_scatter_add_rows": (C.c_int, [vp, vp, vp, i64, i64, C.c_int, vp]),
    "m3ae_mask_ranks": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, vp]),
    "m3ae_mim_targets": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, C.c_int, vp]),
    "m3ae_mim_loss_fwd": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, C.c_int, vp]),
    "m3ae_mim_loss_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, i64, C.c_int, vp]),
    "m3ae_selftest": (C.c_int, [vp, vp]),
}

EXPORTS = tuple(_SIGS)
_lib = None


class M3AEHipError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises if it has not been built (python -m m3ae_amd.build) -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise M3AEHipError(f"{LIB_PATH} is missing: build it with `python -m m3ae_amd.build` "
                               f"(or __graft_entry__.build()); this package has no CPU fallback")
        # PyTorch-ROCm ships its own libamdhip64; a process must run on ONE HIP runtime.  Loading this library first would
        # bring in the system runtime (DT_NEEDED) and torch would then initialise on that one: every launch of this library
        # afterwards fails with hipErrorNoDevice (seen with __graft_entry__.build() followed by smoke() in one process).
        # Importing torch first makes its runtime the process's runtime.
        import torch  # noqa: F401
        if DIAGNOSTIC:
            import warnings
            warnings.warn(f"m3ae_amd: loading the DIAGNOSTIC library {LIB_PATH} (timing-only builds may compute wrong results)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        if l.m3ae_abi_version() != ABI_VERSION:
            raise M3AEHipError(f"ABI version mismatch: library {l.m3ae_abi_version()}, binding {ABI_VERSION}")
        sizes = (i64 * 3)()
        l.m3ae_desc_sizes(sizes)
        mine = (C.sizeof(GemmDesc), C.sizeof(AttnDesc), C.sizeof(XattnDesc))
        if tuple(sizes) != mine:   # a descriptor that is too short would be read past its end
            raise M3AEHipError(f"descriptor layout mismatch: library {tuple(sizes)}, binding {mine}")
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported shape/dtype", -3: "misaligned", -4: "workspace too small"}.get(
            rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
        raise M3AEHipError(f"{what} failed: {kind}")
