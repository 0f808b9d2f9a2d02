"""ctypes binding of libvlhip.so (C ABI declared in include/vlhip.h).

The product path has NO fallback: if the shared library is missing or a symbol cannot be resolved the
import of the native ops fails loudly (RuntimeError), and every native op refuses CPU tensors.
"""
import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLHIP_LIBRARY") or os.path.join(_HERE, "csrc", "libvlhip.so")  # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "vlhip.h")

P = c_void_p
_SIGS = {
    "vl_version": (c_int, []),
    "vl_last_error": (c_char_p, []),
    "vl_gemm_nt": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int64, c_int64, c_int, c_int, P, P, P, c_int64,
                           P, P, P, c_int64, P]),
    "vl_gemm_nt_ex": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int64, c_int64, c_int, c_int, P, P, P, c_int64,
                              P, P, P, c_int64, P, P]),
    "vl_gemm_small_ws_floats": (c_int64, [c_int64, c_int64, c_int64]),
    "vl_gemm_nt_path": (c_int, [c_int64, c_int64, c_int64, c_int, c_int]),
    "vl_gemm_splitk_plan": (c_int64, [c_int64, c_int64, c_int64]),
    "vl_gemm_splitk_ws_floats": (c_int64, [c_int64, c_int64, c_int64]),
    "vl_gemm_nt_splitk": (c_int, [P, c_int64, P, c_int64, c_int64, c_int64, c_int64, c_int64, P, P, P]),
    "vl_gemm_tn_splitk": (c_int, [P, c_int64, P, c_int64, c_int64, c_int64, c_int64, c_int64, P, P, P]),
    "vl_ln_bwd_reduce": (c_int, [P, c_int64, c_int64, P, P, P, P]),
    "vl_ln_bwd_reduce2": (c_int, [P, c_int64, P, P, P, P, c_int64, P, P, P, c_int64, c_int, P]),
    "vl_stack_desc_len": (c_int64, [c_int64]),
    "vl_stack_fwd": (c_int, [P, c_int64, c_int64, P, P]),
    "vl_stack_bwd": (c_int, [P, c_int64, c_int64, P, P]),
    "vl_blocked_elems": (c_int64, [c_int64, c_int64]),
    "vl_transpose_blocked": (c_int, [P, c_int64, c_int64, c_int64, P]),
    "vl_colsum_finalize": (c_int, [P, c_int64, c_int64, P, c_int64, c_int, P]),
    "vl_dw_grouped": (c_int, [P, c_int64, c_int64, c_int, P]),
    "vl_dw_grouped_rowmajor": (c_int, [P, c_int64, c_int64, c_int, P]),
    "vl_dw_grouped_mixed": (c_int, [P, c_int64, c_int64, c_int, c_int, P]),
    "vl_dw_streamk_ws_bytes": (c_int64, [c_int64]),
    "vl_dw_grouped_streamk": (c_int, [P, c_int64, c_int64, c_int, c_int, c_int64, P, c_int64, P]),
    "vl_colreduce_multi": (c_int, [P, c_int64, c_int, P]),
    "vl_qkv_attention_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_float,
                                     c_uint64, P]),
    "vl_qkv_attention_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_uint64,
                                     P]),
    "vl_attn2_fwd": (c_int, [P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_uint64, P]),
    "vl_attn2_bwd": (c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_uint64, P]),
    "vl_ln_fwd": (c_int, [P, P, P, c_int64, P, P, P, P, c_float, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64,
                          c_int64, c_float, c_float, c_uint64, c_int64, c_int64, P]),
    "vl_ln_fwd_rr": (c_int, [P, P, P, P, c_int64, P, P, P, P, c_float, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64,
                             c_int64, c_float, c_float, c_uint64, c_int64, c_int64, P]),
    "vl_ln_bwd_ws_floats": (c_int64, [c_int64, c_int64]),
    "vl_ln_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64,
                          c_float, c_float, c_uint64, c_int64, P]),
    "vl_memset_zero": (c_int, [P, c_int64, P]),
    "vl_gqa_loss_ws_bytes": (c_int64, [c_int64]),
    "vl_gqa_loss": (c_int, [P, P, P, c_int64, c_int64, c_float, P, P, P, P]),
    "vl_mask_mul": (c_int, [P, P, P, c_int64, P]),
    "vl_weight_prep": (c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, P]),
    "vl_imp_ws_bytes": (c_int64, [c_int64]),
    "vl_imp_select": (c_int, [P, P, P, c_int64, c_int64, P, P]),
    "vl_weight_prep_multi": (c_int, [P, c_int64, c_int64, P]),
    "vl_split_f32": (c_int, [P, P, P, c_int64, P]),
    "vl_act_fwd": (c_int, [P, c_int64, c_int64, c_int, c_float, c_uint64, P, P, P, c_int64, P]),
    "vl_act_bwd": (c_int, [P, P, c_int64, c_int64, c_int, c_float, c_uint64, P, P, c_int64, P]),
    "vl_transpose_bf16": (c_int, [P, P, c_int64, c_int64, c_int64, c_int64, P]),
    "vl_colsum_ws_floats": (c_int64, [c_int64, c_int64]),
    "vl_colsum_bf16": (c_int, [P, c_int64, c_int64, c_int64, P, P, P]),
    "vl_addmask": (c_int, [P, P, P, c_int64, c_int64, c_int64, P]),
    "vl_embed_text_fwd": (c_int, [P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, P]),
    "vl_embed_text_bwd": (c_int, [P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, P, P]),
    "vl_embed_gather_fwd": (c_int, [P, P, P, c_int64, c_int64, P]),
    "vl_embed_scatter_add": (c_int, [P, P, P, c_int64, c_int64, c_int64, P, P]),
    "vl_loc_linear_fwd": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, P]),
    "vl_loc_linear_bwd": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, P, P]),
    "vl_loc_bwd_ws_floats": (c_int64, [c_int64, c_int64]),
    "vl_scatter_det_ws_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "vl_scatter_add_det": (c_int, [P, c_int64, P, c_int64, c_int64, P, c_int64, P]),
    "vl_adamw": (c_int, [P, P, P, P, c_int64, P, P, P, c_int64, c_float, c_float, c_float, c_int64, P, c_int, c_float,
                         P, c_float, P, c_float, c_float, P, c_int, P, c_int64, c_int64, c_int64, P]),
    "vl_sumsq_ws_floats": (c_int64, []),
    "vl_sumsq": (c_int, [P, c_int64, P, P, P]),
    "vl_sumsq_flagged": (c_int, [P, c_int64, P, P, c_int64, c_int64, c_int64, P, P]),
}


def header_symbols():
    """Every function name declared in include/vlhip.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(vl_[a-z0-9_]+)\s*\(", text)))


def header_constants():
    """{name: value} of the `NAME = value` enumerators and `#define NAME value` integer constants of include/vlhip.h
    (descriptor field indices of vl_stack_*): the header is the single source of truth for the binding."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    out = {}
    for name, val in re.findall(r"\b(VL_[A-Z0-9_]+)\s*=\s*(-?\d+)", text):
        out[name] = int(val)
    for name, val in re.findall(r"#define\s+(VL_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|-?\d+)(?:ll)?\b", text):
        out[name] = int(val, 0)
    return out


_lib = None


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "clg_vqa_amd: native library %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C clg-vqa_amd/csrc`).  There is no CPU / eager fallback for the product path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise RuntimeError("clg_vqa_amd: %s does not export %s" % (LIB_PATH, name)) from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().vl_last_error().decode(errors="replace")
        raise RuntimeError("vlhip %s failed (code %d): %s" % (what, rc, msg))
