"""ctypes binding of libinfv_ltm.so (C ABI: include/infv_ltm.h).

There is no fallback: if the library is missing or lacks a symbol, importing an operator
raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (hipcc,
``--offload-arch=gfx950``); the built ``.so`` lives next to this file.
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own libamdhip64.so.7; libinfv_ltm.so needs the same SONAME.  torch must be
# imported first so that both share ONE HIP runtime (the stream handles torch hands us belong to
# it); loading ours first would bind /opt/rocm's copy and leave torch's streams foreign.
import torch  # noqa: F401  (side effect: loads torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# INFV_LTM_LIBRARY=exp selects the experiments build of the same sources (-DINFV_EXPERIMENTS: timing, fault-injection and A/B
# knobs compiled in, csrc/knobs.h) -- development tools and the tests of non-default variants only; a path selects that file.
_WHICH = os.environ.get("INFV_LTM_LIBRARY", "")
LIB_PATH = (os.path.join(_HERE, "libinfv_ltm_exp.so") if _WHICH == "exp" else
            (_WHICH if _WHICH else os.path.join(_HERE, "libinfv_ltm.so")))
ABI_VERSION = 5
MAX_LAYERS = 8

i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_basis", "n_heads", "head_size", "d_in", "tokens_per_frame", "n_layers",
        "nb_samples", "sticky", "max_q", "max_batch_chunks")]


class PlanStruct(C.Structure):
    _fields_ = [
        ("T", C.c_int32),
        ("first_rows", C.c_int32),
        ("first_row_box", i32p), ("first_row_begin", i32p), ("first_row_end", i32p),
        ("first_box_val", f32p),
        ("inf_rows", C.c_int32),
        ("inf_row_box", i32p), ("inf_row_begin", i32p), ("inf_row_end", i32p),
        ("inf_box_val", f32p),
        ("inf_old_ptr", i32p), ("inf_old_slot", i32p),
        ("readout_w", f32p), ("readout_w_out", C.c_float),
        ("n_bins", C.c_int32),
        ("edge_box", i32p), ("edge_dx", f32p), ("bin_box", i32p),
        ("uniform_idx", i32p),
    ]


class PsiPlanStruct(C.Structure):
    _fields_ = [
        ("T", C.c_int32), ("n_grid", C.c_int32),
        ("psi_edge", f32p), ("psi_bin", f32p), ("psi_uniform", f32p), ("psi_grid", f32p), ("grid_w", f32p),
    ]


class DensePlanStruct(C.Structure):
    _fields_ = [
        ("T", C.c_int32),
        ("first_K", C.c_int32), ("first_GT", f32p),
        ("inf_K", C.c_int32), ("inf_GT", f32p),
        ("bin_box2", i32p), ("edge_box2", i32p), ("uniform_box2", i32p),
    ]


class Proj(C.Structure):
    _fields_ = [("wk", C.c_void_p), ("bk", C.c_void_p), ("wv", C.c_void_p), ("bv", C.c_void_p)]


class VqfConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_layers", "n_heads", "hidden", "inter", "enc_width", "tokens_per_frame", "n_query", "proj_out",
        "nb_samples")] + [("alpha", C.c_float), ("ln_eps", C.c_float)]


class Linear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p)]


class LayerNorm(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p)]


class VqfLayer(C.Structure):
    _fields_ = [("self_q", Linear), ("self_k", Linear), ("self_v", Linear), ("self_o", Linear), ("self_ln", LayerNorm),
                ("x_q", Linear), ("x_k", Linear), ("x_v", Linear), ("x_o", Linear), ("x_ln", LayerNorm),
                ("ffn_in", Linear), ("ffn_out", Linear), ("ffn_ln", LayerNorm)]


class VqfWeights(C.Structure):
    _fields_ = [("query_tokens", C.c_void_p), ("emb_ln", LayerNorm), ("layer", VqfLayer * MAX_LAYERS),
                ("llama_proj", Linear)]


class LTMError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"infv_ltm error {code}: {msg}")
        self.code = code


_SIGNATURES = {
    "infv_ltm_abi_version": (C.c_int, []),
    "infv_ltm_last_error": (C.c_char_p, []),
    "infv_ltm_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "infv_ltm_destroy": (C.c_int, [C.c_void_p]),
    "infv_ltm_set_plan": (C.c_int, [C.c_void_p, C.POINTER(PlanStruct)]),
    "infv_ltm_set_dense_plan": (C.c_int, [C.c_void_p, C.POINTER(DensePlanStruct)]),
    "infv_ltm_set_psi_plan": (C.c_int, [C.c_void_p, C.POINTER(PsiPlanStruct)]),
    "infv_ltm_has_plan": (C.c_int, [C.c_void_p, C.c_int32]),
    "infv_ltm_reset": (C.c_int, [C.c_void_p]),
    "infv_ltm_has_memory": (C.c_int, [C.c_void_p]),
    "infv_ltm_set_token_dtype": (C.c_int, [C.c_void_p, C.c_int32]),
    "infv_ltm_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "infv_ltm_pool_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "infv_ltm_new_rows": (C.c_int, [C.c_void_p, C.c_int32]),
    "infv_ltm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                C.POINTER(Proj), C.c_void_p, C.c_void_p, C.c_void_p]),
    "infv_ltm_steps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                 C.POINTER(Proj), C.c_void_p, C.c_void_p, C.c_void_p]),
    "infv_ltm_consolidate_q": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                         C.POINTER(Proj), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "infv_ltm_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                   C.POINTER(Proj), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "infv_ltm_forward_into": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.POINTER(Proj), C.c_void_p, C.c_void_p, C.c_void_p]),
    "infv_ltm_consolidate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                       C.POINTER(Proj), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "infv_ltm_consolidate_pooled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                              C.POINTER(Proj), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "infv_ltm_export_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "infv_ltm_import_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(Proj), C.c_void_p]),
    "infv_ltm_chain_state_bytes": (C.c_int64, [C.c_void_p, C.c_int32]),
    "infv_ltm_export_chain_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "infv_ltm_import_chain_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "infv_ltm_reproject": (C.c_int, [C.c_void_p, C.POINTER(Proj), C.c_void_p]),
    "infv_ltm_get_draw": (C.c_int, [C.c_void_p, C.c_int32, i32p, i32p, f32p, f32p, C.c_void_p]),
    "infv_ltm_set_probs": (C.c_int, [C.c_void_p, C.c_int32, f32p]),
    "infv_ltm_set_bins": (C.c_int, [C.c_void_p, C.c_int32, i32p]),
    "infv_ltm_set_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "infv_ltm_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "infv_ltm_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "infv_ltm_profile_read": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "infv_ltm_launch_count": (C.c_int64, []),
    # include/infv_vqf.h
    "infv_vqf_create": (C.c_int, [C.POINTER(VqfConfig), C.POINTER(C.c_void_p)]),
    "infv_vqf_destroy": (C.c_int, [C.c_void_p]),
    "infv_vqf_set_precision": (C.c_int, [C.c_void_p, C.c_int32]),
    "infv_vqf_set_weights_epoch": (C.c_int, [C.c_void_p, C.c_uint64]),
    "infv_vqf_short_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(Linear),
                                           C.POINTER(Linear), C.c_void_p, C.c_void_p, C.c_void_p]),
    "infv_vqf_encode_chunk": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_int32,
                                        C.POINTER(VqfWeights), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "infv_vqf_encode_video": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_int32, C.c_int32,
                                        C.POINTER(VqfWeights), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "infv_vqf_mean": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
}
KERNELS = ("pool", "rows", "project", "draw", "update", "attend", "scores", "chain", "uc")

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it is missing -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the LTM path has no CPU fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.infv_ltm_abi_version() != ABI_VERSION:
        raise ImportError(f"libinfv_ltm.so ABI {lib.infv_ltm_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int) -> int:
    if code < 0:
        raise LTMError(code, load().infv_ltm_last_error().decode(errors="replace"))
    return code
