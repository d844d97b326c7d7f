"""ctypes binding of libals_hip.so (the C ABI in include/als_hip.h).

There is no fallback: if the shared library is missing or a symbol cannot be bound,
importing this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C recommender_amd/csrc`.
"""
import ctypes
import os

import torch  # noqa: F401  must precede the CDLL below: the library has to bind to the HIP runtime
              # (libamdhip64.so.7) that torch loads, or the process ends up with two runtimes

_HERE = os.path.dirname(os.path.abspath(__file__))
# ALS_HIP_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("ALS_HIP_LIB") or os.path.join(_HERE, "libals_hip.so")

ALS_OK = 0
ERROR_NAMES = {1: "invalid argument", 2: "unsupported", 3: "workspace too small", 4: "HIP error"}
DEFAULT_CHUNK_NNZ = 2048
GRAM_CHUNK_ROWS = 256
GRAM_SUPER_ROWS = 65536
MAX_FACTORS = 256
MAX_FACTORS_EXACT = 128
GRAM_WIDE_CHUNK_ROWS = 4096
SHORT_MAX_GROUPS = 8
ABI_VERSION = 8          # als_abi_version() of the library this binding is written for (struct layouts included)
ACCUMULATE_MODES = {"default": 0, "bf16x3": 1, "fp32": 2, "fp16x2": 3}
ROW_KERNELS = {"auto": 0, "wave": 1, "stream": 2}
CG_FORMS = {"auto": 0, "lock_step": 1, "matrix": 2}

c_i32, c_i64, c_f32, c_sz, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/als_hip.h declares
SIGNATURES = {
    "als_last_error_string": (ctypes.c_char_p, []),
    "als_abi_version": (c_i32, []),
    "als_plan_count": (c_i32, [c_vp, c_i64, c_i32, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "als_plan_fill": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "als_gram_tiles_bytes": (c_sz, [c_i32]),
    "als_gram_partial_bytes": (c_sz, [c_i32]),
    "als_gram_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "als_gram_f32": (c_i32, [c_vp, c_i64, c_i32, c_i64, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "als_gram_partials_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "als_gram_finalize_f32": (c_i32, [c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "als_scores_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "als_topk_rows_f32": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "als_recommend_topk_workspace_bytes": (c_sz, [c_i64, c_i64, c_i32]),
    "als_items_prepared_bytes": (c_sz, [c_i64, c_i32]),
    "als_items_prepare_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_sz, c_vp]),
    "als_scores_prepared_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "als_recommend_topk_prepared_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp,
                                                c_vp, c_vp, c_vp, c_sz, c_vp]),
    "als_recommend_topk_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp,
                                       c_vp, c_vp, c_vp, c_sz, c_vp]),
    "als_predict_pairs_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "als_normalize_rows_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "als_half_sweep_workspace_bytes": (c_sz, [c_i64, c_i64, c_i32]),
    "als_half_sweep_cg_workspace_bytes": (c_sz, [c_i64, c_i64, c_i32]),
    "als_half_sweep_cg_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32,
                                      c_vp, c_i64, c_i64, c_vp, c_vp, c_f32, c_i32, c_f32, c_i32, c_i32, c_vp, c_i64, c_vp,
                                      c_vp, c_vp, c_vp, c_sz, c_vp]),
    "als_short_rows_max_groups": (c_i32, [c_i32]),
    "als_plan_short_bounds": (c_i32, [c_vp, c_i64, c_i32, c_vp]),
    "als_whiten_workspace_bytes": (c_sz, [c_i32]),
    "als_whiten_f32": (c_i32, [c_vp, c_f32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "als_static_scratch_bytes": (c_sz, []),
    "als_static_scratch_init": (c_i32, []),
    "als_half_sweep_csr_static_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32,
                                              c_vp, c_i64, c_i64, c_vp, c_i32, c_f32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "als_half_sweep_csr_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32,
                                       c_vp, c_i64, c_i64, c_vp, c_i32, c_f32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz,
                                       c_vp]),
}


class SweepOpts(ctypes.Structure):
    """als_sweep_opts_t of include/als_hip.h."""
    _fields_ = [("accumulate", c_i32), ("row_kernel", c_i32), ("tap_begin", c_vp), ("tap_end", c_vp),
                ("max_rating", c_f32), ("whitened", c_vp), ("ld_whitened", c_i64), ("short_bounds", c_vp),
                ("reserve_cus", c_i32)]


class AlsHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "recommender_amd: %s is missing - the HIP extension has not been built "
            "(run __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.als_abi_version() != ABI_VERSION:
        raise ImportError("recommender_amd: %s has ABI version %d, this binding needs %d - rebuild the extension "
                          "(__graft_entry__.build())" % (LIB_PATH, lib.als_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def check(rc, what):
    if rc != ALS_OK:
        msg = lib.als_last_error_string().decode("utf-8", "replace")
        exc = ValueError if rc in (1, 2, 3) else AlsHipError
        raise exc("%s failed (%s): %s" % (what, ERROR_NAMES.get(rc, rc), msg))
