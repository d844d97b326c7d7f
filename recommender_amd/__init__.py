"""MI355X-native implicit-ALS engine (drop-in for the hot path of reclibwh's RecommenderALS).

The compute path is the HIP library libals_hip.so (recommender_amd/csrc, C ABI in
include/als_hip.h).  Importing the package loads it and raises if it is missing:
there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (loads libals_hip.so, raises ImportError if absent)
from .ops import (DeviceCSR, GramCache, build_plan, gram, gram_finalize, gram_partials, half_sweep,  # noqa: F401
                  half_sweep_cg, normalize_rows, predict_pairs, prepare_items, recommend_topk, scores, short_rows_max_entries,
                  topk_rows, whiten, PreparedItems)

__all__ = ["DeviceCSR", "GramCache", "build_plan", "gram", "gram_partials", "gram_finalize", "half_sweep", "half_sweep_cg",
           "scores", "topk_rows", "predict_pairs", "normalize_rows", "recommend_topk", "whiten", "short_rows_max_entries",
           "prepare_items", "PreparedItems", "RecommenderALS"]

from .recommender import RecommenderALS  # noqa: E402,F401
