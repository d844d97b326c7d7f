"""ctypes binding of oracle/libals_ref.so (the C restatement in als_ref.c).
TEST INFRASTRUCTURE ONLY - never imported by the product package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/als_ref.c -> oracle/libals_ref.so with the committed Makefile."""
    subprocess.run(["make", "-C", _HERE, "libals_ref.so"], check=True, capture_output=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libals_ref.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        for suf, ct in (("f64", ctypes.c_double), ("f32", ctypes.c_float)):
            fn = getattr(L, "als_ref_half_sweep_" + suf)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                           ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_double,
                           ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
            g = getattr(L, "als_ref_gram_" + suf)
            g.restype = None
            g.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        L.als_ref_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def max_threads():
    return int(lib().als_ref_max_threads())


def half_sweep(indptr, indices, data, Yfix, Xold, lamb, alpha, dtype=np.float64, row_begin=0, row_end=None,
               nthreads=1, Y2=None):
    """Half-sweep over rows [row_begin, row_end) in `dtype` arithmetic; returns a new X."""
    suf = {np.float64: "f64", np.float32: "f32"}[np.dtype(dtype).type]
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    data = np.ascontiguousarray(data, dtype=dtype)
    Yfix = np.ascontiguousarray(Yfix, dtype=dtype)
    X = np.array(Xold, dtype=dtype, order="C", copy=True)
    n_rows = len(indptr) - 1
    row_end = n_rows if row_end is None else row_end
    assert X.shape[0] == n_rows and X.shape[1] == Yfix.shape[1]
    assert indices.size == 0 or (indices.min() >= 0 and indices.max() < Yfix.shape[0])
    y2p = None
    if Y2 is not None:
        Y2 = np.ascontiguousarray(Y2, dtype=dtype)
        y2p = Y2.ctypes.data
    bad = getattr(lib(), "als_ref_half_sweep_" + suf)(
        indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, row_begin, row_end, Yfix.ctypes.data,
        Yfix.shape[0], Yfix.shape[1], y2p, float(lamb), float(alpha), X.ctypes.data, int(nthreads))
    if bad:
        raise np.linalg.LinAlgError("%d singular row systems" % bad)
    return X


def gram(Y, dtype=np.float64):
    suf = {np.float64: "f64", np.float32: "f32"}[np.dtype(dtype).type]
    Y = np.ascontiguousarray(Y, dtype=dtype)
    out = np.empty((Y.shape[1], Y.shape[1]), dtype=dtype)
    getattr(lib(), "als_ref_gram_" + suf)(Y.ctypes.data, Y.shape[0], Y.shape[1], out.ctypes.data)
    return out
