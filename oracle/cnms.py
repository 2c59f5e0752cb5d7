"""ORACLE (test infrastructure only): ctypes wrapper of oracle/nms_ref.c."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path():
    return os.path.join(_HERE, "_build", "liboracle_nms.so")


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(lib_path())
        _LIB.oracle_nms.restype = C.c_int
        _LIB.oracle_nms.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_long), C.c_int, C.c_double, C.c_int, C.POINTER(C.c_long)]
    return _LIB


def nms(dets, thresh, predicate=1):
    dets = np.ascontiguousarray(dets, np.float32)
    n = dets.shape[0]
    if n == 0:
        return []
    s = dets[:, 4]
    order = np.ascontiguousarray(np.lexsort((np.arange(n), -s.astype(np.float64))), dtype=np.int64)
    keep = np.zeros((n,), np.int64)
    nk = _lib().oracle_nms(dets.ctypes.data_as(C.POINTER(C.c_float)), order.ctypes.data_as(C.POINTER(C.c_long)), n,
                           float(thresh), int(predicate), keep.ctypes.data_as(C.POINTER(C.c_long)))
    return keep[:nk].tolist()
