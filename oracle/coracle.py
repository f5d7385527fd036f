"""ctypes binding of oracle/rbg_oracle.c (the C restatement; test infrastructure only).

Used by tests (checker for sizes the pure-Python loop cannot reach) and by bench.py's
``cpu_baseline`` leg ("kind": "port").  See rbg_oracle.c for the reference citations.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librbg_oracle.so")
_lib = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "rbg_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.ora_num_threads.restype = ctypes.c_int
        _lib.ora_set_num_threads.argtypes = [ctypes.c_int]
        _lib.ora_build_norm_csr.restype = ctypes.c_int64
        _lib.ora_build_norm_csr.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _i64p, _i64p, _u8p,
                                            _i64p, _i64p, _f32p]
        _lib.ora_spmm_csr_f32.restype = None
        _lib.ora_spmm_csr_f32.argtypes = [ctypes.c_int64, ctypes.c_int64, _i64p, _i64p, _f32p, _f32p, _f32p]
        _lib.ora_lightgcn_forward_f32.restype = None
        _lib.ora_lightgcn_forward_f32.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                                  _i64p, _i64p, _f32p, _f32p, _f32p, _f32p, _f32p]
        _lib.ora_numa_prepare.restype = ctypes.c_void_p
        _lib.ora_numa_prepare.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _i64p, _i64p, _f32p]
        _lib.ora_numa_forward_f32.restype = None
        _lib.ora_numa_forward_f32.argtypes = [ctypes.c_void_p, _f32p, _f32p, _f32p]
        _lib.ora_numa_free.restype = None
        _lib.ora_numa_free.argtypes = [ctypes.c_void_p]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def num_threads():
    return int(lib().ora_num_threads())


def set_num_threads(n):
    lib().ora_set_num_threads(int(n))


def build_norm_csr(uid, iid, n_users, n_items, keep=None):
    uid = np.ascontiguousarray(uid, dtype=np.int64)
    iid = np.ascontiguousarray(iid, dtype=np.int64)
    n = int(n_users) + int(n_items)
    kept = int(uid.shape[0] if keep is None else np.count_nonzero(keep))
    rowptr = np.zeros(n + 1, dtype=np.int64)
    col = np.zeros(2 * kept, dtype=np.int64)
    val = np.zeros(2 * kept, dtype=np.float32)
    k = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    nnz = lib().ora_build_norm_csr(n_users, n_items, uid.shape[0], _p(uid, _i64p), _p(iid, _i64p),
                                   _p(k, _u8p), _p(rowptr, _i64p), _p(col, _i64p), _p(val, _f32p))
    assert nnz == 2 * kept, (nnz, kept)
    return rowptr, col, val


def spmm(rowptr, col, val, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.float32)
    m = rowptr.shape[0] - 1
    out = np.empty((m, x.shape[1]), dtype=np.float32)
    lib().ora_spmm_csr_f32(m, x.shape[1], _p(rowptr, _i64p), _p(col, _i64p), _p(val, _f32p), _p(x, _f32p),
                           _p(out, _f32p))
    return out


def lightgcn_forward(rowptr, col, val, user_w, item_w, n_layers, return_layers=False, buffers=None):
    """buffers: optional (layers [K+1, N, d], out [N, d]) float32 arrays to reuse (a timing loop should not pay for
    72 MB of fresh pages per call)."""
    user_w = np.ascontiguousarray(user_w, dtype=np.float32)
    item_w = np.ascontiguousarray(item_w, dtype=np.float32)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.float32)
    nu, ni, d = user_w.shape[0], item_w.shape[0], user_w.shape[1]
    n = nu + ni
    if buffers is not None:
        layers, out = buffers
        assert layers.shape == (n_layers + 1, n, d) and out.shape == (n, d) and layers.dtype == out.dtype == np.float32
    else:
        layers = np.empty((n_layers + 1, n, d), dtype=np.float32)
        out = np.empty((n, d), dtype=np.float32)
    lib().ora_lightgcn_forward_f32(nu, ni, d, n_layers, _p(rowptr, _i64p), _p(col, _i64p), _p(val, _f32p),
                                   _p(user_w, _f32p), _p(item_w, _f32p), _p(layers, _f32p), _p(out, _f32p))
    if return_layers:
        return out, layers
    return out



class NumaForward:
    """The same propagation with thread-owned row blocks and first-touch placement of everything a thread streams (bench.py's
    cpu_baseline on many-socket hosts; ora_numa_* in rbg_oracle.c).  Built for the CURRENT thread count; bit-identical results."""

    def __init__(self, rowptr, col, val, n_users, n_items, d, n_layers):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.float32)
        self.shape = (n_users + n_items, d)
        self._h = lib().ora_numa_prepare(n_users, n_items, d, n_layers, _p(rowptr, _i64p), _p(col, _i64p), _p(val, _f32p))
        if not self._h:
            raise MemoryError("ora_numa_prepare failed")

    def __call__(self, user_w, item_w, want_result=True):
        user_w = np.ascontiguousarray(user_w, dtype=np.float32)
        item_w = np.ascontiguousarray(item_w, dtype=np.float32)
        out = np.empty(self.shape, dtype=np.float32) if want_result else None
        lib().ora_numa_forward_f32(self._h, _p(user_w, _f32p), _p(item_w, _f32p), _p(out, _f32p))
        return out

    def close(self):
        if self._h:
            lib().ora_numa_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
