"""ORACLE -- ctypes front of oracle/knn_canon.c (canonical-arithmetic kNN).  Test infrastructure."""
import ctypes
import os
import subprocess
import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libknn_canon.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_knn_f32.restype = ctypes.c_int
        _lib.oracle_knn_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def knn_point_major(xp, k, return_pd=False):
    """xp: float32 array/tensor [B,N,C] (point-major) -> int32 ndarray [B,N,k]."""
    a = np.ascontiguousarray(xp.detach().cpu().numpy() if torch.is_tensor(xp) else xp, dtype=np.float32)
    B, N, C = a.shape
    idx = np.empty((B, N, k), dtype=np.int32)
    pd = np.empty((B, N, k), dtype=np.float32) if return_pd else None
    rc = _load().oracle_knn_f32(a.ctypes.data, B, N, C, k, idx.ctypes.data, pd.ctypes.data if return_pd else None)
    if rc != 0:
        raise ValueError("oracle_knn_f32: bad arguments (B=%d N=%d C=%d k=%d)" % (B, N, C, k))
    return (idx, pd) if return_pd else idx


def knn(x, k):
    """Drop-in for the reference signature: x [B,C,N] torch -> idx [B,N,k] int64 torch (CPU)."""
    idx = knn_point_major(x.detach().cpu().transpose(2, 1).contiguous(), k)
    return torch.from_numpy(idx.astype(np.int64)).to(x.device)
