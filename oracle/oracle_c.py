"""ctypes binding of oracle/liboracle.so (the C restatement in oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: see the header of oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64 = ctypes.c_int64
_p = ctypes.c_void_p


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.miq_oracle_quantize_rows_sym_i8.argtypes = [_p, _i64, _i64, _p, _p, _p]
        L.miq_oracle_linear_w8a8.argtypes = [_p, _p, _p, _i64, _i64, _i64]
        L.miq_oracle_linear_i8_nt.argtypes = [_p, _p, _p, _i64, _i64, _i64]
        L.miq_oracle_conv2d_i8.argtypes = [_p, _p, _p] + [_i64] * 9
        L.miq_oracle_epilogue_f32.argtypes = [_p, _p, _p, ctypes.c_float, _p, _i64, _i64, _i64]
        for f in ("miq_oracle_quantize_rows_sym_i8", "miq_oracle_linear_w8a8", "miq_oracle_linear_i8_nt",
                  "miq_oracle_conv2d_i8", "miq_oracle_epilogue_f32"):
            getattr(L, f).restype = None
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(_p)


def quantize_rows_sym_i8(w):
    w = np.ascontiguousarray(w, dtype=np.float32)
    rows = w.shape[0]
    cols = w.size // rows
    q = np.empty(w.shape, dtype=np.int8)
    scale = np.empty(rows, dtype=np.float32)
    zp = np.empty(rows, dtype=np.int64)
    lib().miq_oracle_quantize_rows_sym_i8(_ptr(w), rows, cols, _ptr(q), _ptr(scale), _ptr(zp))
    return q, scale, zp


def linear_w8a8(a, b_kn):
    a = np.ascontiguousarray(a, dtype=np.int8)
    b = np.ascontiguousarray(b_kn, dtype=np.int8)
    m, k = a.shape
    n = b.shape[1]
    assert b.shape[0] == k
    c = np.empty((m, n), dtype=np.int32)
    lib().miq_oracle_linear_w8a8(_ptr(a), _ptr(b), _ptr(c), m, n, k)
    return c


def linear_i8_nt(a, w_nk):
    a = np.ascontiguousarray(a, dtype=np.int8)
    w = np.ascontiguousarray(w_nk, dtype=np.int8)
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k
    c = np.empty((m, n), dtype=np.int32)
    lib().miq_oracle_linear_i8_nt(_ptr(a), _ptr(w), _ptr(c), m, n, k)
    return c


def conv2d_i8(x, w, stride, padding):
    x = np.ascontiguousarray(x, dtype=np.int8)
    w = np.ascontiguousarray(w, dtype=np.int8)
    B, C, H, W = x.shape
    Co, Ci, KH, KW = w.shape
    assert C == Ci
    Ho = (H + 2 * padding - KH) // stride + 1
    Wo = (W + 2 * padding - KW) // stride + 1
    out = np.empty((B, Co, Ho, Wo), dtype=np.int32)
    lib().miq_oracle_conv2d_i8(_ptr(x), _ptr(w), _ptr(out), B, C, H, W, Co, KH, KW, stride, padding)
    return out


def epilogue_f32(acc, bias, alpha, beta):
    """acc int32 [M,N] (Linear) or [B,N,Ho,Wo] (Conv2d) -> fp32, channel axis 1."""
    acc = np.ascontiguousarray(acc, dtype=np.int32)
    outer, channels = acc.shape[0], acc.shape[1]
    inner = acc.size // (outer * channels)
    alpha = np.ascontiguousarray(alpha, dtype=np.float32).reshape(-1)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32).reshape(-1)
    out = np.empty(acc.shape, dtype=np.float32)
    lib().miq_oracle_epilogue_f32(_ptr(acc), None if b is None else _ptr(b), _ptr(alpha),
                                  ctypes.c_float(beta), _ptr(out), outer, channels, inner)
    return out
