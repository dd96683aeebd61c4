"""Helpers for the -m gpu parity tests: device buffers are torch CUDA tensors (plumbing only); every
compute call goes through the C ABI of libparametron_hip.so via parametron_jl_amd._lib."""
import ctypes as C

import numpy as np
import torch

import parametron_jl_amd  # noqa: F401
from parametron_jl_amd import _lib

LT, QT, VAT = _lib.LT, _lib.QT, _lib.VAT
DEV = "cuda:0"


def lib():
    _lib.require_gpu()
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_dev(a, dtype=None):
    a = np.ascontiguousarray(a if dtype is None else np.asarray(a, dtype=dtype))
    if a.dtype.fields is not None:
        t = torch.from_numpy(a.view(np.int64).copy()).to(DEV)
    else:
        t = torch.from_numpy(a.copy()).to(DEV)
    return t


def colmajor(A):
    """numpy (rows, cols) -> device buffer in Julia column-major order."""
    A = np.asarray(A, dtype=np.float64)
    return to_dev(np.ascontiguousarray(A.T).reshape(-1))


def empty_terms(n, dtype):
    words = dtype.itemsize // 8
    return torch.full((max(int(n) * words, 1),), -7, dtype=torch.int64, device=DEV)   # poisoned


def empty_f64(n):
    return torch.full((max(int(n), 1),), float("nan"), dtype=torch.float64, device=DEV)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def terms_to_host(t, n, dtype):
    words = dtype.itemsize // 8
    torch.cuda.synchronize()
    return t[: int(n) * words].cpu().numpy().view(dtype).copy()


def f64_to_host(t, n):
    torch.cuda.synchronize()
    return t[: int(n)].cpu().numpy().copy()


def call(name, *args):
    _lib.call(name, *args)


def same_bits(a, b):
    """bit-exact equality of two float64 arrays (distinguishes -0.0 / NaN payloads)."""
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.int64)
    return a.shape == b.shape and bool(np.all(a == b))


def assert_terms_equal(got, want):
    assert got.dtype == want.dtype and got.shape == want.shape, (got.shape, want.shape)
    for f in got.dtype.names:
        if got.dtype[f].kind == "f":
            assert same_bits(got[f], want[f]), "field %s differs" % f
        else:
            assert np.array_equal(got[f], want[f]), "field %s differs" % f
