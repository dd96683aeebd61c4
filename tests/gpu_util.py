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


def constant_in_the_library_order(r, n, b, sign=-1):
    """c'c restated on the CPU in the fixed order the library reports for an r x n node (pmt_quad_gram_constant_order): the reference's
    left-to-right sum, the 2048 chains, or the fused tall forms' per-workgroup / slice order (2: eight row-pair lanes, 3: sixteen) (include/parametron_hip.h)"""
    order, groups, stage = C.c_int(), C.c_int(), C.c_int()
    call("pmt_quad_gram_constant_order", r, n, C.byref(order), C.byref(groups), C.byref(stage))
    nb = (0.0 - b) if sign < 0 else (0.0 + b)
    if order.value == 0:
        seq = 0.0
        for v in nb:
            seq = seq + v * v
        return order.value, seq
    if order.value == 1:
        chains = np.zeros(2048)
        for i0 in range(0, r, 2048):
            seg = nb[i0:i0 + 2048]
            chains[:len(seg)] = chains[:len(seg)] + seg * seg
        seq = 0.0
        for v in chains:
            seq = seq + v
        return order.value, seq
    G, MR = groups.value, stage.value
    if order.value == 5:
        # mid-size wide shapes (gram_sk.hip: sk_lin_role): thread t of 512 adds rows t, t + 512, .. in order; per wave the __shfl_down tree
        # 32, 16, .., 1; the eight waves in order
        T = MR
        sq = np.zeros(-(-max(r, 1) // T) * T)
        sq[:r] = nb * nb
        lanes = np.zeros(T)
        for k in range(len(sq) // T):
            lanes = lanes + sq[k * T:(k + 1) * T]
        w = lanes.reshape(T // 64, 64)
        h = 32
        while h >= 1:
            w = np.concatenate([w[:, :h] + w[:, h:2 * h], w[:, h:]], axis=1)     # lane i += lane i + h for i < h (the others are not read again)
            h //= 2
        total = w[0, 0]
        for k in range(1, T // 64):
            total = total + w[k, 0]
        return order.value, float(total)
    if order.value == 4:
        # the stream form (gram_stream_kernel): iterations of MR rows dealt out to the 4 G WAVES (wave 4 g + w: iterations 4 g + w, + 4 G, ..);
        # contraction slot lk of a wave adds rows 8 i + 2 lk, + 1 of its iterations in order; slots: (0 + 2) + (1 + 3); waves of a workgroup in
        # order; workgroups in 16 interleaved slices, then the slices in order
        Wv = 4 * G
        nit = -(-r // MR)
        sq = np.zeros((nit + Wv) * MR)
        sq[:r] = nb * nb
        slots = np.zeros((Wv, 4))
        wi = np.arange(Wv)
        for k in range(-(-nit // Wv)):
            base = (wi + k * Wv) * MR
            live = (wi + k * Wv) < nit
            for i in range(MR // 8):
                idx = base[:, None] + 8 * i + 2 * np.arange(4)[None, :]
                slots = np.where(live[:, None], (slots + sq[idx]) + sq[idx + 1], slots)
        wave = (slots[:, 0] + slots[:, 2]) + (slots[:, 1] + slots[:, 3])
        wave = wave.reshape(G, 4)
        part = ((wave[:, 0] + wave[:, 1]) + wave[:, 2]) + wave[:, 3]
        slices = np.zeros(16)
        for g0 in range(0, G, 16):
            seg = part[g0:g0 + 16]
            slices[:len(seg)] = slices[:len(seg)] + seg
        total = slices[0]
        for t in range(1, 16):
            total = total + slices[t]
        return order.value, float(total)
    L = 16 if order.value == 3 else 8                                  # row-pair lanes walking down a column piece of 2 L rows
    nst = -(-r // MR)
    sq = np.zeros((nst + G) * MR)
    sq[:r] = nb * nb                                                   # (rows beyond the matrix add 0.0: exact)
    lanes = np.zeros((G, L))
    gi = np.arange(G)
    for k in range(-(-nst // G)):                                      # workgroup g: stages g, g + G, g + 2G, ..
        base = (gi + k * G) * MR
        live = (gi + k * G) < nst
        for j in range(MR // (2 * L)):
            idx = base[:, None] + 2 * L * j + 2 * np.arange(L)[None, :]
            lanes = np.where(live[:, None], (lanes + sq[idx]) + sq[idx + 1], lanes)
    h = L // 2
    while h >= 1:                                                      # __shfl_down tree: lane i += lane i + h
        lanes = lanes[:, :h] + lanes[:, h:2 * h]
        h //= 2
    part = lanes[:, 0]
    slices = np.zeros(16)
    for g0 in range(0, G, 16):
        seg = part[g0:g0 + 16]
        slices[:len(seg)] = slices[:len(seg)] + seg
    total = slices[0]
    for t in range(1, 16):
        total = total + slices[t]
    return order.value, float(total)
