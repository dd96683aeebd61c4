"""Phase timeline of batch_small_kernel's workgroup 0 (a -DPMT_BS_TRACE=1 library build):
PMT_LIB_PATH=parametron.jl_amd/lib_variants/bs_trace.so python tools/bs_trace.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd  # noqa: E402,F401
from parametron_jl_amd import _lib, batch  # noqa: E402


def main():
    wl = batch.BatchLSQ(torch, 8192, 128, 128, 16)
    for _ in range(10):
        wl.compute()
    torch.cuda.synchronize()
    L = C.CDLL(_lib.LIB_PATH)
    buf = np.zeros((2, 8192), dtype=np.int64)
    assert L.pmt_debug_bs_trace(buf.ctypes.data_as(C.c_void_p)) == 0
    nph = 128
    for who, name, ns, labels in ((0, "matrix wave 0", 4, ("k-steps", "staging (last phase of an instance)", "barrier wait", "barrier -> next phase start")),
                                  (1, "helper wave 0", 8, ("wait + LDS store of the next chunk", "constraint-block loads + q", "copy-out pieces", "staging (last phase)",
                                                           "chunk loads issued", "(stamp)", "barrier wait", "barrier -> next phase start"))):
        t = buf[who, :ns * nph].reshape(nph, ns)
        flat = t.reshape(-1)
        seg = np.diff(np.concatenate([flat, [flat[-1]]])).reshape(nph, ns)
        print(name, "(s_memtime ticks, phases 8..119)")
        body = seg[8:120]
        lastp = body[3::4]
        other = np.delete(body, np.s_[3::4], axis=0)
        for k in range(ns):
            print("   %-40s mean %8.1f   last phase of an instance %8.1f   other phases %8.1f" % (labels[k], body[:, k].mean(), lastp[:, k].mean(), other[:, k].mean()))
        print("   phase total mean %.1f ticks" % np.diff(t[8:120, 0]).mean())


if __name__ == "__main__":
    main()
