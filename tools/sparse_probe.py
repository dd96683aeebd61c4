"""Times the config-5 constraint node (sparse C, 5 % non-zeros, n = 16384, m = 4096): python tools/sparse_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parametron_jl_amd as P  # noqa: E402
from parametron_jl_amd import _lib  # noqa: E402


def dptr(t):
    return C.c_void_p(t.data_ptr())


def main():
    m, n, dens = 4096, 16384, 0.05
    rng = np.random.default_rng(0)
    csc = sp.random(m, n, density=dens, format="csc", random_state=rng, data_rvs=rng.random)
    nnz = csc.nnz
    colptr = csc.indptr.astype(np.int64) + 1
    rowval = csc.indices.astype(np.int64) + 1
    perm, trow, tcol = (np.empty(nnz, dtype=np.int64) for _ in range(3))
    rptr = np.empty(m + 1, dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.call("pmt_sparse_rowmajor_order", m, n, vp(colptr), vp(rowval), vp(perm), vp(trow), vp(tcol), vp(rptr))
    dev = torch.device("cuda:0")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nz = torch.from_numpy(csc.data).to(dev)
    dperm, drow, dvar = torch.from_numpy(perm).to(dev), torch.from_numpy(trow).to(dev), torch.from_numpy(tcol).to(dev)
    varmap = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
    out = torch.empty(nnz * 3, dtype=torch.int64, device=dev)
    outlt = torch.empty(nnz * 2, dtype=torch.int64, device=dev)

    def pack():
        _lib.call("pmt_sparse_pack_vector_f64", dptr(nz), dptr(dperm), dptr(drow), dptr(dvar), nnz, dptr(varmap), 0, dptr(out), stream)

    def assemble():
        _lib.call("pmt_sparse_assemble_f64", dptr(nz), dptr(dperm), dptr(dvar), nnz, dptr(outlt), stream)
    nslab = 8
    slab = np.zeros(m * (nslab + 1), dtype=np.int64)
    _lib.call("pmt_sparse_slab_ptr", m, n, nslab, vp(rptr), vp(tcol), vp(slab))
    dslab = torch.from_numpy(slab).to(dev)
    out2 = torch.empty(nnz * 3, dtype=torch.int64, device=dev)
    outlt2 = torch.empty(nnz * 2, dtype=torch.int64, device=dev)

    def pack_slabs():
        _lib.call("pmt_sparse_pack_vector_slabs_f64", dptr(nz), dptr(dperm), dptr(dvar), dptr(dslab), m, nslab, dptr(varmap), 0, dptr(out2), stream)

    def assemble_slabs():
        _lib.call("pmt_sparse_assemble_slabs_f64", dptr(nz), dptr(dperm), dptr(dvar), dptr(dslab), m, nslab, dptr(outlt2), stream)
    def pack_slabs_premapped():                      # varmap folded into term_var beforehand: no second gather
        _lib.call("pmt_sparse_pack_vector_slabs_f64", dptr(nz), dptr(dperm), dptr(dvar), dptr(dslab), m, nslab, None, 0, dptr(out2), stream)
    dperm32, dvar32 = torch.from_numpy(perm.astype(np.uint32).view(np.int32)).to(dev), torch.from_numpy(tcol.astype(np.uint32).view(np.int32)).to(dev)
    out3 = torch.empty(nnz * 3, dtype=torch.int64, device=dev)

    def pack_slabs_u32_premapped():
        _lib.call("pmt_sparse_pack_vector_slabs_u32_f64", dptr(nz), dptr(dperm32), dptr(dvar32), dptr(dslab), m, nslab, None, 0, dptr(out3), stream)
    cw = C.c_int(0)
    _lib.call("pmt_sparse_blocks_width", m, n, vp(colptr), vp(rowval), C.byref(cw))
    cw = cw.value
    print("block form: band width", cw, flush=True)
    nrb, ncb = -(-m // 128), -(-n // cw)
    desc, idx, band = np.zeros(nrb * n, dtype=np.uint64), np.zeros(nnz, dtype=np.uint32), np.zeros(m * (ncb + 1), dtype=np.int64)
    _lib.call("pmt_sparse_blocks_build", m, n, vp(colptr), vp(rowval), vp(perm), vp(tcol), vp(rptr), cw, vp(desc), vp(idx), vp(band))
    ddesc, didx, dband = torch.from_numpy(desc.view(np.int64)).to(dev), torch.from_numpy(idx.view(np.int32)).to(dev), torch.from_numpy(band).to(dev)
    colvar = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
    out4 = torch.zeros(nnz * 3, dtype=torch.int64, device=dev)
    outlt4 = torch.zeros(nnz * 2, dtype=torch.int64, device=dev)

    def pack_blocks():
        _lib.call("pmt_sparse_pack_vector_blocks_f64", dptr(nz), dptr(ddesc), dptr(didx), dptr(dband), dptr(colvar), m, n, nnz, cw, None, 0, None, 0, dptr(out4), None, stream)

    def assemble_blocks():
        _lib.call("pmt_sparse_assemble_blocks_f64", dptr(nz), dptr(ddesc), dptr(didx), dptr(dband), dptr(colvar), m, n, nnz, cw, None, 0, dptr(outlt4), None, stream)
    pack_blocks(); assemble_blocks()
    pack_slabs_u32_premapped()
    pack(); assemble(); pack_slabs(); assemble_slabs()
    torch.cuda.synchronize()
    print("slab kernels bit-identical to the flat kernels:", bool(torch.equal(out, out2)), bool(torch.equal(outlt, outlt2)), bool(torch.equal(out, out3)),
          " block kernels:", bool(torch.equal(out, out4)), bool(torch.equal(outlt, outlt4)), flush=True)
    only = os.environ.get("SPARSE_PROBE_ONLY")
    for name, fn, bytes_per in (("sparse_pack_vector_kernel", pack, 56), ("sparse_assemble_kernel", assemble, 40),
                                ("sparse_slab_kernel<VAT>", pack_slabs, 48), ("sparse_slab_kernel<LT>", assemble_slabs, 40),
                                ("sparse_slab_kernel<VAT>", pack_slabs_premapped, 48), ("sparse_slab_kernel<VAT,u32>", pack_slabs_u32_premapped, 40),
                                ("sparse_block_kernel<VAT>", pack_blocks, 40), ("sparse_block_kernel<LT>", assemble_blocks, 32)):
        if only and name not in only.split(";"):
            continue
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        P.profile_enable(True)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        rep = P.profile_report()
        P.profile_enable(False)
        ms = rep[name]["avg_ms"]
        print("%s: nnz %d  %.4f ms  %.0f GB/s algorithmic (%d B per non-zero) = %.1f %% of 8 TB/s" %
              (name, nnz, ms, nnz * bytes_per / ms / 1e6, bytes_per, nnz * bytes_per / ms / 1e6 / 80), flush=True)


if __name__ == "__main__":
    main()
