"""config 4 on one GPU, the computation alone (for per-launch PMC passes): python tools/c4_compute_only.py [launches]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from parametron_jl_amd import batch  # noqa: E402
wl = batch.BatchLSQ(torch, batch.TOTAL, batch.N, batch.R, batch.M)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    wl.compute()
torch.cuda.synchronize()
