#!/usr/bin/env python
"""Register / LDS / spill table of the device code, and the build-time gate on it.

The Makefile compiles every .hip with -Rpass-analysis=kernel-resource-usage and keeps the remarks in
parametron.jl_amd/build/<file>.resources.txt.  This script prints one line per kernel and, with --check, FAILS (exit 1) when a hot-path
kernel spills a vector register or uses scratch, or exceeds the register budget its co-residency design depends on:
  gram_sk_kernel*       <= 216 VGPRs (the budget its amdgpu_num_vgpr attribute states)
  sparse_block_kernel*  <= 128 VGPRs (two 512-thread workgroups per CU), sparse_slab_kernel* <= 64 (eight 256-thread workgroups per CU)
  courier / to_host / the Gram node's small reductions   <= 16 VGPRs (co-resident with any contraction build up to 248 VGPRs)
  batch_small_kernel*   no spills
  every kernel          no VGPR spills, no scratch
usage: python tools/kernel_resources.py [--check] [files...]
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]")
BUDGET = {"gram_sk_kernel": 216, "sparse_block_kernel": 128, "sparse_slab_kernel": 64}          # VGPR ceilings that are part of the design
SIDE_KERNELS = {"gram_linear_kernel": 16, "gram_linear_split_kernel": 16, "seq_dot_kernel": 16, "courier_kernel": 16, "to_host_kernel": 16}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def parse(path):
    kernels, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: [^:]*:\d+:\d+: +(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+) \[-Rpass", line)
        if m:
            cur = {"name": m.group(2), "file": os.path.basename(path).replace(".resources.txt", "")}
            kernels.append(cur)
            continue
        for f in FIELDS:
            m = re.search(re.escape(f) + r": (\d+)", line)
            if m and cur is not None:
                cur[f] = int(m.group(1))
    return kernels


def main():
    check = "--check" in sys.argv
    files = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(glob.glob(os.path.join(ROOT, "parametron.jl_amd", "build", "*.resources.txt")))
    kernels = [k for f in files for k in parse(f)]
    dm = demangle([k["name"] for k in kernels])
    bad = []
    print("%-14s %-72s %5s %5s %6s %6s %7s %4s" % ("file", "kernel", "VGPR", "AGPR", "spillV", "scratch", "LDS", "occ"))
    for k in kernels:
        short = re.sub(r"^void ", "", dm[k["name"]]).replace("pmt::", "")
        short = re.sub(r"\(.*$", "", short)
        v, a, sv, sc = k.get("VGPRs", -1), k.get("AGPRs", 0), k.get("VGPRs Spill", 0), k.get("ScratchSize [bytes/lane]", 0)
        print("%-14s %-72s %5d %5d %6d %6d %7d %4d" % (k["file"], short[:72], v, a, sv, sc, k.get("LDS Size [bytes/block]", 0), k.get("Occupancy [waves/SIMD]", 0)))
        base = short.split("<")[0]
        if "rocprim" in k["name"]:
            continue                      # library kernels of the setup-time ordering / pruning helpers (canon_device.hip, prune.hip): not ours, not per solve
        if sv or sc:
            bad.append("%s: %d VGPRs spilled, %d bytes/lane of scratch" % (short, sv, sc))
        if base in BUDGET and v + a > BUDGET[base]:
            bad.append("%s: %d VGPRs > budget %d" % (short, v + a, BUDGET[base]))
        if base in SIDE_KERNELS and v > SIDE_KERNELS[base]:
            bad.append("%s: %d VGPRs > %d (no longer co-resident with the contraction)" % (short, v, SIDE_KERNELS[base]))
    if check and bad:
        print("\nkernel resource check FAILED:\n  " + "\n  ".join(bad), file=sys.stderr)
        return 1
    if check:
        print("kernel resource check ok: %d kernels, no vector spills, no scratch, budgets held" % len(kernels))
    return 0


if __name__ == "__main__":
    sys.exit(main())
