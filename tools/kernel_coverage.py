"""Which kernels of libparametron_hip.so does the GPU test suite launch?  Runs pytest -m gpu in-process with the library's profiler on (every
launch is recorded under its kernel name) and compares the names with the kernels the build produced (parametron.jl_amd/build/kernel_resources.txt).
    python tools/kernel_coverage.py [pytest args]        (GPU box)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402
import torch  # noqa: E402,F401
import parametron_jl_amd as P  # noqa: E402

launched = set()


class Collect:
    def pytest_runtest_setup(self, item):
        P.profile_enable(True)

    def pytest_runtest_teardown(self, item):
        try:
            launched.update(P.profile_report().keys())
        except Exception:
            pass


rc = pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests")] + sys.argv[1:], plugins=[Collect()])
built = set()
for line in open(os.path.join(ROOT, "parametron.jl_amd", "build", "kernel_resources.txt")):
    m = re.match(r"^(\w+)\s+(\S.*?)\s{2,}\d+\s+\d+", line)
    if m and m.group(1) != "file":
        built.add(re.sub(r"<.*", "", m.group(2)).replace("pmt::", "").replace("dma::", "").strip())
names = {re.sub(r"<.*", "", k) for k in launched}
missing = sorted(k for k in built if k not in names and not k.startswith(("rocprim", "void rocprim", "hip", "__")))
print("pytest rc %s; %d kernel names launched by the suite; kernels of the build never launched: %s" % (rc, len(names), missing or "none"))
