"""-m gpu: the Julia host (north_star: "host code stays Julia").  The build image and the GPU boxes of this project have no julia binary,
so this test PROBES for one and, when it is there (with Parametron.jl and MathOptInterface installed), runs README Example 1 through the
reference's own update! and through julia/ParametronHIPBackend.jl on the same values (julia/example1_parity.jl: every MOI array bit for
bit).  Without julia it is skipped with an explicit message; the Julia sources are then only checked statically
(tests/test_cabi_exports.py)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_readme_example1_parity_under_julia():
    julia = shutil.which("julia")
    if julia is None:
        pytest.skip("no `julia` binary on this box (probed with shutil.which): the Julia host cannot be executed here; its ccalls are "
                    "checked statically against include/parametron_hip.h by tests/test_cabi_exports.py")
    probe = subprocess.run([julia, "-e", "using Parametron, MathOptInterface"], capture_output=True, text=True, timeout=600)
    if probe.returncode != 0:
        pytest.skip("julia is present but Parametron.jl / MathOptInterface are not installed (no network here): " + probe.stderr[-300:])
    env = dict(os.environ, PARAMETRON_HIP_LIB=os.path.join(ROOT, "parametron.jl_amd", "lib", "libparametron_hip.so"))
    run = subprocess.run([julia, os.path.join(ROOT, "julia", "example1_parity.jl")], capture_output=True, text=True, timeout=1200, env=env)
    assert run.returncode == 0 and "JULIA_PARITY_OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
