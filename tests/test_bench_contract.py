"""CPU: the parts of bench.py that do not need a GPU — the driver's command-line contract (defaults, flags), the roofline helpers and
the bounded CPU baseline (oracle restatement timed on the host: c*n^3 fit of the literal quadratic node + full-size affine nodes)."""
import json
import sys

import pytest

import bench


def test_command_line_contract_defaults(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert a.gpus == 1 and a.steps >= 20 and a.warmup >= 1 and a.workload == "c2"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)


def test_roofline_helpers_follow_the_contract():
    r = bench.hbm_roofline("k", 0.05, 402653184.0)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] == pytest.approx(402653184.0 / 0.05e-3 / 1e9) and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    m = bench.mfma_roofline("gram_sk_kernel", 1.18, 4096.0 * 4096 * 4097)
    assert m["bound"] == "mfma" and m["unit"] == "TFLOP/s" and 0.7 < m["frac"] < 0.8
    json.dumps(r), json.dumps(m)


def test_cpu_baseline_is_a_bounded_single_core_sample_of_the_oracle():
    class W:                        # a small shape: the function itself decides the sample sizes of the n^3 fit
        n, r, m = 256, 256, 32
    out = bench.cpu_baseline(W)
    assert out["kind"] == "port" and out["cores"] == 1 and out["unit"] == "re-evaluations/s"
    assert out["value"] > 0 and out["seconds_per_reevaluation_extrapolated"] == pytest.approx(1.0 / out["value"])
    assert "cross_check_rows" in out and "EXTRAPOLATED" in out["sample"]
