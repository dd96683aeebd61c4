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


def test_sharded_c4_report_contract():
    """configs.C4_sharded of the default bench line (every N): the fields VERDICT r2 item 2 names, from three measured times"""
    from parametron_jl_amd import batch
    for world in (1, 2, 8):
        r = batch.sharded_report(world, world, steps=20, warmup=5, t_pipe=0.020, t_mono=0.030, t_compute=0.010, nchunks=4)
        json.dumps(r)
        assert r["n_gpus"] == world and r["ranks_seen"] == world and r["scaling"] == "strong" and r["unit"] == "re-evaluations/s"
        assert r["value"] == pytest.approx(8192 * 20 / 0.020) and r["ms_per_step"] == pytest.approx(1.0)
        assert r["compute_only"]["ms_per_step"] == pytest.approx(0.5) and r["compute_then_allgather"]["ms_per_step"] == pytest.approx(1.5)
        assert r["compute_overlapped_exchange"]["value"] == r["value"]
        off, L = batch.slab_layout(128, 16)
        assert r["exchange_bytes_per_rank"] == 8.0 * L * (8192 // world)
        if world == 1:
            assert r["expected_exchange_ms"]["direct_per_link"] == 0.0
        else:
            assert r["expected_exchange_ms"]["direct_per_link"] == pytest.approx(8.0 * L * (8192 // world) / 153e9 * 1e3)
            assert r["expected_exchange_ms"]["ring"] == pytest.approx((world - 1) * r["expected_exchange_ms"]["direct_per_link"])
        assert r["roofline"]["bound"] == "hbm" and r["roofline"]["frac"] == pytest.approx(r["roofline"]["achieved"] / 8000.0)
    # 8 GPUs: 85.6 MB per rank, 0.56 ms per-link bound
    assert r["exchange_bytes_per_rank"] == pytest.approx(85.6e6, rel=0.01) and r["expected_exchange_ms"]["direct_per_link"] == pytest.approx(0.56, abs=0.01)


def test_default_line_documents_the_sharded_config():
    assert "C4_sharded" in bench.__doc__ and "ranks_seen" in bench.__doc__
