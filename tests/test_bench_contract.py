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


def test_line_ends_with_the_contract_objects_and_a_compact_summary():
    """a log that keeps only the tail of stdout must still show what matters: the bulky sections come first, then the contract fields,
    `roofline`, `cpu_baseline`, and `summary` — last and short"""
    out = {"metric": "m", "value": 800.0, "unit": "re-evaluations/s", "ms_per_step": 1.25, "n_gpus": 1,
           "config": {"workload": "C2"}, "roofline": {"frac": 0.74, "avg_ms": 1.18}, "kernels": {"k": {"avg_ms": 1.0, "what": "x" * 4000}},
           "configs": {"C3": {"ms_per_step": 1.27, "workload": "y" * 3000}, "C4": {"ms_per_step": 0.44, "roofline": {"frac": 0.54}},
                       "C5": {"ms_per_step": 0.03, "roofline": {"frac": 0.55}}, "C4_sharded": {"ms_per_step": 0.45, "rccl_calls_made": True}},
           "host_api": {"handoff_device": {"ms_per_solve": 1.29}, "handoff_host_csc": {"ms_per_solve": 1.73}, "handoff_moi": {"ms_per_solve": 4.7},
                        "c3_host_csc": {"ms_per_solve": 1.97}},
           "roofline_constraint_pack": {"frac": 0.63, "in_step": {"hip_events": {"frac": 0.40}, "device_clock": {"frac": 0.44, "measured_in_this_run": True},
                                                                  "rocprofv3_replayed": {"frac": 0.55, "measured_in_this_run": False}}},
           "roofline_affine": {"frac": 0.8, "cold": {"frac": 0.6}}, "cpu_baseline": {"value": 0.004, "sample": "z" * 500}, "ranks_seen": 1}
    res = bench.ordered_for_the_tail(out)
    assert set(res) == set(out) | {"summary"}
    keys = list(res)
    assert keys[-1] == "summary" and keys[-3:-1] == ["cpu_baseline", "roofline"] and keys.index("kernels") < keys.index("metric") < keys.index("roofline")
    s = res["summary"]
    for k in ("C3_ms", "C4_ms", "C4_frac", "C5_ms", "C5_frac", "host_csc_ms", "moi_ms", "device_ms", "pack_in_step_frac", "affine_warm_frac", "affine_cold_frac",
              "ranks_seen", "rccl_calls_made", "gram_frac", "value", "ms_per_step"):
        assert k in s, k
    assert s["pack_in_step_frac"] == 0.44 and s["pack_in_step_frac_rocprof_replayed"] == 0.55 and s["C4_frac"] == 0.54 and s["host_csc_ms"] == 1.73 and s["rccl_calls_made"] is True
    text = json.dumps(res)
    tail = text[-2000:]
    assert '"summary"' in tail and '"roofline"' in tail and len(json.dumps(s)) < 1200
    assert bench.summary_of({})["value"] is None                 # sections may be missing (N > 1, --no-configs): no exception


def test_stdout_carries_the_json_line_only(tmp_path):
    """Libraries underneath (RCCL's version banner, C stdio) must not add lines to stdout: bench.py points descriptor 1 at stderr and writes
    its line to a duplicate of the original one.  Checked in a child process that prints through C stdio behind the line, as RCCL does."""
    import subprocess
    code = ("import bench, ctypes, sys\n"
            "bench.claim_stdout()\n"
            "libc = ctypes.CDLL(None)\n"
            "libc.printf(b'RCCL version : banner\\n')\n"                     # buffered C stdio, flushed at exit
            "print('a python print somewhere')\n"
            "bench.emit_line({'metric': 'm', 'value': 1})\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(__import__("pathlib").Path(bench.__file__).parent))
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "value": 1}
    assert "banner" in r.stderr and "a python print somewhere" in r.stderr


# ---- `--gpus N` launches N ranks by itself ----------------------------------------------------------------------

def _run_bench(*argv, env=None, timeout=180):
    import os
    import subprocess
    root = str(__import__("pathlib").Path(bench.__file__).parent)
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, "bench.py"] + list(argv), capture_output=True, text=True, cwd=root, env=e, timeout=timeout)


def _only_line(r):
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    return json.loads(lines[0])


def test_gpus_n_dry_launch_prints_the_exact_command():
    r = _run_bench("--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-launch")
    assert r.returncode == 0, r.stderr
    line = _only_line(r)
    cmd = line["launch"]
    assert line["n_gpus"] == 2 and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    k = [i for i, a in enumerate(cmd) if a.endswith("bench.py")][0]
    assert cmd[k + 1:] == ["--gpus", "2", "--steps", "20", "--warmup", "5"]          # the same argv, minus --dry-launch


def test_gpus_n_refuses_when_fewer_gpus_are_visible():
    """`python bench.py --gpus 2` on a box with fewer than two GPUs: rc != 0 and ONE JSON object naming the device count — never a line
    that says n_gpus: 1"""
    from parametron_jl_amd import _lib
    have = int(_lib.load().pmt_device_count())
    if have >= 2:
        pytest.skip("two GPUs visible")
    r = _run_bench("--gpus", "2", "--steps", "5", "--warmup", "1")
    assert r.returncode != 0
    line = _only_line(r)
    assert "error" in line and line["gpus_requested"] == 2 and line["gpus_visible"] == have
    assert "value" not in line and "n_gpus" not in line and "metric" not in line


def test_gpus_n_launches_n_ranks_by_itself():
    """the launcher path end to end without a GPU: `--gpus 2 --workload launch-check` starts two ranks (torch.distributed.run, gloo), they
    count themselves, rank 0 alone prints"""
    r = _run_bench("--gpus", "2", "--workload", "launch-check", env={"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = _only_line(r)
    assert line == {"launch_check": True, "n_gpus": 2, "ranks_seen": 2, "gpus_requested": 2}


def test_world_size_must_match_gpus():
    """a launcher that starts a different number of ranks than --gpus names: an error object, rc != 0"""
    r = _run_bench("--gpus", "8", "--workload", "launch-check", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    line = _only_line(r)
    assert "error" in line and line["gpus_requested"] == 8 and line["world_size"] == 1
