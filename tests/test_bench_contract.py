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
    m = bench.mfma_roofline("gram_mid_kernel", 1.18, 4096.0 * 4096 * 4097)
    assert m["bound"] == "mfma" and m["unit"] == "TFLOP/s" and 0.7 < m["frac"] < 0.8 and m["traffic"] is None
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


def _full_size_out():
    """an `out` of the size the default run builds: every section with its prose, per-kernel tables, error strings"""
    kern = {"kernel_%d<with, template, arguments, %d>" % (i, i): {"launches": 20, "avg_ms": 1.2007552499999998, "min_ms": 1.1, "max_ms": 1.3, "measured": "x" * 90}
            for i in range(25)}
    roof = bench.mfma_roofline("gram_mid_kernel", 1.2007552499999998, 4096.0 * 4096 * 4097, launches=20, algorithmic_bytes=335.6e6)
    bench.attach_traffic(roof, {"gram_mid_kernel": {"read": 870123456.789, "write": 184123456.789}, "source": "s"}, "pmt::gram_mid_kernel<")
    shapes = {"%dx%d" % (r, n): {"node_ms": 0.0617123456, "frac": 0.2212345678, "mfma_frac": 0.22, "hbm_frac": 0.1, "binding": "mfma", "kernels_ms": kern}
              for r, n in ((1 << 20, 16), (1 << 20, 64), (1 << 20, 128), (4096, 512), (262144, 512), (300, 300))}
    return {"metric": "QP re-evaluations/sec (Q,q,C,d rebuild) at n=4096", "value": 805.1812345678, "unit": "re-evaluations/s", "n_gpus": 1, "steps": 20,
            "warmup": 5, "ms_per_step": 1.2419512345678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": bench.C2Workload.name, "n": 4096, "r": 4096, "m": 512, "objective_mode": "canonical", "instances_per_gpu": 1,
                       "parallelism": "replicas (independent QP instances, no collective)", "replay": "tape", "step": "s" * 150},
            "config_detail": {"value_inputs_resident": 817.123456789, "boundary": "b" * 600}, "step_algorithmic_bytes": 402800000.0, "step_flops": 68736000000.0,
            "ranks_seen": 1, "kernels": kern, "roofline": roof, "value_200_steps": {"value": 798.5123456, "what": "w" * 200},
            "roofline_affine": {"frac": 0.838825149931128, "cold": {"frac": 0.6634979288064853, "what": "c" * 300}, "note": "n" * 500},
            "roofline_constraint_pack": {"frac": 0.5429215189654949, "warm": {"frac": 0.57}, "note": "n" * 500},
            "pmc_traffic": {"gram_mid_kernel": {"read": 8.7e8, "write": 1.84e8}, "source": "s" * 200},
            "configs": {"C1": {"update_us": 9.123456789, "solve_us_python_host_mock_optimizer": 26.87654321, "model_update_us_c_entry": 12.3456789, "workload": "y" * 300, "kernel_us": kern},
                        "C3": {"ms_per_step": 1.2712345678, "workload": "y" * 3000, "kernels": kern},
                        "C4": {"ms_per_step": 0.4412345678, "roofline": {"frac": 0.5412345678}, "kernels": kern},
                        "C5": {"ms_per_step": 0.0312345678, "roofline": {"frac": 0.5512345678}, "host_updated": {"what": "h" * 400}},
                        "shapes": shapes,
                        "C4_sharded": {"ms_per_step": 0.4512345678, "value": 18123456.789, "rccl_calls_made": True, "ranks_seen": 1, "config": {"workload": "z" * 400}}},
            "cpu_baseline": {"value": 0.00437123456789, "unit": "re-evaluations/s", "cores": 1, "kind": "port", "host_cores": 256, "sample": "z" * 900,
                             "seconds_per_reevaluation_extrapolated": 228.7123456, "fit_times_s": [0.1] * 4, "cross_check_rows": {"sample": "q" * 300}},
            "cpu_canonical_blas": {"value": 2.1320043286220307, "cores": 256, "kind": "k" * 80}}


def test_the_line_fits_4096_bytes_and_carries_the_contract_objects():
    """the driver could not parse a 20.8 KB line.  The stdout line of a FULL-SIZE result is at most 4096 bytes, parses, and
    carries the contract's fields, `roofline` (with frac, traffic and the measured_in_this_run flag), `cpu_baseline` and `summary`; the
    full objects go to the detail file."""
    out = _full_size_out()
    assert len(json.dumps(out)) > 20000                              # the bulk that broke round 5's record
    line = bench.compact_line(out, "/somewhere/bench_detail.json")
    text = json.dumps(line)
    assert len(text.encode()) <= 4096
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in back, k
    assert back["config"]["workload"].startswith("C2") and (back["config"]["n"], back["config"]["r"], back["config"]["m"]) == (4096, 4096, 512)
    r = back["roofline"]
    assert r["bound"] == "mfma" and r["frac"] == pytest.approx(0.7283, abs=1e-3) and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert r["measured_in_this_run"] is True and r["traffic"] == pytest.approx(870123456.789 + 184123456.789, rel=1e-5) and r["avg_ms"] > 0
    c = back["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["host_cores"] == 256 and c["value"] > 0 and len(c["sample"]) <= 400
    s = back["summary"]
    for k in ("C1_update_us", "C1_solve_us", "C3_ms", "C4_ms", "C4_frac", "C5_ms", "C5_frac", "C4_sharded_ms", "rccl_calls_made", "affine_warm_frac",
              "affine_cold_frac", "pack_cold_frac", "gram_shape_frac", "value_inputs_resident", "value_200_steps", "ranks_seen"):
        assert k in s, k
    assert s["gram_shape_frac"]["4096x512"] == 0.221 and back["detail"] == "bench_detail.json"
    assert "kernels" not in back and "configs" not in back


def test_the_line_survives_missing_and_failing_sections():
    """N > 1 / --no-configs / a section that raised: still one parseable line within the limit, the failures named in summary.errors"""
    out = _full_size_out()
    out["configs"]["C3"] = {"error": "RuntimeError: " + "e" * 5000}
    out["cpu_baseline"] = {"error": "OSError: " + "e" * 5000}
    out["roofline"]["traffic_source"] = "t" * 3000
    line = bench.compact_line(out)
    assert len(json.dumps(line).encode()) <= 4096
    assert "C3" in line["summary"]["errors"] and "cpu_baseline" in line["summary"]["errors"] and "error" in line["cpu_baseline"]
    bare = bench.compact_line({"metric": "m", "value": 1.0, "roofline": None, "cpu_baseline": None})
    assert bare["roofline"] is None and bare["summary"] == {} and json.loads(json.dumps(bare))["value"] == 1.0


def test_traffic_is_labelled_measured_or_replayed():
    roof = bench.mfma_roofline("gram_mid_kernel", 1.2, 6.87e10)
    bench.attach_traffic(roof, None, "pmt::no_such_kernel")                  # no rocprofv3, nothing on file
    assert roof["measured_in_this_run"] is False and roof["traffic"] is None and "rocprofv3 not on PATH" in roof["traffic_source"]
    roof = bench.mfma_roofline("gram_mid_kernel", 1.2, 6.87e10)
    bench.attach_traffic(roof, {"error": "rocprofv3 --pmc FETCH_SIZE child: rc 1"}, "pmt::gram_mid_kernel<true>")
    assert roof["measured_in_this_run"] is False and "rc 1" in roof["traffic_source"]
    if roof["traffic"] is not None:                                          # the committed replay, labelled as such
        assert roof["traffic"] == pytest.approx(roof["traffic_read"] + roof["traffic_write"]) and "replay" in roof["traffic_source"]


def test_counter_csv_reduction(tmp_path):
    """rocprofv3's counter_collection.csv -> mean per launch over the LAST `steps` launches of each of the step's kernels"""
    p = tmp_path / "x_counter_collection.csv"
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    for i, v in enumerate([999.0, 100.0, 102.0, 104.0]):
        rows.append('%d,"void pmt::gram_mid_kernel<true>(pmt::MidArgs)",FETCH_SIZE,%r' % (i, v))
    rows.append('9,"void pmt::affine_tile_kernel<1>(pmt::AffArgs)",FETCH_SIZE,50.0')
    rows.append('10,"void pmt::fill_uniform_kernel(double*)",FETCH_SIZE,7.0')
    p.write_text("\n".join(rows) + "\n")
    got = bench.reduce_counter_csv(str(p), 3)
    assert got == {"gram_mid_kernel": pytest.approx(102.0), "affine_tile_kernel<VAT>": pytest.approx(50.0)}


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
