"""bench.py's stdout contract (no GPU): ONE compact JSON line the driver can parse, the full record beside it, and a
`--gpus N` that starts N ranks itself when no launcher did (round 4's 21 KB line came back as `parsed: null`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "launches_per_step",
            "alg_bytes_per_launch", "e2e_frac", "most_time_lost"}


def test_line_is_small_and_complete(tmp_path):
    full = bench.canned_result()
    assert len(json.dumps(full)) > 20000                       # the canned record is as fat as round 4's
    line = bench.compact_line(full, str(tmp_path / "bench_detail.json"))
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT < 8192
    assert CONTRACT <= set(line)
    assert ROOFLINE <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert len(line["dtype"]) <= 80
    assert set(line["config"]) <= {"workload", "global_batch", "utterance_samples", "parallelism"}
    for key in ("cfg3_float32", "cfg2_float32", "cfg1_float32", "cfg5_float32", "cfg5_bfloat16", "off_table_shape"):
        assert "ms_per_step" in line["secondary"][key]
    assert line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
    assert line["value"] == pytest.approx(full["value"], rel=1e-4)


def test_secondary_errors_survive_but_stay_short():
    full = bench.canned_result()
    full["secondary"]["off_table_shape"] = {"error": "ValueError('" + "x" * 2000 + "')"}
    line = bench.compact_line(full)
    assert len(line["secondary"]["off_table_shape"]["error"]) <= 160


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=300)


def test_dry_run_single_process_prints_one_line(tmp_path):
    detail = tmp_path / "d.json"
    r = _run(["--dry-run", "--steps", "3", "--warmup", "1", "--detail", str(detail)])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["detail"] == "d.json"
    assert "per_kernel" in json.load(open(detail))["roofline"]  # the fat part went to the file


def test_gpus_n_without_a_launcher_starts_n_ranks(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset: two ranks (gloo in --dry-run), barrier + MAX over ranks, rank 0
    prints the line with n_gpus = 2."""
    r = _run(["--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1", "--detail", str(tmp_path / "d.json")])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4
    assert line["ms_per_step"] >= 1.0                           # 1 ms sleeps, max over ranks


def test_gpus_n_refuses_fewer_devices():
    """Without --dry-run a 2-GPU figure must not be produced from fewer than 2 visible devices."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_pmc_traffic_lookup_flags_a_stale_summary(tmp_path):
    """`roofline.traffic` quotes a committed counter summary keyed by kernel symbol: when a kernel that
    takes >= 1 % of the step is not in it - added, renamed or re-instantiated since the passes - the line says
    `traffic_stale: true` instead of quoting old counters silently."""
    import json
    import bench
    (tmp_path / "x_pmc_traffic.json").write_text(json.dumps({"conv_hx<2,2,1,4,0,4,1,x1>": 3.0e9, "spk_proj": 6.0e6}))
    shares = {"conv_hx<2,2,1,4,0,4,1,x1>": 0.6, "spk_proj": 0.002}
    t, src, stale = bench.pmc_traffic("conv_hx<2,2,1,4,0,4,1,x1>", shares, ("x_pmc_traffic.json",), root=str(tmp_path))
    assert t == 3.0e9 and stale is False and "STALE" not in src
    shares["conv_wx<3,6,4,2,1,x1>"] = 0.03                   # a new kernel takes 3 % of the step
    t, src, stale = bench.pmc_traffic("conv_hx<2,2,1,4,0,4,1,x1>", shares, ("x_pmc_traffic.json",), root=str(tmp_path))
    assert t == 3.0e9 and stale is True and "conv_wx<3,6,4,2,1,x1>" in src
    shares["tiny_kernel"] = 0.001                            # below 1 %: ignored
    del shares["conv_wx<3,6,4,2,1,x1>"]
    assert bench.pmc_traffic("spk_proj", shares, ("missing.json", "x_pmc_traffic.json"), root=str(tmp_path))[2] is False
    assert bench.pmc_traffic("spk_proj", shares, ("missing.json",), root=str(tmp_path)) == (None, None, False)
    assert "traffic_stale" in bench.ROOFLINE_KEYS
