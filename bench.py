#!/usr/bin/env python
"""bench.py - audio samples/sec of the FastSVC generator forward on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the generator forward over one batch of synthetic utterances already
resident in HBM.  Workloads (BASELINE.json `configs`):

  cfg3 (default at --gpus 1)   64 x 10 s per GPU in bfloat16 storage - BASELINE's largest single-GPU configuration in
                  its stated dtype ("bf16 generator forward"): what `value` is quoted on at N = 1.  `secondary` carries
                  cfg3 in float32, cfg2 (8 x 4 s, float32: the 1e-3 parity configuration), cfg1, cfg4's 1-GPU leg in both
                  storages, the ragged set and an off-table shape.
  cfg4 (default at --gpus N > 1)   the 512-utterance set (10 s each), utterance-parallel STRONG scaling in the same
                  storage: the set is sharded over the ranks (svcc23_fastsvc_amd.distributed.run_utterance_parallel:
                  LPT shards, batches of 64, round-wise asynchronous all-gather over RCCL); a step is one pass over the
                  whole set and `value` = 512 * T / step time (at N = 1 this is `secondary.cfg4_bfloat16_n1`, 8 cfg3
                  batches per pass).  `secondary.weak_cfg2` keeps the weak-scaling figure (every rank a cfg2 batch).
  cfg1 / cfg2 / cfg3 with --gpus N   every rank runs the same-sized batch of different utterances (weak scaling), rank 0
                  packs the weights and broadcasts the blob over RCCL, each step's waveforms are all-gathered
                  asynchronously (overlapping the next step).
  cfg4var         cfg4's variable-length variant (2 - 10 s, ragged length-bucketed batches)
  cfg5            one full training step (see run_cfg5)

Prints ONE JSON line.  `roofline` is measured live with hipEvents on the launch stream
(fastsvc_forward_profile): `roofline.e2e` prices EVERY launch of the step at
max(alg. FLOPs / MFMA peak of its arithmetic, alg. bytes / 8 TB/s) and divides the sum by the measured
step time (time-weighted whole-forward fraction); `roofline.kernel` is the DOMINANT symbol - most time per step, the top
line of `rocprofv3 --stats` for the same command (profiles/) -, `roofline.most_time_lost` the one that loses the most time
against its own roofline (the one to fix first).  `secondary` (N = 1) adds cfg3 in
float32 and bfloat16.  `cpu_baseline` times the CPU oracle's restatement of the reference forward on
the host cores of this box, on bounded samples - a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import svcc23_fastsvc_amd as A  # noqa: E402
from svcc23_fastsvc_amd import synth as S  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense f32-input MFMA = f32 vector peak
PEAK_HALF_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 / f16 MFMA
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak
WEIGHT_SEED = 201


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough for the clocks to settle (a cfg2 step is 1.4 ms: 20 steps after 5 warm-up ones read 3-4 %
    # slower than 200 after 50); the large workloads scale them down below
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg4var", "cfg5"],
                    help="default: cfg3 at --gpus 1, cfg4 (the 512-utterance set, strong scaling) at --gpus N > 1.  "
                         "cfg5: one full training step (generator fwd/bwd, MelGAN MSD, MR-STFT + adversarial losses, RAdam; "
                         "recipe batch 32 x 16000 samples per GPU, data-parallel gradient all-reduce)")
    ap.add_argument("--storage", default=None, choices=["float32", "bfloat16"],
                    help="workspace tensor storage: bfloat16 = BASELINE config 3's dtype (default for the forward "
                         "workloads: what `value` is quoted on); float32 = the 1e-3 parity path (default for cfg5)")
    ap.add_argument("--discriminator", default="melgan", choices=["melgan", "hifigan"],
                    help="cfg5: melgan = the recipe yaml's MelGAN multi-scale discriminator (4.35 M parameters); hifigan = the HiFiGAN "
                         "multi-period + multi-scale discriminator BASELINE config 5 names (fastsvc.py:631-1143 defaults, 70.7 M)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg3 float32 / bfloat16 block")
    ap.add_argument("--autotune", action="store_true",
                    help="run the untimed device-side autotune pass (cf. cudnn.benchmark) instead of "
                         "the shipped launch-shape table / static cost model")
    ap.add_argument("--no-table", action="store_true",
                    help="ignore the shipped launch-shape table (svcc23_fastsvc_amd/tuned_mi355x.json)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--detail", default=os.environ.get("FASTSVC_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="where the full record goes (per-kernel table, secondary workloads' rooflines, notes); stdout carries "
                         "only the compact line")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: exercise the launcher, the gloo rendezvous, the barrier + max-over-ranks timing and the "
                         "line builder with a sleep as the step (tests/test_bench_line.py)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# The ONE stdout line stays small (the driver's parser gave up on round 4's 21 KB line): headline fields, the dominant
# kernel's roofline, the CPU baseline and one {ms_per_step, e2e_frac} pair per secondary workload.  Everything else -
# the per-kernel table, the secondary workloads' own rooflines, notes - goes to --detail (bench_detail.json).
# ---------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 6144
SHORT_DTYPE = {"float32": "f32 (products as split-f16 MFMA x3, f32 accumulate; f32 storage)",
               "bfloat16": "bf16 (bf16 MFMA products, f32 accumulate; bf16 storage)"}
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "secondary", "detail")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "kernel", "avg_launch_us", "launches_per_step",
                 "alg_bytes_per_launch", "alg_flops_per_launch", "e2e_frac", "most_time_lost")


def _sig(x, n=5):
    """Floats to n significant digits, recursively (the line carries measurements, not 17-digit doubles)."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def short_dtype(storage: str, arithmetic: str = "") -> str:
    if arithmetic in ("f32", "f32 arithmetic, bf16 activation storage"):      # FASTSVC_HX=0: the f32-input MFMA kernels
        return arithmetic
    return SHORT_DTYPE[storage]


def compact_roofline(roof):
    if not roof or "error" in roof:
        return roof
    out = {k: roof.get(k) for k in ROOFLINE_KEYS if k not in ("e2e_frac", "most_time_lost")}
    out["e2e_frac"] = (roof.get("e2e") or {}).get("frac")
    lost = roof.get("most_time_lost") or {}
    out["most_time_lost"] = {"kernel": lost.get("kernel"), "frac": lost.get("frac"), "ms_per_step": lost.get("ms_per_step")}
    return out


def compact_secondary(sec):
    if not sec:
        return None
    out = {}
    for k, v in sec.items():
        if not isinstance(v, dict):
            continue
        if "error" in v:
            out[k] = {"error": str(v["error"])[:160]}
        elif "cost_model" in v:                                   # the off-table shape: cost model vs autotuned
            out[k] = {"ms_per_step": v["cost_model"]["ms_per_step"], "autotuned_ms_per_step": v["autotuned"]["ms_per_step"]}
        else:
            e2e = ((v.get("roofline") or {}).get("e2e") or {}).get("frac")
            out[k] = {"ms_per_step": v.get("ms_per_step"), "e2e_frac": e2e}
            if "predicted_speedup_vs_1gpu" in v:                      # (a prediction from single-GPU legs, labelled as one)
                out[k] = {"ms_per_step": v.get("ms_per_step"), "predicted_speedup_vs_1gpu": v["predicted_speedup_vs_1gpu"], "predicted": True}
            if v.get("n_gpus", 1) != 1:
                out[k]["n_gpus"] = v["n_gpus"]
    return out


def compact_cpu(cpu):
    if not cpu:
        return None
    out = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind")}
    out["sample"] = str(cpu.get("sample", ""))[:200]
    if "samples_per_s_by_threads" in cpu:
        out["by_threads"] = cpu["samples_per_s_by_threads"]
    return out


def compact_line(full: dict, detail_path=None) -> dict:
    """The line the driver parses, from the full record: the contract's keys plus `roofline` and `cpu_baseline`."""
    line = {k: full.get(k) for k in LINE_KEYS if k not in ("roofline", "cpu_baseline", "secondary", "detail")}
    line["dtype"] = str(full.get("dtype", ""))[:80]
    cfgd = dict(full.get("config") or {})
    cfgd["workload"] = str(cfgd.get("workload", ""))[:200]
    line["config"] = {k: cfgd[k] for k in ("workload", "global_batch", "utterance_samples", "parallelism") if k in cfgd}
    line["roofline"] = compact_roofline(full.get("roofline"))
    line["cpu_baseline"] = compact_cpu(full.get("cpu_baseline"))
    line["secondary"] = compact_secondary(full.get("secondary"))
    line["detail"] = os.path.basename(detail_path) if detail_path else None
    return _sig(line)


def emit(json_fd: int, full: dict, detail_path):
    """Write the full record to `detail_path` and the compact line to the saved stdout descriptor."""
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(full, f, indent=1)
        except OSError as e:                                     # a read-only tree must not take the line down
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
            detail_path = None
    text = json.dumps(compact_line(full, detail_path), separators=(",", ":"))
    if len(text) > LINE_LIMIT:                                   # never again: drop the optional blocks before the headline
        slim = compact_line(full, detail_path)
        slim["secondary"] = {k: {"ms_per_step": v.get("ms_per_step")} for k, v in (slim.get("secondary") or {}).items()}
        text = json.dumps(slim, separators=(",", ":"))
    sys.stdout.flush()
    os.write(json_fd, (text + "\n").encode())


def self_launch(args, argv):
    """`bench.py --gpus N` (N > 1) started as ONE process with no WORLD_SIZE: start the N ranks here, one per visible GPU,
    the way the contract's torch.distributed.run command would, and pass their single line through."""
    import socket
    import subprocess
    if not args.dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible - refusing to report a "
                             f"{args.gpus}-GPU figure from fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_run(args, json_fd):
    """--dry-run: every rank sleeps 1 ms per step; gloo barrier + MAX over ranks as in the real run; rank 0 prints the
    compact line built from a canned roofline (no GPU, no library)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    steps, warmup = args.steps or 5, args.warmup if args.warmup is not None else 1
    for _ in range(warmup):
        time.sleep(0.001)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        time.sleep(0.001)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        full = canned_result(world, steps, warmup, elapsed)
        emit(json_fd, full, args.detail)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def canned_result(world=1, steps=5, warmup=1, elapsed=0.005):
    """A full record of the shape main() builds (worst-case sizes: long notes, 30 kernels, every secondary block) - for
    --dry-run and for the line-size test."""
    note = "n" * 500
    per_kernel = {f"bf16::conv_hx_kernel<{i},2,1,4,0,4,1,true,false>": {"ms_per_step": 1.234567891234, "launches": 2,
                  "TFLOPs": 123.456789123, "GBs": 4321.123456789, "roofline_ms": 0.123456789123, "frac": 0.123456789123}
                  for i in range(30)}
    roof = {"bound": "hbm", "achieved": 4380.123456789, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": 0.5475154320987,
            "traffic": 3767000000.0, "traffic_source": note, "kernel": "bf16::conv_hx_kernel<2,2,1,4,0,4,1,true,false>",
            "kernel_choice": "most time per step", "avg_launch_us": 842.2123456789, "launches_per_step": 2,
            "alg_bytes_per_launch": 3686400000.0, "alg_flops_per_launch": 1.0e11,
            "most_time_lost": {"kernel": "bf16::cond_stage1_kernel<x1>", "ms_per_step": 1.54, "roofline_ms": 0.19, "frac": 0.125,
                               "lost_ms_per_step": 1.35},
            "e2e": {"frac": 0.3212345678, "note": note, "per_kernel_roofline_ms": 3.73}, "per_kernel": per_kernel}
    sec = {k: {"workload": note, "ms_per_step": 1.3801234567, "dtype": note, "roofline": {"e2e": {"frac": 0.31234567, "note": note},
                                                                                         "top_kernels_by_time": per_kernel}}
           for k in ("cfg3_float32", "cfg2_float32", "cfg2_bfloat16", "cfg1_float32", "cfg1_bfloat16", "cfg4_bfloat16_n1",
                     "cfg4_float32_n1", "cfg4var_float32_n1", "cfg5_float32", "cfg5_bfloat16", "cfg5_hifigan_float32")}
    sec["off_table_shape"] = {"workload": note, "cost_model": {"ms_per_step": 1.1}, "autotuned": {"ms_per_step": 1.0}}
    return {"metric": "audio samples/sec (24 kHz) FastSVC generator fwd", "value": 64 * 240000 * steps / elapsed * world,
            "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": SHORT_DTYPE["bfloat16"], "data": "synthetic",
            "config": {"workload": "dry run (no GPU): " + note, "global_batch": 64 * world, "utterance_samples": 240000,
                       "parallelism": f"utterance-parallel x{world}"},
            "roofline": roof, "secondary": sec,
            "cpu_baseline": {"value": 8.4e5, "unit": "samples/s", "cores": 16, "kind": "port", "sample": note,
                             "samples_per_s_by_threads": {"1": 3.0e5, "8": 7.0e5, "16": 8.4e5, "32": 8.0e5, "64": 7.0e5},
                             "cfg2": {"x": note}}}


def resolve_workload(gpus: int, workload, storage):
    """What a bare `bench.py --gpus N` measures.  N = 1: BASELINE's largest single-GPU configuration (cfg3, 64 x 10 s);
    N > 1: its multi-GPU configuration (cfg4: the 512-utterance set sharded over the ranks, strong scaling, all-gather of
    the waveforms included) - both in cfg3's stated dtype (bfloat16 storage) unless --storage says otherwise, so that
    the N = 1 figure of a scaling run is `secondary.cfg4_bfloat16_n1` of the N = 1 line (8 cfg3 batches per pass)."""
    if workload is None:
        workload = "cfg3" if gpus == 1 else "cfg4"
    if storage is None:
        storage = "float32" if workload == "cfg5" else "bfloat16"
    return workload, storage


def _usable_cores() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(budget_s: float):
    """The reference's CPU PyTorch path as restated by the oracle (checker / reported baseline only).

    `value`: `forward_as_executed` (the reference's op sequence incl. its redundant conditioning
    chains, fp32) on one cfg1-shaped utterance (B=1, F=300 -> 48 000 samples) per iteration.  PyTorch's
    intra-op pool does not scale to a whole 2-socket box on convolutions this small, so a few thread
    counts are tried within the time budget and the BEST is reported with the thread count it used.
    `cfg2`: the same op sequence on the cfg2 batch (8 x 4 s) at that thread count, and `dedup`: the
    de-duplicated dataflow the HIP path implements (`forward_dedup`), so that the algorithmic saving
    (fewer FLOPs) is separable from the hardware speed-up (SURVEY.md §8 d)."""
    from oracle import fastsvc_oracle as O
    cfg = S.FULL_CONFIG
    w = S.fold_weight_norm(S.synth_state_dict(cfg, WEIGHT_SEED))
    b = S.synth_batch(cfg, 1, 300, 4242)
    usable = _usable_cores()
    cands = sorted({1, min(8, usable), min(16, usable), min(32, usable), min(64, usable)})
    per = max(1.5, 0.5 * budget_s / len(cands))
    best = None
    tried = {}

    def med_time(fn, batch, budget, max_runs=40):
        fn(w, cfg.upsampling_scales, batch.ppg, batch.sine, batch.lft, batch.spk_emb)       # warm-up
        times = []
        t_end = time.time() + budget
        while len(times) < 2 or (time.time() < t_end and len(times) < max_runs):
            t = time.time()
            fn(w, cfg.upsampling_scales, batch.ppg, batch.sine, batch.lft, batch.spk_emb)
            times.append(time.time() - t)
            if times[-1] > budget:          # hopelessly oversubscribed / long: one sample is enough
                break
        return float(np.median(times)), len(times)

    for nt in cands:
        torch.set_num_threads(nt)
        med, n = med_time(O.forward_as_executed, b, per)
        tried[str(nt)] = 48000.0 / med
        if best is None or 48000.0 / med > best[0]:
            best = (48000.0 / med, nt, n)
    torch.set_num_threads(best[1])
    rest = max(2.0, 0.5 * budget_s)
    dd1, _ = med_time(O.forward_dedup, b, rest / 6)
    b2 = S.synth_batch(cfg, 8, 600, 4243)
    ex2, n2 = med_time(O.forward_as_executed, b2, rest / 3, max_runs=5)
    dd2, _ = med_time(O.forward_dedup, b2, rest / 3, max_runs=5)
    return {"value": best[0], "unit": "samples/s", "cores": int(best[1]), "kind": "port",
            "host_cores_usable": usable, "samples_per_s_by_threads": tried,
            "sample": f"oracle forward_as_executed (reference op sequence, fp32 CPU PyTorch), "
                      f"1 x 2 s utterance (48000 samples) per run, median of {best[2]} runs at the "
                      f"best of {cands} threads",
            "cfg1_dedup_samples_per_s": 48000.0 / dd1,
            "cfg2": {"as_executed_samples_per_s": 768000.0 / ex2, "dedup_samples_per_s": 768000.0 / dd2,
                     "threads": int(best[1]),
                     "sample": f"8 x 4 s batch (768000 samples) per run, median of {n2} runs"}}


PMC_TAG = "r6"       # profiles/<tag>_<workload>_<storage>_pmc_traffic.json: the counter passes `roofline.traffic` quotes


def kernel_peak_tflops(kernel: str) -> float:
    """Dense MFMA peak of the arithmetic a kernel symbol runs, in ALGORITHMIC (direct-conv, 2*MAC)
    FLOP/s: f32-input MFMA for conv_mfma*; the split-half kernels (conv_hx<... P=3>) spend three
    f16 MFMA products per algorithmic MAC, the bf16 ones (P=1) one."""
    if kernel.startswith(("conv_hx", "conv_wx")):
        nprod = 3 if ",x3" in kernel else 1
        return PEAK_HALF_MFMA_TFLOPS / nprod
    if kernel.startswith("cond_stage"):                      # whole-stage conditioning launch (csrc/fastsvc_cond.hip)
        return PEAK_HALF_MFMA_TFLOPS / (3 if "x3" in kernel else 1)
    return PEAK_FP32_MFMA_TFLOPS


def pmc_traffic(kern, shares, names, root=None):
    """(HBM bytes per launch of kernel `kern`, where the figure comes from, stale?) from the first committed PMC summary of
    `names` under profiles/ that exists.  `shares`: kernel symbol -> share of the step's time, of the kernels THIS run
    launched.  The summary is stale when a kernel that takes >= 1 % of the step is not in it (a kernel was added, renamed or
    re-instantiated since the counter passes were taken): the line then says so instead of quoting old counters silently."""
    for name in names:
        pmc = os.path.join(root or os.path.join(ROOT, "profiles"), name) if name else None
        if not (pmc and os.path.exists(pmc)):
            continue
        try:
            table = json.load(open(pmc))
        except Exception:
            continue
        missing = sorted(k for k, share in shares.items() if share >= 0.01 and k not in table)
        traffic = table.get(kern)
        source = (f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same workload, "
                  "tools/collect_cfg3.sh + tools/summarize_profiles.py; FETCH_SIZE doubled per the guide's gfx950 "
                  "correction; regenerated whenever a kernel changes) - not measured by this run")
        if missing:
            source += f"; STALE: not in the summary: {', '.join(missing[:6])}"
        return traffic, source, bool(missing)
    return None, None, False


def roofline(plan, blob, args_dev, ms_per_step, n_prof=3, lengths=None, pmc_file=None):
    """Per-kernel-symbol aggregation of hipEvent-timed launches.  fastsvc_forward_profile runs the
    forward on ONE stream (no helper streams) so that each kernel is timed running alone;
    `profiles/*_kernel_stats_serial.csv` is rocprofv3's view of the same thing (FASTSVC_SERIAL=1),
    `*_kernel_stats.csv` the default multi-stream schedule."""
    agg = {}
    total_ms = 0.0
    roof_ms_total = 0.0
    roof32_ms_total = 0.0      # the same sum with every conv priced on the f32-input MFMA (round-1 pricing)
    flops_total = 0.0
    bytes_total = 0.0
    for _ in range(n_prof):
        recs = []
        plan.forward(blob, *args_dev, profile=recs)
        for r in recs:
            a = agg.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, roof_ms=0.0))
            t_roof = max(r["flops"] / (kernel_peak_tflops(r["kernel"]) * 1e12), r["bytes"] / (PEAK_HBM_GBS * 1e9)) * 1e3
            a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["launches"] += 1
            a["roof_ms"] += t_roof
            total_ms += r["ms"]; roof_ms_total += t_roof
            roof32_ms_total += max(r["flops"] / (PEAK_FP32_MFMA_TFLOPS * 1e12), r["bytes"] / (PEAK_HBM_GBS * 1e9)) * 1e3
            flops_total += r["flops"]; bytes_total += r["bytes"]
    # the dominant kernel: most time per step (the top line of rocprofv3 --stats for the same command); the kernel that LOSES
    # the most time against its own roofline - the one to fix first - is reported next to it (`most_time_lost`)
    kern, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
    lost_k, lost_a = max(agg.items(), key=lambda kv: kv[1]["ms"] - kv[1]["roof_ms"])
    sec = a["ms"] * 1e-3
    tf = a["flops"] / sec / 1e12
    gbs = a["bytes"] / sec / 1e9
    peak_tf = kernel_peak_tflops(kern)
    t_mfma = a["flops"] / (peak_tf * 1e12)
    t_hbm = a["bytes"] / (PEAK_HBM_GBS * 1e9)
    # HBM bytes per launch from the PMC counters: they need separate rocprofv3 --pmc passes (tools/pmc.sh), so the
    # figure comes from the committed summary of the same command, not from this run - `traffic_source` says which, and
    # `traffic_stale` says whether that summary still describes the kernels this run launched
    traffic, traffic_source, traffic_stale = pmc_traffic(kern, {k: v["ms"] / total_ms for k, v in agg.items()}, (pmc_file, "pmc_traffic.json"))
    if t_mfma >= t_hbm:
        out = {"bound": "mfma", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf / peak_tf}
    else:
        out = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}
    roof_step = roof_ms_total / n_prof
    out.update({
        "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
        "kernel": kern, "kernel_choice": "most time per step",
        "most_time_lost": {"kernel": lost_k, "ms_per_step": lost_a["ms"] / n_prof, "roofline_ms": lost_a["roof_ms"] / n_prof,
                           "frac": lost_a["roof_ms"] / lost_a["ms"], "lost_ms_per_step": (lost_a["ms"] - lost_a["roof_ms"]) / n_prof},
        "avg_launch_us": a["ms"] * 1e3 / a["launches"], "launches_per_step": a["launches"] // n_prof,
        "share_of_step": a["ms"] / total_ms, "lost_ms_per_step": (a["ms"] - a["roof_ms"]) / n_prof,
        "alg_flops_per_launch": a["flops"] / a["launches"], "alg_bytes_per_launch": a["bytes"] / a["launches"],
        "hbm_side_GBs": gbs, "mfma_side_TFLOPs": tf,
        "e2e": {
            "alg_gflop_per_step": flops_total / n_prof / 1e9, "alg_gbyte_per_step": bytes_total / n_prof / 1e9,
            "alg_tflops": flops_total / n_prof / (ms_per_step * 1e-3) / 1e12,
            "frac_of_fp32_mfma_peak": flops_total / n_prof / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "alg_hbm_GBs": bytes_total / n_prof / (ms_per_step * 1e-3) / 1e9,
            "per_kernel_roofline_ms": roof_step,
            "frac": roof_step / ms_per_step,
            "per_kernel_roofline_ms_f32_mfma_pricing": roof32_ms_total / n_prof,
            "frac_f32_mfma_pricing": roof32_ms_total / n_prof / ms_per_step,
            "serial_sum_ms": total_ms / n_prof,
            "note": "per_kernel_roofline_ms = sum over launches of max(alg FLOPs / MFMA peak of the kernel's "
                    "arithmetic, alg bytes / 8 TB/s); frac = that / measured ms_per_step (multi-stream step). "
                    "The split-binary16 / bf16 kernels are priced at 2500/3 resp. 2500 TFLOP/s, which makes nearly "
                    "every layer HBM-bound and the roofline 40 % shorter than with the f32-input MFMA pricing "
                    "(157.3 TFLOP/s) round 1 used: *_f32_mfma_pricing keeps that figure for continuity"},
        "per_kernel": {k: {"ms_per_step": v["ms"] / n_prof, "launches": v["launches"] // n_prof,
                           "TFLOPs": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                           "GBs": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                           "roofline_ms": v["roof_ms"] / n_prof, "frac": v["roof_ms"] / v["ms"]}
                       for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}})
    return out


def time_steps(step, drain, steps, warmup, dist, dev):
    for i in range(warmup):
        step(i)
    drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed


def run_single_gpu_workload(cfg, name, storage, dev, steps, warmup, use_table=True, n_prof=2, batch=None):
    """One extra workload on this GPU (the `secondary` block): device-generated utterances.  batch: another batch size at the
    workload's utterance length (cfg3 at B = 32: what each of 8 ranks runs twice for cfg4)."""
    wl = S.WORKLOADS[name]
    B, F = batch or wl["B"], wl["F"]
    T = F * cfg.hop
    plan = A.Plan(cfg, load_shipped_table=use_table, storage=storage, compact_workspace=True)
    blob = plan.pack(S.synth_state_dict(cfg, WEIGHT_SEED)).to(dev)
    args_dev = list(S.device_batch(cfg, B, F, wl["seed"], dev))
    ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 1, T), dtype=torch.float32, device=dev)
    elapsed = time_steps(lambda i: plan.forward(blob, *args_dev, out=out, workspace=ws),
                         torch.cuda.synchronize, steps, warmup, None, dev)
    ms = elapsed / steps * 1e3
    roof = roofline(plan, blob, args_dev, ms, n_prof=n_prof, pmc_file=f"{PMC_TAG}_{name}_{storage}_pmc_traffic.json")
    top = sorted(roof["per_kernel"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:6]
    res = {"workload": f"{name}: {wl['desc']}, F={F}, T={T}", "storage": storage,
           "dtype": plan.arithmetic, "ms_per_step": ms, "value": B * T * steps / elapsed, "unit": "samples/s",
           "steps": steps, "warmup": warmup, "workspace_GB": plan.workspace_bytes(B, F) / 1e9,
           "data": "synthetic (generated on the device)",
           "roofline": {"e2e": roof["e2e"], "kernel": roof["kernel"], "bound": roof["bound"], "frac": roof["frac"],
                        "achieved": roof["achieved"], "peak": roof["peak"], "unit": roof["unit"],
                        "top_kernels_by_time": {k: v for k, v in top}}}
    del ws, out, args_dev, blob
    torch.cuda.empty_cache()
    return res


def run_cfg4_single_gpu(cfg, dev, steps=3, warmup=1, name="cfg4", storage="float32"):
    """BASELINE config 4's single-GPU leg: the 512 x 10 s set through `run_utterance_parallel` (the multi-GPU code
    path) on a 1-rank RCCL group, inputs resident in HBM, waveforms gathered into the result list.
    name="cfg4var": SURVEY 8(d)'s variable-length variant (2 - 10 s, ragged length-bucketed batches; `value` counts
    real samples only)."""
    import torch.distributed as dist1
    from svcc23_fastsvc_amd import distributed as D
    wl = S.WORKLOADS[name]
    n_utts, F = wl["B"], wl["F"]
    frames = S.workload_frames(name)
    ragged = name == "cfg4var"
    T = F * cfg.hop
    own_group = not dist1.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        dist1.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        plan = A.Plan(cfg, compact_workspace=True, storage=storage)
        blob = plan.pack(S.synth_state_dict(cfg, WEIGHT_SEED)).to(dev)
        utts = []
        for c0 in range(0, n_utts, 64):
            ppg, sine, lft, emb = S.device_batch(cfg, 64, F, wl["seed"] + c0, dev)
            utts += [dict(ppg=ppg[j, :, : frames[c0 + j]], sine=sine[j, :, : frames[c0 + j] * cfg.hop],
                          lft=lft[j, :, : frames[c0 + j] * cfg.hop], spk_emb=emb[j]) for j in range(64)]
        ws = torch.empty(plan.workspace_bytes(64, F), dtype=torch.uint8, device=dev)

        if ragged:
            def fwd(ppg, sine, lft, emb, lens, out=None):
                return plan.forward(blob, ppg, sine, lft, emb, lengths=lens, workspace=ws, out=out)
        else:
            def fwd(ppg, sine, lft, emb, out=None):
                return plan.forward(blob, ppg, sine, lft, emb, workspace=ws, out=out)

        def step(i):
            D.run_utterance_parallel(fwd, utts, dev, max_batch=64, n_frames=frames, hop=cfg.hop, forward_into=True, ragged=ragged)

        elapsed = time_steps(step, torch.cuda.synchronize, steps, warmup, None, dev)
        ms = elapsed / steps * 1e3
        res = {"workload": f"{name}: {wl['desc']} on ONE GPU (batches of <= 64 through run_utterance_parallel, 1-rank RCCL group)",
               "ms_per_step": ms, "value": float(sum(frames)) * cfg.hop * steps / elapsed, "unit": "samples/s", "steps": steps, "warmup": warmup,
               "dtype": plan.arithmetic, "data": "synthetic (generated on the device)"}
        del ws, utts, blob
        torch.cuda.empty_cache()
        return res
    finally:
        if own_group:
            dist1.destroy_process_group()


def run_off_table_shape(cfg, dev, B=5, F=731, steps=50, warmup=10):
    """A shape the shipped launch table has no entries for (real decode batches never are on it): the static cost
    model's launch shapes against on-device autotuning of the same (B, F)."""
    T = F * cfg.hop
    args_dev = list(S.device_batch(cfg, B, F, 4711, dev))
    out = {"workload": f"off-table: {B} x {F} frames ({T} samples each), float32 storage"}
    for mode in ("cost_model", "autotuned"):
        plan = A.Plan(cfg, compact_workspace=True)
        blob = plan.pack(S.synth_state_dict(cfg, WEIGHT_SEED)).to(dev)
        ws = torch.empty(plan.workspace_bytes(B, plan.padded_frames(F)), dtype=torch.uint8, device=dev)
        n_entries = sum(1 for k in plan.tuned_shapes() if k.split("|")[1] == str(B))
        if mode == "autotuned":
            plan.forward(blob, *args_dev, workspace=ws, autotune=True)
        elapsed = time_steps(lambda i: plan.forward(blob, *args_dev, workspace=ws), torch.cuda.synchronize, steps, warmup, None, dev)
        out[mode] = {"ms_per_step": elapsed / steps * 1e3, "value": B * T * steps / elapsed, "unit": "samples/s",
                     "table_entries_for_this_batch_size_before": n_entries,
                     "autotune_trials": plan.last_autotune_trials if mode == "autotuned" else 0}
        del ws, blob
    out["cost_model_over_autotuned"] = out["cost_model"]["ms_per_step"] / out["autotuned"]["ms_per_step"]
    torch.cuda.empty_cache()
    return out


def cpu_train_step_baseline(budget_s: float):
    """The recipe's training step on the host cores (reported baseline for cfg5): the same TrainStep harness with the
    generator as the differentiable PyTorch restatement of the reference dataflow (autograd._forward_torch), yaml-width
    generator and discriminator, on a BOUNDED batch of 2 crops of 16000 samples (the recipe's is 32)."""
    import torch.nn as nn
    from svcc23_fastsvc_amd import autograd as AG
    from svcc23_fastsvc_amd import training as TRN
    cfg = S.FULL_CONFIG
    nt = min(16, _usable_cores())
    torch.set_num_threads(nt)

    class TorchGen(nn.Module):
        def __init__(self):
            super().__init__()
            self.m = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels),
                                        upsampling_scales=list(cfg.upsampling_scales), out_channels=1,
                                        spk_emb_size=cfg.spk_emb_size, use_spk_emb=True)
            self.m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, WEIGHT_SEED).items()})

        def forward(self, x, s, l, spk_emb=None):
            return AG._forward_torch(AG.folded_weights(dict(self.m.named_parameters())), self.m.upsampling_scales, x, s, l, spk_emb)

    B, F = 2, TRN.RECIPE["batch_length"] // cfg.hop
    T = F * cfg.hop
    gen = TorchGen().train()
    disc = TRN.MelGANMultiScaleDiscriminator(**TRN.RECIPE["discriminator_params"]).train()
    trainer = TRN.TrainStep(gen, disc, dict(discriminator_train_start_steps=0), steps=1)
    b = S.synth_batch(cfg, B, F, 5000)
    batch = (tuple(torch.from_numpy(a) for a in (b.ppg, b.sine, b.lft, b.spk_emb)), torch.randn((B, 1, T)) * 0.3)
    trainer.step(batch, log=False)                           # warm-up
    times = []
    t_end = time.time() + budget_s
    while len(times) < 2 or (time.time() < t_end and len(times) < 10):
        t = time.time()
        trainer.step(batch, log=False)
        times.append(time.time() - t)
    med = float(np.median(times))
    return {"value": B * T / med, "unit": "samples/s", "cores": nt, "kind": "port",
            "sample": f"the same training step (generator fwd + bwd as PyTorch CPU operators over the restated dataflow, MelGAN "
                      f"discriminator, MR-STFT + adversarial losses, RAdam), batch {B} x {T} samples (recipe: 32), fp32, median of "
                      f"{len(times)} steps on {nt} threads", "ms_per_step": med * 1e3}


def training_kernel_rooflines(dev, B, T, n=20):
    """The hand-written kernels of the training step's backward side, timed live (HIP events on the launch stream, `n`
    launches each) at the recipe batch's widest tensors - the top up block's 24-channel layers at the full sample rate -
    and priced against their own rooflines: float32-operand MFMA (157.3 TFLOP/s dense) for the convolutions, 8 TB/s for the
    FiLM / InstanceNorm node; the STFT loss is launch- and latency-bound (six resolutions, 0.5 M samples) and carries
    its time only."""
    import ctypes
    from svcc23_fastsvc_amd import conv_grad as CG, training as TRN
    lib = A.load_library()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3                     # us

    C, K, d = 24, 3, 3
    x = torch.randn((B, C, T), device=dev)
    w = torch.randn((C, C, K), device=dev) / (C * K) ** 0.5
    bias = torch.randn((C,), device=dev)
    y, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(bias)
    scratch = torch.empty(int(lib.fastsvc_conv1d_backward_weight_scratch_bytes(B, C, C, T, K)), dtype=torch.uint8, device=dev)
    flops = 2.0 * B * C * C * K * T
    byts = 2.0 * B * C * T * 4
    out = {}
    for name, fn in (
            ("conv1d_forward", lambda: lib.fastsvc_conv1d_forward(vp(x), vp(w), vp(bias), vp(y), B, C, C, T, K, d, 0, st)),
            ("conv1d_backward_data", lambda: lib.fastsvc_conv1d_forward(vp(x), vp(w), None, vp(y), B, C, C, T, K, d, 1, st)),
            ("conv1d_backward_weight", lambda: lib.fastsvc_conv1d_backward_weight(vp(x), vp(y), vp(dw), vp(db), vp(scratch), B, C, C, T, K, d, st))):
        us = timed(fn)
        out[name] = {"shape": f"B{B} {C}->{C} T{T} k{K} d{d} float32", "us": us, "TFLOPs": flops / us / 1e6, "GBs": byts / us / 1e3,
                     "bound": "mfma", "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": flops / us / 1e6 / PEAK_FP32_MFMA_TFLOPS,
                     "hbm_frac": byts / us / 1e3 / PEAK_HBM_GBS}
    sc, sh, o = torch.randn_like(x), torch.randn_like(x), torch.empty_like(x)
    rb = torch.randn(B * C, device=dev)
    mean, rstd = torch.empty_like(rb), torch.empty_like(rb)
    dx, dsc, dsh, dbb = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), torch.empty_like(rb)
    us = timed(lambda: lib.fastsvc_film_norm_forward(vp(x), vp(sc), vp(sh), vp(rb), vp(o), vp(mean), vp(rstd), B * C, T,
                                                     ctypes.c_float(1e-5), ctypes.c_float(0.2), st))
    fb = 4.0 * B * C * T * 4                                      # x, scale, shift read once; out written
    out["film_norm_forward"] = {"shape": f"({B}, {C}, {T}) float32", "us": us, "GBs": fb / us / 1e3, "bound": "hbm",
                                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": fb / us / 1e3 / PEAK_HBM_GBS}
    us = timed(lambda: lib.fastsvc_film_norm_backward(vp(y), vp(x), vp(sc), vp(sh), vp(rb), vp(mean), vp(rstd), vp(dx), vp(dsc), vp(dsh),
                                                      vp(dbb), B * C, T, ctypes.c_float(0.2), st))
    bb = 7.0 * B * C * T * 4                                      # dout, x, scale, shift read; dx, dscale, dshift written
    out["film_norm_backward"] = {"shape": f"({B}, {C}, {T}) float32", "us": us, "GBs": bb / us / 1e3, "bound": "hbm",
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bb / us / 1e3 / PEAK_HBM_GBS}
    crit = A.MultiResolutionSTFTLoss(**TRN.RECIPE["stft_loss_params"]).to(dev)
    yt = torch.randn((B, 1, T), device=dev) * 0.2
    xt = (yt * 0.9 + 0.05 * torch.randn_like(yt)).requires_grad_(True)

    def loss_step():
        a, b = crit(xt, yt)
        torch.autograd.grad(a + b, xt)
    out["stft_loss_forward_backward"] = {"shape": f"({B}, 1, {T}), 6 resolutions", "us": timed(loss_step),
                                         "note": "14 launches (6 forward + fold, 6 backward + gather), through the autograd node"}
    return out


class _Skip(Exception):
    pass


def run_cfg5(args, dist, world, rank, dev, extras=True):
    """BASELINE config 5: the recipe's training step (train_fastsvc.py:157-240) per GPU on a batch of 32 crops of
    16000 samples (fastsvc.yaml:71-72), both sub-networks training, gradients averaged over the ranks (RCCL).
    Generator forward = HIP kernels (twice per step: the trainer's second, no-grad forward feeds the
    discriminator); generator backward, discriminator and losses = PyTorch-ROCm (svcc23_fastsvc_amd/training.py)."""
    from svcc23_fastsvc_amd import training as TRN
    cfg = S.FULL_CONFIG
    B, F = TRN.RECIPE["batch_size"], TRN.RECIPE["batch_length"] // cfg.hop
    T = F * cfg.hop
    torch.manual_seed(1234)                                  # same initial weights on every rank
    gen = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels),
                             upsampling_scales=list(cfg.upsampling_scales), out_channels=1,
                             spk_emb_size=cfg.spk_emb_size, use_spk_emb=True)
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, WEIGHT_SEED).items()})
    gen.activation_storage = args.storage
    gen = gen.to(dev).train()
    hifi = getattr(args, "discriminator", "melgan") == "hifigan"
    disc = (TRN.HiFiGANMultiScaleMultiPeriodDiscriminator() if hifi else
            TRN.MelGANMultiScaleDiscriminator(**TRN.RECIPE["discriminator_params"])).to(dev).train()
    dname = "HiFiGAN multi-period + multi-scale discriminator (reference defaults)" if hifi else "MelGAN multi-scale discriminator (recipe yaml)"
    bf16 = args.storage == "bfloat16"
    trainer = TRN.TrainStep(gen, disc, dict(discriminator_train_start_steps=0, autocast_dtype="bfloat16" if bf16 else None), steps=1)
    ppg, sine, lft, emb = S.device_batch(cfg, B, F, 5000 + rank, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(77 + rank)
    target = torch.randn((B, 1, T), generator=g, device=dev) * 0.3
    batch = ((ppg, sine, lft, emb), target)
    step = lambda i: trainer.step(batch, log=False)
    elapsed = time_steps(step, torch.cuda.synchronize, args.steps, args.warmup, dist if world > 1 else None, dev)
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    # roofline of the HIP share of the step: the generator forward (run twice per step) profiled launch by launch at
    # the recipe batch, priced like every forward line; e2e.frac is against the WHOLE step (PyTorch backward,
    # discriminator and losses included), `hip_share_of_step` says how much of the step the HIP launches are
    roof = None
    try:
        if not extras:
            raise _Skip()
        with torch.no_grad():
            roof = roofline(gen.plan, gen.packed_weights(dev), [ppg, sine, lft, emb], ms, n_prof=2)
        roof["hip_forwards_per_step"] = 2
        roof["hip_share_of_step"] = 2.0 * roof["e2e"]["serial_sum_ms"] / ms
        roof["e2e"]["frac"] = 2.0 * roof["e2e"]["per_kernel_roofline_ms"] / ms
        roof["e2e"]["note"] += ("  cfg5: frac = 2 x the forward's per-kernel roofline / the measured TRAINING step - the backward, "
                                "discriminator and losses are PyTorch-ROCm operators and count as time without roofline.")
    except _Skip:
        roof = None
    except Exception as e:
        roof = {"error": repr(e)}
    try:
        train_kernels = training_kernel_rooflines(dev, B, T) if extras else None
    except Exception as e:
        train_kernels = {"error": repr(e)}
    cpu = None
    if world == 1 and not args.no_cpu_baseline and extras:
        cpu = cpu_train_step_baseline(args.cpu_seconds)
    n_g = sum(p.numel() for p in gen.parameters())
    n_d = sum(p.numel() for p in disc.parameters())
    return {
        "metric": "audio samples/sec (24 kHz) FastSVC training step", "value": world * B * T * args.steps / elapsed,
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": short_dtype(args.storage, gen.plan.arithmetic),
        "dtype_note": f"generator forward: {gen.plan.arithmetic}; " + ("generator backward: float32 HIP convolution / FiLM-norm nodes (f32 master weights, optimizer state, "
                 "InstanceNorm statistics), remaining ops under bf16 autocast; discriminator, STFT and adversarial losses f32 (MIOpen's bf16 "
                 "backward-data of the discriminator's first conv faults intermittently on this ROCm: training.py)" if bf16 else
                 "generator backward: float32 HIP convolution / FiLM-norm nodes (f32-operand MFMA); STFT loss f32 HIP; discriminator f32 PyTorch-ROCm"),
        "data": "synthetic",
        "config": {"workload": f"cfg5: full train step (generator fwd + bwd, {dname}, MR-STFT x6 + adversarial "
                               f"losses, RAdam), batch {B} x {T} samples per GPU, data-parallel x{world} with a flat-bucket gradient all-reduce"
                               + (", bfloat16 (BASELINE config 5's dtype)" if bf16 else ", float32 (--storage bfloat16: config 5's dtype)"),
                   "global_batch": world * B, "utterance_samples": T, "parallelism": f"data-parallel x{world}",
                   "generator_params": n_g, "discriminator_params": n_d,
                   "hip_path": "generator forward (two per step), the generator's backward convolutions (forward recompute, backward "
                               "data, backward weight / bias), its FiLM + InstanceNorm + LeakyReLU nodes (forward and backward) and the "
                               "multi-resolution STFT loss (forward and backward); discriminator, adversarial losses, RAdam and the "
                               "remaining elementwise ops of the generator's graph are PyTorch-ROCm"},
        "roofline": roof, "training_kernels": train_kernels, "cpu_baseline": cpu,
    }


def run_cfg5_secondary(storage, dev, steps=10, warmup=6, discriminator="melgan"):
    """BASELINE config 5's train step inside the default line: 10 timed steps at the recipe batch on this GPU."""
    ns = argparse.Namespace(storage=storage, steps=steps, warmup=warmup, no_cpu_baseline=True, cpu_seconds=0.0, discriminator=discriminator)
    from svcc23_fastsvc_amd import conv_grad as _cg, gconv as _gc
    before = (dict(_cg.ROUTES), dict(_gc.ROUTES))
    full = run_cfg5(ns, None, 1, 0, dev, extras=False)
    torch.cuda.empty_cache()
    # how the step's convolution nodes were routed (HIP kernels vs the stock operator for shapes they decline): a table or shape
    # change that moves the step back onto the stock operators shows here
    routes = {"conv1d": {k: _cg.ROUTES[k] - before[0][k] for k in before[0]}, "grouped_conv1d": {k: _gc.ROUTES[k] - before[1][k] for k in before[1]}}
    return {"workload": full["config"]["workload"], "ms_per_step": full["ms_per_step"], "value": full["value"],
            "unit": full["unit"], "steps": steps, "warmup": warmup, "dtype": full["dtype"], "conv_routes_all_steps": routes}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, argv)                               # never returns
    default_workload = args.workload is None
    args.workload, args.storage = resolve_workload(args.gpus, args.workload, args.storage)
    big = args.workload in ("cfg3", "cfg4", "cfg4var")
    if args.workload == "cfg5":
        args.steps = args.steps or 10
        args.warmup = args.warmup if args.warmup is not None else 6          # (MIOpen settles on its solvers for the discriminator over the first steps)
    if args.steps is None:
        args.steps = 10 if args.workload.startswith("cfg4") else 40 if big else 200
    if args.warmup is None:
        args.warmup = 2 if args.workload.startswith("cfg4") else 10 if big else 50
    # stdout carries exactly ONE line, the JSON: whatever libraries print there (RCCL's version banner on the first
    # collective, for one) goes to stderr - fd 1 is pointed at fd 2 until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, json_fd)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.workload == "cfg5":
        line = run_cfg5(args, dist, world, rank, dev)
        if rank == 0:
            emit(json_fd, line, args.detail)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    cfg = S.FULL_CONFIG
    wl = S.WORKLOADS[args.workload]
    strong = args.workload in ("cfg4", "cfg4var")
    ragged = args.workload == "cfg4var"
    B, F = (64, wl["F"]) if strong else (wl["B"], wl["F"])      # cfg4 runs in batches of 64
    T = F * cfg.hop
    plan = A.Plan(cfg, load_shipped_table=not args.no_table, storage=args.storage, compact_workspace=True)
    n_table = sum(1 for k in plan.tuned_shapes() if k.split("|")[1] == str(B))

    # weights: rank 0 folds + packs, everyone receives the kernel-layout blob (RCCL broadcast)
    if rank == 0:
        blob = plan.pack(S.synth_state_dict(cfg, WEIGHT_SEED)).to(dev)
    else:
        blob = torch.empty(plan.blob_bytes // 4, dtype=torch.float32, device=dev)
    if dist is not None:
        dist.broadcast(blob, src=0)

    if strong:
        # ---- cfg4: the 512-utterance set, sharded over the ranks; inputs of the rank's shard resident in HBM ----
        from svcc23_fastsvc_amd import distributed as D
        if dist is None:                                   # single process: a 1-rank group keeps one code path
            import torch.distributed as dist1
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist1.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            dist = dist1
        n_utts = wl["B"]
        n_frames = S.workload_frames(args.workload)
        mine = D.shard_utterances(n_frames, world)[rank]
        utts = [None] * n_utts
        for c0 in range(0, len(mine), 64):
            chunk = mine[c0: c0 + 64]
            ppg, sine, lft, emb = S.device_batch(cfg, len(chunk), F, wl["seed"] + 7919 * rank + c0, dev)
            for j, i in enumerate(chunk):
                f = n_frames[i]
                utts[i] = dict(ppg=ppg[j, :, :f], sine=sine[j, :, : f * cfg.hop], lft=lft[j, :, : f * cfg.hop], spk_emb=emb[j])
        ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
        def roofline_batch():                           # the longest B utterances as one full batch (profiled after the timing)
            first = sorted(mine, key=lambda i: -n_frames[i])[:B]
            return [torch.stack([torch.nn.functional.pad(utts[i][k], (0, (F - n_frames[i]) * (1 if k == "ppg" else cfg.hop)))
                                 if k != "spk_emb" else utts[i][k] for i in first]) for k in ("ppg", "sine", "lft", "spk_emb")]
        args_dev = None

        if ragged:
            def fwd(ppg, sine, lft, emb, lens, out=None):
                return plan.forward(blob, ppg, sine, lft, emb, lengths=lens, workspace=ws, out=out)
        else:
            def fwd(ppg, sine, lft, emb, out=None):
                return plan.forward(blob, ppg, sine, lft, emb, workspace=ws, out=out)

        def step(i):
            D.run_utterance_parallel(fwd, utts, dev, max_batch=64, n_frames=n_frames, hop=cfg.hop, forward_into=True, ragged=ragged)

        drain = torch.cuda.synchronize
        samples_per_step = float(sum(n_frames)) * cfg.hop            # real samples only: padding is not output
    else:
        if args.workload == "cfg3":
            args_dev = list(S.device_batch(cfg, B, F, wl["seed"] + 1000 * rank, dev))
        else:
            b = S.synth_batch(cfg, B, F, wl["seed"] + 1000 * rank)
            args_dev = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
        ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
        outs = [torch.empty((B, 1, T), dtype=torch.float32, device=dev) for _ in range(2)]
        gathered = [torch.empty((world * B, 1, T), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None
        pending = [None, None]

        def step(i):
            y = outs[i & 1]
            if world > 1 and pending[i & 1] is not None:
                pending[i & 1].wait()                      # the gather that last read this buffer
            plan.forward(blob, *args_dev, out=y, workspace=ws)
            if world > 1:
                pending[i & 1] = dist.all_gather_into_tensor(gathered[i & 1], y, async_op=True)

        def drain():
            if world > 1:
                for h in pending:
                    if h is not None:
                        h.wait()
            torch.cuda.synchronize()

        samples_per_step = float(world) * B * T

    # one-off, untimed: pick the launch shape of every layer for this (B, F) on this device
    # (the reference's recipes run with torch.backends.cudnn.benchmark = True, train_fastsvc.py:617)
    if args.autotune:
        if args_dev is None:
            args_dev = roofline_batch()
        plan.forward(blob, *args_dev, workspace=ws, autotune=True)
    elapsed = time_steps(step, drain, args.steps, args.warmup, dist if world > 1 else None, dev)
    ms_per_step = elapsed / args.steps * 1e3

    weak = None
    if world > 1 and default_workload and not args.no_secondary:
        # the weak-scaling figure next to the strong-scaling headline: every rank a cfg2 batch (8 x 4 s) of its own
        # utterances, waveforms all-gathered asynchronously, same storage
        w2 = S.WORKLOADS["cfg2"]
        B2, F2 = w2["B"], w2["F"]
        T2 = F2 * cfg.hop
        a2 = list(S.device_batch(cfg, B2, F2, w2["seed"] + 1000 * rank, dev))
        ws2 = torch.empty(plan.workspace_bytes(B2, F2), dtype=torch.uint8, device=dev)
        outs2 = [torch.empty((B2, 1, T2), dtype=torch.float32, device=dev) for _ in range(2)]
        gath2 = [torch.empty((world * B2, 1, T2), dtype=torch.float32, device=dev) for _ in range(2)]
        pend2 = [None, None]

        def step2(i):
            if pend2[i & 1] is not None:
                pend2[i & 1].wait()
            plan.forward(blob, *a2, out=outs2[i & 1], workspace=ws2)
            pend2[i & 1] = dist.all_gather_into_tensor(gath2[i & 1], outs2[i & 1], async_op=True)

        def drain2():
            for h in pend2:
                if h is not None:
                    h.wait()
            torch.cuda.synchronize()

        e2 = time_steps(step2, drain2, 100, 20, dist, dev)
        weak = {"workload": f"cfg2: {w2['desc']} per GPU, weak scaling, all-gather of waveforms", "storage": args.storage,
                "ms_per_step": e2 / 100 * 1e3, "value": world * B2 * T2 * 100 / e2, "unit": "samples/s", "n_gpus": world,
                "steps": 100, "warmup": 20, "scaling": "weak"}
        del ws2, outs2, gath2, a2

    if rank == 0:
        # per-batch time for the end-to-end fraction: the step's time scaled to one full B x F batch of samples
        # (cfg4: 1 / batches per step; cfg4var: by the samples of this rank's shard - its batches differ in size)
        if args_dev is None:
            args_dev = roofline_batch()
        batch_share = B * T / (float(sum(n_frames[i] for i in mine)) * cfg.hop) if strong else 1.0
        roof = roofline(plan, blob, args_dev, ms_per_step * batch_share,
                        pmc_file=f"{PMC_TAG}_{'cfg3' if strong else args.workload}_{args.storage}_pmc_traffic.json")
        secondary = None
        if world == 1 and not args.no_secondary and default_workload:
            del ws
            torch.cuda.empty_cache()
            secondary = {}
            runs = [("cfg3_float32", lambda: run_single_gpu_workload(cfg, "cfg3", "float32", dev, steps=20, warmup=5, use_table=not args.no_table)),
                    ("cfg2_float32", lambda: run_single_gpu_workload(cfg, "cfg2", "float32", dev, steps=200, warmup=50, use_table=not args.no_table)),
                    ("cfg2_bfloat16", lambda: run_single_gpu_workload(cfg, "cfg2", "bfloat16", dev, steps=200, warmup=50, use_table=not args.no_table)),
                    ("cfg1_float32", lambda: run_single_gpu_workload(cfg, "cfg1", "float32", dev, steps=200, warmup=50, use_table=not args.no_table)),
                    ("cfg3_b32_bfloat16", lambda: run_single_gpu_workload(cfg, "cfg3", "bfloat16", dev, steps=20, warmup=5, use_table=not args.no_table, batch=32)),
                    ("cfg4_bfloat16_n1", lambda: run_cfg4_single_gpu(cfg, dev, storage="bfloat16")),
                    ("cfg4_float32_n1", lambda: run_cfg4_single_gpu(cfg, dev)),
                    ("cfg4var_float32_n1", lambda: run_cfg4_single_gpu(cfg, dev, name="cfg4var")),
                    ("off_table_shape", lambda: run_off_table_shape(cfg, dev)),
                    ("cfg5_float32", lambda: run_cfg5_secondary("float32", dev)),
                    ("cfg5_bfloat16", lambda: run_cfg5_secondary("bfloat16", dev)),
                    ("cfg5_hifigan_float32", lambda: run_cfg5_secondary("float32", dev, steps=5, warmup=4, discriminator="hifigan"))]
            for key, fn in runs:
                try:
                    secondary[key] = fn()
                except Exception as e:        # an extra block must never take the headline line down
                    secondary[key] = {"error": repr(e)}
            # What 8 ranks would take for cfg4 (512 x 10 s, 64 utterances per rank in two half-batches of 32 so that the first
            # half's all-gather overlaps the second half's kernels, distributed.GatherSchedule): two 32-utterance forwards plus
            # the non-overlapped gather of the LAST half - 8 x 32 x 240000 float32 samples, priced at the per-link bound of a
            # ring over xGMI ((N-1)/N of the gathered bytes at 153 GB/s; a direct all-to-all over 7 links would be 7x
            # shorter).  A prediction from single-GPU measurements: no 8-GPU node was available to this build.
            try:
                t32 = secondary["cfg3_b32_bfloat16"]["ms_per_step"]
                t1 = secondary["cfg4_bfloat16_n1"]["ms_per_step"]
                gather_ms = (7.0 / 8.0) * (8 * 32 * 240000 * 4) / 153e9 * 1e3
                secondary["cfg4_predicted_8rank_bfloat16"] = {
                    "ms_per_step": 2.0 * t32 + gather_ms, "half_batch_ms": t32, "gather_tail_ms": gather_ms,
                    "predicted_speedup_vs_1gpu": t1 / (2.0 * t32 + gather_ms),
                    "note": "2 x (32 x 10 s forward on one GPU) + ring all-gather tail at the xGMI per-link bound; "
                            "measured on ONE GPU, not an 8-GPU run"}
            except Exception:
                pass
        if weak is not None:
            secondary = {"weak_cfg2": weak}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_seconds)
        value = samples_per_step * args.steps / elapsed
        flops_step = plan.flops_per_sample * samples_per_step / world
        on_table = f"shipped table tuned_mi355x.json ({n_table} entries for this batch size)" if n_table else \
                   "static cost model (no table entries for this (B, F): off-table sizes run the slower, untuned shapes)"
        line = {
            "metric": "audio samples/sec (24 kHz) FastSVC generator fwd",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": short_dtype(args.storage, plan.arithmetic), "dtype_note": plan.arithmetic,
            "data": "synthetic",
            "config": {"workload": (f"{args.workload}: {wl['desc']}, sharded over {world} GPU(s) in batches of <= {B}, "
                                    if strong else f"{args.workload}: {wl['desc']} per GPU, ") +
                                   f"F={F} frames, T={T} samples, generator fastsvc.yaml (144->[192,96,48,24], x[2,4,4,5]), spk_emb on",
                       "global_batch": wl["B"] if strong else world * B, "utterance_samples": T if not ragged else
                       f"{wl['Fmin'] * cfg.hop} - {T} (value counts real samples only; padded batches, pad <= 12.5 %)",
                       "parallelism": f"utterance-parallel x{world}" + (" + all-gather of waveforms" if (world > 1 or strong) else "")},
            "rtf_24k": 24000.0 / value,
            "alg_gflop_per_step": flops_step / 1e9,
            "e2e_alg_tflops_per_gpu": flops_step / (ms_per_step * 1e-3) / 1e12,
            "launch_shapes": (f"autotuned on device ({plan.last_autotune_trials} timed trials, untimed)" if args.autotune
                              else on_table),
            "roofline": roof,
            "secondary": secondary,
            "cpu_baseline": cpu,
        }
        emit(json_fd, line, args.detail)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
