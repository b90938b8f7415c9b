#!/usr/bin/env python
"""bench.py - audio samples/sec of the FastSVC generator forward on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one generator forward over one batch of synthetic utterances already resident in HBM
(BASELINE.json configs[1]: 8 x 4 s at 24 kHz, fp32, by default; --workload cfg3 for 64 x 10 s).
With N > 1 every rank runs the same-sized batch of different utterances (utterance-parallel, weak
scaling): rank 0 packs the weights and broadcasts the blob over RCCL, and each step's waveforms
are all-gathered over xGMI (asynchronously, overlapping the next step).  Prints ONE JSON line.

The `roofline` object is measured live with hipEvents on the launch stream
(fastsvc_forward_profile) for the dominant kernel symbol; `cpu_baseline` times the CPU oracle's
as-executed restatement of the reference forward (the reference's CPU PyTorch path) on the host
cores of this box, on a bounded sample - it is a reported baseline, not the target.
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
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak
WEIGHT_SEED = 201


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=["cfg1", "cfg2", "cfg3"])
    ap.add_argument("--storage", default="float32", choices=["float32", "bfloat16"],
                    help="workspace tensor storage: float32 = the parity path (default, what `value` is quoted "
                         "on); bfloat16 = BASELINE config 3's dtype (fp32 arithmetic, bf16-activation accuracy)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--autotune", action="store_true",
                    help="run the untimed device-side autotune pass (cf. cudnn.benchmark) instead of "
                         "the static launch cost model")
    ap.add_argument("--no-table", action="store_true",
                    help="ignore the shipped launch-shape table (svcc23_fastsvc_amd/tuned_mi355x.json)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


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
    """Oracle `forward_as_executed` (the reference's op sequence incl. its redundant chains, fp32
    CPU PyTorch) on the host cores: one cfg1-shaped utterance (B=1, F=300 -> 48 000 samples) per
    iteration.  PyTorch's intra-op pool does not scale to a whole 2-socket box on convolutions
    this small (an all-cores run is slower than 16 threads), so a few thread counts are tried
    within the time budget and the BEST is reported, with the thread count it used."""
    from oracle import fastsvc_oracle as O      # checker / reported baseline only
    cfg = S.FULL_CONFIG
    w = S.fold_weight_norm(S.synth_state_dict(cfg, WEIGHT_SEED))
    b = S.synth_batch(cfg, 1, 300, 4242)
    usable = _usable_cores()
    cands = sorted({1, min(8, usable), min(16, usable), min(32, usable), min(64, usable)})
    per = max(2.0, budget_s / len(cands))
    best = None
    tried = {}
    for nt in cands:
        torch.set_num_threads(nt)
        O.forward_as_executed(w, cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb)   # warm-up
        times = []
        t_end = time.time() + per
        while len(times) < 3 or (time.time() < t_end and len(times) < 40):
            t = time.time()
            O.forward_as_executed(w, cfg.upsampling_scales, b.ppg, b.sine, b.lft, b.spk_emb)
            times.append(time.time() - t)
            if times[-1] > per:          # hopelessly oversubscribed: one sample is enough
                break
        med = float(np.median(times))
        tried[str(nt)] = 48000.0 / med
        if best is None or 48000.0 / med > best[0]:
            best = (48000.0 / med, nt, len(times))
    return {"value": best[0], "unit": "samples/s", "cores": int(best[1]), "kind": "port",
            "host_cores_usable": usable, "samples_per_s_by_threads": tried,
            "sample": f"oracle forward_as_executed (reference op sequence, fp32 CPU PyTorch), "
                      f"1 x 2 s utterance (48000 samples) per run, median of {best[2]} runs at the "
                      f"best of {cands} threads"}


def roofline(plan, blob, args_dev, n_prof=3):
    """Per-kernel-symbol aggregation of hipEvent-timed launches; dominant symbol by time.
    fastsvc_forward_profile runs the forward on ONE stream (no helper streams) so that each
    kernel is timed running alone; `profiles/*_kernel_stats_serial.csv` is rocprofv3's view of the
    same thing (FASTSVC_SERIAL=1), `*_kernel_stats.csv` the default multi-stream schedule."""
    agg = {}
    total_ms = 0.0
    for _ in range(n_prof):
        recs = []
        plan.forward(blob, *args_dev, profile=recs)
        for r in recs:
            a = agg.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["launches"] += 1
            total_ms += r["ms"]
    kern, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
    sec = a["ms"] * 1e-3
    tf = a["flops"] / sec / 1e12
    gbs = a["bytes"] / sec / 1e9
    t_mfma = a["flops"] / (PEAK_FP32_MFMA_TFLOPS * 1e12)
    t_hbm = a["bytes"] / (PEAK_HBM_GBS * 1e9)
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(kern)
        except Exception:
            traffic = None
    if t_mfma >= t_hbm:
        out = {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
               "frac": tf / PEAK_FP32_MFMA_TFLOPS}
    else:
        out = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": gbs / PEAK_HBM_GBS}
    out.update({"traffic": traffic, "kernel": kern,
                "avg_launch_us": a["ms"] * 1e3 / a["launches"], "launches_per_step": a["launches"] // n_prof,
                "share_of_step": a["ms"] / total_ms,
                "alg_flops_per_launch": a["flops"] / a["launches"], "alg_bytes_per_launch": a["bytes"] / a["launches"],
                "hbm_side_GBs": gbs, "mfma_side_TFLOPs": tf,
                "per_kernel": {k: {"ms_per_step": v["ms"] / n_prof, "launches": v["launches"] // n_prof,
                                   "TFLOPs": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                   "GBs": v["bytes"] / (v["ms"] * 1e-3) / 1e9} for k, v in agg.items()}})
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    cfg = S.FULL_CONFIG
    wl = S.WORKLOADS[args.workload]
    B, F = wl["B"], wl["F"]
    T = F * cfg.hop
    plan = A.Plan(cfg, load_shipped_table=not args.no_table, storage=args.storage)
    n_table = sum(1 for k in plan.tuned_shapes() if k.split("|")[1] == str(B))

    # weights: rank 0 folds + packs, everyone receives the kernel-layout blob (RCCL broadcast)
    if rank == 0:
        blob = plan.pack(S.synth_state_dict(cfg, WEIGHT_SEED)).to(dev)
    else:
        blob = torch.empty(plan.blob_bytes // 4, dtype=torch.float32, device=dev)
    if dist is not None:
        dist.broadcast(blob, src=0)

    b = S.synth_batch(cfg, B, F, wl["seed"] + 1000 * rank)
    args_dev = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
    ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    outs = [torch.empty((B, 1, T), dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty((world * B, 1, T), dtype=torch.float32, device=dev) for _ in range(2)] if dist else None
    pending = [None, None]

    def step(i):
        y = outs[i & 1]
        if dist is not None and pending[i & 1] is not None:
            pending[i & 1].wait()                      # the gather that last read this buffer
        plan.forward(blob, *args_dev, out=y, workspace=ws)
        if dist is not None:
            pending[i & 1] = dist.all_gather_into_tensor(gathered[i & 1], y, async_op=True)

    def drain():
        if dist is not None:
            for h in pending:
                if h is not None:
                    h.wait()
        torch.cuda.synchronize()

    # one-off, untimed: pick the launch shape of every layer for this (B, F) on this device
    # (the reference's recipes run with torch.backends.cudnn.benchmark = True, train_fastsvc.py:617)
    if args.autotune:
        plan.forward(blob, *args_dev, out=outs[0], workspace=ws, autotune=True)
    for i in range(args.warmup):
        step(i)
    drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
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

    if rank == 0:
        roof = roofline(plan, blob, args_dev)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_seconds)
        total_samples = float(world) * B * T * args.steps
        value = total_samples / elapsed
        line = {
            "metric": "audio samples/sec (24 kHz) FastSVC generator fwd",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.storage == "float32" else "f32 arithmetic, bf16 activation storage",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']} per GPU, F={F} frames, T={T} samples, "
                                   f"generator fastsvc.yaml (144->[192,96,48,24], x[2,4,4,5]), spk_emb on",
                       "global_batch": world * B, "utterance_samples": T,
                       "parallelism": f"utterance-parallel x{world}" + (" + all-gather of waveforms" if world > 1 else "")},
            "rtf_24k": 24000.0 / value,
            "alg_gflop_per_step": plan.flops_per_sample * B * T / 1e9,
            "e2e_alg_tflops_per_gpu": plan.flops_per_sample * B * T / (elapsed / args.steps) / 1e12,
            "launch_shapes": (f"autotuned on device ({plan.last_autotune_trials} timed trials, untimed)" if args.autotune
                              else f"shipped table tuned_mi355x.json ({n_table} entries for this workload), else static cost model"
                              if n_table else "static cost model"),
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
