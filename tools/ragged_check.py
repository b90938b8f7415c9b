"""Where the time of the variable-length set (bench.py --workload cfg4var) goes: per batch of the ragged schedule,
the forward with per-utterance lengths / without (padded shape, full length) / after on-device autotuning, and the
whole pass through run_utterance_parallel (staging + forward + gather).   python tools/ragged_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S, distributed as D

dev = torch.device("cuda:0")
cfg = S.FULL_CONFIG
frames = S.workload_frames("cfg4var")
sched = D.GatherSchedule(frames, cfg.hop, 1, 64, True, 0.125)
plan = A.Plan(cfg, compact_workspace=True)
blob = plan.pack(S.synth_state_dict(cfg, 1)).to(dev)
ws = torch.empty(plan.workspace_bytes(64, 1500), dtype=torch.uint8, device=dev)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tot = {"ragged": 0.0, "padded": 0.0, "tuned": 0.0, "ideal": 0.0}
for chunk in ([] if os.environ.get('RAGGED_SKIP') else sched.batches[0]):
    lens = [frames[i] for i in chunk]
    B, F = len(chunk), max(lens)
    ins = list(S.device_batch(cfg, B, F, 5, dev))
    out = torch.empty((B, 1, F * cfg.hop), device=dev)
    t_r = timed(lambda: plan.forward(blob, *ins, lengths=lens, workspace=ws, out=out))
    t_p = timed(lambda: plan.forward(blob, *ins, workspace=ws, out=out))
    p2 = A.Plan(cfg, compact_workspace=True)
    p2.forward(blob, *ins, workspace=ws, autotune=True)
    t_t = timed(lambda: p2.forward(blob, *ins, lengths=lens, workspace=ws, out=out))
    ideal = 181.6 / 512 / 1500 * sum(lens)          # the fixed-length set's time per frame (profiles/r3_bench_cfg4_n1.json)
    print(f"B {B:3d} F {F:5d} (min {min(lens):5d}): ragged {t_r:7.2f} ms | no lengths {t_p:7.2f} | autotuned+lengths {t_t:7.2f} | at cfg4's rate {ideal:7.2f}", flush=True)
    for k, v in zip(tot, (t_r, t_p, t_t, ideal)):
        tot[k] += v
print("sum over the 13 batches:", {k: round(v, 1) for k, v in tot.items()})

# ---- the pass around the forwards: staging of device-resident utterances into padded batches ----
utts = []
for c0 in range(0, 512, 64):
    ppg, sine, lft, emb = S.device_batch(cfg, 64, 1500, 77 + c0, dev)
    for j in range(64):
        f = frames[c0 + j]
        utts.append(dict(ppg=ppg[j, :, :f], sine=sine[j, :, : f * cfg.hop], lft=lft[j, :, : f * cfg.hop], spk_emb=emb[j]))
stager = D._Stager(utts, dev, cfg.hop)


def stage_all():
    for chunk in sched.batches[0]:
        stager.stage(chunk, max(frames[i] for i in chunk))


print(f"staging the 13 batches (device-resident utterances): {timed(stage_all):.1f} ms")


def fwd(ppg, sine, lft, emb, lens, out=None):
    return plan.forward(blob, ppg, sine, lft, emb, lengths=lens, workspace=ws, out=out)


import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print(f"whole pass (run_utterance_parallel, ragged): {timed(lambda: D.run_utterance_parallel(fwd, utts, dev, max_batch=64, n_frames=frames, hop=cfg.hop, forward_into=True, ragged=True)):.1f} ms")
mine = sched.batches[0]


def one_pass(sync):
    """The loop of run_utterance_parallel, phase by phase (device-synchronised between phases when `sync`)."""
    T = {"stage": 0.0, "alloc": 0.0, "forward": 0.0, "gather": 0.0, "wait": 0.0}

    def tick():
        if sync:
            torch.cuda.synchronize()
        return time.perf_counter()
    keep = []
    for chunk in mine:
        t0 = tick()
        ppg, sine, lft, emb, _ = stager.stage(chunk, max(frames[i] for i in chunk))
        t1 = tick()
        rows = torch.empty((len(chunk), 1, sine.shape[-1]), device=dev)
        t2 = tick()
        plan.forward(blob, ppg, sine, lft, emb, lengths=[frames[i] for i in chunk], workspace=ws, out=rows)
        t3 = tick()
        keep.append(D._gather_rows(rows, 1, None, async_op=True))
        t4 = tick()
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            T[k] += v
    t5 = tick()
    for g, w in keep:
        w.wait()
    torch.cuda.synchronize()
    T["wait"] = time.perf_counter() - t5
    return {k: round(v * 1e3, 1) for k, v in T.items()}


def gpu_timeline():
    """The asynchronous pass with events on the compute stream around every phase: where the DEVICE spends the pass."""
    evs = []
    mark = lambda tag: (evs.append((tag, torch.cuda.Event(enable_timing=True))), evs[-1][1].record())
    keep = []
    mark("start")
    for chunk in mine:
        ppg, sine, lft, emb, _ = stager.stage(chunk, max(frames[i] for i in chunk)); mark("stage")
        rows = torch.empty((len(chunk), 1, sine.shape[-1]), device=dev)
        plan.forward(blob, ppg, sine, lft, emb, lengths=[frames[i] for i in chunk], workspace=ws, out=rows); mark("forward")
        keep.append(D._gather_rows(rows, 1, None, async_op=True)); mark("gather")
    for g, w in keep:
        w.wait()
    mark("wait")
    torch.cuda.synchronize()
    T = {}
    for (t0, e0), (t1, e1) in zip(evs, evs[1:]):
        T[t1] = T.get(t1, 0.0) + e0.elapsed_time(e1)
    return {k: round(v, 1) for k, v in T.items()}


one_pass(False)
print("asynchronous pass, device time between events on the compute stream (ms):", gpu_timeline())
print("phases, device-synchronised between them (ms):", one_pass(True))
t0 = time.perf_counter()
h = one_pass(False)
print("asynchronous pass: host time per phase (ms)", h, "total %.1f ms" % ((time.perf_counter() - t0) * 1e3))
# ---- the same pass with HOST-resident utterances (what a decode harness reading h5 features has): the stager fills
# page-locked buffers and uploads one batch ahead of the compute stream ----
if os.environ.get("RAGGED_HOST", "1") != "0":
    host_utts = [{k: v.cpu() for k, v in u.items()} for u in utts]
    torch.cuda.synchronize()
    t_host = timed(lambda: D.run_utterance_parallel(fwd, host_utts, dev, max_batch=64, n_frames=frames, hop=cfg.hop, forward_into=True, ragged=True))
    print(f"whole pass, utterances in HOST memory (pinned staging + upload overlapped with compute): {t_host:.1f} ms")
    st2 = D._Stager(host_utts, dev, cfg.hop)
    t0 = time.perf_counter()
    for chunk in mine:
        st2.stage(chunk, max(frames[i] for i in chunk))
    torch.cuda.synchronize()
    print(f"   staging alone (host copies into pinned buffers + H2D): {(time.perf_counter() - t0) * 1e3:.1f} ms for {sum(frames) * (cfg.in_channels + 2 * cfg.hop) * 4 / 1e6:.0f} MB")
dist.destroy_process_group()
