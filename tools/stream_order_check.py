"""Does the ORDER in which streams come to life matter?  The forward forks side work onto two helper streams
(csrc/fastsvc_plan.cpp, ExecCtx).  Measured on MI355X / ROCm 7.0: helper streams created AFTER an RCCL communicator
was initialised in the process make the multi-stream schedule SLOWER than the one-stream schedule (cfg3: 23.5 vs
22.3 ms); created before it (Plan.prepare_stream) they are fine.   python tools/stream_order_check.py cfg2 <variant>
variants: group_first | ctx_after_group | prepare_first, each with / without FASTSVC_SERIAL set (one-stream schedule)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
import torch.distributed as dist

name, variant = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
cfg = S.FULL_CONFIG; wl = S.WORKLOADS[name]; B, F = wl["B"], wl["F"]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
if variant == "group_first":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
plan = A.Plan(cfg, compact_workspace=True)
blob = plan.pack(S.synth_state_dict(cfg, 1)).to(dev)
if variant == "prepare_first":
    plan.prepare_stream(dev)
if variant != "group_first":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ins = list(S.device_batch(cfg, B, F, 5, dev))
ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
out = torch.empty((B, 1, F * cfg.hop), device=dev)
n = 20 if B * F > 50000 else 300
el = bench.time_steps(lambda i: plan.forward(blob, *ins, workspace=ws, out=out), torch.cuda.synchronize, n, n // 5, None, dev)
print(f"{name} {variant:16s} one_stream={'FASTSVC_SERIAL' in os.environ}: {el / n * 1e3:.4f} ms per forward")
dist.destroy_process_group()
