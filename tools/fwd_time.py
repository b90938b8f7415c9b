import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
plan = A.Plan(cfg)
blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
for name in sys.argv[1:] or ["cfg1", "cfg2"]:
    wl = S.WORKLOADS[name]
    ins = list(S.device_batch(cfg, wl["B"], wl["F"], wl["seed"], dev))
    ws = torch.empty(plan.workspace_bytes(wl["B"], wl["F"]), dtype=torch.uint8, device=dev)
    for _ in range(10): plan.forward(blob, *ins, workspace=ws)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): plan.forward(blob, *ins, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    print(f"{name}: {best:.4f} ms", flush=True)
