import sys, time, torch
sys.path.insert(0, "/root/repo")
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
cfg = S.FULL_CONFIG; dev = torch.device("cuda:0")
plan = A.Plan(cfg); blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
b = S.synth_batch(cfg, 8, 600, 1236)
ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
out = torch.empty((8, 1, 96000), device=dev)
for sub in (8, 4, 2, 1):
    ws = torch.empty(plan.workspace_bytes(sub, 600), dtype=torch.uint8, device=dev)
    def run():
        for b0 in range(0, 8, sub):
            plan.forward(blob, *[t[b0:b0+sub] for t in ins], out=out[b0:b0+sub], workspace=ws)
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"sub-batch {sub}: {dt*1e3:.3f} ms per 8 utterances")
