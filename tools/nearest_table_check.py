import sys, time, torch
sys.path.insert(0, ".")
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
cfg = S.FULL_CONFIG
dev = torch.device("cuda:0")
def run(B, F, table, wino_model=None):
    plan = A.Plan(cfg, load_shipped_table=table)
    blob = plan.pack(S.synth_state_dict(cfg, 201)).to(dev)
    b = S.synth_batch(cfg, B, F, 77)
    ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
    ws = torch.empty(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    for _ in range(3): plan.forward(blob, *ins, workspace=ws)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): plan.forward(blob, *ins, workspace=ws)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    t_model = dt
    plan.forward(blob, *ins, workspace=ws, autotune=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): plan.forward(blob, *ins, workspace=ws)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t) / 20
    return t_model * 1e3, dt2 * 1e3
for (B, F) in [(4, 400), (16, 600), (2, 900), (32, 300), (1, 1000)]:
    a = run(B, F, True); b = run(B, F, False)
    print(f"B={B} F={F}: shipped table (no entry for this size) {a[0]:.3f} ms | cost model {b[0]:.3f} ms | autotuned {min(a[1], b[1]):.3f} ms", flush=True)
