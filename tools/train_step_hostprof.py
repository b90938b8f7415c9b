"""Where the HOST time of a cfg5 train step goes (torch.profiler, CPU side): top operators by self CPU time."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S, training as TRN

dev = torch.device("cuda:0")
cfg = S.FULL_CONFIG
B, F = 32, 100
T = F * cfg.hop
torch.manual_seed(1234)
gen = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels), upsampling_scales=list(cfg.upsampling_scales),
                         out_channels=1, spk_emb_size=cfg.spk_emb_size, use_spk_emb=True)
gen.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, 20240607).items()})
gen = gen.to(dev).train()
disc = TRN.MelGANMultiScaleDiscriminator(**TRN.RECIPE["discriminator_params"]).to(dev).train()
tr = TRN.TrainStep(gen, disc, dict(discriminator_train_start_steps=0), steps=1)
ppg, sine, lft, emb = S.device_batch(cfg, B, F, 5000, dev)
y = torch.randn((B, 1, T), device=dev) * 0.3
batch = ((ppg, sine, lft, emb), y)
for _ in range(4):
    tr.step(batch, log=False)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    tr.step(batch, log=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per step {(t1 - t0) / 5 * 1e3:.1f} ms; with drain {(t2 - t0) / 5 * 1e3:.1f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        tr.step(batch, log=False)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
