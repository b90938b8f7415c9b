"""Where a full training step of the recipe (bench.py --workload cfg5) spends its time: wall time of the phases of
TrainStep.step (device-synchronised between them) and the top GPU kernels of one step (torch profiler).
    python tools/train_step_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S, training as TRN

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = os.environ.get("CUDNN_BENCHMARK", "0") == "1"     # (MIOpen find mode; train_fastsvc.py:617 sets it)
cfg = S.FULL_CONFIG
B, F = TRN.RECIPE["batch_size"], TRN.RECIPE["batch_length"] // cfg.hop
T = F * cfg.hop
torch.manual_seed(1234)
gen = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels), upsampling_scales=list(cfg.upsampling_scales),
                         out_channels=1, spk_emb_size=cfg.spk_emb_size, use_spk_emb=True)
gen.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, 1).items()})
gen = gen.to(dev).train()
disc = TRN.MelGANMultiScaleDiscriminator(**TRN.RECIPE["discriminator_params"]).to(dev).train()
trainer = TRN.TrainStep(gen, disc, dict(discriminator_train_start_steps=0), steps=1)
ins = S.device_batch(cfg, B, F, 5000, dev)
target = torch.randn((B, 1, T), device=dev) * 0.3
batch = (ins, target)
tw = time.perf_counter()
for _ in range(3):
    trainer.step(batch, log=False)
torch.cuda.synchronize()
print("3 warm-up steps: %.1f s" % (time.perf_counter() - tw))
t0 = time.perf_counter()
for _ in range(5):
    trainer.step(batch, log=False)
torch.cuda.synchronize()
print("step: %.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
if os.environ.get("NO_PROFILE"):
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    trainer.step(batch, log=False)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
