"""Time the HIP multi-resolution STFT loss against the torch composition at the recipe's batch (32 x 16000)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import training as TR

dev = torch.device("cuda:0")
B, T = 32, 16000
y = torch.randn((B, 1, T), device=dev) * 0.2
x0 = y * 0.9 + torch.randn((B, 1, T), device=dev) * 0.05
for name, crit in (("hip", A.MultiResolutionSTFTLoss(**TR.RECIPE["stft_loss_params"]).to(dev)),
                   ("torch", TR.MultiResolutionSTFTLoss(**TR.RECIPE["stft_loss_params"]).to(dev))):
    def step():
        x = x0.clone().requires_grad_(True)
        sc, mag = crit(x, y)
        (sc + mag).backward()
        return x.grad
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per forward + backward")
