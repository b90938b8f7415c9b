"""HIP conv node (forward + backward data + backward weight) against MIOpen's F.conv1d autograd on the generator's layer
shapes at BASELINE config 5 (32 x 16000)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svcc23_fastsvc_amd import conv_grad as CG

dev = torch.device("cuda:0")
SHAPES = [(32, 24, 24, 16000, 3, 3), (32, 24, 24, 16000, 3, 27), (64, 24, 24, 16000, 3, 2), (32, 48, 48, 3200, 3, 9),
          (32, 96, 96, 800, 3, 3), (32, 192, 192, 200, 3, 27), (32, 144, 192, 100, 3, 1), (64, 24, 48, 3200, 1, 1),
          (32, 24, 1, 16000, 3, 1), (64, 1, 24, 16000, 3, 1)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, Ci, Co, T, K, d) in SHAPES:
    x = torch.randn((B, Ci, T), device=dev, requires_grad=True)
    w = (torch.randn((Co, Ci, K), device=dev) / (Ci * K) ** 0.5).requires_grad_(True)
    b = torch.randn((Co,), device=dev, requires_grad=True)
    gy = torch.randn((B, Co, T), device=dev)
    flops = 2.0 * B * Co * Ci * K * T

    def run(conv):
        def f():
            y = conv()
            torch.autograd.grad(y, (x, w, b), gy)
        return f
    t_hip_f = timeit(lambda: CG.conv1d(x, w, b, d))
    t_mio_f = timeit(lambda: F.conv1d(x, w, b, padding=(K // 2) * d, dilation=d))
    t_hip = timeit(run(lambda: CG.conv1d(x, w, b, d)))
    t_mio = timeit(run(lambda: F.conv1d(x, w, b, padding=(K // 2) * d, dilation=d)))
    print(f"B{B} {Ci}->{Co} T{T} k{K} d{d}: fwd hip {t_hip_f:7.1f} us ({flops / t_hip_f / 1e6:5.1f} TF) miopen {t_mio_f:7.1f} | "
          f"fwd+bwd hip {t_hip:7.1f} us ({3 * flops / t_hip / 1e6:5.1f} TF) miopen {t_mio:7.1f}")
