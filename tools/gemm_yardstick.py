#!/usr/bin/env python
"""Yardstick for the wide (C >= 96) layers: the SAME problems as plain bf16 GEMMs through the vendor library
(torch.matmul -> hipBLASLt / rocBLAS).  Measurement only - never product code.  A k=3 conv over C_in channels on M
time columns is M x N x K with N = C_out, K = 3 C_in; the GEMM reads an (M, K) im2col matrix - three times the bytes
the convolution's input has - so the GB/s column prices the GEMM's own operands, the TFLOP/s column is comparable.

    python tools/gemm_yardstick.py [B F]          (default 64 1500 = cfg3)
"""
import sys
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
dev = torch.device("cuda:0")
# (layer, columns per utterance-signal, signals, C_in, C_out)
LAYERS = [
    ("down.2.c2 / c3 (each)", 8 * F, 2, 96, 96),
    ("film.2.conv", 8 * F, 2, 96, 96),
    ("film.2.heads", 8 * F, 1, 192, 192),
    ("down.3.c2 / c3 (each)", 2 * F, 2, 192, 192),
    ("film.3.conv", 2 * F, 2, 192, 192),
    ("film.3.heads", 2 * F, 1, 384, 384),
    ("up.0.conv_first", F, 1, 144, 192),
    ("up.0.d3 / d9 / d27 (each)", 2 * F, 1, 192, 192),
    ("up.1.conv_first", 2 * F, 1, 192, 96),
    ("up.1.d3 / d9 / d27 (each)", 8 * F, 1, 96, 96),
]


def time_mm(a, b, out, reps=20):
    for _ in range(3):
        torch.matmul(a, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        torch.matmul(a, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"plain bf16 GEMMs of the wide layers' shapes at B = {B}, F = {F} (torch {torch.__version__}, {torch.cuda.get_device_name(0)})")
print(f"{'layer':28s} {'M':>9s} {'N':>4s} {'K':>5s} {'us':>8s} {'TFLOP/s':>8s} {'of 2500':>7s} {'GEMM GB/s':>9s} {'conv-bytes us @8TB/s':>20s}")
for name, cols, nsig, cin, cout in LAYERS:
    M, N, K = cols * B * nsig, cout, 3 * cin
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us = time_mm(a, w, out)
    # the other orientation (out^T = w^T a^T: activations channel-major as the generator holds them)
    at = a.t().contiguous()
    outt = torch.empty(N, M, device=dev, dtype=torch.bfloat16)
    us_t = time_mm(w.t().contiguous(), at, outt)
    best = min(us, us_t)
    fl = 2.0 * M * N * K
    gb = 2.0 * (M * K + K * N + M * N)
    conv_us = 2.0 * (M * cin + M * N) / 8e12 * 1e6
    print(f"{name:28s} {M:9d} {N:4d} {K:5d} {best:8.1f} {fl / best / 1e6:8.1f} {fl / best / 1e6 / 2500:7.2f} {gb / best / 1e3:9.0f} {conv_us:20.1f}"
          f"   (row-major {us:.1f}, channel-major {us_t:.1f})")
    del a, w, out, at, outt
