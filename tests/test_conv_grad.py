"""The convolution node of the generator's training graph: HIP forward / backward-data /
backward-weight kernels (csrc/fastsvc_convgrad.hip, conv_grad.py) against float64 PyTorch convolutions on the CPU for every
(channels, k, dilation) the generator has, ragged time lengths included.  Tolerance: float32 accumulation over
Cin * k (forward, backward data) or B * T (backward weight) terms - 2e-5 of the result's largest magnitude."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import conv_grad as CG


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (and fail loudly without one)"
    A.load_library()
    return torch.device("cuda:0")


def test_conv_grad_rejects_cpu_tensors_and_unsupported_shapes():
    with pytest.raises(A.FastSVCError):
        CG.conv1d(torch.zeros(1, 4, 16), torch.zeros(8, 4, 3))
    lib = A.load_library()
    one = ctypes.c_void_p(256)        # never dereferenced: the argument checks come first
    assert lib.fastsvc_conv1d_forward(one, one, None, one, 1, 4, 8, 16, 5, 1, 0, None) == -5      # k = 5
    assert lib.fastsvc_conv1d_forward(one, one, None, one, 1, 4, 8, 16, 3, 28, 0, None) == -5     # halo 28 > 27
    assert lib.fastsvc_conv1d_forward(one, one, None, one, 1, 4, 8, 16, 2, 1, 0, None) == -1      # even k
    assert lib.fastsvc_conv1d_backward_weight(one, one, one, None, one, 1, 4, 8, 0, 3, 1, None) == -1  # T = 0


# (B, Cin, Cout, T, K, dilation): the generator's layer shapes (fastsvc.py:34-232 at the yaml widths) and odd sizes
CASES = [
    (2, 1, 24, 1000, 3, 1), (2, 1, 24, 1000, 1, 1), (2, 24, 24, 1000, 3, 2), (2, 24, 24, 777, 3, 4),
    (2, 24, 48, 515, 3, 1), (2, 48, 96, 300, 1, 1), (2, 96, 96, 260, 3, 4), (3, 192, 192, 131, 3, 27),
    (2, 192, 384, 75, 3, 1), (2, 144, 192, 50, 3, 1), (2, 24, 1, 1203, 3, 1), (1, 48, 48, 129, 3, 9),
    (2, 96, 48, 400, 3, 3), (1, 7, 5, 33, 3, 2), (5, 20, 70, 128, 1, 1),
    # (time tiles x channel groups not a multiple of 8, several groups: the XCD-aware workgroup permutation's remainder)
    (2, 96, 96, 800, 3, 3), (1, 48, 192, 700, 3, 1), (3, 24, 144, 1300, 1, 1), (2, 192, 96, 330, 3, 9),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_forward_and_gradients_vs_float64(dev, case):
    B, Cin, Cout, T, K, d = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn((B, Cin, T), generator=g)
    w = torch.randn((Cout, Cin, K), generator=g) / np.sqrt(Cin * K)
    b = torch.randn((Cout,), generator=g)
    gy = torch.randn((B, Cout, T), generator=g)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    y64 = F.conv1d(x64, w64, b64, padding=(K // 2) * d, dilation=d)
    y64.backward(gy.double())
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = CG.conv1d(xd, wd, bd, d)
    y.backward(gy.to(dev))
    for name, got, want in (("y", y.detach(), y64.detach()), ("dx", xd.grad, x64.grad), ("dw", wd.grad, w64.grad),
                            ("db", bd.grad, b64.grad)):
        err = (got.double().cpu() - want).abs().max().item()
        assert err <= 2e-5 * want.abs().max().item(), (name, err, want.abs().max().item())
    # without a bias, and with only the input differentiated
    x2 = x.to(dev).requires_grad_(True)
    CG.conv1d(x2, w.to(dev), None, d).backward(gy.to(dev))
    assert (x2.grad - xd.grad).abs().max().item() == 0.0
    # slabs are added in a fixed order: the weight gradient is bit-reproducible
    w3, b3 = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    CG.conv1d(x.to(dev), w3, b3, d).backward(gy.to(dev))
    assert torch.equal(w3.grad, wd.grad) and torch.equal(b3.grad, bd.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 24, 4000), (2, 48, 801), (4, 192, 50), (1, 5, 16001), (2, 7, 1)], ids=str)
def test_film_norm_lrelu_node_vs_float64(dev, shape):
    """`_feature_affine` with a speaker embedding + LeakyReLU (fastsvc.py:115-139, 60-83) as one HIP node: output and all four
    gradients against the float64 composition on the CPU.  Rows with a large mean against a small spread included (the
    two-pass variance): 1e-5 of each result's largest magnitude."""
    B, C, T = shape
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = torch.randn(shape, generator=g)
    sc = 1.0 + 0.3 * torch.randn(shape, generator=g)
    sh = 0.2 * torch.randn(shape, generator=g)
    sh[0, 0] += 300.0                                                     # mean >> spread in one row
    bias = torch.randn((B, C, 1), generator=g)
    gy = torch.randn(shape, generator=g)
    leaves64 = [t.double().requires_grad_(True) for t in (x, sc, sh, bias)]
    u = leaves64[1] * leaves64[0] + leaves64[2]
    z = F.instance_norm(u, eps=1e-5) + leaves64[3] if T > 1 else (u - u.mean(-1, keepdim=True)) / (u.var(-1, unbiased=False, keepdim=True) + 1e-5).sqrt() + leaves64[3]
    y64 = F.leaky_relu(z, 0.2)
    y64.backward(gy.double())
    leaves = [t.to(dev).requires_grad_(True) for t in (x, sc, sh, bias)]
    y = CG.film_norm_lrelu(*leaves, 1e-5, 0.2)
    y.backward(gy.to(dev))
    for name, got, want in [("y", y.detach(), y64.detach())] + [(n, a.grad, b.grad) for n, a, b in zip("x sc sh bias".split(), leaves, leaves64)]:
        err = (got.double().cpu() - want).abs().max().item()
        # (the row with mean 300: u itself carries float32 rounding of 3e-5, divided by a spread of ~1)
        assert err <= 1e-4 * max(want.abs().max().item(), 1.0), (name, err, want.abs().max().item())


@pytest.mark.gpu
def test_weight_norm_fold_all_layers_in_one_launch_vs_torch(dev):
    """w = g v / ||v|| (torch.nn.utils.weight_norm, fastsvc.py:354-362) for 60 layers of mixed shapes (more than one table of
    56), values and both gradients against float64 autograd."""
    g = torch.Generator().manual_seed(9)
    shapes = [(24, 1, 3), (24, 24, 3), (48, 24, 1), (192, 192, 1, 3), (1, 24, 3), (96, 48, 3), (384, 192, 3), (5, 7, 1)] * 8
    shapes = shapes[:60]
    vs = [torch.randn(s, generator=g) for s in shapes]
    gs = [torch.rand((s[0],) + (1,) * (len(s) - 1), generator=g) + 0.5 for s in shapes]
    grads = [torch.randn(s, generator=g) for s in shapes]
    v64 = [v.double().requires_grad_(True) for v in vs]
    g64 = [t.double().requires_grad_(True) for t in gs]
    w64 = [v * (gg / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))) for v, gg in zip(v64, g64)]
    torch.autograd.backward(w64, [d.double() for d in grads])
    vd = [v.to(dev).requires_grad_(True) for v in vs]
    gd = [t.to(dev).requires_grad_(True) for t in gs]
    wd = CG.weight_norm_fold(vd, gd)
    torch.autograd.backward(wd, [d.to(dev) for d in grads])
    for i in range(len(shapes)):
        for name, got, want in (("w", wd[i].detach(), w64[i].detach()), ("dv", vd[i].grad, v64[i].grad), ("dg", gd[i].grad, g64[i].grad)):
            assert got.shape == want.shape
            err = (got.double().cpu() - want).abs().max().item()
            assert err <= 2e-6 * max(want.abs().max().item(), 1e-3), (i, name, err)
