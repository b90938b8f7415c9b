"""The grouped strided convolutions of the recipe's discriminator (csrc/fastsvc_gconv.hip; MelGANDiscriminator's
`Conv1d(c, min(4c, 512), 41, stride=4, padding=20, groups=c // 4)` + LeakyReLU, harana/models/fastsvc.py:386-520): forward,
backward data and backward weight / bias against the stock operators in float64, on the recipe's three layer shapes (incl. row
lengths that are not a multiple of 4), without the activation, and inside the discriminator module (same outputs and
parameter gradients with the kernels on and off)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import svcc23_fastsvc_amd  # noqa: F401
from svcc23_fastsvc_amd import gconv as GC
from svcc23_fastsvc_amd import training as TR


def test_module_tree_and_cpu_route_are_the_stock_operators():
    conv = GC.GroupedConv1d(16, 64, kernel_size=41, stride=4, padding=20, groups=4)
    seq = GC.ConvAct(conv, torch.nn.LeakyReLU(0.2))
    assert list(seq.state_dict()) == ["0.weight", "0.bias"]
    x = torch.randn(2, 16, 203)
    want = F.leaky_relu(F.conv1d(x, conv.weight, conv.bias, stride=4, padding=20, groups=4), 0.2)
    assert torch.equal(seq(x), want) and torch.equal(conv(x), F.conv1d(x, conv.weight, conv.bias, stride=4, padding=20, groups=4))
    torch.nn.utils.weight_norm(conv)                        # the reference applies weight-norm to every conv: hooks must still run
    assert torch.allclose(seq(x), F.leaky_relu(F.conv1d(x, conv.weight, conv.bias, stride=4, padding=20, groups=4), 0.2))
    d = TR.MelGANMultiScaleDiscriminator(**TR.RECIPE["discriminator_params"])
    assert sum(isinstance(m, GC.GroupedConv1d) for m in d.modules()) == 9


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 16, 64, 4, 1003), (2, 64, 256, 16, 1000), (2, 256, 512, 64, 250), (2, 4, 16, 1, 403),
                                   (1, 16, 64, 4, 41)])
@pytest.mark.parametrize("slope", [0.2, 1.0])
def test_forward_and_gradients_match_float64(shape, slope):
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    B, Cin, Cout, G, T = shape
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn((B, Cin, T), generator=g)
    w = torch.randn((Cout, Cin // G, 41), generator=g) * 0.1
    b = torch.randn((Cout,), generator=g) * 0.1
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    pre = F.conv1d(xd, wd, bd, stride=4, padding=20, groups=G)
    yd = F.leaky_relu(pre, slope) if slope != 1.0 else pre
    r = torch.randn(yd.shape, generator=g, dtype=torch.float64)
    (yd * r).sum().backward()
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = GC._GroupedConvActFn.apply(xg, wg, bg, G, 4, 20, slope)
    assert tuple(y.shape) == tuple(yd.shape)
    (y * r.to(dev).float()).sum().backward()
    for name, got, want in (("y", y, yd), ("dx", xg.grad, xd.grad), ("dw", wg.grad, wd.grad), ("db", bg.grad, bd.grad)):
        err = float((got.detach().cpu().double() - want.detach()).abs().max())
        assert err <= 2e-5 * max(1.0, float(want.detach().abs().max())), (name, err)
    # the weight gradient is a fixed-order sum: bit-identical when run again
    xg.grad = wg.grad = bg.grad = None
    y2 = GC._GroupedConvActFn.apply(xg, wg, bg, G, 4, 20, slope)
    first = None
    for _ in range(2):
        xg.grad = wg.grad = bg.grad = None
        y2 = GC._GroupedConvActFn.apply(xg, wg, bg, G, 4, 20, slope)
        (y2 * r.to(dev).float()).sum().backward()
        cur = (wg.grad.clone(), bg.grad.clone(), xg.grad.clone())
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, c) for a, c in zip(first, cur))


@pytest.mark.gpu
def test_discriminator_same_outputs_and_gradients_with_the_kernels_on_and_off():
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    d = TR.MelGANMultiScaleDiscriminator(**TR.RECIPE["discriminator_params"]).to(dev)
    x = (torch.randn((4, 1, 16000), device=dev) * 0.3).requires_grad_(True)
    res = {}
    for on in (True, False):
        GC.USE_HIP = on
        try:
            d.zero_grad()
            x.grad = None
            before = dict(GC.ROUTES)
            outs = d(x)
            # (with the switch on, the nine grouped k = 41 layers must really take the HIP node - were they declined,
            # both runs would be the stock operators and the comparison below would pass on nothing)
            took = {k: GC.ROUTES[k] - before[k] for k in before}
            assert took == ({"hip": 9, "stock": 0} if on else {"hip": 0, "stock": 9}), (on, took)
            loss = sum(((o[-1] - 1.0) ** 2).mean() for o in outs) + sum(f.abs().mean() for o in outs for f in o[:-1])
            loss.backward()
            res[on] = (float(loss), [o[-1].detach().clone() for o in outs], x.grad.clone(),
                       {k: p.grad.clone() for k, p in d.named_parameters()})
        finally:
            GC.USE_HIP = True
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * max(1.0, abs(res[False][0]))
    for a, c in zip(res[True][1], res[False][1]):
        assert float((a - c).abs().max()) <= 1e-5 * max(1.0, float(c.abs().max()))
    assert float((res[True][2] - res[False][2]).abs().max()) <= 1e-4 * float(res[False][2].abs().max())
    for k, gr in res[False][3].items():
        assert float((res[True][3][k] - gr).abs().max()) <= 1e-4 * max(1e-6, float(gr.abs().max())), k


@pytest.mark.gpu
def test_slopes_outside_zero_one_run_the_stock_pair():
    """The fused backward recovers LeakyReLU' from the sign of the ACTIVATED output: right for slopes in (0, 1] only.
    A negative slope or slope 0 must take the stock operators - and give their gradients."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    conv = GC.GroupedConv1d(16, 64, kernel_size=41, stride=4, padding=20, groups=4).to(dev)
    assert GC.supported(conv)
    x = (torch.randn((2, 16, 1024), device=dev) * 0.5).requires_grad_(True)
    for slope in (-0.3, 0.0, 0.2):
        before = dict(GC.ROUTES)
        x.grad = None
        conv.zero_grad()
        y = conv(x, act_slope=slope)
        y.square().sum().backward()
        took = {k: GC.ROUTES[k] - before[k] for k in before}
        assert took == ({"hip": 1, "stock": 0} if 0.0 < slope <= 1.0 else {"hip": 0, "stock": 1}), (slope, took)
        gx, gw = x.grad.clone(), conv.weight.grad.clone()
        x.grad = None
        conv.zero_grad()
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv1d(x, conv.weight, conv.bias, stride=4, padding=20, groups=4), slope)
        ref.square().sum().backward()
        assert float((y - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
        assert float((gx - x.grad).abs().max()) <= 1e-4 * max(1e-6, float(x.grad.abs().max())), slope
        assert float((gw - conv.weight.grad).abs().max()) <= 1e-4 * max(1e-6, float(conv.weight.grad.abs().max())), slope
