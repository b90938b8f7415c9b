"""Training-mode slice: gradients against the LIVE reference's autograd.

`tests/golden/tiny_grads.npz` holds d sum(r * y) / d (every parameter, every input) of the reference generator in
train() mode (weight-norm parametrisation), produced by `tests/golden/make_golden.py grads`.
CPU: the differentiable restatement the backward is built on (`svcc23_fastsvc_amd/autograd.py`) reproduces them.
GPU: `model.train()(x, s, l, emb)` - HIP forward through the C ABI - followed by `.backward()` reproduces them."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import autograd as AG
from svcc23_fastsvc_amd import synth as S

REL = 1e-3          # the bar VERDICT r1 set for this slice: 1e-3 relative


def _module(cfg, seed):
    g = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels),
                           upsampling_scales=list(cfg.upsampling_scales), out_channels=cfg.out_channels,
                           spk_emb_size=cfg.spk_emb_size, use_spk_emb=cfg.use_spk_emb)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, seed).items()}, strict=True)
    return g


def _check(gold, tag, grads_in, named_grads):
    """max-abs error of every gradient relative to that tensor's own maximum - floored at 1e-3 of the largest
    parameter gradient, because some gradients are mathematically ZERO (weight_v of the C_in = 1 1x1 convs:
    w = g * v / |v| does not depend on |v|) and hold only rounding noise on both sides."""
    worst = 0.0
    gmax = max(float(np.abs(gold[k]).max()) for k in gold.files if k.startswith(f"{tag}/p/"))
    for name, got in list(grads_in.items()) + list(named_grads.items()):
        key = f"{tag}/in/{name}" if name in ("ppg", "sine", "lft", "spk_emb") else f"{tag}/p/{name}"
        if key not in gold.files:
            assert got is None or float(got.abs().max()) == 0.0, key
            continue
        want = torch.from_numpy(gold[key])
        assert got is not None, key
        err = float((got.cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-3 * gmax)
        worst = max(worst, err)
        assert err <= REL, (key, err)
    return worst


@pytest.mark.parametrize("tag", ["spk", "nospk"])
def test_restated_backward_matches_reference_gradients_cpu(tag):
    gold = load_golden("tiny_grads.npz")
    cfg = S.TINY_CONFIG
    seed_w, seed_x, B, F = (int(v) for v in gold["meta"])
    g = _module(cfg, seed_w)
    b = S.synth_batch(cfg, B, F, seed_x)
    ins = [torch.from_numpy(a).clone().requires_grad_(True) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
    params = dict(g.named_parameters())
    y = AG._forward_torch(AG.folded_weights(params), cfg.upsampling_scales, ins[0], ins[1], ins[2],
                          ins[3] if tag == "spk" else None)
    assert float((y.detach() - torch.from_numpy(gold[f"{tag}/y"])).abs().max()) <= 1e-4
    (y * torch.from_numpy(gold["r"])).sum().backward()
    _check(gold, tag, dict(zip(("ppg", "sine", "lft", "spk_emb"), (t.grad for t in ins))),
           {n: p.grad for n, p in params.items()})
    # both state-dict layouts: after remove_weight_norm the same function, gradients on `.weight`
    g.remove_weight_norm()
    w2 = AG.folded_weights(dict(g.named_parameters()))
    y2 = AG._forward_torch(w2, cfg.upsampling_scales, ins[0].detach(), ins[1].detach(), ins[2].detach(),
                           ins[3].detach() if tag == "spk" else None)
    assert float((y2 - y).abs().max()) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["spk", "nospk"])
def test_train_mode_hip_forward_and_backward_match_reference_gradients(tag):
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    gold = load_golden("tiny_grads.npz")
    cfg = S.TINY_CONFIG
    seed_w, seed_x, B, F = (int(v) for v in gold["meta"])
    g = _module(cfg, seed_w).to(dev).train()
    b = S.synth_batch(cfg, B, F, seed_x)
    ins = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
    y = g(ins[0], ins[1], ins[2], ins[3] if tag == "spk" else None)          # HIP forward, graph recorded
    assert y.requires_grad and float((y.detach().cpu() - torch.from_numpy(gold[f"{tag}/y"])).abs().max()) <= 1e-4
    (y * torch.from_numpy(gold["r"]).to(dev)).sum().backward()
    worst = _check(gold, tag, dict(zip(("ppg", "sine", "lft", "spk_emb"), (t.grad for t in ins))),
                   {n: p.grad for n, p in g.named_parameters()})
    assert worst <= REL
    # the trainer's next steps work on it (train_fastsvc.py:202-205): clip + optimizer step, then a new forward
    # sees the updated weights (packed-weight cache invalidated by the parameters' version counters)
    torch.nn.utils.clip_grad_norm_(g.parameters(), 10.0)
    opt = torch.optim.SGD(g.parameters(), lr=1e-2)
    opt.step()
    with torch.no_grad():
        y2 = g(ins[0].detach(), ins[1].detach(), ins[2].detach(), ins[3].detach() if tag == "spk" else None)
    assert float((y2 - y.detach()).abs().max()) > 1e-6


@pytest.mark.gpu
def test_train_step_shape_at_the_recipe_batch():
    """BASELINE configs[4]'s generator leg at the recipe's batch (fastsvc.yaml: batch 32, 100-frame crops),
    yaml-width generator: one forward + backward; gradients finite and non-zero for every parameter."""
    dev = torch.device("cuda:0")
    cfg = S.FULL_CONFIG
    g = _module(cfg, 5).to(dev).train()
    B, F = 32, 100
    ppg, sine, lft, emb = S.device_batch(cfg, B, F, 6, dev)
    y = g(ppg, sine, lft, emb)
    assert tuple(y.shape) == (B, 1, F * cfg.hop)
    y.square().mean().backward()
    for n, p in g.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0.0, n


@pytest.mark.gpu
def test_eval_mode_keeps_the_graph_like_any_module():
    """nn.Module semantics: with grad enabled and parameters that require grad the output carries a graph in
    eval() as in train() - a generator loss computed through a discriminator in eval() must reach the generator's
    parameters; no_grad, `inference_only` and the inference extensions run the plain HIP forward."""
    dev = torch.device("cuda:0")
    cfg = S.TINY_CONFIG
    g = _module(cfg, 3).to(dev).eval()
    b = S.synth_batch(cfg, 2, 8, 4)
    ins = [torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb)]
    y = g(*ins)
    assert y.requires_grad
    y.square().mean().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in g.parameters())
    with torch.no_grad():
        assert not g(*ins).requires_grad
    assert not g(*ins, lengths=[8, 5]).requires_grad
    g.inference_only = True
    assert not g(*ins).requires_grad
