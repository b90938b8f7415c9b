"""Training-mode entry of ``FastSVCGenerator`` (first slice).

The reference trainer calls the generator under autograd (``harana/bin/train_fastsvc.py:157-240``:
``y_ = self.model["generator"](*x)``, losses, ``gen_loss.backward()``, ``clip_grad_norm_``, optimizer step).
What exists here:

* **forward**: the HIP path (``Plan.forward`` through the C ABI), exactly as in inference;
* **backward**: PyTorch-ROCm autograd over a differentiable restatement of the SAME de-duplicated dataflow
  (``_forward_torch`` below; MIOpen convolutions), re-evaluated from the saved inputs and the module's own
  parameters - weight-norm included, so gradients reach ``weight_g`` / ``weight_v`` as the reference's
  optimizer expects (``fastsvc.py:354-362``).  The HIP output and the restatement agree to ~1e-5, so the
  gradient is that of the function the forward computed, to that accuracy.

This makes ``model.train()(x, s, l, emb)`` + ``.backward()`` work unchanged for the trainer; it is NOT a
hand-written backward (conv backward-data / -weight, InstanceNorm backward on the MFMA kernels are not built -
DESIGN.md §9), and it costs a PyTorch forward + backward per step.  It never runs in inference and is not a
fallback for it: without the HIP library the forward raises as everywhere else.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Optional

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2
IN_EPS = 1e-5
# GPU: the convolutions and the FiLM / InstanceNorm / LeakyReLU chains of the differentiated dataflow are HIP graph nodes
# (False: MIOpen / aten ops, for A/B runs and tests)
HIP_CONV_BACKWARD = True


def folded_weights(named_params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``<layer>.weight`` / ``.bias`` tensors from either state-dict layout, differentiably:
    legacy weight-norm ``w = g * v / ||v||`` (norm over all dims but 0; ``fastsvc.py:342-362``);
    Conv2d (Cout, Cin, 1, 3) kernels are viewed as Conv1d (Cout, Cin, 3)."""
    out: Dict[str, torch.Tensor] = {}
    bases = [k[: -len("_v")] for k in named_params if k.endswith(".weight_v")]
    if bases and HIP_CONV_BACKWARD and all(named_params[b + "_v"].is_cuda for b in bases):
        # all layers in ONE launch each way (conv_grad.weight_norm_fold) instead of ~8 small aten launches per layer and direction
        from .conv_grad import weight_norm_fold
        ws = weight_norm_fold([named_params[b + "_v"] for b in bases], [named_params[b + "_g"] for b in bases])
        fused = dict(zip(bases, ws))
    else:
        fused = {}
    for k, v in named_params.items():
        if k.endswith(".weight_v"):
            base = k[: -len("_v")]
            if base in fused:
                out[base] = fused[base]
                continue
            g = named_params[base + "_g"]
            norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
            out[base] = v * (g / norm)
        elif k.endswith(".weight_g"):
            continue
        else:
            out[k] = v
    for k in list(out):
        if out[k].dim() == 4:
            out[k] = out[k].squeeze(2)
    return out


def _conv(x, w, prefix, dilation=1):
    weight = w[prefix + ".weight"]
    k = weight.shape[-1]
    if x.is_cuda and HIP_CONV_BACKWARD:
        # the graph node with the hand-written backward kernels (conv_grad.py: backward data / weight / bias in HIP)
        from .conv_grad import conv1d
        return conv1d(x, weight, w[prefix + ".bias"], dilation)
    return F.conv1d(x, weight, w[prefix + ".bias"], padding=(k // 2) * dilation, dilation=dilation)


def _lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def _forward_torch(w: Dict[str, torch.Tensor], scales, x, s, l, spk_emb: Optional[torch.Tensor]):
    """The dataflow of DESIGN.md §2 in differentiable torch ops (each conditioning chain once, the 1x1 after
    the decimation, FiLM sums formed once per stage)."""
    n = len(scales)
    down_scales = [1] + [int(v) for v in reversed(list(scales)[1:])]
    cond = {}
    for name, sig in (("lft", l), ("sine", s)):
        h = sig
        outs = []
        for k in range(n):
            p = f"downsampling_{name}.{k}"
            hd = h[..., :: down_scales[k]][..., : h.shape[-1] // down_scales[k]]
            r = _conv(hd, w, f"{p}.residual_block.0")
            t = _conv(_lrelu(hd), w, f"{p}.downsample_block.2", 1)
            t = _conv(_lrelu(t), w, f"{p}.downsample_block.4", 2)
            h = _conv(_lrelu(t), w, f"{p}.downsample_block.6", 4) + r
            f = f"film_{name}.{k}"
            u = _lrelu(_conv(h, w, f"{f}.conv"))
            outs.append((_conv(u, w, f"{f}.conv_scale"), _conv(u, w, f"{f}.conv_shift")))
        cond[name] = outs
    y = x
    for i in range(n):
        k = n - 1 - i
        sc = cond["lft"][k][0] + cond["sine"][k][0]
        sh = cond["lft"][k][1] + cond["sine"][k][1]
        p = f"upsampling_nets.{i}"
        bias = None
        if spk_emb is not None:
            e = spk_emb / spk_emb.norm(dim=1, keepdim=True).clamp_min(1e-12)
            bias = F.linear(e, w[f"{p}.emb_projector.weight"], w[f"{p}.emb_projector.bias"]).unsqueeze(-1)

        def aff(t):
            t = sc * t + sh
            if bias is None:
                return t
            return F.instance_norm(t, eps=IN_EPS) + bias       # (one fused kernel each way instead of six reductions / elementwise ops)

        def aff_lrelu(t):
            if bias is not None and t.is_cuda and HIP_CONV_BACKWARD:
                # FiLM affine + InstanceNorm + speaker bias + LeakyReLU as ONE node, HIP forward and backward (conv_grad.py)
                from .conv_grad import film_norm_lrelu
                return film_norm_lrelu(t, sc, sh, bias, IN_EPS, LRELU_SLOPE)
            return _lrelu(aff(t))

        a = _conv(y, w, f"{p}.conv_first")
        st = int(scales[i])
        xr = _conv(torch.repeat_interleave(a, st, dim=-1), w, f"{p}.residual_block.1")
        t = _lrelu(_conv(torch.repeat_interleave(_lrelu(a), st, dim=-1), w, f"{p}.upsample_block0.2"))
        xm = _conv(aff_lrelu(t), w, f"{p}.conv_block1.1", 3) + xr
        t = _conv(aff_lrelu(xm), w, f"{p}.conv_block2.1", 9)
        y = _conv(aff_lrelu(t), w, f"{p}.conv_block3.1", 27) + xm
    return _conv(y, w, "conv_last")


class GeneratorFunction(torch.autograd.Function):
    """forward = HIP path; backward = autograd of ``_forward_torch`` at the saved inputs / parameters."""

    @staticmethod
    def forward(ctx, module, names, x, s, l, spk_emb, *params):
        with torch.no_grad():
            y = module._forward_device(x, s, l, spk_emb, lengths=None)
        ctx.module = module
        ctx.names = names
        ctx.has_emb = spk_emb is not None
        ctx.save_for_backward(x, s, l, *(() if spk_emb is None else (spk_emb,)), *params)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        saved = list(ctx.saved_tensors)
        x, s, l = saved[:3]
        emb = saved[3] if ctx.has_emb else None
        params = saved[4:] if ctx.has_emb else saved[3:]
        needs = ctx.needs_input_grad            # (module, names, x, s, l, spk_emb, *params)
        # bfloat16 activation storage: the FORWARD ran with bfloat16 tensors and bf16 MFMA products; the backward is a FLOAT32
        # recompute of the same dataflow - the HIP convolution / FiLM-norm nodes take and return float32 (conv_grad.py casts),
        # only the remaining F.linear / elementwise ops see the bf16 autocast below - so gradients are taken at a float32
        # re-evaluation of a forward that was rounded to bfloat16: the mismatch is the forward's own bf16 error (1e-2 of the
        # output's rms, test_workloads_gpu.py) carried into the loss gradient; tests/test_training.py holds the bfloat16 step to
        # the float32 step within that.  Gradients come back in the float32 of the master parameters.
        amp = (torch.autocast(device_type="cuda", dtype=torch.bfloat16)
               if getattr(ctx.module, "activation_storage", "float32") == "bfloat16" and x.is_cuda else contextlib.nullcontext())
        with torch.enable_grad(), amp:
            ins = [t.detach().requires_grad_(needs[2 + i]) for i, t in enumerate((x, s, l))]
            e = None if emb is None else emb.detach().requires_grad_(needs[5])
            leaves = [p.detach().requires_grad_(needs[6 + i]) for i, p in enumerate(params)]
            w = folded_weights(dict(zip(ctx.names, leaves)))
            y = _forward_torch(w, ctx.module.upsampling_scales, ins[0], ins[1], ins[2], e)
            wanted = [t for t in ins + ([e] if e is not None else []) + leaves if t.requires_grad]
            grads = torch.autograd.grad(y, wanted, grad_y.to(y.dtype), allow_unused=True) if wanted else []
        it = iter(grads)
        out = [None, None]
        for t in ins:
            out.append(next(it) if t.requires_grad else None)
        out.append(next(it) if (e is not None and e.requires_grad) else None)
        for t in leaves:
            out.append(next(it) if t.requires_grad else None)
        return tuple(out)


def forward_with_grad(module, x, s, l, spk_emb):
    names = tuple(n for n, _ in module.named_parameters())
    params = tuple(p for _, p in module.named_parameters())
    return GeneratorFunction.apply(module, names, x, s, l, spk_emb, *params)
