"""``FastSVCGenerator`` - host-side mirror of the reference ``nn.Module`` surface.

Same constructor kwargs, attributes, ``state_dict`` keys (reference checkpoints load with
``strict=True``), ``forward / inference / remove_weight_norm / apply_weight_norm`` as
``harana.models.FastSVCGenerator`` (``harana/models/fastsvc.py:235-383``), so the reference's
``harana/bin/decode_fastsvc.py`` (``:140-143,187-189``) and ``load_model``
(``harana/utils/utils.py:243-280``) call it unchanged.  The arithmetic is NOT PyTorch: ``forward``
hands raw device pointers to the gfx950 library through the C ABI (``include/fastsvc_hip.h``).
The sub-modules below only own parameters under the reference's names; they have no forward.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .engine import FastSVCError, Plan
from .synth import GeneratorConfig


class _ParamHolder(nn.Module):
    """Owns parameters under the reference's names; the math lives in the HIP library."""

    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise FastSVCError("parameter container only: call FastSVCGenerator.forward (HIP path)")


class _Slot(nn.Module):
    """Parameter-free placeholder that keeps nn.Sequential indices equal to the reference's
    (LeakyReLU / Stretch2d / Squeeze2d positions in fastsvc.py:56-78,164-178)."""

    def __init__(self, what: str):
        super().__init__()
        self.what = what

    def extra_repr(self) -> str:
        return self.what


def _conv1x3_2d(cin: int, cout: int, dilation: int) -> nn.Conv2d:
    # Conv2d1x3 (upsample.py:99-106): kernel (1,3), padding (0,d), dilation d, torch default init
    return nn.Conv2d(cin, cout, kernel_size=(1, 3), padding=(0, dilation), dilation=dilation)


def _conv1x3_1d(cin: int, cout: int, dilation: int) -> nn.Conv1d:
    # Conv1d1x3 (upsample.py:76-83)
    return nn.Conv1d(cin, cout, kernel_size=3, padding=dilation, dilation=dilation)


def _conv1x1(cin: int, cout: int) -> nn.Conv1d:
    # Conv1d1x1 (residual_block.py:27-48): kaiming-normal weight, zero bias
    m = nn.Conv1d(cin, cout, kernel_size=1)
    nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
    nn.init.constant_(m.bias, 0.0)
    return m


class _UpBlockParams(_ParamHolder):
    def __init__(self, cin: int, c: int, scale: int, spk_emb_size: int, use_spk_emb: bool):
        super().__init__()
        self.conv_first = _conv1x3_2d(cin, c, 1)
        self.upsample_block0 = nn.Sequential(_Slot("LeakyReLU(0.2)"), _Slot(f"Stretch x{scale}"),
                                             _conv1x3_2d(c, c, 1), _Slot("LeakyReLU(0.2)"))
        self.conv_block1 = nn.Sequential(_Slot("LeakyReLU(0.2)"), _conv1x3_2d(c, c, 3))
        self.conv_block2 = nn.Sequential(_Slot("LeakyReLU(0.2)"), _conv1x3_2d(c, c, 9))
        self.conv_block3 = nn.Sequential(_Slot("LeakyReLU(0.2)"), _conv1x3_2d(c, c, 27))
        self.residual_block = nn.Sequential(_Slot(f"Stretch x{scale}"), _conv1x3_2d(c, c, 1))
        self.instance_norm = _Slot("InstanceNorm over T (fused into the conv prologue)")
        if use_spk_emb:
            self.emb_projector = nn.Linear(spk_emb_size, c)


class _DownBlockParams(_ParamHolder):
    def __init__(self, cin: int, c: int, scale: int):
        super().__init__()
        self.residual_block = nn.Sequential(_conv1x1(cin, c), _Slot(f"Squeeze /{scale}"))
        self.downsample_block = nn.Sequential(
            _Slot(f"Squeeze /{scale}"), _Slot("LeakyReLU(0.2)"), _conv1x3_1d(cin, c, 1),
            _Slot("LeakyReLU(0.2)"), _conv1x3_1d(c, c, 2),
            _Slot("LeakyReLU(0.2)"), _conv1x3_1d(c, c, 4))


class _FiLMParams(_ParamHolder):
    def __init__(self, c: int):
        super().__init__()
        self.conv = _conv1x3_1d(c, c, 1)
        self.relu = _Slot("LeakyReLU(0.2)")
        self.conv_scale = _conv1x3_1d(c, c, 1)
        self.conv_shift = _conv1x3_1d(c, c, 1)


class FastSVCGenerator(nn.Module):
    """FastSVC waveform generator, MI355X-native forward.

    Args (identical to the reference, fastsvc.py:238-246):
        in_channels, mid_channels, upsampling_scales, out_channels, spk_emb_size, use_spk_emb.
    """

    # sub-batch so that one launch sequence never needs more scratch than this (bytes)
    max_workspace_bytes = 48 << 30
    # True: the first forward of every new (batch, frames) shape times the candidate launch shapes
    # of each layer on the device and keeps the fastest (the role cudnn.benchmark plays for the
    # reference, train_fastsvc.py:617).  False: static cost model.
    autotune = False
    # "float32" (the parity path) or "bfloat16": workspace tensors stored as bf16 (BASELINE config 3:
    # half the HBM traffic of the narrow layers, bf16-activation accuracy).  Set before the first forward.
    activation_storage = "float32"
    # True: never build an autograd graph (the plain HIP forward whatever the grad mode says) - for inference code
    # that does not wrap its calls in torch.no_grad()
    inference_only = False

    def __init__(self, in_channels: int = 144, mid_channels: Sequence[int] = (192, 96, 48, 24),
                 upsampling_scales: Sequence[int] = (2, 4, 4, 5), out_channels: int = 1,
                 spk_emb_size: int = 512, use_spk_emb: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.upsampling_scales = upsampling_scales      # kept as passed, never mutated
        self.mid_channels = mid_channels
        self.out_channels = out_channels
        self.use_spk_emb = use_spk_emb
        self._cfg = GeneratorConfig.from_kwargs(
            in_channels=in_channels, mid_channels=mid_channels, upsampling_scales=upsampling_scales,
            out_channels=out_channels, spk_emb_size=spk_emb_size, use_spk_emb=use_spk_emb)
        cfg = self._cfg

        self.upsampling_nets = nn.ModuleList()
        cin = in_channels
        for scale, c in zip(cfg.upsampling_scales, cfg.mid_channels):
            self.upsampling_nets.append(_UpBlockParams(cin, c, scale, spk_emb_size, use_spk_emb))
            cin = c
        lft, sine = [], []
        cin = 1
        for scale, c in zip(cfg.down_scales, cfg.down_channels):
            lft.append(_DownBlockParams(cin, c, scale))
            sine.append(_DownBlockParams(cin, c, scale))
            cin = c
        self.downsampling_lft = nn.Sequential(*lft)
        self.downsampling_sine = nn.Sequential(*sine)
        self.film_lft = nn.ModuleList(_FiLMParams(c) for c in cfg.down_channels)
        self.film_sine = nn.ModuleList(_FiLMParams(c) for c in cfg.down_channels)
        self.conv_last = _conv1x1(cfg.mid_channels[-1], out_channels)
        self.apply_weight_norm()

        self._tuned_shapes = set()
        self._plan: Optional[Plan] = None
        self._blob: Optional[torch.Tensor] = None
        self._blob_key = None

    # ------------------------------------------------------------------ weight norm
    def apply_weight_norm(self):
        """Legacy weight-norm on every Conv1d / Conv2d (fastsvc.py:354-362); not on Linear."""
        def _apply(m):
            if isinstance(m, (nn.Conv1d, nn.Conv2d)) and not hasattr(m, "weight_g"):
                torch.nn.utils.weight_norm(m)
        self.apply(_apply)
        if getattr(self, "_blob", None) is not None:
            self.invalidate_packed_weights()

    def remove_weight_norm(self):
        """Fold g * v / ||v|| back into ``.weight`` (fastsvc.py:342-352)."""
        def _remove(m):
            try:
                torch.nn.utils.remove_weight_norm(m)
            except ValueError:
                return
        self.apply(_remove)
        self.invalidate_packed_weights()

    # ------------------------------------------------------------------ packed-weight cache
    # The kernels read a device-resident, kernel-layout copy of the parameters.  It is rebuilt when a
    # parameter object or its autograd version counter changes (optimizer steps, `p.copy_()`,
    # `load_state_dict`, weight-norm removal), and dropped explicitly by everything that re-homes or
    # replaces parameters (`_apply`: .to()/.half()/.cuda(); `load_state_dict`; weight-norm changes).
    # LIMITATION: writes through `p.data` (`p.data.copy_()`, `p.data.mul_()`, EMA swaps done that
    # way) bump no version counter; call `invalidate_packed_weights()` after them, or set
    # `FastSVCGenerator.checksum_weights = True` to fingerprint the parameter storage on every
    # forward (one device reduction per parameter: safe, slower).
    checksum_weights = False

    def invalidate_packed_weights(self):
        """Forget the packed device blob; the next forward folds and packs the parameters again."""
        self._blob = None
        self._blob_key = None
        self._prefetch = None

    def _weights_key(self, device):
        key = (str(device),) + tuple((id(p), p._version) for p in self.parameters())
        if self.checksum_weights:
            with torch.no_grad():
                key += tuple(float(p.detach().double().sum()) + float(p.detach().double().abs().sum())
                             for p in self.parameters())
        return key

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, "_blob", None) is not None:
            self.invalidate_packed_weights()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed_weights()
        return out

    # the plan / blob hold a ctypes handle and a device cache: never copied or pickled, always rebuilt
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_plan"] = None
        state["_blob"] = None
        state["_blob_key"] = None
        state["_prefetch"] = None
        state["_tuned_shapes"] = set()
        return state

    def __deepcopy__(self, memo):
        import copy
        # legacy weight-norm leaves a NON-LEAF `weight` attribute (g * v / ||v||, recomputed by its
        # pre-forward hook) on every conv, which torch refuses to deep-copy: detach it first - the HIP
        # path folds weight_g / weight_v itself and never reads that attribute
        for m in self.modules():
            w = m.__dict__.get("weight")
            if isinstance(w, torch.Tensor) and w.grad_fn is not None:
                m.__dict__["weight"] = w.detach()
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def packed_weights(self, device) -> torch.Tensor:
        """Device-resident kernel-layout weight blob; rebuilt when any parameter changed."""
        if self._plan is None:
            self._plan = Plan(self._cfg, storage=self.activation_storage, compact_workspace=True)
        key = self._weights_key(device)
        if self._blob is None or self._blob_key != key:
            on_gpu = torch.device(device).type != "cpu"
            pre = getattr(self, "_prefetch", None)
            pre = pre[1] if (pre is not None and pre[0] == key) else None
            host = self._plan.pack(self.state_dict(), reuse_pinned=on_gpu, prefetched=pre)   # staging buffer: uploaded right here
            self._blob = host.to(device)
            self._blob_key = key
        self._prefetch = None
        return self._blob

    def prefetch_packed_weights(self):
        """Call right after an optimizer update: starts the asynchronous device-to-host copy of the parameters that the next
        forward's re-pack needs (``Plan.pack_prefetch``), so that the host packs while the GPU works on whatever is
        enqueued in between (the train step puts the discriminator's real-batch forward there) instead of draining the
        stream at the next forward."""
        p = next(self.parameters())
        if not p.is_cuda:
            return
        if self._plan is None:
            self._plan = Plan(self._cfg, storage=self.activation_storage, compact_workspace=True)
        key = self._weights_key(p.device)
        if self._blob is not None and self._blob_key == key:
            return
        self._prefetch = (key, self._plan.pack_prefetch(self.state_dict()))

    def load_packed_weights(self, blob: torch.Tensor):
        """Adopt an already packed device blob (e.g. received by RCCL broadcast)."""
        if self._plan is None:
            self._plan = Plan(self._cfg, storage=self.activation_storage, compact_workspace=True)
        if blob.numel() * blob.element_size() != self._plan.blob_bytes:
            raise ValueError("packed blob has the wrong size for this configuration")
        self._blob = blob
        self._blob_key = self._weights_key(blob.device)

    @property
    def plan(self) -> Plan:
        if self._plan is None:
            self._plan = Plan(self._cfg, storage=self.activation_storage, compact_workspace=True)
        return self._plan

    # ------------------------------------------------------------------ forward
    def forward(self, x, s, l, spk_emb=None, *, lengths=None, out=None):
        """x (B, in_channels, F) PPG - s (B, 1, T) sine - l (B, 1, T) loudness -
        spk_emb (B, spk_emb_size) or None  ->  (B, out_channels, T), T = F * prod(scales).
        Same contract as fastsvc.py:305-332 (raw conv_last output, no tanh).

        Extension (keyword only, not in the reference): ``lengths`` = per-utterance frame counts of
        a padded ragged batch; utterance b is computed as if run alone with lengths[b] frames and
        the padding of the output is zero.  ``out`` (inference only): a contiguous float32 (B, out_channels, T)
        tensor the waveform is written into (e.g. a collective's send buffer).

        Autograd: whenever grad mode is enabled and a parameter or an input requires grad, the output carries a graph
        (``train()`` and ``eval()`` alike, as for any ``nn.Module``).  The plain HIP forward runs under
        ``torch.no_grad()`` / ``torch.inference_mode()``, with ``self.inference_only = True``, or - in ``eval()`` - when an
        inference extension (``lengths``, ``out=``) is used."""
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise FastSVCError("FastSVCGenerator (HIP) needs GPU tensors; there is no CPU fallback "
                               "(the CPU oracle lives in oracle/ and is test infrastructure only)")
        hop = self._cfg.hop
        if s.shape[-1] != l.shape[-1] or s.shape[-1] != x.shape[-1] * hop:
            raise ValueError(f"length mismatch: {x.shape[-1]} frames x hop {hop} vs sine "
                             f"{s.shape[-1]} / loudness {l.shape[-1]} samples")
        if spk_emb is not None and not self.use_spk_emb:
            raise ValueError("spk_emb given but the generator was built with use_spk_emb=False")
        # nn.Module semantics (and the reference's): eval() does not detach - with grad enabled and parameters that
        # require grad the output carries a graph in either mode, so that a loss computed through a discriminator in
        # eval() still reaches the generator's parameters.  Only the inference extensions (`lengths`, `out=`) and the
        # explicit `inference_only` switch run the plain HIP forward under an enabled grad mode.
        needs_grad = torch.is_grad_enabled() and not self.inference_only and (
            any(p.requires_grad for p in self.parameters()) or
            any(isinstance(t, torch.Tensor) and t.requires_grad for t in (x, s, l, spk_emb)))
        if needs_grad and not self.training and (lengths is not None or out is not None):
            needs_grad = False
        if needs_grad:
            # training (train_fastsvc.py:157-240 calls the module under autograd): HIP forward, PyTorch-ROCm
            # autograd backward over a restatement of the same dataflow - see autograd.py (first slice)
            if lengths is not None:
                raise NotImplementedError("ragged batches (`lengths`) are an inference extension: no backward")
            if out is not None:
                raise ValueError("`out=` is an inference extension: no backward through it")
            from .autograd import forward_with_grad
            return forward_with_grad(self, x, s, l, spk_emb)
        return self._forward_device(x, s, l, spk_emb, lengths, out)

    def _forward_device(self, x, s, l, spk_emb, lengths, out=None):
        """The HIP forward proper (no autograd): packed weights, workspace sub-batching, C-ABI call."""
        hop = self._cfg.hop
        blob = self.packed_weights(x.device)
        plan = self.plan
        B, _, F = x.shape
        step = B
        Fw = plan.padded_frames(F)                 # (frame counts are run padded to a multiple of 4, as a ragged batch)
        while step > 1 and plan.workspace_bytes(step, Fw) > self.max_workspace_bytes:
            step = (step + 1) // 2
        if step == B:
            tune = bool(self.autotune) and lengths is None and (B, F, str(x.device)) not in self._tuned_shapes
            y = plan.forward(blob, x, s, l, spk_emb, autotune=tune, lengths=lengths, out=out)
            if tune:
                self._tuned_shapes.add((B, F, str(x.device)))
        else:
            y = out if out is not None else torch.empty((B, self.out_channels, F * hop), dtype=torch.float32, device=x.device)
            ws = torch.empty(plan.workspace_bytes(step, Fw), dtype=torch.uint8, device=x.device)
            for b0 in range(0, B, step):
                b1 = min(B, b0 + step)
                plan.forward(blob, x[b0:b1], s[b0:b1], l[b0:b1],
                             None if spk_emb is None else spk_emb[b0:b1], out=y[b0:b1], workspace=ws,
                             lengths=None if lengths is None else
                             (lengths[b0:b1] if isinstance(lengths, torch.Tensor) else list(lengths)[b0:b1]))
        # (`out=` given: the caller's float32 buffer IS the result - no cast copy behind its back)
        return y.to(x.dtype) if (x.dtype != torch.float32 and out is None) else y

    def inference(self, x, f0, l, signal_generator, pad_fn, spk_emb=None):
        """Time-major single-utterance entry used by decode_fastsvc.py:187-189
        (fastsvc.py:364-383): x (F, C), f0 (F, 1), l (T, 1) -> (T, out_channels)."""
        x = pad_fn(x.transpose(1, 0).unsqueeze(0))
        l = l.transpose(1, 0).unsqueeze(0)
        f0 = f0.transpose(1, 0).unsqueeze(0)
        s = signal_generator(f0)
        return self.forward(x, s, l, spk_emb).squeeze(0).transpose(1, 0)


def install_into_harana() -> None:
    """Make ``getattr(harana.models, "FastSVCGenerator")`` resolve to this class
    (``train_fastsvc.py:700-704``, ``utils.py:266-275`` look it up by that name).  If the
    reference package is importable it is patched; otherwise a minimal namespace is created."""
    import importlib
    import sys
    import types
    try:
        models = importlib.import_module("harana.models")
    except Exception:
        pkg = sys.modules.get("harana") or types.ModuleType("harana")
        pkg.__path__ = getattr(pkg, "__path__", [])
        models = types.ModuleType("harana.models")
        pkg.models = models
        sys.modules["harana"] = pkg
        sys.modules["harana.models"] = models
    models.FastSVCGenerator = FastSVCGenerator
