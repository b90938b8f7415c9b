"""Deterministic synthetic weights and inputs for the FastSVC generator path.

There is no network for checkpoints or datasets, so parity tests, goldens and ``bench.py`` all use
weights and features regenerated on every machine from a counter-based integer hash (splitmix64
finaliser).  Only integer arithmetic and exact power-of-two scalings are used up to the final
float32 cast, so the tensors are bit-identical in the build container (where the goldens are made
from the live reference) and on the GPU box.

Layer table and tensor shapes follow the reference generator
(``harana/models/fastsvc.py:238-303``; state-dict keys SURVEY.md §8(b)); feature distributions
follow SURVEY.md §8(d) (PPG standard-scaled, ``preprocess_fastsvc.py:41-75``; sine excitation by the
``SignalGenerator.sinusoid`` formula ``harana/utils/features.py:177-197``).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_u64(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit hashes of counters offset..offset+n-1 under (seed, stream)."""
    with np.errstate(over="ignore"):
        key = _mix64(np.array([np.uint64(seed & 0xFFFFFFFFFFFFFFFF)], dtype=np.uint64))
        key = _mix64(key ^ np.uint64(stream & 0xFFFFFFFFFFFFFFFF))
        ctr = np.arange(offset, offset + n, dtype=np.uint64)
        return _mix64(_mix64(ctr + key[0]) ^ key[0])


def hash_uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """float64 uniforms in [0, 1): top 53 bits scaled by 2**-53 (exact)."""
    return (hash_u64(seed, stream, n, offset) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def hash_normalish(seed: int, stream: int, n: int) -> np.ndarray:
    """Zero-mean unit-variance bell-shaped values: Irwin-Hall sum of four uniforms (no libm)."""
    u = hash_uniform(seed, stream, 4 * n).reshape(4, n)
    return (u.sum(axis=0) - 2.0) * np.sqrt(3.0)


def stream_id(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


# --------------------------------------------------------------------------------------------
# Generator configuration and layer table
# --------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class GeneratorConfig:
    """Constructor kwargs of ``FastSVCGenerator`` (fastsvc.py:238-246; yaml fastsvc.yaml:23-29)."""

    in_channels: int = 144
    mid_channels: Tuple[int, ...] = (192, 96, 48, 24)
    upsampling_scales: Tuple[int, ...] = (2, 4, 4, 5)
    out_channels: int = 1
    spk_emb_size: int = 512
    use_spk_emb: bool = True

    @staticmethod
    def from_kwargs(**kw) -> "GeneratorConfig":
        kw = dict(kw)
        if "mid_channels" in kw:
            kw["mid_channels"] = tuple(int(c) for c in kw["mid_channels"])
        if "upsampling_scales" in kw:
            kw["upsampling_scales"] = tuple(int(s) for s in kw["upsampling_scales"])
        return GeneratorConfig(**kw)

    @property
    def n_stages(self) -> int:
        return len(self.mid_channels)

    @property
    def hop(self) -> int:
        return int(np.prod(self.upsampling_scales))

    @property
    def down_scales(self) -> Tuple[int, ...]:
        """Down-sampling chain scales, fastsvc.py:270-272: reversed up-scales, last dropped, 1 first."""
        rev = list(self.upsampling_scales[::-1])
        rev.pop()
        return tuple([1] + rev)

    @property
    def down_channels(self) -> Tuple[int, ...]:
        return tuple(self.mid_channels[::-1])


TINY_CONFIG = GeneratorConfig(
    in_channels=8, mid_channels=(16, 8, 8, 4), upsampling_scales=(2, 4, 4, 5), out_channels=1,
    spk_emb_size=16, use_spk_emb=True,
)
FULL_CONFIG = GeneratorConfig()


@dataclass(frozen=True)
class LayerSpec:
    """One parameterised layer of the generator, by its state-dict prefix."""

    name: str                 # state-dict prefix, e.g. "upsampling_nets.0.conv_first"
    kind: str                 # "conv2d1x3" | "conv1d" | "linear"
    cout: int
    cin: int
    ksize: int                # 3 or 1 (0 for linear)
    dilation: int = 1
    kaiming_normal: bool = False   # Conv1d1x1 init (residual_block.py:32-37) vs torch default
    weight_norm: bool = True       # apply_weight_norm covers Conv1d/Conv2d only (fastsvc.py:354-362)

    @property
    def weight_shape(self) -> Tuple[int, ...]:
        if self.kind == "conv2d1x3":
            return (self.cout, self.cin, 1, 3)
        if self.kind == "conv1d":
            return (self.cout, self.cin, self.ksize)
        return (self.cout, self.cin)

    @property
    def g_shape(self) -> Tuple[int, ...]:
        return (self.cout,) + (1,) * (len(self.weight_shape) - 1)


def layer_table(cfg: GeneratorConfig) -> List[LayerSpec]:
    """All parameterised layers in reference ``state_dict`` order."""
    layers: List[LayerSpec] = []
    cin = cfg.in_channels
    for i, c in enumerate(cfg.mid_channels):
        p = f"upsampling_nets.{i}"
        layers += [
            LayerSpec(f"{p}.conv_first", "conv2d1x3", c, cin, 3, 1),
            LayerSpec(f"{p}.upsample_block0.2", "conv2d1x3", c, c, 3, 1),
            LayerSpec(f"{p}.conv_block1.1", "conv2d1x3", c, c, 3, 3),
            LayerSpec(f"{p}.conv_block2.1", "conv2d1x3", c, c, 3, 9),
            LayerSpec(f"{p}.conv_block3.1", "conv2d1x3", c, c, 3, 27),
            LayerSpec(f"{p}.residual_block.1", "conv2d1x3", c, c, 3, 1),
        ]
        if cfg.use_spk_emb:
            layers.append(LayerSpec(f"{p}.emb_projector", "linear", c, cfg.spk_emb_size, 0,
                                    weight_norm=False))
        cin = c
    for sig in ("lft", "sine"):
        cin = 1
        for k, c in enumerate(cfg.down_channels):
            p = f"downsampling_{sig}.{k}"
            layers += [
                LayerSpec(f"{p}.residual_block.0", "conv1d", c, cin, 1, 1, kaiming_normal=True),
                LayerSpec(f"{p}.downsample_block.2", "conv1d", c, cin, 3, 1),
                LayerSpec(f"{p}.downsample_block.4", "conv1d", c, c, 3, 2),
                LayerSpec(f"{p}.downsample_block.6", "conv1d", c, c, 3, 4),
            ]
            cin = c
    # ModuleList registration order in the reference: film_lft then film_sine (fastsvc.py:290-299)
    for sig in ("lft", "sine"):
        for k, c in enumerate(cfg.down_channels):
            p = f"film_{sig}.{k}"
            layers += [
                LayerSpec(f"{p}.conv", "conv1d", c, c, 3, 1),
                LayerSpec(f"{p}.conv_scale", "conv1d", c, c, 3, 1),
                LayerSpec(f"{p}.conv_shift", "conv1d", c, c, 3, 1),
            ]
    layers.append(LayerSpec("conv_last", "conv1d", cfg.out_channels, cfg.mid_channels[-1], 1, 1,
                            kaiming_normal=True))
    return layers


def state_dict_keys(cfg: GeneratorConfig, weight_norm: bool = True) -> List[str]:
    keys: List[str] = []
    for L in layer_table(cfg):
        if L.kind == "linear":
            keys += [f"{L.name}.weight", f"{L.name}.bias"]
        elif weight_norm:
            keys += [f"{L.name}.bias", f"{L.name}.weight_g", f"{L.name}.weight_v"]
        else:
            # remove_weight_norm re-registers .weight after .bias
            keys += [f"{L.name}.bias", f"{L.name}.weight"]
    return keys


def synth_state_dict(cfg: GeneratorConfig, seed: int, weight_norm: bool = True,
                     perturb_g: bool = True) -> Dict[str, np.ndarray]:
    """Deterministic float32 parameters with the reference's key names, shapes and init scales.

    Conv/Linear default init is U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias; the 1x1
    convs use N(0, 2/fan_in) with zero bias (``harana/layers/residual_block.py:32-37``) - here with
    small non-zero biases so that bias handling is exercised.  With ``weight_norm`` the tensors are
    ``weight_v`` plus ``weight_g = ||v|| * (0.75 + 0.5 u)`` (``perturb_g``) so the fold is
    non-trivial.
    """
    out: Dict[str, np.ndarray] = {}
    for L in layer_table(cfg):
        sid = stream_id(L.name)
        shape = L.weight_shape
        n = int(np.prod(shape))
        fan_in = L.cin * max(L.ksize, 1)
        if L.kaiming_normal:
            w = hash_normalish(seed, sid, n) * np.sqrt(2.0 / fan_in)
            b = (hash_uniform(seed, sid + 1, L.cout) - 0.5) * 0.1
        else:
            bound = 1.0 / np.sqrt(fan_in)
            w = (hash_uniform(seed, sid, n) * 2.0 - 1.0) * bound
            b = (hash_uniform(seed, sid + 1, L.cout) * 2.0 - 1.0) * bound
        w = w.reshape(shape).astype(np.float32)
        b = b.astype(np.float32)
        if L.kind == "linear" or not (weight_norm and L.weight_norm):
            out[f"{L.name}.weight"] = w
            out[f"{L.name}.bias"] = b
        else:
            norm = np.sqrt((w.astype(np.float64) ** 2).reshape(L.cout, -1).sum(axis=1))
            g = norm
            if perturb_g:
                g = norm * (0.75 + 0.5 * hash_uniform(seed, sid + 2, L.cout))
            out[f"{L.name}.bias"] = b
            out[f"{L.name}.weight_g"] = g.reshape(L.g_shape).astype(np.float32)
            out[f"{L.name}.weight_v"] = w
    return out


def fold_weight_norm(sd: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """``w = g * v / ||v||`` per output channel (legacy ``weight_norm`` dim=0).

    float32 arithmetic in the same order as torch's ``_weight_norm`` (norm over all but dim 0).
    Returns a dict with ``.weight``/``.bias`` keys only (post-``remove_weight_norm`` layout).
    """
    out: Dict[str, np.ndarray] = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            base = k[: -len(".weight_v")]
            g = np.asarray(sd[base + ".weight_g"], dtype=np.float32)
            vv = np.asarray(v, dtype=np.float32)
            norm = np.sqrt((vv.astype(np.float32) ** 2).reshape(vv.shape[0], -1)
                           .sum(axis=1, dtype=np.float32)).reshape(g.shape)
            out[base + ".weight"] = (vv * (g / norm)).astype(np.float32)
        else:
            out[k] = np.asarray(v, dtype=np.float32)
    return out


# --------------------------------------------------------------------------------------------
# Synthetic features
# --------------------------------------------------------------------------------------------
@dataclass
class SynthBatch:
    ppg: np.ndarray       # (B, in_channels, F)  float32
    sine: np.ndarray      # (B, 1, T)            float32,  T = hop * F
    lft: np.ndarray       # (B, 1, T)            float32
    spk_emb: np.ndarray   # (B, spk_emb_size)    float32
    f0: np.ndarray = field(default=None)  # (B, 1, F) float32 frame-level F0 in Hz (0 = unvoiced)


def synth_f0(B: int, F: int, seed: int) -> np.ndarray:
    """Log random walk (sigma 0.02 / frame) around 220 Hz, clamped to [80, 600], ~30 % unvoiced
    in 10-frame runs; zeros mark unvoiced frames as WORLD harvest does."""
    step = hash_normalish(seed, stream_id("f0.walk"), B * F).reshape(B, F) * 0.02
    start = (hash_uniform(seed, stream_id("f0.start"), B) - 0.5) * 0.6
    logf0 = np.log(220.0) + start[:, None] + np.cumsum(step, axis=1)
    f0 = np.clip(np.exp(logf0), 80.0, 600.0)
    nrun = (F + 9) // 10
    uv = hash_uniform(seed, stream_id("f0.uv"), B * nrun).reshape(B, nrun) < 0.3
    uv = np.repeat(uv, 10, axis=1)[:, :F]
    f0[uv] = 0.0
    return f0.astype(np.float32).reshape(B, 1, F)


def sine_from_f0(f0: np.ndarray, hop: int, sample_rate: int, seed: int,
                 sine_amp: float = 0.1, noise_amp: float = 0.003) -> np.ndarray:
    """Sine excitation by the reference formula (features.py:188-197): nearest up-sampling by
    ``hop``, ``rad = (f0 / sr) % 1``, ``sine_amp * vuv * sin(2 pi cumsum(rad))`` plus noise of
    amplitude ``noise_amp`` voiced / ``noise_amp / 3`` unvoiced.  Generated once on the host and
    fed to every implementation under test (it is an *input* of the parity boundary)."""
    B, _, F = f0.shape
    f0u = np.repeat(f0.astype(np.float64), hop, axis=2)
    vuv = (f0u > 0).astype(np.float64)
    rad = (f0u / float(sample_rate)) % 1.0
    sine = vuv * np.sin(np.cumsum(rad, axis=2) * 2.0 * np.pi) * sine_amp
    if noise_amp > 0:
        T = F * hop
        amp = vuv * noise_amp + (1.0 - vuv) * noise_amp / 3.0
        noise = hash_normalish(seed, stream_id("sine.noise"), B * T).reshape(B, 1, T)
        sine = sine + noise * amp
    return sine.astype(np.float32)


def synth_batch(cfg: GeneratorConfig, B: int, F: int, seed: int,
                sample_rate: int = 24000) -> SynthBatch:
    hop = cfg.hop
    T = hop * F
    ppg = hash_normalish(seed, stream_id("ppg"), B * cfg.in_channels * F)
    ppg = ppg.reshape(B, cfg.in_channels, F).astype(np.float32)
    f0 = synth_f0(B, F, seed)
    sine = sine_from_f0(f0, hop, sample_rate, seed)
    nblk = (T + 63) // 64
    lft = hash_uniform(seed, stream_id("lft"), B * nblk).reshape(B, 1, nblk) * 10.0 - 9.0
    lft = np.repeat(lft, 64, axis=2)[:, :, :T].astype(np.float32)
    emb = (hash_normalish(seed, stream_id("spk_emb"), B * cfg.spk_emb_size) * 5.0)
    emb = emb.reshape(B, cfg.spk_emb_size).astype(np.float32)
    return SynthBatch(ppg=np.ascontiguousarray(ppg), sine=np.ascontiguousarray(sine),
                      lft=np.ascontiguousarray(lft), spk_emb=np.ascontiguousarray(emb), f0=f0)


# BASELINE.json configs as (B, F) at 24 kHz (150 PPG frames per second)
WORKLOADS = {
    "cfg1": dict(B=1, F=300, seed=1235, desc="1 x 2 s utterance"),
    "cfg2": dict(B=8, F=600, seed=1236, desc="8 x 4 s utterances"),
    "cfg3": dict(B=64, F=1500, seed=1237, desc="64 x 10 s utterances"),
    "cfg4": dict(B=512, F=1500, seed=1238, desc="512 x 10 s utterances (sharded)"),
    # SURVEY 8(d)'s optional variant of cfg4: lengths uniform in 2 - 10 s, run as ragged length-bucketed batches
    "cfg4var": dict(B=512, F=1500, Fmin=300, seed=1239, desc="512 utterances of 2 - 10 s (uniform), ragged batches (sharded)"),
}


def workload_frames(name: str):
    """Frame count of every utterance of a sharded workload (deterministic)."""
    wl = WORKLOADS[name]
    if "Fmin" not in wl:
        return [wl["F"]] * wl["B"]
    rng = np.random.default_rng(wl["seed"])
    return [int(v) for v in rng.integers(wl["Fmin"], wl["F"] + 1, size=wl["B"])]


def device_batch(cfg: GeneratorConfig, B: int, F: int, seed: int, device, sample_rate: int = 24000):
    """The same kind of synthetic utterances as `synth_batch` (SURVEY.md §8 d), generated with torch
    ops ON `device` - for the large workloads (cfg3 / cfg4: tens of GB of features) where the
    integer-hash host generator would take minutes.  Not bit-identical to `synth_batch` (torch's
    generator, not the hash streams): use it for timing and for self-consistency properties only;
    parity fixtures always come from `synth_batch`.  Returns (ppg, sine, lft, spk_emb) tensors."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    hop = cfg.hop
    T = hop * F
    ppg = torch.randn((B, cfg.in_channels, F), generator=g, device=device, dtype=torch.float32)
    step = torch.randn((B, F), generator=g, device=device, dtype=torch.float64) * 0.02
    start = (torch.rand((B, 1), generator=g, device=device, dtype=torch.float64) - 0.5) * 0.6
    f0 = torch.exp(np.log(220.0) + start + torch.cumsum(step, dim=1)).clamp(80.0, 600.0)
    nrun = (F + 9) // 10
    uv = torch.rand((B, nrun), generator=g, device=device) < 0.3
    f0 = torch.where(uv.repeat_interleave(10, dim=1)[:, :F], torch.zeros_like(f0), f0)
    f0u = f0.repeat_interleave(hop, dim=1)
    vuv = (f0u > 0).to(torch.float64)
    phase = torch.cumsum((f0u / float(sample_rate)) % 1.0, dim=1)
    sine = vuv * torch.sin(phase * (2.0 * np.pi)) * 0.1
    noise = torch.randn((B, T), generator=g, device=device, dtype=torch.float32)
    sine = (sine + noise.to(torch.float64) * (vuv * 0.003 + (1.0 - vuv) * 0.001)).to(torch.float32).unsqueeze(1)
    nblk = (T + 63) // 64
    lft = (torch.rand((B, 1, nblk), generator=g, device=device, dtype=torch.float32) * 10.0 - 9.0)
    lft = lft.repeat_interleave(64, dim=2)[:, :, :T].contiguous()
    emb = torch.randn((B, cfg.spk_emb_size), generator=g, device=device, dtype=torch.float32) * 5.0
    return ppg.contiguous(), sine.contiguous(), lft, emb


def fill_module_from_hash(module, seed: int) -> None:
    """Deterministic, library-independent parameter values for ANY torch module, by parameter NAME (a discriminator of the
    recipe has 4.35 M parameters: too large for a fixture, and framework initialisers are not reproducible across
    versions): weight_g = 1 + 0.1 n, biases 0.01 n, everything else 0.1 n, n = hash_normalish of the name's stream.
    Used on both sides of the recipe-size training golden (tests/golden/make_golden.py train_recipe: the reference's
    discriminator; tests/test_training.py: this package's)."""
    import torch
    with torch.no_grad():
        for name, p in module.named_parameters():
            n = hash_normalish(seed, stream_id("fill." + name), p.numel()).reshape(tuple(p.shape))
            if name.endswith("weight_g"):
                v = 1.0 + 0.1 * n
            elif name.endswith("bias"):
                v = 0.01 * n
            else:
                v = 0.1 * n
            p.copy_(torch.from_numpy(v.astype(np.float32)))


# ---------------------------------------------------------------------------
# HiFiGAN discriminator fixture configuration (tests/golden/hifigan_disc.npz): reduced widths - the default widths hold
# 70.7 M parameters - chosen so that every structural feature is exercised: three scales with pooling, grouped strided
# k = 41 convolutions whose group count grows to its cap, five periods with reflect padding (the length below is a multiple
# of none of them), hidden widths that reach their caps (the period stack's output conv reads the capped width).
# ---------------------------------------------------------------------------
HIFIGAN_FIXTURE_PARAMS = dict(
    scales=3,
    scale_downsample_pooling="AvgPool1d",
    scale_downsample_pooling_params=dict(kernel_size=4, stride=2, padding=2),
    scale_discriminator_params=dict(in_channels=1, out_channels=1, kernel_sizes=[15, 41, 5, 3], channels=16,
                                    max_downsample_channels=64, max_groups=16, bias=True, downsample_scales=[2, 2, 4, 4, 1],
                                    nonlinear_activation="LeakyReLU", nonlinear_activation_params=dict(negative_slope=0.1)),
    follow_official_norm=True,
    periods=[2, 3, 5, 7, 11],
    period_discriminator_params=dict(in_channels=1, out_channels=1, kernel_sizes=[5, 3], channels=4,
                                     downsample_scales=[3, 3, 3, 3, 1], max_downsample_channels=16, bias=True,
                                     nonlinear_activation="LeakyReLU", nonlinear_activation_params=dict(negative_slope=0.1),
                                     use_weight_norm=True, use_spectral_norm=False),
)
HIFIGAN_FIXTURE_INPUT = (2, 1, 4001)       # (B, 1, T)


def key_hashes(keys) -> np.ndarray:
    """int64 CRC32 of every state-dict key, in order: module-tree equality as integer data."""
    import zlib
    return np.array([zlib.crc32(k.encode()) for k in keys], dtype=np.int64)


# ---------------------------------------------------------------------------
# Multi-resolution STFT loss test signals (tests/golden/stft_loss.npz: the expected values come from the reference)
# ---------------------------------------------------------------------------
def stft_loss_cases(recipe_params: Dict) -> List[Tuple[str, np.ndarray, np.ndarray, Dict]]:
    """(tag, x, y, constructor kwargs): predicted / target waveforms (B, T) float32 from the integer-hash generator.
    ``recipe``: the yaml's six resolutions; ``default``: the class defaults (window shorter than the frame, hops that do
    not divide it, odd length); ``floor``: a silent and a very quiet prediction, a target with a silent stretch (bins
    at the 1e-7 power floor, where clamp passes no gradient)."""
    def noise(seed, B, T, amp):
        return (hash_normalish(seed, 7, B * T).reshape(B, T) * amp).astype(np.float32)

    def tone(B, T, f):
        t = np.arange(T, dtype=np.float64)[None, :]
        return (0.4 * np.sin(2 * np.pi * f * (1.0 + 0.1 * np.arange(B)[:, None]) * t / 24000.0)).astype(np.float32)

    recipe = dict(recipe_params)
    default = dict(fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50], win_lengths=[600, 1200, 240], window="hann_window")
    cases = []
    y = tone(2, 4000, 220.0) + noise(31, 2, 4000, 0.05)
    cases.append(("recipe", (y + noise(32, 2, 4000, 0.1)).astype(np.float32), y, recipe))
    y = tone(2, 3001, 330.0) + noise(33, 2, 3001, 0.1)
    cases.append(("default", (0.8 * y + noise(34, 2, 3001, 0.05)).astype(np.float32), y, default))
    y = tone(3, 2500, 440.0) + noise(35, 3, 2500, 0.02)
    y[1, 700:1900] = 0.0
    x = (y + noise(36, 3, 2500, 0.1)).astype(np.float32)
    x[0] = 0.0
    x[1] = noise(37, 1, 2500, 1e-5)[0]
    cases.append(("floor", x, y.astype(np.float32), recipe))
    return cases
