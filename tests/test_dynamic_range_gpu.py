"""Dynamic range of the float32-storage path (-m gpu).  Its k=3 convolutions multiply binary16 PIECES of the float32
operands (csrc/fastsvc_hx.hip); binary16 has an absolute floor (2^-24) and ceiling (65504) that float32 - what the
reference computes in - does not.  The kernels therefore scale weights (per output channel, packer) and activations
(per tensor and utterance, from the running max |value| the producing kernel records) by exact powers of two.
These tests move inputs and weights across the exponent range and hold the HIP path to the float64 oracle:

  * relative to the output's rms, <= 1e-3 (the north-star bar) AND no worse than 8x a float32 CPU run of the same
    dataflow (+ 2e-5): the scaled cases are ill-conditioned for float32 itself, and that is the honest yardstick;
  * every value finite (the unscaled split produced NaN for 4 of these cases - profiles/r3_range_probe_*.txt).
"""
import numpy as np
import pytest
import torch

import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S
from range_cases import CASES, build_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (and fail loudly without one)"
    A.load_library()
    return torch.device("cuda:0")


def _run(dev, cfg, sd, b, spk, **plan_kw):
    plan = A.Plan(cfg, **plan_kw)
    blob = plan.pack(sd).to(dev)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    y = plan.forward(blob, t(b.ppg), t(b.sine), t(b.lft), t(b.spk_emb if spk else None))
    torch.cuda.synchronize()
    return y.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("name", CASES)
def test_scaled_weights_and_inputs_stay_fp32_class(dev, name):
    from oracle import fastsvc_oracle as O
    cfg = S.FULL_CONFIG
    sd, b, spk = build_case(cfg, name)
    folded = S.fold_weight_norm(sd)
    emb = b.spk_emb if spk else None
    y64 = O.forward_dedup(folded, cfg.upsampling_scales, b.ppg, b.sine, b.lft, emb, dtype=torch.float64).numpy()
    y32 = O.forward_dedup(folded, cfg.upsampling_scales, b.ppg, b.sine, b.lft, emb, dtype=torch.float32).numpy()
    rms = float(np.sqrt((y64 ** 2).mean()))
    e_f32 = float(np.abs(y32.astype(np.float64) - y64).max()) / rms
    # both workspace layouts: the default one (separate launches, every tap inspectable) and the compact one the module
    # and bench.py run, whose conditioning stages 0 / 1 are whole-stage launches with their own bound-derived scales
    # (csrc/fastsvc_cond.hip: every LDS-resident tensor scaled from the measured input maximum through (l1, bmax))
    for compact in (False, True):
        y = _run(dev, cfg, sd, b, spk, compact_workspace=compact)
        assert np.isfinite(y).all(), (name, compact)
        e_hip = float(np.abs(y - y64).max()) / rms
        assert e_hip <= 1e-3, (name, compact, e_hip, e_f32)
        assert e_hip <= 8.0 * e_f32 + 2e-5, (name, compact, e_hip, e_f32)


def test_amax_rows_hold_the_tensors_maxima(dev):
    """The scale of a staged tensor comes from the amax row a producer filled: each row must equal the largest
    magnitude of the tensor in the workspace (an upper bound that is tight), per utterance and signal."""
    cfg = S.FULL_CONFIG
    sd = S.synth_state_dict(cfg, 31)
    B, F = 2, 24
    b = S.synth_batch(cfg, B, F, 32)
    plan = A.Plan(cfg)
    plan.keep_last_block_output(B, F)
    blob = plan.pack(sd).to(dev)
    ws = torch.zeros(plan.workspace_bytes(B, F), dtype=torch.uint8, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    plan.forward(blob, t(b.ppg), t(b.sine), t(b.lft), t(b.spk_emb), workspace=ws)
    torch.cuda.synchronize()
    amax_in = plan.tap("amax_in", B, F, ws).cpu().numpy().reshape(3, B, 256).max(axis=-1)   # an entry: 8 slots, 128 bytes apart
    assert np.array_equal(amax_in[0], np.abs(b.lft).reshape(B, -1).max(axis=1))
    assert np.array_equal(amax_in[1], np.abs(b.sine).reshape(B, -1).max(axis=1))
    assert np.array_equal(amax_in[2], np.abs(b.ppg).reshape(B, -1).max(axis=1))
    checked = 0
    n = cfg.n_stages
    for i in range(n):
        # the MEASURED tensors: each conditioning stage's output and the block outputs that feed another block;
        # everything else is bounded from these through the layers' (l1, bmax) or sits behind an InstanceNorm
        for name in (f"down_h.{i}",) + ((f"up.{i}.out",) if i + 1 < n else ()):
            x = plan.tap(name, B, F, ws).cpu().numpy()
            want = np.abs(x).reshape(x.shape[0], -1).max(axis=1)
            got = plan.tap("amax:" + name, B, F, ws).cpu().numpy().reshape(2 * B, 256).max(axis=-1)[: x.shape[0]]
            # (lanes past the end of a row contribute bias-sized values computed from zero padding: >=, and tight)
            assert np.all(got >= want) and np.all(got <= np.maximum(want * 1.5, want + 1.0)), (name, got, want)
            checked += 1
    assert checked == 2 * n - 1
