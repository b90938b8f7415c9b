"""HiFiGAN multi-scale + multi-period discriminator (harana/models/fastsvc.py:631-1143) - the discriminator BASELINE
config 5 names - against vectors of the LIVE reference (tests/golden/hifigan_disc.npz, `make_golden.py hifigan`):
the default-width module tree (key by key, 70.7 M parameters, incl. the reference's quirk that its scale discriminators
carry no weight / spectral norm, fastsvc.py:957-975), outputs / feature maps / adversarial losses / input gradient of a
reduced-width instance, and one `Trainer._train_step` against it (train_fastsvc.py:157-240)."""
import copy

import numpy as np
import pytest
import torch

from conftest import load_golden
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import autograd as AG
from svcc23_fastsvc_amd import synth as S
from svcc23_fastsvc_amd import training as TR
from test_training import _TorchGenerator


def _hash_signal(seed, stream, shape):
    n = int(np.prod(shape))
    return torch.from_numpy((0.3 * S.hash_normalish(seed, S.stream_id(stream), n)).reshape(shape).astype(np.float32))


def _small(seed):
    D = TR.HiFiGANMultiScaleMultiPeriodDiscriminator(**copy.deepcopy(S.HIFIGAN_FIXTURE_PARAMS))
    S.fill_module_from_hash(D, seed)
    return D.train()


def test_default_width_module_tree_matches_reference():
    g = load_golden("hifigan_disc.npz")
    D = TR.HiFiGANMultiScaleMultiPeriodDiscriminator()
    sd = D.state_dict()
    assert np.array_equal(S.key_hashes(sd.keys()), g["full/key_crc"])      # same keys in the same order: checkpoints load
    shapes = np.array([list(v.shape) + [0] * (4 - v.dim()) for v in sd.values()], dtype=np.int64)
    assert np.array_equal(shapes, g["full/shapes"])
    assert sum(p.numel() for p in D.parameters()) == int(g["full/numel"]) == 70711277      # 70.7 M
    # the reference's quirk: the scale discriminators hold plain weights (no weight_g / weight_orig), the period ones weight-norm
    assert all(k.endswith((".weight", ".bias")) for k in sd if k.startswith("msd."))
    assert all(k.endswith((".weight_g", ".weight_v", ".bias")) for k in sd if k.startswith("mpd."))
    # ... and the norm the flags ask for when the quirk is switched off: spectral norm on scale 0, weight norm on the others
    fixed = TR.HiFiGANMultiScaleDiscriminator(scales=2, follow_official_norm=True, reference_norm_quirk=False,
                                              discriminator_params=dict(channels=8, max_downsample_channels=16, max_groups=4))
    keys = list(fixed.state_dict())
    assert any(k.startswith("discriminators.0.") and k.endswith("weight_orig") for k in keys)
    assert any(k.startswith("discriminators.1.") and k.endswith("weight_g") for k in keys)
    assert not TR.discriminator_is_per_sample_stateless(fixed) and TR.discriminator_is_per_sample_stateless(D)


def test_outputs_feature_maps_losses_and_input_gradient_match_reference():
    g = load_golden("hifigan_disc.npz")
    seed_d, seed_x = int(g["meta"][0]), int(g["meta"][1])
    D = _small(seed_d)
    assert np.array_equal(S.key_hashes(D.state_dict().keys()), g["small/key_crc"])
    x = _hash_signal(seed_x, "hifigan.x", S.HIFIGAN_FIXTURE_INPUT)
    x_hat = _hash_signal(seed_x, "hifigan.x_hat", S.HIFIGAN_FIXTURE_INPUT).requires_grad_(True)
    outs, fmaps = D(x, return_fmaps=True)
    outs_hat = D(x_hat)
    assert len(outs) == 8                                      # three scales, then five periods
    for i in range(8):
        for tag, o in (("real", outs[i]), ("fake", outs_hat[i])):
            want = g[f"small/{tag}.{i}"]
            assert tuple(o.shape) == want.shape, (tag, i)
            assert np.abs(o.detach().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), (tag, i)
    assert [f.numel() for f in fmaps] == [int(v) for v in g["small/fmap_numel"]]
    got = np.array([float(f.detach().abs().mean()) for f in fmaps])
    assert np.abs(got - g["small/fmap_mean_abs"]).max() <= 1e-5 * max(1.0, g["small/fmap_mean_abs"].max())
    adv = TR.generator_adversarial_loss(outs_hat)
    adv.backward()
    assert abs(float(adv) - float(g["small/gen_adv"])) <= 1e-6 * max(1.0, float(g["small/gen_adv"]))
    want = g["small/d_gen_adv_d_x"]
    assert np.abs(x_hat.grad.numpy() - want).max() <= 1e-5 * np.abs(want).max()
    real, fake = TR.discriminator_adversarial_loss(outs_hat, outs)
    assert abs(float(real) - float(g["small/dis_real"])) <= 1e-6 and abs(float(fake) - float(g["small/dis_fake"])) <= 1e-6
    # one batch of [fake ; real] = two calls (what TrainStep's batched path relies on)
    both = D(torch.cat([x_hat.detach(), x], dim=0))
    for i in range(8):
        assert np.abs(both[i][:2].detach().numpy() - g[f"small/fake.{i}"]).max() <= 1e-5
        assert np.abs(both[i][2:].detach().numpy() - g[f"small/real.{i}"]).max() <= 1e-5


def _one_step(generator, params_module, dev, rel):
    g = load_golden("hifigan_disc.npz")
    _, _, seed_w, seed_b, seed_t, seed_d2, B, F = (int(v) for v in g["meta"])
    cfg = S.TINY_CONFIG
    T = F * cfg.hop
    D = _small(seed_d2).to(dev)
    step = TR.TrainStep(generator, D, dict(discriminator_train_start_steps=0), steps=1)
    b = S.synth_batch(cfg, B, F, seed_b)
    x = tuple(torch.from_numpy(a).to(dev) for a in (b.ppg, b.sine, b.lft, b.spk_emb))
    target = _hash_signal(seed_t, "train.target", (B, 1, T)).to(dev)
    before = {"g": {k: v.detach().cpu().clone() for k, v in params_module.state_dict().items()},
              "d": {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}}
    log = step.step((x, target))
    for k, v in log.items():
        want = float(g[f"step/loss/{k}"])
        assert abs(v - want) <= 2e-4 * max(1.0, abs(want)), (k, v, want)
    for tag, module in (("g", params_module), ("d", D)):
        for k, v in module.state_dict().items():
            v = v.detach().cpu()
            moved = float(g[f"step/{tag}/step_norm/{k}"])
            # the step each tensor took: same size (norm) as the reference's, and its leading elements where the reference put them
            got = float((v - before[tag][k]).double().norm())
            assert abs(got - moved) <= rel * moved + 1e-6, (tag, k, got, moved)
            head = v.flatten()[:4].numpy()
            scale = max(moved / max(1.0, v.numel() ** 0.5), 1e-7)
            assert np.abs(head - g[f"step/{tag}/head/{k}"]).max() <= 40 * rel * scale + 2e-6, (tag, k)


def _tiny_module(seed_w):
    cfg = S.TINY_CONFIG
    m = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels),
                           upsampling_scales=list(cfg.upsampling_scales), out_channels=cfg.out_channels,
                           spk_emb_size=cfg.spk_emb_size, use_spk_emb=cfg.use_spk_emb)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, seed_w).items()})
    return m.train()


def test_train_step_against_hifigan_discriminator_matches_reference_cpu():
    m = _tiny_module(int(load_golden("hifigan_disc.npz")["meta"][2]))
    _one_step(_TorchGenerator(m), m, torch.device("cpu"), rel=2e-2)


@pytest.mark.gpu
def test_train_step_against_hifigan_discriminator_matches_reference_hip():
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    m = _tiny_module(int(load_golden("hifigan_disc.npz")["meta"][2])).to(dev)
    _one_step(m, m, dev, rel=5e-2)
