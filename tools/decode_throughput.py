"""Throughput of the decode harness (svcc23_fastsvc_amd.decode.decode_utterances) from HOST-resident features - what
`python -m svcc23_fastsvc_amd.decode` does after reading the dumps: 512 utterances of 2 - 10 s (SURVEY 8d's variant of
cfg4), time-major numpy features, F0 shift on, waveforms back as numpy arrays.   python tools/decode_throughput.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import svcc23_fastsvc_amd as A
from svcc23_fastsvc_amd import synth as S, decode as Dc

dev = torch.device("cuda:0")
cfg = S.FULL_CONFIG
frames = S.workload_frames("cfg4var")
n = int(os.environ.get("DECODE_UTTS", "512"))
frames = frames[:n]
rng = np.random.default_rng(3)
feats = []
for f in frames:
    f0 = np.where(rng.random((f, 1)) < 0.3, 0.0, rng.uniform(80, 400, (f, 1)))
    feats.append({"ppg": rng.standard_normal((f, cfg.in_channels), dtype=np.float32), "f0": f0,
                  "lft": rng.uniform(-9, 1, (f * cfg.hop, 1)).astype(np.float32)})
m = A.FastSVCGenerator(in_channels=cfg.in_channels, mid_channels=list(cfg.mid_channels), upsampling_scales=list(cfg.upsampling_scales),
                       out_channels=cfg.out_channels, spk_emb_size=cfg.spk_emb_size, use_spk_emb=True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synth_state_dict(cfg, 1).items()})
m.remove_weight_norm()
m = m.eval().to(dev)
sg = A.SignalGenerator(sample_rate=24000, hop_size=cfg.hop, noise_amp=0.0, signal_types=["sine"])
emb = rng.standard_normal(cfg.spk_emb_size).astype(np.float32)
src = [[5.0, 1.0]] * len(feats)
fns = {"decode_utterances": Dc.decode_utterances}
try:
    from svcc23_fastsvc_amd import _decode_old
    fns["one batch at a time (before)"] = _decode_old.decode_utterances
except ImportError:
    pass
samples = sum(frames) * cfg.hop
for name, fn in fns.items():
    ys = fn(m, feats, sg, dev, trg_emb=emb, src_f0_stats=src, trg_f0_stats=[5.3, 1.0], max_batch=64)   # warm (packs, loads)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ys = fn(m, feats, sg, dev, trg_emb=emb, src_f0_stats=src, trg_f0_stats=[5.3, 1.0], max_batch=64)
    dt = (time.perf_counter() - t0) / 3
    assert all(y.shape == (f * cfg.hop,) and np.isfinite(y).all() for y, f in zip(ys, frames))
    print(f"{name}: {dt * 1e3:.1f} ms per pass over {len(feats)} utterances = {samples / dt / 1e6:.1f} M samples/s "
          f"({samples / 24000 / dt:.0f} x real time)")
