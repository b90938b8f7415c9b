"""Dynamic-range probe of the split-binary16 path (VERDICT r2 "what's weak" #2): weights and inputs scaled
by powers of two, per-channel weight_g spread, speaker-less path with large FiLM scales.  Prints, per case, the HIP
path's error and a float32 CPU torch reference's error, both against the float64 oracle, relative to the output's
rms.  (tests/test_dynamic_range_gpu.py holds the same cases as assertions.)

    python tools/range_probe.py [--hx0]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svcc23_fastsvc_amd as A                      # noqa: E402
from svcc23_fastsvc_amd import synth as S           # noqa: E402
from oracle import fastsvc_oracle as O              # noqa: E402
from tests.range_cases import CASES, build_case     # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = S.FULL_CONFIG
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    worst = 0.0
    for name in CASES:
        if only and not any(o in name for o in only):
            continue
        sd, b, spk = build_case(cfg, name)
        folded = S.fold_weight_norm(sd)
        emb = b.spk_emb if spk else None
        y64 = O.forward_dedup(folded, cfg.upsampling_scales, b.ppg, b.sine, b.lft, emb, dtype=torch.float64).numpy()
        y32 = O.forward_dedup(folded, cfg.upsampling_scales, b.ppg, b.sine, b.lft, emb, dtype=torch.float32).numpy()
        plan = A.Plan(cfg)
        blob = plan.pack(sd).to(dev)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        y = plan.forward(blob, t(b.ppg), t(b.sine), t(b.lft), t(emb)).cpu().numpy().astype(np.float64)
        rms = float(np.sqrt((y64 ** 2).mean())) + 1e-300
        e_hip = float(np.abs(y - y64).max()) / rms
        e_f32 = float(np.abs(y32 - y64).max()) / rms
        worst = max(worst, e_hip)
        flag = "" if (e_hip <= 1e-3 and np.isfinite(y).all()) else "   <-- FAIL"
        print(f"{name:34s} rms {rms:10.3e}  hip/rms {e_hip:9.2e}  torch-f32/rms {e_f32:9.2e}{flag}", flush=True)
    print(f"worst {worst:.2e}")


if __name__ == "__main__":
    main()
