"""Decode harness - the caller on the other side of the generator boundary.

Mirrors what ``harana/bin/decode_fastsvc.py:120-206`` does per utterance - F0 mean shift
(``harana/utils/features.py:41-108``, std forced to 1 at ``decode_fastsvc.py:165,176``), sine
excitation, ``inference()``, PCM-16 wav - but batches the utterances: they are grouped by similar
length, zero-padded, and run as ragged batches (``lengths``), with the excitation synthesised on the
device.  Each utterance's waveform equals what the reference's one-at-a-time loop produces
(``tests/golden/decode_chain.npz`` is made by that loop on the live reference).

Feature containers follow the reference's dump layout (``audio_feats_dataset.py:30-34``), time-major:
``f0 (F, 1)``, ``ppg (F, C)``, ``lft (T, 1)``, optionally ``spk_emb``; ``.npz`` files with those keys
are read here (``.h5`` as well when h5py is importable - it is not a dependency).

    python -m svcc23_fastsvc_amd.decode --dumpdir feats/ --checkpoint ckpt.pkl --config conf.yaml \\
        --outdir wav/ --spk-emb embs.npz --srcf0stats src_stats/ --trgf0stats trg_stats/
"""
from __future__ import annotations

import argparse
import glob
import os
import time
import wave
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .distributed import bucket_ragged


class F0Statistics:
    """Same surface as ``harana.utils.features.F0Statistics`` (features.py:41-108)."""

    def estimate(self, f0list: Sequence[np.ndarray]) -> np.ndarray:
        """[mean, std] of log F0 over the voiced (non-zero) frames of all sequences."""
        logs = [np.log(np.asarray(f0)[np.nonzero(f0)]) for f0 in f0list]
        f0s = np.concatenate(logs) if logs else np.zeros(0)
        return np.array([np.mean(f0s), np.std(f0s)])

    def convert(self, f0: np.ndarray, orgf0stats: Sequence[float], tarf0stats: Sequence[float]) -> np.ndarray:
        """Gaussian-normalised log-F0 transform of the voiced frames; unvoiced frames stay 0."""
        f0 = np.asarray(f0)
        cvf0 = np.zeros(len(f0))
        voiced = f0 > 0
        cvf0[voiced] = np.exp((tarf0stats[1] / orgf0stats[1]) * (np.log(f0[voiced]) - orgf0stats[0]) + tarf0stats[0])
        return cvf0


def convert_f0_device(f0: torch.Tensor, orgf0stats: Sequence[float], tarf0stats: Sequence[float]) -> torch.Tensor:
    """``F0Statistics.convert`` on a device tensor of any shape (float64 inside, like numpy)."""
    f = f0.to(torch.float64)
    voiced = f > 0
    safe = torch.where(voiced, f, torch.ones_like(f))
    cv = torch.exp((float(tarf0stats[1]) / float(orgf0stats[1])) * (torch.log(safe) - float(orgf0stats[0]))
                   + float(tarf0stats[0]))
    return torch.where(voiced, cv, torch.zeros_like(cv)).to(torch.float32)


def to_pcm16(y) -> np.ndarray:
    """float waveform -> int16: round-to-nearest of y * 32767 (libsndfile's float normalisation, which
    is what ``sf.write(..., "PCM_16")`` at decode_fastsvc.py:195-200 applies), saturated instead of
    wrapped when |y| > 1."""
    if isinstance(y, torch.Tensor):
        y = y.detach().to("cpu", torch.float32).numpy()
    return np.clip(np.rint(np.asarray(y, dtype=np.float64).reshape(-1) * 32767.0), -32768, 32767).astype(np.int16)


def write_wav(path: str, y, sample_rate: int) -> None:
    pcm = to_pcm16(y)
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(pcm.tobytes())


def load_features(path: str) -> Dict[str, np.ndarray]:
    """One utterance's dump: {"f0": (F,1), "ppg": (F,C), "lft": (T,1)[, "spk_emb", "wave"]}."""
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    if path.endswith(".h5"):
        try:
            import h5py                     # optional; the reference's own container
        except ImportError as e:            # pragma: no cover - not installed in the build image
            raise RuntimeError("reading .h5 dumps needs h5py; convert to .npz with the same keys") from e
        with h5py.File(path, "r") as f:     # pragma: no cover
            return {k: f[k][()] for k in f.keys()}
    raise ValueError(f"unsupported feature container: {path}")


class _PinnedSet:
    """One batch worth of page-locked staging (inputs up, waveforms down), grown on demand and reused."""

    def __init__(self):
        self.bufs: Dict[str, torch.Tensor] = {}

    def get(self, key: str, shape) -> torch.Tensor:
        need = int(np.prod(shape))
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.empty(max(need, 1), dtype=torch.float32, pin_memory=True)
            self.bufs[key] = buf
        return buf[:need].view(*shape)


@torch.no_grad()
def decode_utterances(model, feats: Sequence[Dict[str, np.ndarray]], signal_generator, device,
                      trg_emb=None, src_f0_stats: Optional[Sequence[Sequence[float]]] = None,
                      trg_f0_stats: Optional[Sequence[float]] = None, max_batch: int = 32,
                      pad_tolerance: float = 0.125) -> List[np.ndarray]:
    """Waveforms (float32, (T,)) for every utterance, in order.

    model             ``FastSVCGenerator`` (eval, weight-norm removed or not) on ``device``
    feats             per utterance: time-major ``f0 (F,1)``, ``ppg (F,C)``, ``lft (T,1)``
    signal_generator  ``svcc23_fastsvc_amd.SignalGenerator`` (device excitation)
    trg_emb           target-speaker embedding (E,) / (1,E), or None
    src_f0_stats      per utterance [mean, std] of its SOURCE speaker's log F0, or None for no shift
    trg_f0_stats      [mean, std] of the target speaker (the reference forces both stds to 1)

    A three-stage pipeline over the length-bucketed batches, so that the GPU is the only thing the pass waits for:
    while batch k computes, the host assembles batch k+1 in page-locked memory (numpy row copies, F0 shift) and
    uploads it on a copy stream, and takes batch k-1's waveforms out of the page-locked buffer they were downloaded
    to.  Two staging sets each way; one at a time it was host-bound 5:1 (tools/decode_throughput.py)."""
    hop = signal_generator.hop_size
    frames = [int(np.asarray(u["ppg"]).shape[0]) for u in feats]
    out: List[Optional[np.ndarray]] = [None] * len(feats)
    if not feats:
        return []
    device = torch.device(device)
    emb_row = None
    if trg_emb is not None:
        emb_row = torch.as_tensor(np.asarray(trg_emb), dtype=torch.float32).reshape(1, -1).to(device)
    batches = list(bucket_ragged(range(len(feats)), frames, max_batch, pad_tolerance))
    cuda = device.type == "cuda"
    up_sets, down_sets = [_PinnedSet(), _PinnedSet()], [_PinnedSet(), _PinnedSet()]
    copy_stream = torch.cuda.Stream(device) if cuda else None
    up_free = [None, None]                    # event: the upload that last read this input set is done
    staged = {}

    def stage(k: int) -> None:
        chunk = batches[k]
        fmax, B = frames[chunk[0]], len(chunk)
        C = int(np.asarray(feats[chunk[0]]["ppg"]).shape[1])
        slot = k & 1
        if cuda and up_free[slot] is not None:
            up_free[slot].synchronize()
        shapes = {"ppg": (B, C, fmax), "f0": (B, 1, fmax), "lft": (B, 1, fmax * hop)}
        if cuda:
            host = {key: up_sets[slot].get(key, shp) for key, shp in shapes.items()}
        else:
            host = {key: torch.empty(shp, dtype=torch.float32) for key, shp in shapes.items()}
        hp, hf, hl = (host[key].numpy() for key in ("ppg", "f0", "lft"))
        for j, i in enumerate(chunk):
            u, n = feats[i], frames[i]
            hp[j, :, :n] = np.asarray(u["ppg"], dtype=np.float32).T
            hp[j, :, n:] = 0
            f = np.asarray(u["f0"], dtype=np.float64).reshape(-1)
            if src_f0_stats is not None and trg_f0_stats is not None:
                f = F0Statistics().convert(f, src_f0_stats[i], trg_f0_stats)
            hf[j, 0, :n] = f
            hf[j, 0, n:] = 0
            hl[j, 0, : n * hop] = np.asarray(u["lft"], dtype=np.float32).reshape(-1)[: n * hop]
            hl[j, 0, n * hop:] = 0
        if not cuda:
            staged[k] = (host["ppg"], host["f0"], host["lft"], None)
            return
        compute = torch.cuda.current_stream(device)
        with torch.cuda.stream(copy_stream):
            dev = {key: host[key].to(device, non_blocking=True) for key in shapes}
            ready = torch.cuda.Event()
            ready.record(copy_stream)
        for t in dev.values():
            t.record_stream(compute)            # allocated on the copy stream's pool, consumed on the compute stream
        up_free[slot] = ready
        staged[k] = (dev["ppg"], dev["f0"], dev["lft"], ready)

    pending = {}

    def compute(k: int) -> None:
        chunk = batches[k]
        ppg, f0, lft, ready = staged.pop(k)
        if ready is not None:
            torch.cuda.current_stream(device).wait_event(ready)
        sine = signal_generator(f0)
        emb = None if emb_row is None else emb_row.expand(len(chunk), -1).contiguous()
        y = model(ppg, sine, lft, emb, lengths=[frames[i] for i in chunk]).to(torch.float32)
        if cuda:
            host_y = down_sets[k & 1].get("y", tuple(y.shape))
            host_y.copy_(y, non_blocking=True)          # download queued behind the forward on the compute stream
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(device))
            pending[k] = (host_y, done)
        else:
            pending[k] = (y, None)

    def finish(k: int) -> None:
        host_y, done = pending.pop(k)
        if done is not None:
            done.synchronize()
        y = host_y.numpy()
        for j, i in enumerate(batches[k]):
            out[i] = y[j].reshape(-1)[: frames[i] * hop].copy()

    stage(0)
    for k in range(len(batches)):
        compute(k)                               # asynchronous: the host goes on while the GPU runs batch k
        if k + 1 < len(batches):
            stage(k + 1)
        if k >= 1:
            finish(k - 1)                        # (before compute(k + 1) reuses that download set)
    finish(len(batches) - 1)
    return out  # type: ignore[return-value]


def _read_f0_mean(stats_dir: str, name: str) -> np.ndarray:
    import yaml
    with open(os.path.join(stats_dir, f"{name}.yml")) as f:
        y = yaml.safe_load(f)
    return np.array([float(y["stats"]["mean"]), 1.0])       # std forced to 1 (decode_fastsvc.py:165,176)


def main(argv=None) -> None:                                  # pragma: no cover - exercised on a GPU box
    import yaml
    from . import FastSVCGenerator, SignalGenerator
    ap = argparse.ArgumentParser(description="Batched FastSVC decoding (cf. harana-decode-fastsvc)")
    ap.add_argument("--dumpdir", required=True, help="directory of per-utterance feature dumps (.npz / .h5)")
    ap.add_argument("--checkpoint", required=True, help="reference checkpoint (.pkl with ['model']['generator'])")
    ap.add_argument("--config", required=True, help="recipe yaml (generator_params, hop_size, sampling_rate, ...)")
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--spk-emb", default=None, help=".npz of target-speaker embeddings keyed by speaker")
    ap.add_argument("--srcf0stats", default=None)
    ap.add_argument("--trgf0stats", default=None)
    ap.add_argument("--max-batch", type=int, default=32)
    args = ap.parse_args(argv)
    with open(args.config) as f:
        config = yaml.safe_load(f)
    device = torch.device("cuda")
    model = FastSVCGenerator(**config["generator_params"])
    model.load_state_dict(torch.load(args.checkpoint, map_location="cpu")["model"]["generator"])
    model.remove_weight_norm()
    model = model.eval().to(device)
    sg_conf = config.get("signal_generator", {})
    sg = SignalGenerator(sample_rate=config["sampling_rate"], hop_size=config["hop_size"],
                         sine_amp=sg_conf.get("sine_amp", 0.1), noise_amp=sg_conf.get("noise_amp", 0.003),
                         signal_types=sg_conf.get("signal_types", ["sine"]))
    files = sorted(glob.glob(os.path.join(args.dumpdir, "*.npz")) + glob.glob(os.path.join(args.dumpdir, "*.h5")))
    feats = [load_features(p) for p in files]
    utt_ids = [os.path.splitext(os.path.basename(p))[0] for p in files]
    os.makedirs(args.outdir, exist_ok=True)
    embs = dict(np.load(args.spk_emb)) if args.spk_emb else {}
    for trgspk in config.get("convert_to_speakers", [None]):
        trg_emb = embs.get(trgspk) if config["generator_params"].get("use_spk_emb") else None
        src_stats = trg_stats = None
        if args.srcf0stats and args.trgf0stats and trgspk is not None:
            trg_stats = _read_f0_mean(args.trgf0stats, trgspk)
            src_stats = [_read_f0_mean(args.srcf0stats, u.split("_")[0]) for u in utt_ids]
        t0 = time.time()
        ys = decode_utterances(model, feats, sg, device, trg_emb, src_stats, trg_stats, args.max_batch)
        torch.cuda.synchronize()
        dt = time.time() - t0
        total = sum(len(y) for y in ys)
        for utt, y in zip(utt_ids, ys):
            write_wav(os.path.join(args.outdir, f"{utt}_{trgspk}_gen.wav"), y, config["sampling_rate"])
        print(f"{len(ys)} utterances -> {trgspk}: RTF = {dt / (total / config['sampling_rate']):.5f}")


if __name__ == "__main__":                                    # pragma: no cover
    main()
