"""Checkpoint I/O in the reference trainer's on-disk schema.

``Trainer.save_checkpoint`` / ``load_checkpoint`` (``harana/bin/train_fastsvc.py:104-155``) write ONE
``torch.save`` dict::

    {"model":     {"generator": state_dict, "discriminator": state_dict},
     "optimizer": {"generator": ..., "discriminator": ...},
     "scheduler": {"generator": ..., "discriminator": ...},
     "steps": int, "epochs": int}

and ``load_model`` (``harana/utils/utils.py:243-280``) reads ``["model"]["generator"]`` next to a
``config.yml`` holding ``generator_type`` / ``generator_params``.  These helpers read and write that
layout so that checkpoints move between the reference and this package in both directions; the
generator state dict is the 251-key layout ``FastSVCGenerator.state_dict()`` produces (weight-norm
``weight_g`` / ``weight_v`` keys) or the folded ``.weight`` layout after ``remove_weight_norm()``.
"""
from __future__ import annotations

import os
from typing import Mapping, Optional

import torch


def save_checkpoint(path: str, generator: torch.nn.Module, discriminator: Optional[torch.nn.Module] = None,
                    optimizer: Optional[Mapping[str, object]] = None,
                    scheduler: Optional[Mapping[str, object]] = None, steps: int = 0, epochs: int = 0,
                    config: Optional[dict] = None) -> None:
    """Write `path` in the trainer schema; with `config` also `config.yml` beside it (what
    ``load_model`` and ``decode_fastsvc.py:120-129`` read when no --config is passed)."""
    def _sd(obj):
        return obj.state_dict() if hasattr(obj, "state_dict") else ({} if obj is None else dict(obj))

    state = {
        "optimizer": {"generator": _sd((optimizer or {}).get("generator")),
                      "discriminator": _sd((optimizer or {}).get("discriminator"))},
        "scheduler": {"generator": _sd((scheduler or {}).get("generator")),
                      "discriminator": _sd((scheduler or {}).get("discriminator"))},
        "steps": int(steps), "epochs": int(epochs),
        "model": {"generator": generator.state_dict(),
                  "discriminator": _sd(discriminator)},
    }
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    torch.save(state, path)
    if config is not None:
        import yaml
        with open(os.path.join(d, "config.yml"), "w") as f:
            yaml.dump(config, f, Dumper=yaml.Dumper)


def load_checkpoint(path: str, generator: torch.nn.Module, discriminator: Optional[torch.nn.Module] = None,
                    optimizer: Optional[Mapping[str, object]] = None,
                    scheduler: Optional[Mapping[str, object]] = None, load_only_params: bool = False) -> dict:
    """Mirror of ``Trainer.load_checkpoint`` (``train_fastsvc.py:130-155``): strict parameter load,
    then - unless `load_only_params` - optimizer / scheduler states; returns {"steps", "epochs"}."""
    state = torch.load(path, map_location="cpu")
    generator.load_state_dict(state["model"]["generator"])
    if discriminator is not None:
        discriminator.load_state_dict(state["model"]["discriminator"])
    if not load_only_params:
        for kind, objs in (("optimizer", optimizer), ("scheduler", scheduler)):
            for who in ("generator", "discriminator"):
                obj = (objs or {}).get(who)
                if obj is not None:
                    obj.load_state_dict(state[kind][who])
    return {"steps": state.get("steps", 0), "epochs": state.get("epochs", 0)}


def load_generator(path: str, config: Optional[dict] = None) -> torch.nn.Module:
    """What ``load_model`` does for this path (``utils.py:243-280``): config.yml beside the
    checkpoint unless given, class by ``generator_type``, strict load of ["model"]["generator"]."""
    from .generator import FastSVCGenerator
    if config is None:
        import yaml
        with open(os.path.join(os.path.dirname(os.path.abspath(path)), "config.yml")) as f:
            config = yaml.load(f, Loader=yaml.Loader)
    kind = config.get("generator_type", "FastSVCGenerator")
    if kind != "FastSVCGenerator":
        raise ValueError(f"generator_type {kind!r}: only FastSVCGenerator is implemented here")
    model = FastSVCGenerator(**config["generator_params"])
    model.load_state_dict(torch.load(path, map_location="cpu")["model"]["generator"])
    return model
