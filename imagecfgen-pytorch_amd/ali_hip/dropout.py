"""Dropout2d mask source for the Discriminator stacks (reference mnist.py:99-134).

Production: counter-based masks generated on the device (``ali_dropout_mask``),
keyed by (seed, running offset) -- one Bernoulli(1-p)/(1-p) value per (sample, channel).
Parity runs: ``with injected_masks(list_of_[B,C]_tensors):`` replays masks recorded
from the CPU generator in call order (dx, dz, dxz -- mnist.py:152-154), because the
CPU Bernoulli stream cannot be regenerated on the GPU.
"""
import contextlib

import torch

from . import ops

_state = {"seed": 0x5EED, "offset": 0, "inject": None, "pos": 0, "dev_counter": None}


def manual_seed(seed: int):
    _state["seed"], _state["offset"] = int(seed), 0


@contextlib.contextmanager
def injected_masks(masks):
    prev = (_state["inject"], _state["pos"])
    _state["inject"], _state["pos"] = list(masks), 0
    try:
        yield
    finally:
        _state["inject"], _state["pos"] = prev


def begin_iteration(dev_counter):
    """Stepper hook: masks of this iteration are keyed by the device-side iteration counter (graph replays
    read its current value) and by positions that restart at 0 every iteration."""
    _state["dev_counter"], _state["offset"] = dev_counter, 0


def masks_consumed() -> int:
    return _state["pos"]


def next_mask(B: int, C: int, p: float, device) -> torch.Tensor:
    inj = _state["inject"]
    if inj is not None:
        if _state["pos"] >= len(inj):
            raise RuntimeError("injected dropout masks exhausted")
        m = inj[_state["pos"]]
        _state["pos"] += 1
        if tuple(m.shape) != (B, C):
            raise RuntimeError(f"injected mask shape {tuple(m.shape)} != {(B, C)}")
        return m.to(device=device, dtype=torch.float32).contiguous()
    m = ops.dropout_mask(_state["seed"], _state["offset"], p, B, C, device, _state["dev_counter"])
    _state["offset"] += B * C
    return m
