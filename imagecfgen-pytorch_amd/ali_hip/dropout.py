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

_state = {"seed": 0x5EED, "offset": 0, "inject": None, "pos": 0, "dev_counter": None,
          "plan": None, "record": None, "req": 0, "pair": 0, "plans": {}, "anon_plans": {}, "tag": None}


def manual_seed(seed: int):
    _state["seed"], _state["offset"] = int(seed), 0
    _state["plan"] = _state["record"] = None
    _state["anon_plans"].clear()


@contextlib.contextmanager
def injected_masks(masks):
    prev = (_state["inject"], _state["pos"])
    _state["inject"], _state["pos"] = list(masks), 0
    try:
        yield
    finally:
        _state["inject"], _state["pos"] = prev


@contextlib.contextmanager
def paired_passes(n_per_pass: int):
    """Two forward passes of one stack batched along the sample axis (rows [0,B) = first pass, [B,2B) = second):
    every mask request inside is for 2B samples.  Replayed (injected) masks were recorded pass after pass, so request
    k is served by masks k and n_per_pass + k of the tape; generated masks are simply drawn for 2B samples."""
    prev, _state["pair"] = _state["pair"], int(n_per_pass)
    try:
        yield
    finally:
        if _state["inject"] is not None and _state["pair"]:
            _state["pos"] += _state["pair"]          # the second pass's masks were consumed alongside the first's
        _state["pair"] = prev


def begin_iteration(dev_counter, owner=None, tag=None):
    """Stepper hook: masks of this iteration are keyed by the device-side iteration counter (graph replays
    read its current value) and by positions that restart at 0 every iteration.

    The first iteration of a stepper (per ``tag`` = batch shape / schedule variant) records its (B, C, p) request
    sequence; from then on all masks of such an iteration are produced by ONE launch (``ali_dropout_mask_multi``,
    bit-identical draws) and handed out as views.  The plans live on the owner, one per tag, for its whole life: a
    captured HIP graph keeps writing into its plan's buffer."""
    _state["dev_counter"], _state["offset"], _state["req"] = dev_counter, 0, 0
    _state["plans"] = owner.__dict__.setdefault("_mask_plans", {}) if owner is not None else _state["anon_plans"]
    _state["tag"] = tag
    plan = _state["plan"] = _state["plans"].get(tag)
    if _state["inject"] is not None:
        _state["plan"] = _state["record"] = None
        return
    if plan is None:
        _state["record"] = {"req": []}
        return
    _state["record"] = None
    ops.dropout_mask_multi(_state["seed"], dev_counter, plan["ends"], plan["ps"], plan["clog"], plan["cpad"],
                           plan["buf"])


def end_iteration():
    rec = _state["record"]
    _state["record"] = None
    if rec is None or not rec["req"] or len(rec["req"]) > 64:
        return
    ends, off = [], 0
    for B, C, p, cpad in rec["req"]:
        off += B * cpad
        ends.append(off)
    dev = rec["device"]
    _state["plans"][_state["tag"]] = {"req": rec["req"], "ends": ends, "ps": [r[2] for r in rec["req"]],
                                      "clog": [r[1] for r in rec["req"]], "cpad": [r[3] for r in rec["req"]],
                                      "buf": torch.empty(off, dtype=torch.float32, device=dev)}


def masks_consumed() -> int:
    return _state["pos"]


class Lane:
    """A sub-sequence of the iteration's mask requests that starts at a FIXED request index: lets two layer chains that
    the reference runs one after the other (D.dx, then D.dz: mnist.py:152-153) be advanced side by side while each
    still receives exactly the masks of its own turn.  ``next_mask(..., lane=lane)`` serves request ``lane.i`` and
    advances the lane only; the caller moves the global position past both chains afterwards (``advance``)."""

    def __init__(self, start):
        self.i = int(start)


def lanes_start(n_ahead: int):
    """The request index the next ``next_mask`` call would serve, if the coming ``n_ahead`` requests can be looked up
    by index (an injected tape, or the iteration's recorded mask plan); None otherwise (first iteration of a stepper:
    masks are drawn one launch at a time, in order)."""
    inj = _state["inject"]
    if inj is not None:
        i = _state["pos"]
        return i if i + n_ahead - 1 + _state["pair"] < len(inj) else None
    plan, rec = _state["plan"], _state["record"]
    if plan is None or rec is not None:
        return None
    i = _state["req"]
    return i if i + n_ahead <= len(plan["req"]) else None


def advance(n: int):
    """move the global position past ``n`` requests that were served through lanes"""
    if _state["inject"] is not None:
        _state["pos"] += n
        return
    plan = _state["plan"]
    for i in range(_state["req"], _state["req"] + n):
        B, C, _, _ = plan["req"][i]
        _state["offset"] += B * C
    _state["req"] += n


def _pad(mask, cpad):
    if mask.shape[1] == cpad:
        return mask
    out = torch.ones(mask.shape[0], cpad, device=mask.device)
    out[:, :mask.shape[1]] = mask
    return out


def peek_mask(skip: int, B: int, C: int, p: float, device, cpad=None):
    """The mask ``next_mask(B, C, p, device, cpad)`` will return after ``skip`` more requests, without consuming
    anything -- or None when it cannot be known yet (masks drawn one launch at a time: first iteration of a stepper,
    changed request sequence).  The caller must still request it in its turn."""
    cpad = C if cpad is None else cpad
    inj = _state["inject"]
    if inj is not None:
        i = _state["pos"] + skip
        if i + _state["pair"] >= len(inj):
            return None
        m = inj[i]
        if _state["pair"]:
            m = torch.cat([m, inj[i + _state["pair"]]], dim=0)
        if tuple(m.shape) != (B, C):
            return None
        return _pad(m.to(device=device, dtype=torch.float32).contiguous(), cpad)
    plan, rec = _state["plan"], _state["record"]
    if plan is None or rec is not None:
        return None
    i = _state["req"] + skip
    if i >= len(plan["req"]) or plan["req"][i] != (B, C, p, cpad):
        return None
    lo = plan["ends"][i] - B * cpad
    return plan["buf"][lo:lo + B * cpad].view(B, cpad)


def next_mask(B: int, C: int, p: float, device, cpad=None, lane=None) -> torch.Tensor:
    """[B, cpad] mask of the next Dropout2d: columns < C are Bernoulli(1-p)/(1-p), channel-padding columns are 1.
    ``lane``: serve the lane's next request instead of the global one (see ``Lane``)."""
    cpad = C if cpad is None else cpad
    if lane is not None:
        base = _state["pos"] if _state["inject"] is not None else _state["req"]
        m = peek_mask(lane.i - base, B, C, p, device, cpad)
        if m is None:
            raise RuntimeError(f"dropout lane: request {lane.i} is not a ({B}, {C}, p={p}) mask of this iteration's sequence")
        lane.i += 1
        return m
    inj = _state["inject"]
    if inj is not None:
        if _state["pos"] >= len(inj):
            raise RuntimeError("injected dropout masks exhausted")
        m = inj[_state["pos"]]
        if _state["pair"]:
            if _state["pos"] + _state["pair"] >= len(inj):
                raise RuntimeError("injected dropout masks exhausted (paired passes)")
            m = torch.cat([m, inj[_state["pos"] + _state["pair"]]], dim=0)
        _state["pos"] += 1
        if tuple(m.shape) != (B, C):
            raise RuntimeError(f"injected mask shape {tuple(m.shape)} != {(B, C)}")
        return _pad(m.to(device=device, dtype=torch.float32).contiguous(), cpad)
    plan, rec = _state["plan"], _state["record"]
    if plan is not None and rec is None:
        i = _state["req"]
        if i < len(plan["req"]) and plan["req"][i] == (B, C, p, cpad):
            lo = plan["ends"][i] - B * cpad
            _state["req"] = i + 1
            _state["offset"] += B * C
            return plan["buf"][lo:lo + B * cpad].view(B, cpad)
        _state["plan"] = None          # request sequence changed: one launch per mask for the rest of this iteration
    m = ops.dropout_mask(_state["seed"], _state["offset"], p, B, C, device, _state["dev_counter"])
    _state["offset"] += B * C
    if rec is not None:
        rec["req"].append((B, C, p, cpad))
        rec["device"] = device
    return _pad(m, cpad)
