"""Tensor-level wrappers over the C ABI (device pointers from ``tensor.data_ptr()``,
stream from ``torch.cuda.current_stream()``).  PyTorch is plumbing here: memory,
streams, autograd bookkeeping.  Every function requires CUDA fp32 contiguous
tensors and raises otherwise -- there is no fallback path.
"""
import ctypes
from ctypes import byref, c_void_p

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_TANH, AliConvGeom, AliEpilogue  # noqa: F401

_WS = {}
_WS_BYTES = 256 << 20


def set_workspace_bytes(n: int):
    global _WS_BYTES
    _WS_BYTES = int(n)
    _WS.clear()


def workspace(device, slot: int = 0) -> torch.Tensor:
    """One scratch slab per device, reused stream-ordered by every kernel.  Zero-filled once: its first 4 KiB are the
    split-K arrival counters of the GEMM kernels, which every launch leaves at zero (include/ali_hip.h).
    ``slot`` > 0: the additional slabs the jobs of a multi-job GEMM launch use (one each: ``gemm_batch``)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), slot)
    ws = _WS.get(key)
    if ws is None or ws.numel() < _WS_BYTES:
        ws = torch.zeros(_WS_BYTES, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name="tensor"):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: need a contiguous fp32 CUDA tensor, got {t.dtype} {t.device} "
                         f"contiguous={t.is_contiguous()}")
    return c_void_p(t.data_ptr())


def _opt(t, name="tensor"):
    return None if t is None else _chk(t, name)


_PRECISION = {"f16": False}
_DACT16 = {"on": __import__("os").environ.get("ALI_NO_DACT16", "0") in ("", "0")}   # (A/B switch of AliEpilogue.dact_y16)


class precision:
    """``with ops.precision("f16"):`` -- the GEMMs launched inside run their contraction on fp16 MFMA
    (v_mfma_f32_32x32x16_f16, fp32 accumulation; include/ali_hip.h: AliEpilogue.mfma_f16) where their fast path
    applies; "f32" (default) is the exact fp32 MFMA everywhere."""

    def __init__(self, name):
        if name not in ("f32", "f16"):
            raise ValueError(f"precision must be 'f32' or 'f16', got {name!r}")
        self.f16 = name == "f16"

    def __enter__(self):
        self.prev, _PRECISION["f16"] = _PRECISION["f16"], self.f16
        return self

    def __exit__(self, *exc):
        _PRECISION["f16"] = self.prev


def geom(B, H, W, C, P, Q, K, R, S, stride, pad) -> AliConvGeom:
    return AliConvGeom(B, H, W, C, P, Q, K, R, S, stride, pad)


def epilogue(bias=None, act=ACT_NONE, slope=0.0, mask=None, dact_y=None, dact=ACT_NONE, dslope=0.0,
             bn_fwd=None, bn_bwd=None) -> AliEpilogue:
    """``bn_fwd`` = (part, groups, stat_mask or None[, slots]): also leave the per-tile (sum, sum of squares) of the
    output; ``bn_bwd`` = (part, x, mean, invstd, mask_in or None, mask_pre or None[, slots]): per-tile (sum g~*xhat,
    sum g~) of a data gradient (include/ali_hip.h, AliEpilogue).  ``slots`` = the slot count ``part`` was sized for
    (``conv_mtiles``): the launch refuses to run when its own M-tile count differs."""
    ep = AliEpilogue()
    # (python-side references of the operands behind the raw pointers: keeps them alive, lets a launch hook see them)
    ep.refs = {"bias": bias, "mask": mask, "dact_y": dact_y, "bn_fwd": bn_fwd, "bn_bwd": bn_bwd}
    ep.bias = _opt(bias, "bias")
    ep.act, ep.slope = act, slope
    ep.mask = _opt(mask, "mask")
    ep.mask_ld = mask.shape[1] if mask is not None else 0
    ep.dact_y = _opt(dact_y, "dact_y")
    ep.dact, ep.dslope = dact, dslope
    tw = getattr(dact_y, "_ali16", None) if (dact_y is not None and _PRECISION["f16"] and _DACT16["on"]) else None
    if tw is not None and tw.shape == dact_y.shape and tw.is_contiguous():
        ep.dact_y16 = c_void_p(tw.data_ptr())     # act' from the fp16 twin: half the bytes of the epilogue's read
        ep.refs["dact_y16"] = tw
    ep.mfma_f16 = int(_PRECISION["f16"])
    if bn_fwd is not None:
        part, groups, smask = bn_fwd[:3]
        ep.bn_slots = int(bn_fwd[3]) if len(bn_fwd) > 3 else 0
        ep.bn_part, ep.bn_mode, ep.bn_groups = _chk(part, "bn_part"), 1, groups
        ep.bn_stat_mask = _opt(smask, "bn_stat_mask")
        ep.bn_mask_ld = smask.shape[1] if smask is not None else 0
    elif bn_bwd is not None:
        part, x, mean, invstd, m_in, m_pre = bn_bwd[:6]
        ep.bn_slots = int(bn_bwd[6]) if len(bn_bwd) > 6 else 0
        ep.bn_part, ep.bn_mode, ep.bn_groups = _chk(part, "bn_part"), 2, 1
        ep.bn_x, ep.bn_mean, ep.bn_invstd = _chk(x, "bn_x"), c_void_p(mean.data_ptr()), c_void_p(invstd.data_ptr())
        ep.bn_mask_in, ep.bn_mask_pre = _opt(m_in, "bn_mask_in"), _opt(m_pre, "bn_mask_pre")
        m = m_in if m_in is not None else m_pre
        ep.bn_mask_ld = m.shape[1] if m is not None else 0
    return ep


_MTILES = {}


def reload_tuning():
    """Make the library re-read its developer tuning variables (ALI_BM, ALI_BN, ALI_SPLITK, ALI_TILE_M_SCALE, ...:
    csrc/ali_common.h) and drop every host-side cache that was derived under the old ones (M-tile counts, dispatch-order
    tables, deferrable flags).  Tests and sweeps only; never while a captured graph that used the old tuning is alive."""
    _lib.load().ali_reload_tuning()
    _MTILES.clear()
    _TILE_ORDER.clear()
    _DEFERRABLE.clear()


class tuning:
    """``with ops.tuning(ALI_BM=128, ALI_BN=128): ...`` -- set tuning variables for the block, restore them after."""

    def __init__(self, **env):
        self.env = {k: str(v) for k, v in env.items()}

    def __enter__(self):
        import os
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
        reload_tuning()
        return self

    def __exit__(self, *exc):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        reload_tuning()


def conv_mtiles(g: AliConvGeom, which: int):
    """(M-tiles, tile rows, rows ordered (pixel, image)?) of the launch ali_conv_fwd (0) / ali_conv_bwd_data (1) makes."""
    f16 = int(_PRECISION["f16"])
    key = (which, f16) + tuple(getattr(g, n) for n, _ in AliConvGeom._fields_)
    hit = _MTILES.get(key)
    if hit is None:
        rows, pm = ctypes.c_int32(0), ctypes.c_int32(0)
        n = _lib.load().ali_conv_mtiles(byref(g), which, f16, byref(rows), byref(pm))
        hit = _MTILES[key] = (n, rows.value, bool(pm.value))
    return hit


class KernelProfile:
    """Live timing of the GEMM / direct kernel launches of ONE eager iteration with HIP events on the launch stream
    (bench.py's roofline leg).  Every launch is timed once, where it runs in the iteration: ``[e0] k [e1]``.

    What makes that interval the kernel's duration as a rocprofv3 kernel trace reports it (checked against the trace of
    the same process: 4.764 vs 4.770 ms for the 88 GEMM launches of a MorphoMNIST iteration):
      * the caller keeps the stream busy with real work (graph replays of the same iteration) while the host enqueues
        the instrumented one, so its launches execute back to back -- as in a replay -- and at the clocks of the timed
        region (behind an idle or spin-waiting stream the chip clocks down: the same kernels then ran 10 % longer);
      * ``calibrate()`` measures what an ``[e0] k [e1]`` interval contains besides the kernel -- event markers plus
        dispatch latency -- as the zero-length intercept of spin kernels of two known lengths timed the same way, and
        that overhead is subtracted from every record (3.4 us plain, 5.7 us under rocprofv3).
    (An event pair attached to the dispatch itself, hipExtLaunchKernelGGL, was tried instead: its interval is 4.5 us
    LONGER than the trace's duration per launch, and it needs library support; dropped.)
    Each record: (family, algorithmic flops, algorithmic bytes, (e0, e1), shape)."""

    def __init__(self):
        self.records = []
        self.overhead_ms = 0.0
        self.spin_cycles_per_ms = None

    @staticmethod
    def _spin_pair(cycles):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(int(cycles))
        e1.record()
        return e0, e1

    def spin_rate(self):
        if self.spin_cycles_per_ms is None:
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            torch.cuda._sleep(20_000_000)
            t1.record()
            torch.cuda.synchronize()
            self.spin_cycles_per_ms = 20_000_000 / max(t0.elapsed_time(t1), 1e-3)
        return self.spin_cycles_per_ms

    def calibrate(self, busy=None, reps=24):
        """overhead_ms = the part of an ``[e0] k [e1]`` interval that is not the kernel.  ``busy()`` (optional) enqueues
        real work first, so that the pairs are measured behind a busy stream like the iteration itself."""
        rate = self.spin_rate()
        short, long_ = int(0.02 * rate), int(0.10 * rate)          # 20 us, 100 us
        if busy is not None:
            busy()
        else:
            torch.cuda._sleep(int(20.0 * rate))
        pairs = [(self._spin_pair(short), self._spin_pair(long_)) for _ in range(reps)]
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for (a, b), _ in pairs)[reps // 2]
        tl = sorted(a.elapsed_time(b) for _, (a, b) in pairs)[reps // 2]
        slope = (tl - ts) / (long_ - short)                       # ms per spin cycle
        self.overhead_ms = max(ts - slope * short, 0.0)
        return self.overhead_ms

    def hold(self, ms):
        """park the current stream for about ``ms`` milliseconds behind a spin kernel (fallback when the caller has no
        real work to keep the stream busy with: the chip clocks down meanwhile)"""
        torch.cuda._sleep(int(ms * self.spin_rate()))

    def _ms(self, ev):
        e0, e1 = ev
        return max(e0.elapsed_time(e1) - self.overhead_ms, 1e-4)

    def _table(self, key):
        torch.cuda.synchronize()
        tab = {}
        for name, flops, nbytes, ev, desc in self.records:
            f = tab.setdefault(key(name, desc), {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0})
            f["launches"] += 1
            f["flops"] += flops
            f["bytes"] += nbytes
            f["ms"] += self._ms(ev)
        return tab

    def summary(self):
        return self._table(lambda name, desc: name)

    def by_shape(self):
        return self._table(lambda name, desc: (name,) + desc)


_PROFILE = None


def set_profile(p):
    global _PROFILE
    _PROFILE = p


_HOOK = None


class launch_hook:
    """``with ops.launch_hook(obj):`` -- tests only.  Every GEMM launch made inside the block is reported to ``obj``
    right after it has been issued (``obj.scatter(...)`` for ``tconv_scatter``):
    ``obj.gemm(kind, g, a, w, out, ep, in_ld, out_ld)`` for ``conv_fwd`` ("fwd") /
    ``conv_bwd_data`` ("bwd_data") and ``obj.wgrad(g, x, dy, dst, cg_log, cd_log, strides, db, dy_ld)`` for
    ``conv_bwd_weight`` -- the latter returns a callable (or None) that is run once the result is final (at once, or
    after ``FoldQueue.flush`` for deferred launches).  tests/launch_audit.py re-computes each launch with torch on the
    CPU from the very operands the kernel read."""

    def __init__(self, obj):
        self.obj = obj

    def __enter__(self):
        global _HOOK
        self.prev, _HOOK = _HOOK, self.obj
        return self.obj

    def __exit__(self, *exc):
        global _HOOK
        _HOOK = self.prev


def _geom_cost(g: AliConvGeom, live=None):
    """(algorithmic FLOP, algorithmic bytes, shape key) of one GEMM launch: 2*B*P*Q*K*C*R*S and both activations + the
    weights once, fp32.  ``live`` = (c, k): the channels of x / y that carry data where a channel stride is padded
    (5 of 8 image planes, 771 of 800 Generator inputs) -- multiplications by the zero padding are not algorithmic work."""
    c = live[0] if (live and live[0]) else g.C
    k = live[1] if (live and live[1]) else g.K
    return (2.0 * g.B * g.P * g.Q * k * c * g.R * g.S,
            4.0 * (g.B * g.H * g.W * c + g.B * g.P * g.Q * k + k * c * g.R * g.S),
            (g.B, g.H, g.W, g.C, g.P, g.Q, g.K, g.R, g.stride, g.pad))


_NO_SHAPE = (0,) * 10


def _launch(name, cost, fn):
    """Run one kernel launch; under a KernelProfile bracket it with two event records (see KernelProfile).
    ``cost`` = (algorithmic flops, algorithmic bytes, shape key)."""
    if _PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    _PROFILE.records.append((name, cost[0], cost[1], (e0, e1), cost[2]))


PAIR_GEMMS = True          # tests / A-B measurements: False = ``gemm_batch`` launches every GEMM on its own, at once
_GEMM_BATCH = None
GEMM_BATCH_SLOTS = 4


class gemm_batch:
    """``with ops.gemm_batch():`` -- the ``conv_fwd`` / ``conv_bwd_data`` calls made inside are RECORDED (include/ali_hip.h:
    AliGemmJob) and issued together when the block ends: one launch per kernel variant instead of one per call.  The
    calls of one block must be mutually independent and nothing inside the block may read a result of one of them
    (``chain.run_parallel`` advances independent layer chains one GEMM at a time under such a block).  Each recorded
    job works in a workspace slab of its own.  Launches of a kind no multi-job kernel exists for go out at once."""

    def __init__(self):
        self.jobs, self.keep, self.after = [], [], []
        self.flops = self.bytes = 0.0

    def __enter__(self):
        global _GEMM_BATCH
        self.prev, _GEMM_BATCH = _GEMM_BATCH, (self if PAIR_GEMMS else None)
        return self

    def __exit__(self, *exc):
        global _GEMM_BATCH
        _GEMM_BATCH = self.prev
        if exc[0] is None:
            self.flush()

    def flush(self):
        if self.jobs:
            n = len(self.jobs)
            arr = (_lib.AliGemmJob * n)(*self.jobs)

            def go():
                _lib.check(_lib.load().ali_gemm_launch_multi(n, arr, _stream()), "ali_gemm_launch_multi")
            _launch("gconv", (self.flops, self.bytes, _NO_SHAPE), go)
        todo = self.after
        self.jobs, self.keep, self.after, self.flops, self.bytes = [], [], [], 0.0, 0.0
        for fn in todo:
            fn()


def _gemm(which, name, g, a, w_packed, out, ep, in_ld, out_ld, live):
    """one forward (which = 0) / data-gradient (1) GEMM: launched, or recorded into the open ``gemm_batch``"""
    lib = _lib.load()
    ep.in_ld, ep.out_ld = in_ld, out_ld
    _f16_operands(g, which, a, w_packed, out, ep)
    _set_tile_order(g, which, ep, a.device)
    cost = _geom_cost(g, live)
    kind = "fwd" if which == 0 else "bwd_data"
    hook = _HOOK
    b = _GEMM_BATCH
    if b is not None and len(b.jobs) >= GEMM_BATCH_SLOTS:
        b.flush()
    if b is None:
        ws = workspace(a.device)
        fn = lib.ali_conv_fwd if which == 0 else lib.ali_conv_bwd_data

        def go():
            _lib.check(fn(byref(g), _view_ptr(a, in_ld, "a"), _chk(w_packed, "w"), _view_ptr(out, out_ld, "out"),
                          byref(ep), c_void_p(ws.data_ptr()), ws.numel(), _stream()), "ali_conv_" + kind)
        _launch(name, cost, go)
        if hook is not None:
            hook.gemm(kind, g, a, w_packed, out, ep, in_ld, out_ld)
        return out
    ws = workspace(a.device, 1 + len(b.jobs))
    job = _lib.AliGemmJob()
    fn = lib.ali_conv_fwd_job if which == 0 else lib.ali_conv_bwd_data_job
    ev = None
    if _PROFILE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _lib.check(fn(byref(g), _view_ptr(a, in_ld, "a"), _chk(w_packed, "w"), _view_ptr(out, out_ld, "out"), byref(ep),
                  c_void_p(ws.data_ptr()), ws.numel(), byref(job), _stream()), "ali_conv_" + kind + "_job")
    if job.opaque[0] == 1:
        b.jobs.append(job)
        b.keep.append((a, w_packed, out, ep, ws))
        b.flops += cost[0]
        b.bytes += cost[1]
        if hook is not None:
            b.after.append(lambda: hook.gemm(kind, g, a, w_packed, out, ep, in_ld, out_ld))
    else:                               # a launch of another kind: it went out right away
        if ev is not None:
            ev[1].record()
            _PROFILE.records.append((name, cost[0], cost[1], ev, cost[2]))
        if hook is not None:
            hook.gemm(kind, g, a, w_packed, out, ep, in_ld, out_ld)
    return out


def shadow16(t):
    """the fp16 twin a tensor carries (set by the fp16-MFMA launch that produced it, or by ``ensure_shadow16``)"""
    h = getattr(t, "_ali16", None)
    if h is not None and (h.shape != t.shape or h.device != t.device):
        return None
    return h


_TWIN_OF = {}          # data_ptr of a packed-weight tensor -> weakref of its fp16 twin (the pack kernel writes both)
_TWIN_WRITTEN = set()  # data_ptrs whose twin the pending / last pack launch wrote: no separate re-rounding needed


def ensure_shadow16(t):
    """fp16 twin of a (weight) tensor, created once and kept on the tensor object; ``refresh_shadow16`` re-rounds it
    in place after the tensor changed (fixed addresses: captured graphs stay valid)."""
    import weakref
    h = shadow16(t)
    if h is None:
        h = t.detach().to(torch.float16)
        t._ali16 = h
        if len(_TWIN_OF) > 4096:
            for k in [k for k, r in _TWIN_OF.items() if r() is None]:
                del _TWIN_OF[k]
        _TWIN_OF[t.data_ptr()] = weakref.ref(h)
    return h


def refresh_shadow16(t):
    h = shadow16(t)
    if h is None:
        return
    if t.data_ptr() in _TWIN_WRITTEN:          # the pack launch that rewrote t wrote the twin as well
        _TWIN_WRITTEN.discard(t.data_ptr())
        return
    h.copy_(t.detach())


def _twin_for_pack(dst):
    r = _TWIN_OF.get(dst.data_ptr())
    h = r() if r is not None else None
    if h is not None and h.numel() == dst.numel() and h.device == dst.device:
        return h
    return None


def _f16_operands(g, which, x, w_packed, y, ep):
    """fp16 twins for a precision("f16") launch: read them where both operands have one, leave one of the output.
    (Column-range operands -- in_ld / out_ld -- neither read nor leave twins.)"""
    if not _PRECISION["f16"]:
        return
    x16, w16 = (None if ep.in_ld else shadow16(x)), shadow16(w_packed)
    if x16 is not None and w16 is not None:
        ep.in16, ep.w16 = c_void_p(x16.data_ptr()), c_void_p(w16.data_ptr())
    if not ep.out_ld and _lib.load().ali_conv_writes_out16(byref(g), which):
        y16 = shadow16(y)                # (a persistent output buffer keeps its twin: fixed addresses for captured graphs)
        if y16 is None or y16.dtype != torch.float16:
            y16 = torch.empty(y.shape, dtype=torch.float16, device=y.device)
        ep.out16 = c_void_p(y16.data_ptr())
        y._ali16 = y16


_TILE_ORDER = {}


def conv_tile_order(g: AliConvGeom, which: int, device):
    """Cached device table of the launch's M-tile ids, longest k-loop first (include/ali_hip.h: ali_conv_tile_order),
    or None when every tile costs the same.  Built on the host: a miss while the stream is capturing returns None
    (the eager warm-up iteration in front of every capture fills the cache)."""
    f16 = int(_PRECISION["f16"])
    key = (device.index, which, f16) + tuple(getattr(g, n) for n, _ in AliConvGeom._fields_)
    if key in _TILE_ORDER:
        return _TILE_ORDER[key]
    if torch.cuda.is_current_stream_capturing():
        return None
    cap = 1 << 16
    buf = (ctypes.c_int32 * cap)()
    n = _lib.load().ali_conv_tile_order(byref(g), which, f16, ctypes.cast(buf, c_void_p), cap)
    tab = torch.tensor(list(buf[:n]), dtype=torch.int32, device=device) if n > 0 else None
    _TILE_ORDER[key] = tab
    return tab


def _set_tile_order(g, which, ep, device):
    tab = conv_tile_order(g, which, device)
    if tab is not None:
        ep.tile_order, ep.tile_order_n = tab.data_ptr(), tab.numel()


def _view_ptr(t, ld, name):
    """pointer of a dense tensor (ld == 0) or of a column range of rows ``ld`` floats apart (a strided view)"""
    return _ptr(t) if ld else _chk(t, name)


def conv_fwd(g: AliConvGeom, x, w_packed, y, ep: AliEpilogue, in_ld=0, out_ld=0, live=None):
    """``in_ld`` / ``out_ld`` > 0: ``x`` / ``y`` are column ranges (strided views) of wider row-major buffers whose
    pixels are that many floats apart (AliEpilogue.in_ld / out_ld).  ``live`` = (channels of x, channels of y) that
    carry data when a stride is padded: accounting only (``_geom_cost``)."""
    return _gemm(0, "gconv", g, x, w_packed, y, ep, in_ld, out_ld, live)


def conv_bwd_data(g: AliConvGeom, dy, w_packed, dx, ep: AliEpilogue, in_ld=0, out_ld=0, live=None):
    return _gemm(1, "gconv_t", g, dy, w_packed, dx, ep, in_ld, out_ld, live)


_PIXTAB = {}


def wgrad_pixtab(g: AliConvGeom, device):
    """Cached per-geometry pixel table of the weight-gradient kernel (include/ali_hip.h: ali_wgrad_pixtab)."""
    key = (device.index, g.B, g.H, g.W, g.C, g.P, g.Q, g.stride, g.pad)
    tab = _PIXTAB.get(key)
    if tab is None:
        if len(_PIXTAB) > 256:
            _PIXTAB.clear()
        tab = torch.empty(2 * g.B * g.P * g.Q, dtype=torch.int32, device=device)
        _lib.check(_lib.load().ali_wgrad_pixtab(byref(g), c_void_p(tab.data_ptr()), _stream()), "ali_wgrad_pixtab")
        _PIXTAB[key] = tab
    return tab


_DEFERRABLE = {}


def wgrad_deferrable(g: AliConvGeom) -> int:
    f16 = int(_PRECISION["f16"])
    key = (f16,) + tuple(getattr(g, n) for n, _ in AliConvGeom._fields_)
    hit = _DEFERRABLE.get(key)
    if hit is None:
        hit = _DEFERRABLE[key] = int(_lib.load().ali_wgrad_deferrable(byref(g), f16))
    return hit


DEFER_WGRAD_LAUNCH = True     # tests / A-B measurements: False = every weight-gradient GEMM is its own launch


class FoldQueue:
    """Deferred slab reductions of the weight-gradient launches of one backward pass (include/ali_hip.h: AliWgradFold):
    ``conv_bwd_weight(..., defer=queue)`` gives each launch its own region of the queue's arena and skips the second
    (reduction) launch; ``flush()`` folds them all at once.  The arena is allocated once per device and shared (one
    backward pass is in flight at a time; captured graphs keep pointing into it)."""
    _arena = {}
    arena_bytes = 512 << 20

    def __init__(self, device):
        self.device = device
        self.jobs = []          # deferred slab reductions
        self.launches = []      # deferred GEMM launches (AliWgradJob) ...
        self.keep = []          # ... and the tensors they read / write, alive until flush
        self.off = 0
        self.flops = 0.0
        self.bytes = 0.0
        self.n_jobs = 1         # deferrable launches of the pass in progress (expect()): what a combined launch will hold
        self.after_flush = []   # launch_hook callbacks of deferred weight gradients (tests)

    def arena(self):
        a = FoldQueue._arena.get(self.device.index)
        if a is None or a.numel() * 4 < FoldQueue.arena_bytes:
            a = FoldQueue._arena[self.device.index] = torch.zeros(FoldQueue.arena_bytes // 4, dtype=torch.float32,
                                                                   device=self.device)
        return a

    def expect(self, geoms):
        """Called before a pass's first weight gradient with the geometries of all of them: the number of launches the
        pass will combine decides how far each is split (a function of the pass alone -- not of what ran before)."""
        self.n_jobs = max(1, sum(wgrad_deferrable(g) for g in geoms))

    def abandon(self):
        """forget recorded work without launching it (an iteration that raised half-way)"""
        self.jobs, self.launches, self.keep, self.flops, self.bytes, self.off = [], [], [], 0.0, 0.0, 0
        self.after_flush = []

    def split_target(self):
        """blocks a deferred GEMM should split into: about 2048 in the whole combined launch"""
        return max(256, min(1024, 2048 // self.n_jobs))

    def flush(self):
        if self.launches:
            arr = (_lib.AliWgradJob * len(self.launches))(*self.launches)
            n = len(self.launches)

            def go_l():
                _lib.check(_lib.load().ali_wgrad_launch_multi(n, arr, _stream()), "ali_wgrad_launch_multi")
            _launch("wgrad_multi", (self.flops, self.bytes, _NO_SHAPE), go_l)
        self.launches, self.keep, self.flops, self.bytes = [], [], 0.0, 0.0
        if self.jobs:
            arr = (_lib.AliWgradFold * len(self.jobs))(*self.jobs)
            n = len(self.jobs)

            def go():
                _lib.check(_lib.load().ali_wgrad_fold_multi(n, arr, _stream()), "ali_wgrad_fold_multi")
            _launch("wgrad_fold", (0.0, 0.0, _NO_SHAPE), go)
        self.jobs, self.off = [], 0
        todo, self.after_flush = self.after_flush, []
        for fn in todo:
            fn()


def conv_bwd_weight(g: AliConvGeom, x, dy, dst, cg_log, cd_log, s_dc, s_gc, s_tap, db=None, dy_ld=0, defer=None):
    """``db`` (optional, [cd_log]): also produce the column sums of ``dy`` (Conv2d bias gradient) in the same launch.
    ``dy_ld`` > 0: ``dy`` is a column range (a strided view) of rows that are ``dy_ld`` floats apart.
    ``defer`` (FoldQueue): leave the slab reduction of a split launch to ``defer.flush()``."""
    lib = _lib.load()
    ws = workspace(x.device)
    tab = wgrad_pixtab(g, x.device) if (g.C % 4 == 0 and g.K % 4 == 0) else None
    f16 = int(_PRECISION["f16"])
    x16, dy16 = (shadow16(x), shadow16(dy)) if (f16 and not dy_ld) else (None, None)
    if x16 is None or dy16 is None:
        x16 = dy16 = None
    ws_ptr, ws_n, job, lj = ws.data_ptr(), ws.numel(), None, None
    if defer is not None:
        arena = defer.arena()
        left = arena.numel() * 4 - defer.off
        if left >= (32 << 20):                    # a region of its own (else: the shared workspace, immediate fold)
            ws_ptr, ws_n, job = arena.data_ptr() + defer.off, left, _lib.AliWgradFold()
            if DEFER_WGRAD_LAUNCH:
                lj = _lib.AliWgradJob()

    def go():
        _lib.check(lib.ali_conv_bwd_weight(byref(g), _chk(x, "x"), _ptr(dy) if dy_ld else _chk(dy, "dy"),
                                           _ptr(dst), cg_log, cd_log,        # (dst: any strides, see s_dc / s_gc / s_tap)
                                           s_dc, s_gc, s_tap, _opt(db, "db"),
                                           None if tab is None else c_void_p(tab.data_ptr()), f16,
                                           None if x16 is None else c_void_p(x16.data_ptr()),
                                           None if dy16 is None else c_void_p(dy16.data_ptr()), dy_ld,
                                           None if job is None else byref(job), None if lj is None else byref(lj),
                                           0 if lj is None else defer.split_target(),
                                           c_void_p(ws_ptr), ws_n, _stream()), "ali_conv_bwd_weight")
    cost = _geom_cost(g, (cg_log, cd_log))
    if lj is None:
        _launch("wgrad", cost, go)
    else:                     # records the job, or -- a launch of another kind (large tiles, fp16) -- runs it right away
        ev = None
        if _PROFILE is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        go()
        if lj.opaque[0] == 1:
            defer.launches.append(lj)
            defer.keep.append((x, dy, dst, db, tab))
            defer.flops += cost[0]
            defer.bytes += cost[1]
        elif ev is not None:
            ev[1].record()
            _PROFILE.records.append(("wgrad", cost[0], cost[1], ev, cost[2]))
    deferred = job is not None and job.S > 0
    if deferred:
        defer.jobs.append(job)
        defer.off += (int(job.ws_used) + 255) // 256 * 256
    if _HOOK is not None:
        done = _HOOK.wgrad(g, x, dy, dst, cg_log, cd_log, (s_dc, s_gc, s_tap), db, dy_ld)
        if done is not None:
            if deferred or (lj is not None and lj.opaque[0] == 1):
                defer.after_flush.append(done)
            else:
                done()
    return dst


def _ptr(t):
    """raw pointer of a (possibly strided) fp32 CUDA view"""
    if not (t.is_cuda and t.dtype == torch.float32):
        raise ValueError("need an fp32 CUDA tensor")
    return c_void_p(t.data_ptr())


def tconv1_fwd(big, w_tk, bias, out, B, P, Q, K, R, S, pad, ostride, act, slope, rowscale=None):
    """``rowscale``: optional [B] view (any stride) of per-sample factors applied to the output."""
    lib = _lib.load()

    def go():
        _lib.check(lib.ali_tconv1_fwd(_chk(big, "big"), _chk(w_tk, "w"), None if bias is None else _ptr(bias),
                                      _ptr(out), B, P, Q, K, R, S, pad, ostride, act, slope,
                                      None if rowscale is None else _ptr(rowscale),
                                      0 if rowscale is None else rowscale.stride(0), _stream()),
                   "ali_tconv1_fwd")
    _launch("tconv1_fwd", (2.0 * B * P * Q * K * R * S, 4.0 * B * (P * Q * K + (P + R - 1 - 2 * pad) * (Q + S - 1 - 2 * pad)), _NO_SHAPE), go)
    return out


def tconv1_dgrad(small, sstride, w_tk, dact_y, dact, dslope, gbig, B, P, Q, K, R, S, pad):
    lib = _lib.load()

    def go():
        _lib.check(lib.ali_tconv1_dgrad(_ptr(small), sstride, _chk(w_tk, "w"), _opt(dact_y), dact, dslope,
                                        _chk(gbig, "gbig"), B, P, Q, K, R, S, pad, _stream()), "ali_tconv1_dgrad")
    _launch("tconv1_dgrad", (2.0 * B * P * Q * K * R * S, 4.0 * B * (2 * P * Q * K + (P + R - 1 - 2 * pad) * (Q + S - 1 - 2 * pad)), _NO_SHAPE), go)
    return gbig


def tconv1_wgrad(big, small, sstride, nc, dw, s_k, s_tap, s_c, B, P, Q, K, R, S, pad):
    lib = _lib.load()
    ws = workspace(big.device)

    def go():
        _lib.check(lib.ali_tconv1_wgrad(_chk(big, "big"), _ptr(small), sstride, nc, _ptr(dw), s_k, s_tap, s_c, B, P,
                                        Q, K, R, S, pad, c_void_p(ws.data_ptr()), ws.numel(), _stream()),
                   "ali_tconv1_wgrad")
    _launch("tconv1_wgrad", (2.0 * B * P * Q * K * R * S * nc, 4.0 * B * (P * Q * K + nc * (P + R - 1 - 2 * pad) * (Q + S - 1 - 2 * pad)), _NO_SHAPE), go)
    return dw


_PACK_BATCH = None
_PACK_AFTER = []          # callbacks to run once the pending pack jobs have been launched (fp16 twins of the packs)


def after_packs(fn):
    """Run ``fn`` now, or -- inside ``batched_packs`` -- right after the collected pack jobs have been launched."""
    if _PACK_BATCH is None:
        fn()
    else:
        _PACK_AFTER.append(fn)


class batched_packs:
    """Collect the ``pack_weights`` calls issued inside the block and run them as ONE launch (at most 40 jobs per
    launch; a job that reads the destination of a pending job flushes first)."""

    def __enter__(self):
        global _PACK_BATCH
        self.prev, _PACK_BATCH = _PACK_BATCH, []
        return self

    def __exit__(self, *exc):
        global _PACK_BATCH
        _flush_packs()
        _PACK_BATCH = self.prev
        if self.prev is None:
            todo, _PACK_AFTER[:] = list(_PACK_AFTER), []
            for fn in todo:
                fn()


def _launch_packs(jobs):
    """jobs: (src, dst, N, T, C, Cpad, s_n, s_tap, s_c) tuples; one launch per 40 (include/ali_hip.h)."""
    lib = _lib.load()
    for lo in range(0, len(jobs), 40):
        part = jobs[lo:lo + 40]
        n = len(part)
        twins = [_twin_for_pack(j[1]) for j in part]
        srcs = (c_void_p * n)(*[j[0].data_ptr() for j in part])
        dsts = (c_void_p * n)(*[j[1].data_ptr() for j in part])
        d16 = (c_void_p * n)(*[None if h is None else h.data_ptr() for h in twins])
        dims = (ctypes.c_int32 * (4 * n))(*[v for j in part for v in j[2:6]])
        strides = (ctypes.c_int64 * (3 * n))(*[v for j in part for v in j[6:9]])
        _lib.check(lib.ali_pack_weights_multi(n, srcs, dsts, d16, dims, strides, _stream()), "ali_pack_weights_multi")
        for j, h in zip(part, twins):
            if h is not None:
                _TWIN_WRITTEN.add(j[1].data_ptr())


def _flush_packs():
    jobs = _PACK_BATCH
    if not jobs:
        return
    _launch_packs(jobs)
    jobs.clear()


def pack_weights(src, dst, N, T, C, Cpad, s_n, s_tap, s_c):
    _chk(src, "src"), _chk(dst, "dst")
    if _PACK_BATCH is not None:
        if any(j[1].data_ptr() == src.data_ptr() for j in _PACK_BATCH):
            _flush_packs()
        _PACK_BATCH.append((src, dst, N, T, C, Cpad, s_n, s_tap, s_c))
        return dst
    _launch_packs([(src, dst, N, T, C, Cpad, s_n, s_tap, s_c)])
    return dst


def head_fwd(x2d, w, bias, y):
    """y[b] = bias + x2d[b, :] @ w  (the Discriminator's Conv2d(C, 1, 1) head; include/ali_hip.h: ali_head_fwd)"""
    B, C = x2d.shape
    _lib.check(_lib.load().ali_head_fwd(_ptr(x2d), x2d.stride(0), _chk(w, "w"), _opt(bias, "bias"), _chk(y, "y"), B, C,
                                        _stream()), "ali_head_fwd")
    return y


def head_wgrad(x2d, g, dw, db=None):
    """dw[c] = sum_b g[b] * x2d[b, c], db = sum_b g[b]  (include/ali_hip.h: ali_head_wgrad)"""
    B, C = x2d.shape
    _lib.check(_lib.load().ali_head_wgrad(_ptr(x2d), x2d.stride(0), _chk(g, "g"), _chk(dw, "dw"), _opt(db, "db"), B, C,
                                          _stream()), "ali_head_wgrad")
    return dw


def copy_multi(pairs):
    """[(dst, src), ...]: dst.copy_(src) for all pairs -- same-dtype contiguous CUDA pairs share one launch
    (include/ali_hip.h: ali_copy_multi), anything else falls back to Tensor.copy_."""
    fast = []
    for dst, src in pairs:
        if (dst.is_cuda and src.is_cuda and dst.device == src.device and dst.dtype == src.dtype and dst.shape == src.shape
                and dst.is_contiguous() and src.is_contiguous() and dst.element_size() % 4 == 0 and dst.numel() > 0):
            fast.append((dst, src))
        else:
            dst.copy_(src)
    if fast:
        n = len(fast)
        srcs = (c_void_p * n)(*[s_.data_ptr() for _, s_ in fast])
        dsts = (c_void_p * n)(*[d_.data_ptr() for d_, _ in fast])
        sizes = (ctypes.c_int64 * n)(*[d_.numel() * d_.element_size() for d_, _ in fast])
        _lib.check(_lib.load().ali_copy_multi(n, srcs, dsts, sizes, _stream()), "ali_copy_multi")


def act_bwd(gy, y, act, slope, out=None):
    lib = _lib.load()
    out = torch.empty_like(gy) if out is None else out
    _lib.check(lib.ali_act_bwd(_chk(gy, "gy"), _chk(y, "y"), _chk(out), gy.numel(), act, slope, _stream()),
               "ali_act_bwd")
    return out


def colsum(x2d_rows, C, ld, x, out=None):
    lib = _lib.load()
    ws = workspace(x.device)
    out = torch.empty(C, dtype=torch.float32, device=x.device) if out is None else out
    _lib.check(lib.ali_colsum(_chk(x, "x"), x2d_rows, C, ld, _chk(out), c_void_p(ws.data_ptr()), ws.numel(),
                              _stream()), "ali_colsum")
    return out


def rowmask_mul(x, mask, B, rows_per_img, C, out=None):
    lib = _lib.load()
    out = torch.empty_like(x) if out is None else out
    _lib.check(lib.ali_rowmask_mul(_chk(x, "x"), _chk(mask, "mask"), _chk(out), B, rows_per_img, C, _stream()),
               "ali_rowmask_mul")
    return out


def dropout_mask(seed, offset, p, B, C, device, dev_counter=None):
    lib = _lib.load()
    out = torch.empty(B, C, dtype=torch.float32, device=device)
    ctr = None
    if dev_counter is not None:
        assert dev_counter.is_cuda and dev_counter.dtype == torch.int64
        ctr = c_void_p(dev_counter.data_ptr())
    _lib.check(lib.ali_dropout_mask(seed, offset, ctr, p, _chk(out), out.numel(), _stream()), "ali_dropout_mask")
    return out


def dropout_mask_multi(seed, dev_counter, seg_end, seg_p, seg_clog, seg_cpad, out):
    """one launch for all masks of an iteration; seg_* are host lists (see ali_hip.h)."""
    lib = _lib.load()
    n = len(seg_end)
    ends = (ctypes.c_int64 * n)(*seg_end)
    ps = (ctypes.c_float * n)(*seg_p)
    cl = (ctypes.c_int32 * n)(*seg_clog)
    cp = (ctypes.c_int32 * n)(*seg_cpad)
    ctr = None if dev_counter is None else c_void_p(dev_counter.data_ptr())
    _lib.check(lib.ali_dropout_mask_multi(seed, ctr, ends, ps, cl, cp, n, _chk(out), _stream()),
               "ali_dropout_mask_multi")
    return out


def bn_stats(x, mask, B, rows_per_img, C, gamma, beta, running_mean, running_var, momentum, eps, training, groups=1):
    """Returns st [4, C] (mean, invstd, sc, sh), or [groups, 4, C] for ``groups`` batched passes."""
    lib = _lib.load()
    ws = workspace(x.device)
    st = torch.empty(groups, 4, C, dtype=torch.float32, device=x.device)
    _lib.check(lib.ali_bn_stats(_chk(x, "x"), _opt(mask), B, rows_per_img, C, _opt(gamma), _opt(beta),
                                _opt(running_mean), _opt(running_var), momentum, eps, int(training),
                                c_void_p(st[0, 0].data_ptr()), c_void_p(st[0, 1].data_ptr()),
                                c_void_p(st[0, 2].data_ptr()), c_void_p(st[0, 3].data_ptr()), groups, 4 * C,
                                c_void_p(ws.data_ptr()), ws.numel(), _stream()), "ali_bn_stats")
    return st[0] if groups == 1 else st


def bn_stats_from_partials(part, slots, groups, C, count, gamma, beta, running_mean, running_var, momentum, eps):
    """bn_stats (training mode) from the per-tile partial sums a convolution epilogue left in ``part``."""
    lib = _lib.load()
    st = torch.empty(groups, 4, C, dtype=torch.float32, device=part.device)
    _lib.check(lib.ali_bn_stats_from_partials(_chk(part, "part"), slots, groups, C, count, _opt(gamma), _opt(beta),
                                              _opt(running_mean), _opt(running_var), momentum, eps,
                                              c_void_p(st[0, 0].data_ptr()), c_void_p(st[0, 1].data_ptr()),
                                              c_void_p(st[0, 2].data_ptr()), c_void_p(st[0, 3].data_ptr()), 4 * C,
                                              _stream()), "ali_bn_stats_from_partials")
    return st[0] if groups == 1 else st


def bn_bwd_from_partials(part, slots, x, g, mask_in, mask_pre, st, gamma, B, rows_per_img, C, batch_stats, slope,
                         want_gx=True, out_dgamma=None, out_dbeta=None):
    """bn_bwd from the per-tile partial sums the data-gradient GEMM's epilogue left in ``part``."""
    lib = _lib.load()
    if out_dgamma is None or out_dbeta is None:
        dg = torch.empty(2, C, dtype=torch.float32, device=x.device)
        out_dgamma, out_dbeta = dg[0], dg[1]
    gx = torch.empty_like(x) if want_gx else None
    _lib.check(lib.ali_bn_bwd_from_partials(_chk(part, "part"), slots, _chk(x, "x"), _chk(g, "g"), _opt(mask_in),
                                            _opt(mask_pre), c_void_p(st[0].data_ptr()), c_void_p(st[1].data_ptr()),
                                            _opt(gamma), B, rows_per_img, C, int(batch_stats), float(slope),
                                            _chk(out_dgamma, "dgamma"), _chk(out_dbeta, "dbeta"), _opt(gx), _stream()),
               "ali_bn_bwd_from_partials")
    return out_dgamma, out_dbeta, gx


def bn_apply(x, st, mask_in, mask_post, B, rows_per_img, C, out=None, groups=1):
    lib = _lib.load()
    out = torch.empty_like(x) if out is None else out
    st0 = st if groups == 1 else st[0]
    _lib.check(lib.ali_bn_apply(_chk(x, "x"), c_void_p(st0[2].data_ptr()), c_void_p(st0[3].data_ptr()), _opt(mask_in),
                                _opt(mask_post), _chk(out), B, rows_per_img, C, groups, 4 * C, _stream()),
               "ali_bn_apply")
    return out


def bn_bwd(x, g, mask_in, mask_pre, st, gamma, B, rows_per_img, C, batch_stats, slope, want_gx=True, out_dgamma=None,
           out_dbeta=None):
    """returns (dgamma, dbeta, gx); out_dgamma / out_dbeta: optional contiguous [C] destinations."""
    lib = _lib.load()
    ws = workspace(x.device)
    if out_dgamma is None or out_dbeta is None:
        dg = torch.empty(2, C, dtype=torch.float32, device=x.device)
        out_dgamma, out_dbeta = dg[0], dg[1]
    gx = torch.empty_like(x) if want_gx else None
    _lib.check(lib.ali_bn_bwd(_chk(x, "x"), _chk(g, "g"), _opt(mask_in), _opt(mask_pre), c_void_p(st[0].data_ptr()),
                              c_void_p(st[1].data_ptr()), _opt(gamma), B, rows_per_img, C, int(batch_stats),
                              float(slope), _chk(out_dgamma, "dgamma"), _chk(out_dbeta, "dbeta"), _opt(gx),
                              c_void_p(ws.data_ptr()), ws.numel(), _stream()), "ali_bn_bwd")
    return out_dgamma, out_dbeta, gx


def plane_table_grad(g, g_ch, x0, x_ch, idx, idx_col, n_rows, out=None, table=None):
    """Embedding-table gradient from the gradient ``g`` [B,H,W,Cg] of the assembled planes ``x0`` [B,H,W,Cx]
    (include/ali_hip.h: ali_plane_table_grad).  Returns / fills ``out`` [n_rows, 256].  ``table`` [n_rows, 256]: take
    the plane values from the table instead of ``x0`` (which may then be None, or carry a Dropout2d mask)."""
    lib = _lib.load()
    B, H, W, Cg = g.shape
    if out is None:
        out = torch.empty(n_rows, 256, dtype=torch.float32, device=g.device)
    if not (idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous()):
        raise ValueError("idx must be a contiguous int32 CUDA tensor")
    _lib.check(lib.ali_plane_table_grad(_chk(g, "g"), Cg, g_ch, None if table is not None else _chk(x0, "x0"),
                                        0 if table is not None else x0.shape[3], x_ch,
                                        c_void_p(idx.data_ptr()), idx.shape[1], idx_col, B, H, W, n_rows, _chk(out, "out"),
                                        _opt(table, "table"), _stream()), "ali_plane_table_grad")
    return out


def col2im(contrib, ldc, bias, out, B, H, W, Hout, Wout, NC, ostride, R, S, stride, pad, act=ACT_NONE, slope=0.0):
    """Gather half of a scatter-form transposed convolution (include/ali_hip.h: ali_col2im)."""
    lib = _lib.load()
    _lib.check(lib.ali_col2im(_chk(contrib, "contrib"), ldc, _opt(bias, "bias"), _chk(out, "out"), B, H, W, Hout, Wout,
                              NC, ostride, R, S, stride, pad, act, float(slope), _stream()), "ali_col2im")
    return out


def tconv_scatter_ok(C, NC, R, S, stride):
    return bool(_lib.load().ali_tconv_scatter_ok(C, NC, R, S, stride))


def tconv_scatter(x, w_nc, bias, out, B, H, W, C, Hout, Wout, NC, ostride, R, S, stride, pad, act=ACT_NONE, slope=0.0):
    """Scatter-form transposed convolution to 1-2 channels in one launch (include/ali_hip.h: ali_tconv_scatter)."""
    lib = _lib.load()

    def go():
        _lib.check(lib.ali_tconv_scatter(_chk(x, "x"), _chk(w_nc, "w"), _opt(bias, "bias"), _ptr(out), B, H, W, C, Hout,
                                         Wout, NC, ostride, R, S, stride, pad, act, float(slope), _stream()),
                   "ali_tconv_scatter")
    _launch("tconv_scatter", (2.0 * B * H * W * C * NC * R * S, 4.0 * B * (H * W * C + Hout * Wout * NC), _NO_SHAPE), go)
    if _HOOK is not None and hasattr(_HOOK, "scatter"):
        _HOOK.scatter(x, w_nc, bias, out, (B, H, W, C, Hout, Wout, NC, ostride, R, S, stride, pad), act, slope)
    return out


def tconv_scatter_wgrad(big, small, sstride, dw, s_k, s_tap, B, P, Q, K, H, W, R, S, stride, pad):
    """Weight gradient of a strided ConvTranspose2d(64 -> 1) in one launch + slab fold (include/ali_hip.h:
    ali_tconv_scatter_wgrad); returns None when the shape / workspace does not allow it (caller: conv_bwd_weight)."""
    lib = _lib.load()
    need = int(lib.ali_tconv_scatter_wgrad_ws(B, P, Q, K, R, S, stride))
    ws = workspace(big.device)
    if need == 0 or ws.numel() < need:
        return None

    def go():
        _lib.check(lib.ali_tconv_scatter_wgrad(_chk(big, "big"), _ptr(small), sstride, _ptr(dw), s_k, s_tap, B, P, Q, K, H,
                                               W, R, S, stride, pad, c_void_p(ws.data_ptr()), ws.numel(), _stream()),
                   "ali_tconv_scatter_wgrad")
    _launch("tconv_scatter_wgrad", (2.0 * B * P * Q * K * R * S, 4.0 * B * (P * Q * K + H * W), _NO_SHAPE), go)
    return dw


def spect_post(y, B, T, F, out, mean=None, std=None, clip_k=3.0):
    lib = _lib.load()
    _lib.check(lib.ali_spect_post(_chk(y, "y"), B, T, F, _opt(mean, "mean"), _opt(std, "std"), float(clip_k),
                                  _chk(out, "out"), _stream()), "ali_spect_post")
    return out


def bce_logits(logit, target, gscale=1.0, want_grad=True):
    """returns (out2 = [loss, mean sigmoid] device tensor, glogit or None)."""
    lib = _lib.load()
    B = logit.numel()
    out2 = torch.empty(2, dtype=torch.float32, device=logit.device)
    gl = torch.empty_like(logit) if want_grad else None
    _lib.check(lib.ali_bce_logits(_chk(logit, "logit"), B, float(target), float(gscale), _chk(out2), _opt(gl),
                                  _stream()), "ali_bce_logits")
    return out2, gl


def bce_logits_pair(logit, B, target_a, target_b, gscale=1.0, want_grad=True):
    """BCE of two passes batched along the rows ([0,B) vs target_a, [B,2B) vs target_b) in one launch.
    Returns (out3 = [(loss_a+loss_b)/2, mean sigmoid(a), mean sigmoid(b)], glogit [2B,1] or None)."""
    lib = _lib.load()
    assert logit.numel() == 2 * B
    out3 = torch.empty(3, dtype=torch.float32, device=logit.device)
    gl = torch.empty_like(logit) if want_grad else None
    _lib.check(lib.ali_bce_logits_pair(_chk(logit, "logit"), B, float(target_a), float(target_b), float(gscale),
                                       _chk(out3), _opt(gl), _stream()), "ali_bce_logits_pair")
    return out3, gl


def _attr_arrays(tensors):
    """ctypes views of a list of [B, n] one-hot tensors (fp32 or int32, contiguous CUDA)."""
    n = len(tensors)
    for t in tensors:
        if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.int32) and t.dim() == 2):
            raise ValueError(f"categorical attribute: need a contiguous [B, n] fp32 / int32 CUDA tensor, got "
                             f"{t.dtype} {tuple(t.shape)}")
    ptrs = (c_void_p * max(n, 1))(*[t.data_ptr() for t in tensors])
    ncls = (ctypes.c_int32 * max(n, 1))(*[t.shape[1] for t in tensors])
    isint = (ctypes.c_int32 * max(n, 1))(*[int(t.dtype == torch.int32) for t in tensors])
    return ptrs, ncls, isint


def attr_pack(cats, conts, B, device):
    """arg-max class of every categorical attribute -> idx [B, n_cat] int32; continuous attributes ([B] / [B,1] fp32)
    -> cont [B, n_cont] (or None).  One launch (include/ali_hip.h: ali_attr_pack)."""
    lib = _lib.load()
    ptrs, ncls, isint = _attr_arrays(cats)
    idx = torch.empty(B, max(len(cats), 1), dtype=torch.int32, device=device)
    for t in conts:
        _chk(t, "continuous attribute")
    cptr = (c_void_p * max(len(conts), 1))(*[t.data_ptr() for t in conts])
    cont = torch.empty(B, len(conts), dtype=torch.float32, device=device) if conts else None
    _lib.check(lib.ali_attr_pack(ptrs, ncls, isint, len(cats), cptr, len(conts), B, c_void_p(idx.data_ptr()),
                                 None if cont is None else c_void_p(cont.data_ptr()), _stream()), "ali_attr_pack")
    return idx, cont


def g_input(z, onehots, tables, cont, ld, out=None):
    """Generator input rows [B, ld] = [z | onehot_j @ table_j | cont | 0] in one launch (ali_g_input)."""
    lib = _lib.load()
    B, zdim = z.shape
    ptrs, ncls, isint = _attr_arrays(onehots)
    for t in tables:
        _chk(t, "embedding table")
    tptr = (c_void_p * max(len(tables), 1))(*[t.data_ptr() for t in tables])
    if out is None:
        out = torch.empty(B, ld, dtype=torch.float32, device=z.device)
    _lib.check(lib.ali_g_input(_chk(z, "z"), zdim, ptrs, ncls, isint, tptr, len(tables), _opt(cont, "cont"),
                               0 if cont is None else cont.shape[1], B, ld, _chk(out), _stream()), "ali_g_input")
    return out


def g_input_table_grad(onehot, g, off, out):
    """out [n_classes, 256] = onehot^T @ g[:, off:off+256]  (ali_g_input_table_grad)."""
    lib = _lib.load()
    ptrs, ncls, isint = _attr_arrays([onehot])
    _lib.check(lib.ali_g_input_table_grad(c_void_p(onehot.data_ptr()), isint[0], ncls[0], _chk(g, "g"), g.shape[1], off,
                                          g.shape[0], _chk(out, "out"), _stream()), "ali_g_input_table_grad")
    return out


def adam(p, g, m, v, lr, beta1, beta2, eps, step, dev_step=None, grad_scale=1.0, arrive=None, p16=None):
    """``arrive`` (zeroed int32 device tensor): ``dev_step`` counts COMPLETED steps and the launch advances it itself;
    ``p16`` (fp16, same numel): the launch also leaves the fp16 twin of the updated parameters (include/ali_hip.h: ali_adam)."""
    lib = _lib.load()
    ds = ar = None
    if dev_step is not None:
        assert dev_step.is_cuda and dev_step.dtype == torch.int32
        ds = c_void_p(dev_step.data_ptr())
    if arrive is not None:
        assert arrive.is_cuda and arrive.dtype == torch.int32 and dev_step is not None
        ar = c_void_p(arrive.data_ptr())
    _lib.check(lib.ali_adam(_chk(p, "p"), _chk(g, "g"), _chk(m, "m"), _chk(v, "v"), p.numel(), lr, beta1, beta2, eps,
                            step, ds, ar, grad_scale, None if p16 is None else c_void_p(p16.data_ptr()), _stream()),
               "ali_adam")


def add_i64_multi(counters, incs):
    """counters[i] += incs[i] for int64 device scalars, one launch (include/ali_hip.h: ali_add_i64_multi)"""
    n = len(counters)
    if n == 0:
        return
    for t in counters:
        if not (t.is_cuda and t.dtype == torch.int64 and t.numel() == 1):
            raise ValueError("add_i64_multi: need one-element int64 CUDA tensors")
    ptrs = (c_void_p * n)(*[t.data_ptr() for t in counters])
    inc = (ctypes.c_int64 * n)(*[int(v) for v in incs])
    _lib.check(_lib.load().ali_add_i64_multi(n, ptrs, inc, _stream()), "ali_add_i64_multi")


def assemble_planes(X, idx, tables, cont, B, H, W, Cpad, out=None, mask=None):
    """X [B,H,W] fp32; idx [B,n_emb] int32; tables: list of [n,256] fp32; cont [B,n_cont] or None.
    ``out``: optional contiguous [B,H,W,Cpad] destination (e.g. one half of a batched-pass buffer).
    ``mask`` [B, >= Cpad] (any row stride): multiply the planes by it per (sample, channel)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(B, H, W, Cpad, dtype=torch.float32, device=X.device)
    n_emb = len(tables)
    arr = (c_void_p * max(n_emb, 1))(*[t.data_ptr() for t in tables])
    for t in tables:
        _chk(t, "embedding table")
    if idx is not None and not (idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous()):
        raise ValueError("idx must be a contiguous int32 CUDA tensor")
    n_cont = 0 if cont is None else cont.shape[1]
    _lib.check(lib.ali_assemble_planes(_chk(X, "X"), None if idx is None else c_void_p(idx.data_ptr()),
                                       ctypes.cast(arr, ctypes.POINTER(c_void_p)), n_emb, _opt(cont), n_cont,
                                       _chk(out), B, H, W, Cpad, None if mask is None else _ptr(mask),
                                       0 if mask is None else mask.stride(0), _stream()), "ali_assemble_planes")
    return out
