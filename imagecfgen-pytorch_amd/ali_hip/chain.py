"""Fused execution of an ``nn.Sequential`` conv stack on the HIP kernels.

A stack (``Encoder.layers``, ``Generator.layers``, ``Discriminator.dx/dz/dxz``;
reference image_scms/mnist.py:30-40,63-74,98-136) is parsed once into *stages*

    [Dropout2d] [BatchNorm2d] [Dropout2d]  ->  Conv2d | ConvTranspose2d | Linear(+Unflatten)  ->  LeakyReLU | Tanh

and executed as ONE autograd node: forward runs the NHWC kernels stage by stage,
backward is hand scheduled -- the activation derivative of stage i-1 (and the
Dropout2d mask in front of stage i) is folded into the epilogue of stage i's
data-gradient GEMM, BatchNorm backward is fused with the LeakyReLU derivative.
The ``nn`` modules only own the parameters (reference layouts, reference
``state_dict`` keys); their own ``forward`` is never called on CUDA tensors.
"""
import weakref
from typing import List, Optional

import torch
import torch.nn as nn

from . import dropout as _dropout
from . import ops
from .ops import ACT_LEAKY, ACT_NONE, ACT_TANH


def _pad4(c: int) -> int:
    return (c + 3) // 4 * 4


class Stage:
    __slots__ = ("kind", "mod", "act", "slope", "pre", "unflat", "index")

    def __init__(self, kind, mod, pre, index):
        self.kind, self.mod, self.pre, self.index = kind, mod, pre, index
        self.act, self.slope, self.unflat = ACT_NONE, 0.0, None


class PackCache:
    """Kernel-layout copies of one parameter (the reference layouts stay the master copies: Conv
    [Cout,Cin,kh,kw], ConvT [Cin,Cout,kh,kw], Linear [out,in]).

    A copy is rebuilt when the parameter's version / storage changed since it was built (``load_state_dict``, a
    ``torch.optim`` step, any in-place edit bumps the version).  Static mode (AliStepper, graph capture): the rebuild
    happens IN PLACE so buffers keep their addresses, and ``refresh()`` rewrites every copy right after the optimiser
    kernel that changed the parameters (that kernel works on raw pointers and does not bump versions)."""

    def __init__(self):
        self.store = {}
        self.static = False

    @staticmethod
    def _tag(param):
        return (param.data_ptr(), param._version, tuple(param.shape))

    def get(self, key, param: torch.Tensor, builder):
        hit = self.store.get(key)
        tag = self._tag(param)
        if hit is not None and hit[0] == tag:
            return hit[1]
        with torch.no_grad():
            val = builder(hit[1] if (hit is not None and self.static) else None)
            ops.after_packs(lambda v=val: ops.refresh_shadow16(v))   # the fp16 twin (precision "f16"), if it has one
        self.store[key] = (tag, val, builder, param)
        return val

    def refresh(self):
        with torch.no_grad():
            for key, (tag, val, builder, param) in list(self.store.items()):
                out = builder(val)
                assert out.data_ptr() == val.data_ptr()
                ops.after_packs(lambda v=val: ops.refresh_shadow16(v))
                self.store[key] = (self._tag(param), val, builder, param)


class ChainPlan:
    def __init__(self, seq: nn.Sequential):
        self.seq = weakref.ref(seq)          # the plan lives in a WeakKeyDictionary keyed by seq: no strong cycle
        self.stages: List[Stage] = []
        self.cache = PackCache()
        pre = []
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            name = m.__class__.__name__
            if isinstance(m, nn.Dropout2d) or name in ("Dropout2d", "TapedDropout2d"):
                pre.append(("drop", float(m.p)))
            elif isinstance(m, nn.BatchNorm2d):
                pre.append(("bn", m))
            elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
                kind = "convT" if isinstance(m, nn.ConvTranspose2d) else ("conv" if isinstance(m, nn.Conv2d)
                                                                          else "linear")
                st = Stage(kind, m, pre, len(self.stages))
                pre = []
                if kind == "linear" and i + 1 < len(mods) and isinstance(mods[i + 1], nn.Unflatten):
                    st.unflat = tuple(mods[i + 1].unflattened_size)
                    i += 1
                if i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU):
                    st.act, st.slope = ACT_LEAKY, float(mods[i + 1].negative_slope)
                    i += 1
                elif i + 1 < len(mods) and isinstance(mods[i + 1], nn.Tanh):
                    st.act = ACT_TANH
                    i += 1
                self.stages.append(st)
            else:
                raise NotImplementedError(f"ali_hip.chain: unsupported layer {name}")
            i += 1
        if pre:
            raise NotImplementedError("trailing Dropout2d/BatchNorm2d without a convolution")
        for st in self.stages:
            kinds = [p[0] for p in st.pre]
            if kinds not in ([], ["drop"], ["bn"], ["drop", "bn"], ["bn", "drop"]):
                raise NotImplementedError(f"unsupported pre-op pattern {kinds}")

    def params(self) -> List[torch.Tensor]:
        out = []
        for st in self.stages:
            for kind, m in st.pre:
                if kind == "bn":
                    out += [m.weight, m.bias]
            out.append(st.mod.weight)
            if st.mod.bias is not None:
                out.append(st.mod.bias)
        return out

    # ------------------------------------------------------------------ packing
    def packed(self, st: Stage, which, cin_stride: int):
        """which = 'fwd' | 'dgrad' kernel-layout weights of a stage (see include/ali_hip.h); 'scatter' and
        ('scatter_dgrad', planes): the 1x1 GEMM weights of the scatter-form transposed convolutions (ali_col2im)."""
        w = st.mod.weight
        dev = w.device

        def build(dst=None):
            def buf(*shape, zero=False):
                if dst is not None:
                    return dst.reshape(shape)
                return (torch.zeros if zero else torch.empty)(*shape, device=dev)
            if which == "scatter":            # ConvT [Ci][Co][R][S] -> rows n = tap*Co + co, columns ci (pad: 0)
                Ci, Co, R, S = w.shape
                out = buf(R * S * Co, 1, cin_stride, zero=True)
                # as a pack: "row" tap, "tap" co, channel ci -- joins the batched re-pack launch of an optimiser step
                ops.pack_weights(w.detach(), out, R * S, Co, Ci, cin_stride, 1, R * S, Co * R * S)
                return out
            if isinstance(which, tuple) and which[0] == "scatter_dgrad":   # Conv [K][C][R][S] -> rows tap*NP + j, cols k
                K, C, R, S = w.shape
                if len(which[1]) == 1:        # one plane: rows = taps, channel k; source = the plane's [K][.][R][S] slice
                    out = buf(R * S, 1, K)
                    src = w.detach().reshape(-1)[which[1][0] * R * S:]
                    ops.pack_weights(src, out, R * S, 1, K, K, 1, 0, C * R * S)
                    return out
                sel = torch.stack([w.detach()[:, c] for c in which[1]], dim=0)          # [NP][K][R][S], slices only
                out = buf(R * S * len(which[1]), 1, K)
                out[:, 0, :].copy_(sel.permute(2, 3, 0, 1).reshape(R * S * len(which[1]), K))
                return out
            if w.dim() == 4:
                src = _storage_view(w)     # (weights may live in the forward pack's order already: FlatGroup.layouts)
                s0, s1, _, s3 = w.stride()
            if st.kind == "conv":
                K, C, R, S = w.shape
                T = R * S
                if which == "fwd":      # [K][T][Cpad]
                    return ops.pack_weights(src, buf(K, T, cin_stride), K, T, C, cin_stride, s0, s3, s1)
                out = buf(cin_stride, T, K, zero=True)             # [Cpad][T][K]; rows >= C stay zero
                ops.pack_weights(src, out, C, T, K, K, s1, s3, s0)
                return out
            if st.kind == "convT":
                Ci, Co, R, S = w.shape
                T = R * S
                if which == "fwd":      # convT forward == data-gradient GEMM: [Co][T][Ci_pad]
                    return ops.pack_weights(src, buf(Co, T, cin_stride), Co, T, Ci, cin_stride, s1, s3, s0)
                # convT dgrad == conv forward GEMM: [Ci_pad][T][Co]; rows >= Ci stay zero
                out = buf(cin_stride, T, Co, zero=True)
                ops.pack_weights(src, out, Ci, T, Co, Co, s0, s3, s1)
                return out
            # linear (+Unflatten(C,h,w)): 1x1 conv whose output channel n' = t*C + co is NHWC [B,h,w,C]
            O, I = w.shape
            Cc, hh, ww = st.unflat if st.unflat else (O, 1, 1)
            T = hh * ww
            if which == "fwd":
                return ops.pack_weights(w.detach(), buf(O, 1, cin_stride), T, Cc, I, cin_stride, I, T * I, 1)
            fwd = self.packed(st, "fwd", cin_stride)
            out = buf(cin_stride, 1, O, zero=True)                 # [I_pad][1][O] = transpose of the fwd pack
            ops.pack_weights(fwd, out, I, 1, O, O, 1, 0, cin_stride)
            return out

        twin = getattr(w, "_ali_flat16", None)
        if which == "fwd" and (not ops._PRECISION["f16"] or twin is not None) and w.dim() == 4 \
                and tuple(w.stride()) == _fwd_pack_strides(st) \
                and cin_stride == (w.shape[1] if st.kind == "conv" else w.shape[0]):
            # the master weights ARE the forward pack (FlatGroup.layouts): no copy to refresh.  fp16-MFMA path: the Adam
            # launch keeps an fp16 twin of the flat parameter buffer (FlatGroup.flat16); its segment is the pack's twin
            n_out = w.shape[0] if st.kind == "conv" else w.shape[1]
            alias = _storage_view(w).view(n_out, w.shape[2] * w.shape[3], cin_stride)
            if ops._PRECISION["f16"]:
                alias._ali16 = twin.view(n_out, w.shape[2] * w.shape[3], cin_stride)
            return alias
        val = self.cache.get((st.index, which, cin_stride), w, build)
        if ops._PRECISION["f16"] and which in ("fwd", "dgrad"):
            with torch.no_grad():
                ops.ensure_shadow16(val)         # fp16 twin of the packed weights, re-rounded whenever they are re-packed
        return val

    def packed_bias(self, st: Stage):
        b = st.mod.bias
        if b is None:
            return None
        if st.kind != "linear" or not st.unflat:
            return b.detach()
        Cc, hh, ww = st.unflat
        def build(dst=None):
            val = b.detach().reshape(Cc, hh * ww).t().contiguous().reshape(-1)
            if dst is not None:
                dst.copy_(val)
                return dst
            return val
        return self.cache.get((st.index, "bias"), b, build)


# ---------------------------------------------------------------------- shapes
def _out_shape(st: Stage, B, H, W, C):
    m = st.mod
    if st.kind == "conv":
        R, S = m.kernel_size
        s, p = m.stride[0], m.padding[0]
        return B, (H + 2 * p - R) // s + 1, (W + 2 * p - S) // s + 1, m.out_channels
    if st.kind == "convT":
        R, S = m.kernel_size
        s, p, op = m.stride[0], m.padding[0], m.output_padding[0]
        return B, (H - 1) * s - 2 * p + R + op, (W - 1) * s - 2 * p + S + op, m.out_channels
    if st.unflat:
        Cc, hh, ww = st.unflat
        return B, hh, ww, Cc
    return B, 1, 1, m.out_features


def _geom(st: Stage, xin_shape, out_shape):
    """AliConvGeom of the stage's equivalent Conv2d (ConvT: roles of x / y swapped)."""
    B, H, W, C = xin_shape
    _, P, Q, K = out_shape
    m = st.mod
    if st.kind == "conv":
        return ops.geom(B, H, W, C, P, Q, K, m.kernel_size[0], m.kernel_size[1], m.stride[0], m.padding[0])
    if st.kind == "convT":   # x := convT output, y := convT input
        return ops.geom(B, P, Q, K, H, W, C, m.kernel_size[0], m.kernel_size[1], m.stride[0], m.padding[0])
    # linear: 1x1 conv on the [B,1,1,I] map with O output channels
    return ops.geom(B, 1, 1, C, 1, 1, P * Q * K, 1, 1, 1, 0)


def _is_tconv1(st: Stage, cin_stride: int) -> bool:
    """ConvTranspose2d(C -> 1), stride 1, no output padding: served by the direct one-channel kernels."""
    m = st.mod
    return (st.kind == "convT" and m.out_channels == 1 and m.stride[0] == 1 and m.output_padding[0] == 0
            and m.kernel_size[0] <= 5 and cin_stride in (32, 64, 128, 256) and st.act in (ACT_NONE, ACT_LEAKY, ACT_TANH))


def _is_head(st: Stage, in_shape) -> bool:
    """Conv2d(C, 1, 1) on a 1x1 map without activation (the Discriminator's last layer, mnist.py:127): a GEMV"""
    m = st.mod
    return (st.kind == "conv" and m.out_channels == 1 and tuple(m.kernel_size) == (1, 1) and m.stride[0] == 1
            and m.padding[0] == 0 and in_shape[1] == 1 and in_shape[2] == 1 and in_shape[3] % 4 == 0
            and st.act == ACT_NONE)


def _scatter_fwd(st: Stage, cin_stride: int) -> bool:
    """ConvTranspose2d with one or two output channels that the direct kernels do not cover (stride 2 Generator tails of
    the spectrogram models): per-input-pixel tap contributions by a 1x1 GEMM with N = Cout*R*S columns, then
    ali_col2im -- an implicit GEMM over the output pixels would use 1/32 of every MFMA tile."""
    m = st.mod
    return (st.kind == "convT" and m.out_channels <= 2 and m.out_channels * m.kernel_size[0] * m.kernel_size[1] <= 64
            and cin_stride % 4 == 0)


def _scatter_dgrad(st: Stage, planes) -> bool:
    """The few consumed input planes of a first Conv2d's data gradient, same scatter form.  The contribution tensor
    (planes * taps floats per pixel, written and read once) makes it HBM bound: measured break-even with the implicit
    GEMM at about 6 planes of a 5x5 filter, a clear win below."""
    m = st.mod
    return (st.kind == "conv" and planes is not None and 1 <= len(planes) <= 8 and m.out_channels % 4 == 0
            and len(planes) * m.kernel_size[0] * m.kernel_size[1] <= 128)


def _first_conv_direct(st: Stage, c_in_log: int) -> bool:
    """First Conv2d of a stack (<= 8 real input channels, stride 1): per-channel direct weight gradient and
    single-plane data gradient (ali_tconv1_*)."""
    m = st.mod
    return (st.kind == "conv" and m.stride[0] == 1 and c_in_log <= 8 and m.out_channels in (32, 64, 128, 256)
            and m.kernel_size[0] <= 5)


_NBT = {"pending": None}


def _count_batch(bn):
    """num_batches_tracked += 1 (nn.BatchNorm2d training forward).  Inside a stepper iteration the increments are
    collected and applied by one multi-tensor launch (``flush_batch_counts``) instead of one launch per forward."""
    pend = _NBT["pending"]
    if pend is None:
        bn.num_batches_tracked += 1
    else:
        ent = pend.setdefault(id(bn), [bn.num_batches_tracked, 0])
        ent[1] += 1


def defer_batch_counts():
    _NBT["pending"] = {}


def drop_pending_batch_counts():
    if _NBT["pending"] is not None:
        _NBT["pending"] = {}


def abort_batch_counts():
    _NBT["pending"] = None


def flush_batch_counts(extra=()):
    """apply the collected num_batches_tracked increments (+ ``extra`` = [(int64 counter tensor, increment), ...]) in
    one launch"""
    pend, _NBT["pending"] = _NBT["pending"], None
    jobs = [(e[0], e[1]) for e in (pend or {}).values()] + list(extra)
    cuda = [(t, k) for t, k in jobs if t.is_cuda]
    for t, k in jobs:
        if not t.is_cuda:
            t += k
    if cuda:
        ops.add_i64_multi([t.reshape(1) if t.dim() == 0 else t for t, _ in cuda], [k for _, k in cuda])


class _Saved:
    __slots__ = ("x_in", "t", "y", "mask", "bn_stats", "bn", "pattern", "geom", "in_shape", "out_shape", "training")


def slice_saved(saved, group: int, groups: int):
    """The saved state of one of the ``groups`` passes that ``chain_forward(..., groups=...)`` ran as one batch
    (contiguous row ranges of every activation, that pass's BatchNorm statistics)."""
    out = []
    for sv in saved:
        B = sv.in_shape[0] // groups
        lo, hi = group * B, (group + 1) * B
        s2 = _Saved()
        s2.x_in, s2.t, s2.y = sv.x_in[lo:hi], sv.t[lo:hi], sv.y[lo:hi]
        s2.mask = None if sv.mask is None else sv.mask[lo:hi]
        s2.bn, s2.pattern, s2.training = sv.bn, sv.pattern, sv.training
        s2.bn_stats = None if sv.bn_stats is None else (sv.bn_stats[group] if isinstance(sv.bn_stats, list)
                                                        else sv.bn_stats)
        s2.in_shape, s2.out_shape = (B,) + tuple(sv.in_shape[1:]), (B,) + tuple(sv.out_shape[1:])
        g = sv.geom
        s2.geom = ops.geom(B, g.H, g.W, g.C, g.P, g.Q, g.K, g.R, g.S, g.stride, g.pad)
        out.append(s2)
    return out


def _fwd_pack_strides(st: Stage):
    """Strides that make a 4-d conv weight's storage order the forward GEMM's pack ([N][R*S][Cin]: for a Conv2d weight
    torch's channels_last), or None when the stage's forward pack is not a permutation of the weight (padded input
    channels, one-channel / scatter tails, Linear)."""
    w = st.mod.weight
    if w.dim() != 4 or st.index == 0:      # (first layers: padded planes, direct / scatter kernels with their own packs)
        return None
    if st.kind == "conv":
        K, C, R, S = w.shape
        return (R * S * C, 1, S * C, C) if C % 4 == 0 else None
    if st.kind == "convT":
        Ci, Co, R, S = w.shape
        if Ci % 32 != 0 or _is_tconv1(st, Ci) or _scatter_fwd(st, Ci):
            return None
        return (1, R * S * Ci, S * Ci, Ci)
    return None


def pack_layouts(plans):
    """{id(weight): strides} for FlatGroup(layouts=...): every conv weight whose forward pack is a permutation of it"""
    out = {}
    for pl in plans:
        for st in pl.stages:
            strides = _fwd_pack_strides(st)
            if strides is not None:
                out[id(st.mod.weight)] = strides
    return out


def _storage_view(w):
    """the dense storage range of a (possibly permuted) parameter as a contiguous 1-d tensor"""
    return torch.as_strided(w.detach(), (w.numel(),), (1,))


def join_ok(plan: ChainPlan) -> bool:
    """Can the chain's last stage write its output straight into a column range of a wider row-major buffer
    (``chain_forward(join=...)``) and take its output gradient from one (``chain_backward(gy_ld=...)``)?  A plain
    Conv2d GEMM ending in no activation / LeakyReLU (the ends of D.dx and D.dz, mnist.py:116-123)."""
    st = plan.stages[-1]
    return (len(plan.stages) > 1 and st.kind == "conv" and st.act in (ACT_NONE, ACT_LEAKY)
            and st.mod.out_channels % 4 == 0 and st.mod.in_channels % 4 == 0)


def drive(gen):
    """run a chain generator to its end on its own (every GEMM is launched where it is requested)"""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value


def run_parallel(*gens):
    """Advance independent chain generators in lock step, one GEMM each per round, and issue the GEMMs of a round as
    ONE multi-job launch per kernel variant (``ops.gemm_batch``).  A chain generator (``chain_forward_gen``,
    ``chain_backward_gen``, or a generator that ``yield from``-s several of them) yields right after each GEMM request
    and before anything that reads its result, so whatever else it launches between two yields only depends on GEMMs
    of earlier rounds.  The generators must not depend on each other.  Returns their return values."""
    res = [None] * len(gens)
    live = list(range(len(gens)))
    while live:
        nxt = []
        with ops.gemm_batch():
            for i in live:
                try:
                    next(gens[i])
                    nxt.append(i)
                except StopIteration as e:
                    res[i] = e.value
        live = nxt
    return res


def delayed(gen, rounds: int):
    """``gen`` entering ``run_parallel`` ``rounds`` rounds late (pairs a short chain's GEMMs with later, larger layers of
    its partner instead of with the partner's first one)"""
    for _ in range(rounds):
        yield
    return (yield from gen)


def chain_forward(*args, **kwargs):
    """``chain_forward_gen`` run on its own: returns (y_last, saved list)."""
    return drive(chain_forward_gen(*args, **kwargs))


def chain_backward(*args, **kwargs):
    """``chain_backward_gen`` run on its own: returns (gx or None, {param tensor id -> grad})."""
    return drive(chain_backward_gen(*args, **kwargs))


def chain_forward_gen(plan: ChainPlan, x: torch.Tensor, training: bool, c_log_in: int, save: bool, groups: int = 1,
                      join=None, first_mask_applied: bool = False, lane=None, alloc=None):
    """Generator form of the forward pass (see ``run_parallel``): yields after every GEMM request.
    x: NHWC [B,H,W,Cp] fp32 CUDA.  Returns (y_last, saved list).

    ``join`` = (joint [B, Ctot] fp32, column offset, mask [B, Ctot] or None): the last stage (``join_ok``; output map
    1x1) writes act(conv) * mask[:, off:off+K] into joint[:, off:off+K] instead of a tensor of its own -- the
    concatenation and the Dropout2d in front of the consuming chain cost no launch (mnist.py:152-154).
    ``first_mask_applied``: the input already carries the first stage's Dropout2d mask (it was folded into the
    producers that way); the mask is still drawn, in order, and saved for the backward pass.
    ``lane`` (dropout.Lane): the chain's Dropout2d masks are the requests of that lane (chains advanced side by side).
    ``alloc(tag, shape)``: where the stage outputs live instead of fresh tensors (persistent buffers of a forward pass
    that is computed ahead of its iteration, AliStepper.pipeline_reduce).

    ``groups`` > 1: the batch holds that many independent forward passes back to back (equal sample counts).  The
    convolutions run once over all of them; BatchNorm takes its batch statistics -- and updates the running ones --
    pass by pass, in order, exactly as separate calls would."""
    saved = []
    cur = x
    c_log = c_log_in
    folded = None        # mask of the coming stage, already applied by the previous stage's GEMM epilogue
    early = None         # mask of the coming stage, requested early (its BatchNorm statistics needed it), not applied
    bn_part = None       # (partials, slots) of the coming stage's BatchNorm, left by the previous stage's GEMM epilogue
    for si, st in enumerate(plan.stages):
        B, H, W, Cp = cur.shape
        rows = H * W
        kinds = [p[0] for p in st.pre]
        mask = None
        bn = None
        mask_applied = folded is not None or (si == 0 and first_mask_applied)
        if folded is not None:
            mask, folded = folded, None
        elif early is not None:
            mask, early = early, None
        for kind, arg in st.pre:
            if kind == "drop" and training and mask is None:
                mask = _dropout.next_mask(B, c_log, arg, cur.device, Cp, lane=lane)
            elif kind == "bn":
                bn = arg
        sv = _Saved()
        sv.x_in, sv.mask, sv.bn, sv.training = cur, mask, bn, training
        sv.pattern = kinds
        sv.bn_stats = None
        t = cur
        if bn is not None:
            mask_in = mask if kinds == ["drop", "bn"] else None
            mask_post = mask if kinds == ["bn", "drop"] else None
            use_batch = training or bn.running_mean is None
            momentum = bn.momentum if bn.momentum is not None else 0.1
            if bn_part is not None:     # batch statistics from the partial sums the producing conv left behind
                part, slots = bn_part
                bn_part = None
                stg = ops.bn_stats_from_partials(part, slots, groups, Cp, (B // groups) * rows, bn.weight.detach(),
                                                 bn.bias.detach(), bn.running_mean, bn.running_var, momentum, bn.eps)
            else:                       # per-pass statistics / running-stat updates, one launch for all passes
                stg = ops.bn_stats(cur, mask_in, B, rows, Cp, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                   bn.running_var, momentum, bn.eps, use_batch, groups=groups)
            if use_batch and bn.num_batches_tracked is not None:
                for _ in range(groups):
                    _count_batch(bn)
            t = ops.bn_apply(cur, stg, mask_in, mask_post, B, rows, Cp, groups=groups)
            sv.bn_stats = stg if groups == 1 else [stg[gi] for gi in range(groups)]
        elif mask is not None and not mask_applied:
            t = ops.rowmask_mul(cur, mask, B, rows, Cp)
        out_shape = _out_shape(st, B, H, W, Cp)
        g = _geom(st, (B, H, W, Cp), out_shape)
        nxt = plan.stages[si + 1] if si + 1 < len(plan.stages) else None
        out_ld = 0
        jmask = None
        if nxt is None and join is not None:
            joint, joff, jm = join
            if out_shape[1] != 1 or out_shape[2] != 1 or joint.shape[0] != B or not join_ok(plan):
                raise ValueError("chain_forward(join=...): the last stage must be a plain conv GEMM onto a 1x1 map")
            out_ld = joint.shape[1]
            y = joint[:, joff:joff + out_shape[3]].unflatten(1, (1, 1, out_shape[3]))     # [B,1,1,K] view, rows out_ld apart
            jmask = None if jm is None else jm[:, joff:joff + out_shape[3]]
        else:
            y = (alloc(("y", si), out_shape) if alloc is not None
                 else torch.empty(out_shape, dtype=torch.float32, device=cur.device))
        nk = [p[0] for p in nxt.pre] if nxt is not None else []
        plain_gemm = st.kind in ("conv", "convT") and not _is_tconv1(st, Cp) and not _scatter_fwd(st, Cp)
        # A lone Dropout2d in front of the next stage multiplies this stage's output by a per-(sample, channel)
        # mask: the GEMM epilogue does it (act(.)*mask), so y is stored masked.  LeakyReLU'(y) only needs the sign
        # of y, which the kept entries preserve and the dropped ones do not need (their gradient is masked to 0).
        if (training and nk == ["drop"] and st.act in (ACT_NONE, ACT_LEAKY) and st.kind in ("conv", "convT")
                and not _is_tconv1(st, Cp)):
            folded = _dropout.next_mask(B, out_shape[3], nxt.pre[0][1], cur.device, out_shape[3], lane=lane)
        # A BatchNorm behind this conv: its batch statistics are column sums of this stage's output -- accumulated
        # per M-tile by the GEMM epilogue (times the Dropout2d mask that may sit in between), no extra pass
        bn_fwd = None
        if "bn" in nk and st.kind == "conv" and plain_gemm and folded is None:
            nbn = [a for k, a in nxt.pre if k == "bn"][0]
            if training or nbn.running_mean is None:
                if training and nk == ["drop", "bn"]:
                    early = _dropout.next_mask(B, out_shape[3], nxt.pre[0][1], cur.device, out_shape[3], lane=lane)
                slots, tile_rows, pixel_major = ops.conv_mtiles(g, 0)
                rows_g = (B // groups) * (1 if pixel_major else out_shape[1] * out_shape[2])
                if slots > 0 and (groups == 1 or (B % groups == 0 and rows_g % tile_rows == 0
                                                  and (not pixel_major or B % tile_rows == 0))):
                    part = torch.empty(2 * out_shape[3] * slots, dtype=torch.float32, device=cur.device)
                    bn_part = (part, slots)
                    bn_fwd = (part, groups, early, slots)
        ep = ops.epilogue(bias=plan.packed_bias(st), act=st.act, slope=st.slope, mask=folded, bn_fwd=bn_fwd)
        if jmask is not None:
            ep.mask, ep.mask_ld = jmask.data_ptr(), jmask.stride(0)
            ep.refs["mask"] = jmask
        if si == 0 and st.kind == "conv" and c_log < Cp:
            ep.in_ch_live = c_log          # channel padding of a first layer: kernels that can skip it do
        # (the register-blocked tconv1_fwd beats the scatter form on the MNIST tail by 17 us per launch; the scatter
        # form serves the stride-2 / two-channel tails of the spectrogram Generators)
        if _is_head(st, (B, H, W, Cp)) and folded is None and bn_fwd is None and not out_ld:
            ops.head_fwd(t.reshape(B, Cp), plan.packed(st, "fwd", Cp).reshape(-1), plan.packed_bias(st), y.reshape(B))
        elif _scatter_fwd(st, Cp) and folded is None and not _is_tconv1(st, Cp):
            m = st.mod
            R, S = m.kernel_size
            Co = m.out_channels
            if ops.tconv_scatter_ok(Cp, Co, R, S, m.stride[0]) and t.is_contiguous():
                # contributions kept in LDS per output tile: one launch, no [pixels][taps] tensor
                ops.tconv_scatter(t, plan.packed(st, "scatter", Cp), plan.packed_bias(st), y, B, H, W, Cp, out_shape[1],
                                  out_shape[2], Co, Co, R, S, m.stride[0], m.padding[0], st.act, st.slope)
            else:
                contrib = (alloc(("contrib", si), (B, H, W, Co * R * S)) if alloc is not None
                           else torch.empty(B, H, W, Co * R * S, dtype=torch.float32, device=cur.device))
                ops.conv_fwd(ops.geom(B, H, W, Cp, H, W, Co * R * S, 1, 1, 1, 0), t, plan.packed(st, "scatter", Cp),
                             contrib, ops.epilogue(), live=(c_log, None))
                yield
                ops.col2im(contrib, Co * R * S, plan.packed_bias(st), y, B, H, W, out_shape[1], out_shape[2], Co, Co, R,
                           S, m.stride[0], m.padding[0], st.act, st.slope)
        elif _is_tconv1(st, Cp):
            m = st.mod
            ops.tconv1_fwd(t, plan.packed(st, "fwd", Cp), plan.packed_bias(st), y, B, H, W, Cp, m.kernel_size[0],
                           m.kernel_size[1], m.padding[0], 1, st.act, st.slope)
        elif st.kind == "convT":
            ops.conv_bwd_data(g, t, plan.packed(st, "fwd", Cp), y, ep, live=(None, c_log))
            yield
        else:
            ops.conv_fwd(g, t, plan.packed(st, "fwd", Cp), y, ep, out_ld=out_ld, live=(c_log, None))
            yield
        sv.t, sv.y, sv.geom, sv.in_shape, sv.out_shape = t, y, g, (B, H, W, Cp), out_shape
        if save:
            saved.append(sv)
        cur = y
        c_log = out_shape[3]
    return cur, saved


def wgrad_geoms(plan: ChainPlan, saved):
    """geometries of the weight-gradient GEMMs chain_backward(plan, saved, ..., need_params=True) launches"""
    return [sv.geom for st, sv in zip(plan.stages, saved) if not _is_tconv1(st, sv.in_shape[3])]


def chain_backward_gen(plan: ChainPlan, saved, gy: torch.Tensor, c_log_in: int, need_gx: bool, need_params: bool = True,
                       grad_dst=None, gx_planes=None, gy_ld: int = 0, gy_pre: bool = False, in_act=None, fold=None):
    """Generator form of the backward pass (see ``run_parallel``): yields after every data-gradient GEMM request (the
    weight gradients are deferred to ``fold`` or launched at once: nothing in the chain reads them).
    Returns (gx or None, {param tensor id -> grad}).  ``grad_dst`` optionally maps id(param) to a
    preallocated destination (a view of a flat gradient buffer) that the kernels write directly.
    ``gx_planes`` (hand-scheduled step only): instead of the full input gradient return only these input
    channels of it, as a [B,H,W,len(gx_planes)] tensor -- the first layer's data gradient is consumed one plane
    at a time (image plane towards G, embedding plane towards the digit table).
    ``gy_ld`` > 0 (needs ``join_ok``): ``gy`` is a column range (strided view, rows ``gy_ld`` floats apart) of a wider
    buffer; ``gy_pre``: it already is the gradient of the last stage's PRE-activation.  ``in_act`` = (act, slope) of the
    activation that produced this chain's input ``x`` (the ends of the chains whose outputs were joined): the returned
    gradient is then the one of their pre-activations, computed by the first stage's data-gradient epilogue.
    ``fold`` (ops.FoldQueue): the weight-gradient launches leave their slab reductions to ``fold.flush()`` -- the
    parameter gradients are complete only after the caller has flushed."""
    grads = {}
    grad_dst = grad_dst or {}
    n = len(plan.stages)
    last = plan.stages[-1]
    if gy_ld:
        if not (gy_pre or last.act == ACT_NONE) or not join_ok(plan):
            raise ValueError("chain_backward(gy_ld=...): needs join_ok(plan) and a pre-activation gradient")
        g_pre = gy
    else:
        gy = gy.contiguous()
        g_pre = gy if (gy_pre or last.act == ACT_NONE) else ops.act_bwd(gy, saved[-1].y, last.act, last.slope)
    gx = None
    for i in range(n - 1, -1, -1):
        st, sv = plan.stages[i], saved[i]
        assert not isinstance(sv.bn_stats, list), "grouped forward state: backpropagate slice_saved(saved, g, groups)"
        B, H, W, Cp = sv.in_shape
        _, P, Q, K = sv.out_shape
        rows_out = B * P * Q
        m = st.mod
        g = sv.geom
        ld = gy_ld if i == n - 1 else 0          # pixel pitch of g_pre (0 = dense)
        prev = plan.stages[i - 1] if i > 0 else None
        c_in_log = prev.mod.out_channels if (prev is not None and prev.kind != "linear") else (
            (prev.unflat[0] if prev.unflat else prev.mod.out_features) if prev is not None else c_log_in)
        # a first conv's weight gradient (<= 8 input planes): direct per-channel kernel on its own, but as one more job of
        # the pass's combined weight-gradient launch when there is one (its 200 x 32 GEMM fills gaps there, and the bias
        # gradient comes along: measured 7.15 -> 7.08 ms per MNIST iteration)
        gemm_first = fold is not None and Cp % 4 == 0 and K % 4 == 0
        if need_params:
            # ---- parameter gradients of this stage
            fused_db = None
            if m.bias is not None:
                if st.kind == "linear" and st.unflat:
                    Cc, hh, ww = st.unflat
                    db = ops.colsum(B, hh * ww * Cc, hh * ww * Cc, g_pre).reshape(hh * ww, Cc).t().reshape(-1)
                    if id(m.bias) in grad_dst:
                        db = grad_dst[id(m.bias)].copy_(db)
                    grads[id(m.bias)] = db
                elif st.kind == "conv" and not (i == 0 and _first_conv_direct(st, c_in_log) and not gemm_first):
                    # Conv2d: the bias gradient is the column sum of the dense wgrad operand -> fused into that launch
                    fused_db = grad_dst.get(id(m.bias))
                    if fused_db is None:
                        fused_db = torch.empty(K, dtype=torch.float32, device=gy.device)
                    grads[id(m.bias)] = fused_db
                else:
                    grads[id(m.bias)] = ops.colsum(rows_out, K, K, g_pre, out=grad_dst.get(id(m.bias)))
            dw = grad_dst[id(m.weight)] if id(m.weight) in grad_dst else torch.empty_like(m.weight)
            if _is_head(st, sv.in_shape) and not ld:
                ops.head_wgrad(sv.t.reshape(B, Cp)[:, :c_in_log], g_pre.reshape(B), dw.reshape(-1), db=fused_db)
            elif st.kind == "conv" and i == 0 and _first_conv_direct(st, c_in_log) and not gemm_first:
                T = m.kernel_size[0] * m.kernel_size[1]
                # dW[k][c][tap] = sum big=g_pre[..,k] * small=t[..,c], all input channels in one launch
                ops.tconv1_wgrad(g_pre, sv.t, Cp, c_in_log, dw, c_in_log * T, 1, T, B, P, Q, K,
                                 m.kernel_size[0], m.kernel_size[1], m.padding[0])
            elif st.kind == "conv":
                T = m.kernel_size[0] * m.kernel_size[1]
                ops.conv_bwd_weight(g, sv.t, g_pre, dw, c_in_log, K, dw.stride(0), dw.stride(1), dw.stride(3), db=fused_db,
                                    dy_ld=ld, defer=fold)
            elif _is_tconv1(st, Cp):
                T = m.kernel_size[0] * m.kernel_size[1]
                ops.tconv1_wgrad(sv.t, g_pre, 1, 1, dw, T, 1, 0, B, H, W, Cp, m.kernel_size[0], m.kernel_size[1],
                                 m.padding[0])
            elif st.kind == "convT":
                T = m.kernel_size[0] * m.kernel_size[1]
                done = None
                if (m.out_channels == 1 and Cp == 64 and c_in_log == 64 and not ld and sv.t.is_contiguous()
                        and g_pre.is_contiguous() and dw.stride(2) == m.kernel_size[1] * dw.stride(3)):
                    # one-channel tail (stride 2): pixel-contraction kernel instead of a GEMM with one gathered channel
                    done = ops.tconv_scatter_wgrad(sv.t, g_pre, 1, dw, dw.stride(0), dw.stride(3), B, H, W, Cp, P, Q,
                                                   m.kernel_size[0], m.kernel_size[1], m.stride[0], m.padding[0])
                if done is None:
                    # gathered operand = convT output-grad (channels K), dense = convT input (channels Cp)
                    ops.conv_bwd_weight(g, g_pre, sv.t, dw, K, c_in_log, dw.stride(0), dw.stride(1), dw.stride(3),
                                        defer=fold)
            else:
                O, I = m.weight.shape
                Cc, hh, ww = st.unflat if st.unflat else (O, 1, 1)
                T = hh * ww
                if T == 1:
                    ops.conv_bwd_weight(g, sv.t, g_pre, dw, I, O, I, 1, 0, defer=fold)
                else:
                    tmp = torch.empty(O, I, device=dw.device)   # rows in n' = t*C + co order
                    ops.conv_bwd_weight(g, sv.t, g_pre, tmp, I, O, I, 1, 0)      # (re-packed right away: not deferred)
                    ops.pack_weights(tmp, dw, Cc, T, I, I, I, Cc * I, 1)
            grads[id(m.weight)] = dw
        # ---- data gradient, folded with what sits between y_{i-1} and this conv
        if i == 0 and gx_planes is not None and sv.bn is None and _first_conv_direct(st, c_in_log):
            planes = torch.empty(B, H, W, len(gx_planes), dtype=torch.float32, device=gy.device)
            wd = plan.packed(st, "dgrad", Cp)                  # [Cpad][T][K]: row c is the [T][K] filter of plane c
            for j, c in enumerate(gx_planes):     # (the input's Dropout2d mask column scales the plane in the same launch)
                ops.tconv1_fwd(g_pre, wd[c], None, planes[..., j], B, P, Q, K, m.kernel_size[0], m.kernel_size[1],
                               m.padding[0], len(gx_planes), ACT_NONE, 0.0,
                               rowscale=None if sv.mask is None else sv.mask[:, c])
            gx = planes
            break
        if i == 0 and sv.bn is None and _scatter_dgrad(st, gx_planes):
            R, S = m.kernel_size
            NP = len(gx_planes)
            if ops.tconv_scatter_ok(K, NP, R, S, m.stride[0]) and g_pre.is_contiguous():
                planes = torch.empty(B, H, W, NP, dtype=torch.float32, device=gy.device)
                ops.tconv_scatter(g_pre, plan.packed(st, ("scatter_dgrad", tuple(gx_planes)), Cp), None, planes, B, P, Q, K,
                                  H, W, NP, NP, R, S, m.stride[0], m.padding[0])
                if sv.mask is not None:
                    cols = torch.cat([sv.mask[:, c:c + 1] for c in gx_planes], dim=1)
                    planes = planes * cols.reshape(B, 1, 1, -1)
                gx = planes
                break
            contrib = torch.empty(B, P, Q, NP * R * S, dtype=torch.float32, device=gy.device)
            ops.conv_fwd(ops.geom(B, P, Q, K, P, Q, NP * R * S, 1, 1, 1, 0), g_pre,
                         plan.packed(st, ("scatter_dgrad", tuple(gx_planes)), Cp), contrib, ops.epilogue())
            yield
            planes = torch.empty(B, H, W, NP, dtype=torch.float32, device=gy.device)
            ops.col2im(contrib, NP * R * S, None, planes, B, P, Q, H, W, NP, NP, R, S, m.stride[0], m.padding[0])
            if sv.mask is not None:
                cols = torch.cat([sv.mask[:, c:c + 1] for c in gx_planes], dim=1)
                planes = planes * cols.reshape(B, 1, 1, -1)
            gx = planes
            break
        if i == 0 and not need_gx and sv.bn is None:
            break
        pact, pslope = (prev.act, prev.slope) if prev is not None else (in_act or (ACT_NONE, 0.0))
        gt = torch.empty(sv.in_shape, dtype=torch.float32, device=gy.device)
        bn_red = None
        if sv.bn is None:
            ep = ops.epilogue(mask=sv.mask, dact_y=sv.x_in if pact != ACT_NONE else None, dact=pact, dslope=pslope)
        else:
            # BatchNorm backward needs sum(g~ * xhat) and sum(g~) of the gradient this GEMM produces: its epilogue
            # accumulates them per M-tile while the values are in registers
            mask_in = sv.mask if sv.pattern == ["drop", "bn"] else None
            mask_pre = sv.mask if sv.pattern == ["bn", "drop"] else None
            slots = ops.conv_mtiles(g, 0 if st.kind == "convT" else 1)[0]
            if slots > 0 and st.kind in ("conv", "convT") and not _is_tconv1(st, Cp):
                part = torch.empty(2 * Cp * slots, dtype=torch.float32, device=gy.device)
                bn_red = (part, slots)
                ep = ops.epilogue(bn_bwd=(part, sv.x_in, sv.bn_stats[0], sv.bn_stats[1], mask_in, mask_pre, slots))
            else:
                ep = ops.epilogue()
        if _is_tconv1(st, Cp) and sv.bn is None and sv.mask is None:
            ops.tconv1_dgrad(g_pre, 1, plan.packed(st, "fwd", Cp), sv.x_in if pact != ACT_NONE else None, pact, pslope,
                             gt, B, H, W, Cp, m.kernel_size[0], m.kernel_size[1], m.padding[0])
        elif st.kind == "convT":
            ops.conv_fwd(g, g_pre, plan.packed(st, "dgrad", Cp), gt, ep, live=(None, c_in_log))
            yield
        else:
            ops.conv_bwd_data(g, g_pre, plan.packed(st, "dgrad", Cp), gt, ep, in_ld=ld, live=(c_in_log, None))
            yield
        if sv.bn is not None:
            bn = sv.bn
            use_batch = sv.training or bn.running_mean is None
            slope = pslope if pact == ACT_LEAKY else -1.0
            want = (i > 0) or need_gx
            out_dg = grad_dst.get(id(bn.weight)) if need_params else None
            out_db = grad_dst.get(id(bn.bias)) if need_params else None
            if bn_red is not None:
                dgam, dbet, gprev = ops.bn_bwd_from_partials(bn_red[0], bn_red[1], sv.x_in, gt, mask_in, mask_pre,
                                                             sv.bn_stats, bn.weight.detach(), B, H * W, Cp, use_batch,
                                                             slope, want_gx=want, out_dgamma=out_dg, out_dbeta=out_db)
            else:
                dgam, dbet, gprev = ops.bn_bwd(sv.x_in, gt, mask_in, mask_pre, sv.bn_stats, bn.weight.detach(), B, H * W,
                                               Cp, use_batch, slope, want_gx=want, out_dgamma=out_dg, out_dbeta=out_db)
            if need_params:
                grads[id(bn.weight)] = dgam
                grads[id(bn.bias)] = dbet
            if pact == ACT_TANH and gprev is not None:
                gprev = ops.act_bwd(gprev, sv.x_in, ACT_TANH, 0.0)
            gt = gprev
        if i == 0:
            gx = gt
        else:
            g_pre = gt
    return gx, grads


class ChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan: ChainPlan, training: bool, c_log_in: int, x, *params):
        need_grad = any(ctx.needs_input_grad[3:])
        y, saved = chain_forward(plan, x, training, c_log_in, save=need_grad)
        ctx.plan, ctx.saved_stages, ctx.c_log_in = plan, saved, c_log_in
        ctx.param_ids = [id(p) for p in plan.params()]
        return y

    @staticmethod
    def backward(ctx, gy):
        need_gx = ctx.needs_input_grad[3]
        need_params = any(ctx.needs_input_grad[4:])
        gx, grads = chain_backward(ctx.plan, ctx.saved_stages, gy, ctx.c_log_in, need_gx, need_params)
        ctx.saved_stages = None
        pg = tuple(grads.get(pid) if need else None for pid, need in zip(ctx.param_ids, ctx.needs_input_grad[4:]))
        return (None, None, None, gx) + pg


_PLANS = weakref.WeakKeyDictionary()   # nn.Sequential -> ChainPlan; kept OUT of the module so that the reference's
#                                        checkpoint style torch.save({'E': E, ...}) (train_mnist_image_scm.py:61-67) still pickles


def get_plan(seq: nn.Sequential) -> ChainPlan:
    plan = _PLANS.get(seq)
    if plan is None or plan.seq() is not seq:
        plan = ChainPlan(seq)
        _PLANS[seq] = plan
    return plan


def run_chain(seq: nn.Sequential, x: torch.Tensor, c_log_in: Optional[int] = None) -> torch.Tensor:
    """Execute ``seq`` on NHWC ``x`` through the HIP kernels (autograd aware)."""
    if not x.is_cuda:
        raise RuntimeError("ali_hip.chain.run_chain needs CUDA tensors")
    plan = get_plan(seq)
    x = x.contiguous()
    if c_log_in is None:
        c_log_in = x.shape[-1]
    return ChainFn.apply(plan, seq.training, c_log_in, x, *plan.params())
