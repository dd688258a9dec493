"""Hand-scheduled ALI/BiGAN iteration on the HIP kernels (no autograd engine).

Reproduces the reference loop body (image_scms/mnist.py:224-248 and its copies
audio_mnist.py:384-420, whalecalls.py:462-498, esrf_acoustic.py:341-377) with the
work the reference computes and then throws away removed:

  EG step   E(x), D(x,E(x)), G(z), D(G(z),z); backward only along the paths that reach
            E and G: dxz -> dz -> E for the real branch, dxz -> dx -> G for the fake branch;
            no Discriminator weight gradients (the reference zeroes them before use).
  D step a  E'(x) forward only, D full backward, Adam(D).
  D step b  G'(z) forward only, D full backward, Adam(D).
  scores    D''(G'(z), z), D''(x, E'(x)) forward only, re-using G'(z) and E'(x) (the
            reference recomputes identical values) -- train mode, so Dropout2d is active and
            the BatchNorm running statistics get their 6 updates per iteration.

Parameters live in two flat fp32 buffers (E+G, D) with flat gradient / Adam-moment
twins, so an optimiser step is one kernel launch and a data-parallel gradient exchange is
one RCCL all-reduce per group.  Kernel-layout weight copies have fixed addresses and are
re-packed right after the Adam launch that changed them, which makes the whole iteration
capturable in a HIP graph (``capture=True``).
"""
from typing import Dict

import torch

from . import dp
from . import dropout as _dropout
from . import ops
from . import chain as _chain
from .chain import (chain_backward, chain_backward_gen, chain_forward, chain_forward_gen, drive, get_plan,
                    run_parallel, slice_saved)


class FlatGroup:
    """Parameters re-pointed into one flat buffer + flat grad / exp_avg / exp_avg_sq buffers.

    ``layouts`` (optional, {id(param): strides}): the element order a parameter takes inside its segment of the flat
    buffers -- a dense permutation given as the strides of the (logically unchanged) parameter tensor.  The stepper
    asks for the forward GEMM's weight layout (`[K][R*S][C]`: torch's channels_last for a Conv2d weight), so that the
    optimiser step itself leaves the weights in kernel layout and half of the re-pack disappears; Adam is elementwise
    and does not care.  Gradients, exp_avg and exp_avg_sq use the same order (``*_views``)."""

    def __init__(self, params, lr, betas, eps, layouts=None, twin16=False):
        """``twin16``: keep an fp16 twin of the whole flat parameter buffer, written by the Adam launch itself (fp16-MFMA
        path: a conv weight whose segment is laid out as the forward GEMM's pack needs no re-pack launch at all -- the
        fp32 segment IS the pack and ``flat16``'s segment its twin)."""
        self.params = list(params)
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.n = n
        self.flat = torch.empty(n, device=dev)
        self.flat16 = torch.empty(n, dtype=torch.float16, device=dev) if twin16 else None
        self.grad = torch.zeros(n, device=dev)
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.grad_views, self.m_views, self.v_views = {}, [], []
        layouts = layouts or {}
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                st = layouts.get(id(p))

                def view(buf):
                    seg = buf[off:off + k]
                    return seg.view(p.shape) if st is None else torch.as_strided(seg, p.shape, st)
                w = view(self.flat)
                w.copy_(p.detach())
                p.data = w
                if self.flat16 is not None:      # the segment's twin, found by chain.packed through the parameter
                    p._ali_flat16 = self.flat16[off:off + k]
                gv = view(self.grad)
                p.grad = gv
                self.grad_views[id(p)] = gv
                self.m_views.append(view(self.m))
                self.v_views.append(view(self.v))
                off += k
        self.sync16()
        self.lr, self.betas, self.eps = lr, betas, eps
        self.steps = 0
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)   # device-side count of completed steps (graph replays)
        self.arrive = torch.zeros(1, dtype=torch.int32, device=dev) if dev.type == "cuda" else None

    def sync16(self):
        """re-round the fp16 twin after the parameters were changed by anything but ``adam`` (load_state, restore)"""
        if self.flat16 is not None:
            self.flat16.copy_(self.flat)

    @staticmethod
    def logical(views):
        """the buffer behind ``views`` as one flat tensor in the parameters' own (row-major) element order"""
        return torch.cat([v.reshape(-1) for v in views])

    def grad_logical(self):
        """the flat gradient in the parameters' own element order (tests, diagnostics)"""
        return self.logical(list(self.grad_views.values()))

    def load_logical(self, views, flat):
        off = 0
        for v in views:
            v.copy_(flat[off:off + v.numel()].view(v.shape))
            off += v.numel()

    def adam(self, grad_scale=1.0):
        self.steps += 1
        # the kernel runs step step_t + 1 and its last block advances step_t: no launch for the counter
        ops.adam(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.steps,
                 dev_step=self.step_t, grad_scale=grad_scale, arrive=self.arrive, p16=self.flat16)

    # torch.optim-like surface for callers that keep the returned optimisers
    def zero_grad(self):
        self.grad.zero_()

    def step(self):
        self.adam()

    def state_dict(self):
        # the device-side count is the authoritative one: HIP-graph replays do not run the host-side increment
        self.steps = int(self.step_t.item())
        return {"step": self.steps, "exp_avg": self.logical(self.m_views), "exp_avg_sq": self.logical(self.v_views),
                "lr": self.lr, "betas": self.betas, "eps": self.eps}


class MnistFamily:
    """Input assembly of the MorphoMNIST models (mnist.py:46-55, 76-85, 142-151)."""
    name = "mnist"
    hw = (28, 28)

    def __init__(self, E, G, D):
        self.e_tables = [E.digit_embedding[0].weight]
        self.d_tables = [D.digit_embedding[0].weight]
        self.g_tables = [G.digit_embedding.weight]

    @staticmethod
    def conditioning(c):
        """(arg-max class index [B,1] int32, continuous attributes [B,n] in sorted key order, one-hot tensors)"""
        keys = sorted(k for k in c if k != "digit")
        B = c["digit"].shape[0]
        onehots = [_attr2d(c["digit"], B)]
        idx, cont = ops.attr_pack(onehots, [_attr2d(c[k], B, True) for k in keys], B, onehots[0].device)
        return idx, cont, onehots

    @staticmethod
    def used(c):
        """the entries of an attribute dict the models read (mnist.py:47-51: every key)"""
        return {k: v for k, v in c.items() if torch.is_tensor(v)}


class SpectFamily:
    """Input assembly of the spectrogram models (audio_mnist.py:203-210,250-256,309-318 and the whale / ESRF copies):
    categorical attributes -> embedding planes (E, D) / one-hot @ table (G), optional continuous plane (ESRF)."""
    name = "spect"

    def __init__(self, E, G, D):
        self.hw = tuple(E.image_hw)
        self.cat_keys = tuple(E.cat_keys)
        self.cont_key = E.cont_key
        self.e_tables = [E.plane_module(k)[0].weight for k in self.cat_keys]
        self.d_tables = [D.plane_module(k)[0].weight for k in self.cat_keys]
        self.g_tables = [G.table(k).weight for k in self.cat_keys]

    def conditioning(self, c):
        B = c[self.cat_keys[0]].shape[0]
        onehots = [_attr2d(c[k], B) for k in self.cat_keys]
        conts = [_attr2d(c[self.cont_key], B, True)] if self.cont_key is not None else []
        idx, cont = ops.attr_pack(onehots, conts, B, onehots[0].device)
        return idx, cont, onehots

    def used(self, c):
        """the entries of an attribute dict the models read: callers may pass whole batch dicts with extra keys
        (finetune_whale_bigan.py:63-64: 'audio', 'path', 'time')"""
        keys = self.cat_keys + ((self.cont_key,) if self.cont_key is not None else ())
        return {k: c[k] for k in keys}


def _attr2d(t, B, as_float=False):
    """attribute tensor as the kernels take it: contiguous [B, n], fp32 (or int32 one-hots, whalecalls.py:455)"""
    if t.dtype != torch.float32 and (as_float or t.dtype != torch.int32):
        t = t.float()
    return t.reshape(B, -1).contiguous()


def _g_input(fam, z, onehots, cont, alloc=None):
    """[z | onehot @ table ... | cont | 0 pad] rows of the Generator's first layer (mnist.py:76-85), one launch;
    channel stride % 32 == 0 -> uniform-tap fast path of the GEMM kernel.  Returns ([B,1,1,ld], logical width)."""
    B = z.shape[0]
    z = z.reshape(B, -1).float().contiguous()
    n_log = z.shape[1] + 256 * len(fam.g_tables) + (0 if cont is None else cont.shape[1])
    ld = n_log + (-n_log) % 32
    out = alloc("gin", (B, ld)) if alloc is not None else None
    return ops.g_input(z, onehots, [t.detach() for t in fam.g_tables], cont, ld, out=out).reshape(B, 1, 1, ld), n_log


def family_of(E, G, D):
    return SpectFamily(E, G, D) if hasattr(E, "cat_keys") else MnistFamily(E, G, D)


class AliStepper:
    def __init__(self, E, G, D, lr=1e-4, betas=(0.5, 0.999), eps=1e-8, family=None, process_group=None,
                 capture=False, precision="f32", loss_scale=None, pipeline_reduce=False):
        """``precision="f16"``: the convolutions' forward and data-gradient GEMMs contract fp16 operands on
        v_mfma_f32_32x32x16_f16 with fp32 accumulation (BASELINE config 5); activations, master weights, weight
        gradients' accumulation and Adam stay fp32.  The three losses' gradients are multiplied by ``loss_scale``
        (default 1024 for f16, 1 otherwise; a power of two) so that small gradients survive the fp16 rounding of the
        GEMM operands, and Adam divides it out again."""
        self.E, self.G, self.D = E, G, D
        # Data parallel only: overlap the LAST all-reduce of an iteration (D step b) with the NEXT iteration's E(x) / G(z)
        # forward passes, which need neither that reduction nor the D update behind it.  ``step(...,
        # ahead=(images, c, z))`` names the next batch; its forward is computed into persistent buffers while the
        # collective runs and consumed by the next ``step``.  Same arithmetic, same results (tests/test_gpu_dp.py).
        self.pipeline_reduce = bool(pipeline_reduce)
        self._ahead = None            # {"key": ..., "fwd": (ex, sE, gz, sG, n_log, g_log)} of the batch computed ahead
        self._arena = {}              # persistent outputs of the ahead-of-time forward: {(tag, shape): tensor}
        self.precision = precision
        self.loss_scale = float(loss_scale if loss_scale is not None else (1024.0 if precision == "f16" else 1.0))
        self.family = family or family_of(E, G, D)
        self.pE, self.pG = get_plan(E.layers), get_plan(G.layers)
        self.pDx, self.pDz, self.pDxz = get_plan(D.dx), get_plan(D.dz), get_plan(D.dxz)
        # Master weights live in the forward GEMM's layout (FlatGroup.layouts; chain.packed then aliases them).  fp16 path:
        # the Adam launch also writes the fp16 twin of the flat buffer, so the forward packs need no launch either; the
        # data-gradient packs (transposes) are still re-packed, fp16 twins included.
        lay = _chain.pack_layouts([self.pE, self.pG, self.pDx, self.pDz, self.pDxz])
        t16 = precision == "f16" and self.E is not None and next(E.parameters()).is_cuda
        self.opt_eg = FlatGroup(list(E.parameters()) + list(G.parameters()), lr, betas, eps, layouts=lay, twin16=t16)
        self.opt_d = FlatGroup(list(D.parameters()), lr, betas, eps, layouts=lay, twin16=t16)
        for pl in (self.pE, self.pG, self.pDx, self.pDz, self.pDxz):
            pl.cache.store.clear()
            pl.cache.static = True
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.dist = process_group is not None      # a 1-rank group still issues every collective (tests the RCCL path)
        self.capture = capture
        self.segmented = False     # tests: force the data-parallel (segmented) replay on a single rank
        self._capture_snapshot = None
        self._graph = {}           # captured graphs by (input shapes, do_eg): a ragged last batch keeps its own

        self.bn_buffers = [b for n_, b in D.named_buffers() if "running" in n_]
        self.iter_t = torch.zeros(1, dtype=torch.int64, device=self.opt_d.flat.device)
        self._emb_planes = tuple(range(1, 1 + len(self.family.d_tables)))
        self._n_drop = sum(1 for pl in (self.pDx, self.pDz, self.pDxz) for st in pl.stages
                           if any(k == "drop" for k, _ in st.pre))
        # D's joint rows [dx | dz] (mnist.py:152-154): when both branches end in a plain conv GEMM + the same
        # activation and dxz starts with a lone Dropout2d, the branch ends write straight into the joint buffer
        # (masked), and dxz's first data gradient returns the branches' pre-activation gradients -- no torch.cat, no
        # mask pass, no slice copies, no act' passes around the join
        lx, lz, f0 = self.pDx.stages[-1], self.pDz.stages[-1], self.pDxz.stages[0]
        self._join = (_chain.join_ok(self.pDx) and _chain.join_ok(self.pDz) and (lx.act, lx.slope) == (lz.act, lz.slope)
                      and f0.kind == "conv" and [k for k, _ in f0.pre] == ["drop"]
                      and f0.mod.in_channels == lx.mod.out_channels + lz.mod.out_channels
                      and tuple(f0.mod.kernel_size) == (1, 1))
        self._fold = ops.FoldQueue(self.opt_d.flat.device) if self.opt_d.flat.is_cuda else None
        if self._fold is not None:
            self._fold.arena()
        self._join_act = (lx.act, lx.slope)
        self._join_skip = sum(1 for pl in (self.pDx, self.pDz) for st in pl.stages if any(k == "drop" for k, _ in st.pre))
        self._n_drop_dx = sum(1 for st in self.pDx.stages if any(k == "drop" for k, _ in st.pre))

    # ------------------------------------------------------------------ pieces
    def _planes(self, X, idx, cont, tables, out=None, mask=None):
        B = X.shape[0]
        H, W = self.family.hw
        n_log = 1 + len(tables) + (0 if cont is None else cont.shape[1])
        out = ops.assemble_planes(X.reshape(B, H, W), idx, [t.detach() for t in tables], cont, B, H, W,
                                  (n_log + 3) // 4 * 4, out=out, mask=mask)
        return out, n_log

    def _d_input_mask(self, B, device, cont):
        """The Dropout2d mask in front of D.dx's first conv (mnist.py:118) for a pass that is about to start, looked up
        ahead of its turn like the join's (None: not known yet, or no such layer) -- assemble_planes then applies it
        and no mask pass is needed."""
        st = self.pDx.stages[0]
        if [k for k, _ in st.pre] != ["drop"]:
            return None
        n_log = 1 + len(self.family.d_tables) + (0 if cont is None else cont.shape[1])
        return _dropout.peek_mask(0, B, n_log, st.pre[0][1], device, (n_log + 3) // 4 * 4)

    def _d_planes(self, X, idx, cont):
        """D's conv input for one pass: (x0, n_log, whether x0 already carries the first Dropout2d mask)"""
        mask = self._d_input_mask(X.shape[0], X.device, cont)
        x0, n_log = self._planes(X, idx, cont, self.family.d_tables, mask=mask)
        return x0, n_log, mask is not None

    def _d_planes_pair(self, Xa, Xb, idx, cont):
        """... of two passes that will run as one 2B batch, assembled straight into its two halves (call inside
        dropout.paired_passes: the mask is the one of the 2B-row request)"""
        B = Xa.shape[0]
        H, W = self.family.hw
        tables = self.family.d_tables
        n_log = 1 + len(tables) + (0 if cont is None else cont.shape[1])
        mask = self._d_input_mask(2 * B, Xa.device, cont)
        buf = torch.empty(2 * B, H, W, (n_log + 3) // 4 * 4, dtype=torch.float32, device=Xa.device)
        self._planes(Xa, idx, cont, tables, out=buf[:B], mask=None if mask is None else mask[:B])
        self._planes(Xb, idx, cont, tables, out=buf[B:], mask=None if mask is None else mask[B:])
        return buf, n_log, mask is not None

    def _plane_grads(self, g0, idx, tables, dst):
        """Embedding-table gradients from the gradient of the assembled planes (tiny tensors).  ``g0`` is either the
        full input gradient [B,H,W,Cpad] or only its embedding planes [B,H,W,len(tables)].  tanh' comes from the tables
        themselves (the stored planes may carry a Dropout2d mask)."""
        gofs = 0 if g0.shape[3] == len(tables) else 1
        g0 = g0.contiguous()
        for j, t in enumerate(tables):
            ops.plane_table_grad(g0, gofs + j, None, 1 + j, idx, j, t.shape[0], out=dst[id(t)], table=t.detach())

    def _g_input(self, z, onehots, cont, alloc=None):
        return _g_input(self.family, z, onehots, cont, alloc)

    # ---- the E(x) / G(z) forward of the E+G phase, computable ahead of its iteration (pipeline_reduce)
    def _arena_alloc(self, chain):
        def alloc(tag, shape):
            key = (chain, tag, tuple(int(v) for v in shape))
            t = self._arena.get(key)
            if t is None:
                t = self._arena[key] = torch.empty(key[2], dtype=torch.float32, device=self.opt_d.flat.device)
            return t
        return alloc

    def _eg_forward(self, images, idx, cont, onehots, zin, alloc_e=None, alloc_g=None):
        """E(x) and G(z) with their saved activations: independent chains, layer pairs share a launch"""
        fam = self.family
        x0e, n_log = self._planes(images, idx, cont, fam.e_tables,
                                  out=None if alloc_e is None else alloc_e("x0", (images.shape[0],) + tuple(fam.hw) + (
                                      (1 + len(fam.e_tables) + (0 if cont is None else cont.shape[1]) + 3) // 4 * 4,)))
        gin, g_log = self._g_input(zin, onehots, cont, alloc_g)
        (ex, sE), (gz, sG) = run_parallel(chain_forward_gen(self.pE, x0e, True, n_log, True, alloc=alloc_e),
                                          chain_forward_gen(self.pG, gin, True, g_log, True, alloc=alloc_g))
        return ex, sE, gz, sG, n_log, g_log

    @staticmethod
    def _batch_key(images, z):
        return (images.data_ptr(), tuple(images.shape), z.data_ptr(), tuple(z.shape))

    def _prefetch(self, images, c, z):
        """compute the E+G phase's forward passes for a batch ahead of its iteration, into persistent buffers"""
        B = images.shape[0]
        idx, cont, onehots = self.family.conditioning(c)
        zin = z.reshape(B, -1).float().contiguous()
        with ops.precision(self.precision):
            fwd = self._eg_forward(images, idx, cont, onehots, zin, self._arena_alloc("E"), self._arena_alloc("G"))
        self._ahead = {"key": self._batch_key(images, z), "fwd": fwd}

    def _join_begin(self, B, device):
        """(joint buffer [B, n_dx + nz], the Dropout2d mask of dxz's first stage) for a D forward that is about to
        start with its dx chain, or None: the mask has to be known before the branches run, i.e. it is looked up ahead
        of its turn (injected tapes, the per-iteration mask plan) -- never drawn out of order."""
        if not self._join:
            return None
        ctot = self.pDxz.stages[0].mod.in_channels
        mask = _dropout.peek_mask(self._join_skip, B, ctot, self.pDxz.stages[0].pre[0][1], device, ctot)
        if mask is None:
            return None
        return torch.empty(B, ctot, dtype=torch.float32, device=device), mask

    def _dx_forward_gen(self, x0, n_log, save, groups=1, x_masked=False, join=None, lane=None):
        """D.dx, writing its end into the joint buffer when the chains can join: (dx_pre, join) for _d_forward"""
        if join is None:                     # (False: the caller knows there is none)
            join = self._join_begin(x0.shape[0], x0.device)
        join = join or None
        dx_pre = yield from chain_forward_gen(self.pDx, x0, True, n_log, save, groups,
                                              join=None if join is None else (join[0], 0, join[1]),
                                              first_mask_applied=x_masked, lane=lane)
        return dx_pre, join

    def _dx_forward(self, *args, **kwargs):
        return drive(self._dx_forward_gen(*args, **kwargs))

    def _d_forward(self, x0, n_log, zin, save, groups=1, dx_pre=None, join=None, x_masked=False):
        B = x0.shape[0]
        # D.dx and D.dz are independent chains (mnist.py:152-153): advanced side by side, their GEMMs share launches.
        # Each keeps the Dropout2d masks of its own turn in the reference's order (dropout.Lane) -- which needs the
        # iteration's mask sequence to be known ahead (not in a stepper's very first iteration: then one after the other)
        pair, lanes = dx_pre is None, (None, None)
        if pair and self._join_skip:
            base = _dropout.lanes_start(self._join_skip)
            pair = base is not None
            if pair:
                lanes = (_dropout.Lane(base), _dropout.Lane(base + self._n_drop_dx))
        if pair:
            join = self._join_begin(B, x0.device)
            n_dx = _chain._out_shape(self.pDx.stages[-1], B, 1, 1, 1)[3]
            (dx_pre, _), (dz, s_dz) = run_parallel(
                self._dx_forward_gen(x0, n_log, save, groups, x_masked, join=join if join is not None else False,
                                     lane=lanes[0]),
                # (one round late: D.dx's first conv runs on a kernel of its own, D.dz's two GEMMs ride with the next two)
                _chain.delayed(chain_forward_gen(self.pDz, zin.reshape(B, 1, 1, -1), True, zin.numel() // B, save, groups,
                                                 join=None if join is None else (join[0], n_dx, join[1]), lane=lanes[1]),
                               1))
            if self._join_skip:
                _dropout.advance(self._join_skip)
            dx, s_dx = dx_pre
            assert dx.shape[-1] == n_dx
        else:
            if dx_pre is None:
                dx_pre, join = self._dx_forward(x0, n_log, save, groups, x_masked)
            dx, s_dx = dx_pre
            n_dx = dx.shape[-1]
            dz, s_dz = chain_forward(self.pDz, zin.reshape(B, 1, 1, -1), True, zin.numel() // B, save, groups,
                                     join=None if join is None else (join[0], n_dx, join[1]))
        if join is None:
            joint = torch.cat([dx.reshape(B, -1), dz.reshape(B, -1)], dim=1).reshape(B, 1, 1, -1)
            logit, s_dxz = chain_forward(self.pDxz, joint, True, joint.shape[-1], save, groups)
        else:
            joint = join[0].reshape(B, 1, 1, -1)
            logit, s_dxz = chain_forward(self.pDxz, joint, True, joint.shape[-1], save, groups, first_mask_applied=True)
        return logit.reshape(B, 1), (s_dx, s_dz, s_dxz, n_dx, n_log)

    def _d_forward_pair(self, Xa, zina, Xb, zinb, idx, cont, save):
        """D(a) and D(b) with the same weights as ONE batch of 2B samples (rows [0,B) = a): half the launches, and
        the small 1x1 layers see twice the rows.  BatchNorm statistics, running-stat updates and Dropout2d masks stay
        per pass, in the order a, b (chain_forward groups / dropout.paired_passes)."""
        with _dropout.paired_passes(self._n_drop):
            x0, n_log, masked = self._d_planes_pair(Xa, Xb, idx, cont)
            zin2 = torch.empty((2 * zina.shape[0],) + tuple(zina.shape[1:]), dtype=zina.dtype, device=zina.device)
            ops.copy_multi([(zin2[:zina.shape[0]], zina), (zin2[zina.shape[0]:], zinb)])
            logit, saved = self._d_forward(x0, n_log, zin2, save, 2, x_masked=masked)
        return logit, saved

    def _d_backward(self, saved, glogit, need_params, need_x, need_z, planes=None):
        s_dx, s_dz, s_dxz, n_dx, n_log = saved
        B = glogit.shape[0]
        dst = self.opt_d.grad_views if need_params else None
        fold = self._fold if need_params else None
        if fold is not None:
            fold.expect(_chain.wgrad_geoms(self.pDxz, s_dxz) + _chain.wgrad_geoms(self.pDx, s_dx)
                        + _chain.wgrad_geoms(self.pDz, s_dz))
        gjoint, _ = chain_backward(self.pDxz, s_dxz, glogit.reshape(B, 1, 1, 1), s_dxz[0].in_shape[3], True,
                                   need_params, dst, fold=fold, **self._join_in())
        gjoint = gjoint.reshape(B, -1)
        gx0 = gz = None
        gens = []
        if need_params or need_x:
            gens.append(chain_backward_gen(self.pDx, s_dx, *self._branch_grad(gjoint, 0, B, 0, n_dx), n_log,
                                           need_x, need_params, dst, gx_planes=planes if need_x else None, fold=fold,
                                           **self._join_out(gjoint)))
        if need_params or need_z:
            nz = gjoint.shape[1] - n_dx
            gens.append(chain_backward_gen(self.pDz, s_dz, *self._branch_grad(gjoint, 0, B, n_dx, nz), nz, need_z,
                                           need_params, dst, fold=fold, **self._join_out(gjoint)))
        res = run_parallel(*gens)          # the two branches are independent: their GEMMs go out pairwise
        if need_params or need_x:
            gx0 = res[0][0]
        if need_params or need_z:
            gz = res[-1][0]
        if fold is not None:
            fold.flush()          # one launch sums the slabs of all of D's weight gradients
        return gx0, gz

    # the join in the backward pass: dxz's first data-gradient epilogue applies the branch ends' act', the branches
    # read their column range of the joint gradient in place (otherwise: slice copies + act' passes)
    def _join_in(self):
        return {"in_act": self._join_act} if self._join else {}

    def _join_out(self, gjoint):
        return {"gy_ld": gjoint.shape[1], "gy_pre": True} if self._join else {}

    def _branch_grad(self, gjoint, r0, rows, c0, cols):
        g = gjoint[r0:r0 + rows, c0:c0 + cols]
        return ((g if self._join else g.contiguous()).unflatten(1, (1, 1, cols)),)

    # ------------------------------------------------------------------ the iteration, phase by phase
    def _begin(self, images, c, z, do_eg=True):
        B = images.shape[0]
        # (iter_t = iterations completed so far: it keys this iteration's Dropout2d masks and advances at the end of the
        # iteration, in the launch that also applies the BatchNorm batch counts)
        if self._fold is not None:
            self._fold.abandon()       # (nothing is pending after a completed iteration)
        _dropout.begin_iteration(self.iter_t, owner=self, tag=(B, bool(do_eg)))
        _chain.defer_batch_counts()
        idx, cont, onehots = self.family.conditioning(c)
        return {"images": images, "B": B, "idx": idx, "cont": cont, "onehots": onehots,
                "zin": z.reshape(B, -1).float().contiguous(), "out": {}}

    def _eg_grads(self, cx):
        """E+G gradients (reference mnist.py:224-229)."""
        fam, images, idx, cont, onehots, zin, B = (self.family, cx["images"], cx["idx"], cx["cont"], cx["onehots"],
                                                   cx["zin"], cx["B"])
        # E(x) and G(z): computed here, or ahead of the iteration (pipeline_reduce)
        fwd = cx.pop("eg_fwd", None)
        ex, sE, gz, sG, n_log, g_log = fwd if fwd is not None else self._eg_forward(images, idx, cont, onehots, zin)
        # D(x, E(x)) and D(G(z), z) share the weights: one batch of 2B samples (reference order: real pass first)
        logits, (s_dx, s_dz, s_dxz, n_dx, _) = self._d_forward_pair(images, ex.reshape(zin.shape), gz, zin, idx, cont, True)
        # (bce(D_valid, 0) + bce(D_fake, 1)) / 2 and its gradient for both halves: one launch
        l3, gl = ops.bce_logits_pair(logits, B, 0.0, 1.0, 0.5 * self.loss_scale)
        cx["out"]["loss_eg"] = l3[0]
        # backward: dxz for both passes at once (data gradient only: D is not updated in this phase) ...
        gjoint, _ = chain_backward(self.pDxz, s_dxz, gl.reshape(2 * B, 1, 1, 1), s_dxz[0].in_shape[3], True, False,
                                   **self._join_in())
        gjoint = gjoint.reshape(2 * B, -1)
        dst = self.opt_eg.grad_views
        nz = gjoint.shape[1] - n_dx
        if self._fold is not None:
            self._fold.expect(_chain.wgrad_geoms(self.pE, sE) + _chain.wgrad_geoms(self.pG, sG))

        def real_branch():
            # ... real pass: only the z-side path (dxz -> dz) reaches E
            g_ex, _ = yield from chain_backward_gen(self.pDz, slice_saved(s_dz, 0, 2),
                                                    *self._branch_grad(gjoint, 0, B, n_dx, nz), nz, True, False,
                                                    **self._join_out(gjoint))
            # of E's input gradient only the embedding planes are consumed (their tables are parameters of E)
            g_x0e, _ = yield from chain_backward_gen(self.pE, sE, g_ex.reshape(ex.shape), n_log, bool(self._emb_planes),
                                                     True, dst, gx_planes=self._emb_planes or None, fold=self._fold)
            if self._emb_planes:
                self._plane_grads(g_x0e, idx, fam.e_tables, dst)

        def fake_branch():
            # ... fake pass: only the image path (dxz -> dx) reaches G
            g_x0f, _ = yield from chain_backward_gen(self.pDx, slice_saved(s_dx, 1, 2),
                                                     *self._branch_grad(gjoint, B, B, 0, n_dx), n_log, True, False,
                                                     gx_planes=(0,), **self._join_out(gjoint))
            g_gz = g_x0f[..., 0].contiguous().reshape(gz.shape)
            g, _ = yield from chain_backward_gen(self.pG, sG, g_gz, g_log, True, True, dst, fold=self._fold)
            return g

        # the two branches are independent from here on: their GEMMs go out pairwise
        _, g_gin = run_parallel(real_branch(), fake_branch())
        if self._fold is not None:
            self._fold.flush()    # one launch sums the slabs of all of E's and G's weight gradients
        g_gin = g_gin.reshape(B, -1)
        off = zin.shape[1]
        for oh, t in zip(onehots, fam.g_tables):
            ops.g_input_table_grad(oh, g_gin, off, dst[id(t)])
            off += 256

    def _apply_eg(self):
        self.opt_eg.adam(1.0 / (self.world * self.loss_scale))
        with ops.batched_packs():
            self.pE.cache.refresh()
            self.pG.cache.refresh()

    def _apply_d(self):
        self.opt_d.adam(1.0 / (self.world * self.loss_scale))
        self._refresh_d()

    def _phase_eg(self, cx):
        self._eg_grads(cx)
        if self.dist:
            dp.allreduce_sum_(self.opt_eg.grad, self.pg)
        self._apply_eg()

    def _phase_d_real(self, cx):
        self._d_real_pre(cx)
        self._d_real_rest(cx)
        if self.dist:
            dp.allreduce_sum_(self.opt_d.grad, self.pg)
        self._apply_d()

    def _phase_d_fake(self, cx):
        self._d_fake_pre(cx)
        self._d_fake_rest(cx)
        if self.dist:
            dp.allreduce_sum_(self.opt_d.grad, self.pg)
        self._apply_d()

    # The two D phases are cut where their first dependence on the previous optimiser step sits, so that the
    # data-parallel all-reduce of that step's gradients overlaps with the part in front of the cut:
    # D.dx(x) needs neither E' nor a new D; G'(z) needs G' (already stepped) but not the D update in flight.
    def _d_real_pre(self, cx):
        """D.dx on the real batch: independent of the E+G update whose gradients may still be in the all-reduce."""
        fam, images, idx, cont = self.family, cx["images"], cx["idx"], cx["cont"]
        x0d, n_log, masked = self._d_planes(images, idx, cont)
        cx["x0d"], cx["n_log"] = x0d, n_log
        cx["dx_pre"] = self._dx_forward(x0d, n_log, True, x_masked=masked)

    def _d_real_all(self, cx):
        """The whole D-real phase on one GPU: nothing has to hide an all-reduce, so E'(x) and D.dx(x) -- independent
        chains -- run side by side (chain.run_parallel) instead of one in front of the E+G optimiser step."""
        fam, images, idx, cont = self.family, cx["images"], cx["idx"], cx["cont"]
        x0d, n_log, masked = self._d_planes(images, idx, cont)
        cx["x0d"], cx["n_log"] = x0d, n_log
        x0e, _ = self._planes(images, idx, cont, fam.e_tables)
        cx["dx_pre"], (cx["ex_pre"], _) = run_parallel(self._dx_forward_gen(x0d, n_log, True, x_masked=masked),
                                                       chain_forward_gen(self.pE, x0e, True, n_log, False))
        self._d_real_rest(cx)

    def _d_real_rest(self, cx):
        """D gradients on (x, E'(x)) (reference mnist.py:232-235); E' forward only."""
        fam, images, idx, cont = self.family, cx["images"], cx["idx"], cx["cont"]
        x0d, n_log = cx["x0d"], cx["n_log"]
        ex = cx.pop("ex_pre", None)
        if ex is None:
            x0e, _ = self._planes(images, idx, cont, fam.e_tables)
            ex, _ = chain_forward(self.pE, x0e, True, n_log, False)
        dx_pre, join = cx.pop("dx_pre")
        d_valid, sD = self._d_forward(x0d, n_log, ex, True, dx_pre=dx_pre, join=join)
        l, gl = ops.bce_logits(d_valid, 1.0, self.loss_scale)
        cx["out"]["loss_d_real"] = l[0]
        g_x0, _ = self._d_backward(sD, gl, True, True, False, planes=self._emb_planes)
        self._plane_grads(g_x0, idx, fam.d_tables, self.opt_d.grad_views)
        cx["ex"] = ex

    def _d_fake_pre(self, cx):
        """G'(z): independent of the D update whose gradients may still be in the all-reduce."""
        gin, g_log = self._g_input(cx["zin"], cx["onehots"], cx["cont"])
        cx["gz"], _ = chain_forward(self.pG, gin, True, g_log, False)

    def _d_fake_rest(self, cx):
        """D gradients on (G'(z), z) (reference mnist.py:237-240); G' forward only."""
        fam, idx, cont, zin = self.family, cx["idx"], cx["cont"], cx["zin"]
        x0f, _, masked = self._d_planes(cx["gz"], idx, cont)
        d_fake, sD = self._d_forward(x0f, cx["n_log"], zin, True, x_masked=masked)
        l, gl = ops.bce_logits(d_fake, 0.0, self.loss_scale)
        cx["out"]["loss_d_fake"] = l[0]
        g_x0, _ = self._d_backward(sD, gl, True, True, False, planes=self._emb_planes)
        self._plane_grads(g_x0, idx, fam.d_tables, self.opt_d.grad_views)

    def _phase_scores(self, cx, average_bn=True):
        """sigma(D(G(z),z)).mean(), sigma(D(x,E(x))).mean() (reference mnist.py:243-248): forward only, train mode,
        re-using G'(z) and E'(x) of the D phases (the reference recomputes identical values)."""
        fam, images, idx, cont, zin = self.family, cx["images"], cx["idx"], cx["cont"], cx["zin"]
        logits, _ = self._d_forward_pair(cx["gz"], zin, images, cx["ex"].reshape(zin.shape), idx, cont, False)
        s3 = ops.bce_logits_pair(logits, cx["B"], 0.0, 0.0, 1.0, want_grad=False)[0]
        cx["out"]["dg"], cx["out"]["de"] = s3[1], s3[2]
        _dropout.end_iteration()
        # all BatchNorm num_batches_tracked increments of the iteration and the iteration counter: one launch
        _chain.flush_batch_counts(extra=[(self.iter_t, 1)])
        if self.dist and average_bn:
            # replicas use local batch statistics; keep the running buffers (state_dict) identical
            dp.average_buffers_(self.bn_buffers, self.pg)

    def _pipelined(self, do_eg):
        return self.pipeline_reduce and do_eg and (self.dist or self.segmented)

    def _segments(self, do_eg, ahead=False):
        """The iteration as a list of (work, reduce, wait) steps.  ``work(cx)`` is pure device work (capturable in a HIP
        graph); ``reduce`` names the parameter group whose flat gradient buffer is all-reduced -- asynchronously, on
        the collective's own stream -- right after the step; a step with ``wait`` first makes the compute stream wait
        for the all-reduce in flight.  The steps without ``wait`` that follow a reduce overlap with it."""
        segs = []
        if not (self.dist or self.segmented):
            # one GPU, no collectives to hide: E'(x) runs beside D.dx(x), behind the E+G optimiser step
            if do_eg:
                segs.append((lambda cx: self._eg_grads(cx), self.opt_eg, False))
                segs.append((lambda cx: (self._apply_eg(), self._d_real_all(cx)), self.opt_d, True))
            else:
                segs.append((lambda cx: self._d_real_all(cx), self.opt_d, False))
        elif do_eg:
            segs.append((lambda cx: self._eg_grads(cx), self.opt_eg, False))
            segs.append((lambda cx: self._d_real_pre(cx), None, False))
            segs.append((lambda cx: (self._apply_eg(), self._d_real_rest(cx)), self.opt_d, True))
        else:
            segs.append((lambda cx: (self._d_real_pre(cx), self._d_real_rest(cx)), self.opt_d, False))
        segs.append((lambda cx: self._d_fake_pre(cx), None, False))
        segs.append((lambda cx: (self._apply_d(), self._d_fake_rest(cx)), self.opt_d, True))
        if ahead and self._pipelined(do_eg):
            # the NEXT iteration's E(x) / G(z) forward passes run under the all-reduce that was just issued: they need
            # neither it nor the D update behind it (pipeline_reduce)
            segs.append((lambda cx: self._prefetch(*cx["ahead"]), None, False))
        segs.append((lambda cx: (self._apply_d(), self._phase_scores(cx, average_bn=False)), None, True))
        return segs

    def _take_ahead(self, cx, images, c, z):
        """pipeline_reduce: the E+G phase consumes the forward passes computed ahead (by the previous ``step``'s tail,
        or right here when this batch was not announced)"""
        if self._ahead is None or self._ahead["fwd"] is None or self._ahead["key"] != self._batch_key(images, z):
            self._prefetch(images, c, z)
        cx["eg_fwd"], self._ahead = self._ahead["fwd"], None

    def _iteration(self, images, c, z, do_eg=True, ahead=None):
        pipe = self._pipelined(do_eg)
        if pipe and (self._ahead is None or self._ahead["fwd"] is None or self._ahead["key"] != self._batch_key(images, z)):
            self._prefetch(images, c, z)       # (in front of _begin: it runs its own attribute plumbing)
        cx = self._begin(images, c, z, do_eg)
        if pipe:
            self._take_ahead(cx, images, c, z)
            cx["ahead"] = ahead
        pending = None
        for work, group, wait in self._segments(do_eg, ahead is not None):
            if wait and pending is not None:
                pending.wait()
                pending = None
            work(cx)
            if group is not None and self.dist:
                pending = dp.allreduce_sum_async_(group.grad, self.pg)
        if self.dist:
            dp.average_buffers_(self.bn_buffers, self.pg)
        return cx["out"]

    def _state_tensors(self):
        ts = [self.iter_t]
        for g in (self.opt_eg, self.opt_d):
            ts += [g.flat, g.m, g.v, g.step_t]
        ts += [b for _, b in self.D.named_buffers()]
        return ts

    def _snapshot(self):
        return [t.clone() for t in self._state_tensors()], (self.opt_eg.steps, self.opt_d.steps)

    def _restore(self, snap):
        vals, (se, sd) = snap
        for t, v in zip(self._state_tensors(), vals):
            t.copy_(v)
        self.opt_eg.steps, self.opt_d.steps = se, sd
        self.opt_eg.sync16(), self.opt_d.sync16()
        for pl in (self.pE, self.pG, self.pDx, self.pDz, self.pDxz):
            pl.cache.refresh()

    def load_state(self, E_src, G_src, D_src, opt_e=None, opt_d=None):
        """Adopt the weights / buffers of another (e.g. CPU, reference-trained) E, G, D and, optionally, the
        exp_avg / exp_avg_sq / step of their torch.optim.Adam optimisers; kernel-layout copies are refreshed."""
        for src, dst in ((E_src, self.E), (G_src, self.G), (D_src, self.D)):
            sd = src.state_dict()
            for k, v in dst.state_dict().items():
                v.copy_(sd[k])
        _chain.drop_pending_batch_counts()      # the adopted num_batches_tracked already include them
        for group, opt, mods in ((self.opt_eg, opt_e, (E_src, G_src)), (self.opt_d, opt_d, (D_src,))):
            if opt is None:
                continue
            src_params = [p for m in mods for p in m.parameters()]
            step = 0
            for p_src, mv, vv in zip(src_params, group.m_views, group.v_views):
                st = opt.state.get(p_src, {})
                if st:
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                    step = int(st["step"])
                else:
                    mv.zero_()
                    vv.zero_()
            group.steps = step
            group.step_t.fill_(step)
        self.opt_eg.sync16(), self.opt_d.sync16()
        for pl in (self.pE, self.pG, self.pDx, self.pDz, self.pDxz):
            pl.cache.refresh()

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """Resumable checkpoint: the reference's state-dict format (``{E,G,D}_state_dict``, what ``mnist.load_model``
        / the audio scripts read, mnist.py:302-313) plus what the reference never saves: both Adam states and the
        iteration counter that keys the dropout masks.  Tensors are cloned to the CPU."""
        def opt(g):
            # the device-side count is the authoritative one: graph replays do not run the host-side increment
            return {"step": int(g.step_t.item()), "exp_avg": g.logical(g.m_views).cpu(),
                    "exp_avg_sq": g.logical(g.v_views).cpu()}
        sd = {f"{n}_state_dict": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
              for n, m in (("E", self.E), ("G", self.G), ("D", self.D))}
        sd.update(optimizer_E=opt(self.opt_eg), optimizer_D=opt(self.opt_d), iteration=int(self.iter_t.item()),
                  dropout_seed=_dropout._state["seed"])
        return sd

    def load_state_dict(self, sd):
        """Inverse of ``state_dict`` (also accepts a checkpoint that only holds the three module state dicts):
        everything is copied in place, so captured HIP graphs stay valid."""
        with torch.no_grad():
            for n, m in (("E", self.E), ("G", self.G), ("D", self.D)):
                src = sd[f"{n}_state_dict"]
                for k, v in m.state_dict().items():
                    v.copy_(src[k])
            for g, key in ((self.opt_eg, "optimizer_E"), (self.opt_d, "optimizer_D")):
                o = sd.get(key)
                if o is None:
                    g.m.zero_(), g.v.zero_()
                    g.steps = 0
                else:
                    g.load_logical(g.m_views, o["exp_avg"].to(g.m.device))
                    g.load_logical(g.v_views, o["exp_avg_sq"].to(g.v.device))
                    g.steps = int(o["step"])
                g.step_t.fill_(g.steps)
            self.iter_t.fill_(int(sd.get("iteration", 0)))
            self.opt_eg.sync16(), self.opt_d.sync16()
            if "dropout_seed" in sd and int(sd["dropout_seed"]) != _dropout._state["seed"]:
                # the seed is a launch argument of the mask kernel: graphs captured with the old one must go
                _dropout._state["seed"] = int(sd["dropout_seed"])
                self._graph.clear()
                self.__dict__.pop("_mask_plans", None)
            for pl in (self.pE, self.pG, self.pDx, self.pDz, self.pDxz):
                pl.cache.refresh()

    def _refresh_d(self):
        with ops.batched_packs():
            self.pDx.cache.refresh()
            self.pDz.cache.refresh()
            self.pDxz.cache.refresh()

    # ------------------------------------------------------------------ public
    @torch.no_grad()
    def step(self, images, c: Dict[str, torch.Tensor], z, do_eg=True, masks=None, ahead=None):
        """One iteration.  ``masks``: optional list of host-recorded Dropout2d masks (parity mode).
        ``ahead`` = (images, c, z) of the NEXT call (``pipeline_reduce``, data parallel only; ignored otherwise): pass
        the very tensors the next call will be given."""
        try:
            return self._step(images, c, z, do_eg, masks, ahead)
        finally:
            _chain.abort_batch_counts()     # no-op after a completed iteration

    def _step(self, images, c, z, do_eg, masks, ahead=None):
        with ops.precision(self.precision):
            return self._step_impl(images, c, z, do_eg, masks, ahead)

    def _step_impl(self, images, c, z, do_eg, masks, ahead=None):
        if not self._pipelined(do_eg):
            ahead = None
        if masks is not None:
            with _dropout.injected_masks(masks):
                return self._iteration(images, c, z, do_eg, ahead)
        if not self.capture:
            return self._iteration(images, c, z, do_eg, ahead)
        try:
            if self.dist or self.segmented:
                return self._replay_segments(images, c, z, do_eg, ahead)
            return self._replay(images, c, z, do_eg)
        except RuntimeError as e:  # graph capture refused (driver / RCCL combination): keep training, eagerly
            if self._graph or not any(w in str(e).lower() for w in ("captur", "graph")):
                raise              # a failure of a graph that already ran, or an error that is not about capture
            import warnings
            warnings.warn(f"AliStepper: HIP graph capture failed ({e!r}); continuing with eager launches")
            self.capture = False
            if self._capture_snapshot is not None:
                self._restore(self._capture_snapshot)
            return self._iteration(images, c, z, do_eg)

    def _replay_segments(self, images, c, z, do_eg, ahead=None):
        """Data-parallel replay: one HIP graph per segment, the gradient all-reduces in between launched eagerly."""
        pipe = self._pipelined(do_eg)
        key = ("seg", tuple(images.shape), do_eg, pipe, ahead is not None)
        if key not in self._graph:
            st = {"images": images.clone(), "z": z.clone(), "c": {k: v.clone() for k, v in c.items()}}
            nxt = None
            if ahead is not None:
                nxt = (ahead[0].clone(), {k: v.clone() for k, v in ahead[1].items()}, ahead[2].clone())
            snap = self._snapshot()
            self._capture_snapshot = snap
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):          # warm-up outside capture (collectives included: all ranks do this)
                self._iteration(st["images"], st["c"], st["z"], do_eg, nxt)
                if pipe and self._ahead is None:
                    self._prefetch(st["images"], st["c"], st["z"])     # the persistent buffers the capture will name
            torch.cuda.current_stream().wait_stream(s)
            self._restore(snap)
            pool = torch.cuda.graph_pool_handle()
            graphs, cx = [], None
            for i, (fn, group, wait) in enumerate(self._segments(do_eg, ahead is not None)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    if i == 0:
                        cx = self._begin(st["images"], st["c"], st["z"], do_eg)
                        if pipe:
                            cx["eg_fwd"], cx["ahead"] = self._ahead["fwd"], nxt      # (the persistent buffers)
                    fn(cx)
                graphs.append((g, group, wait))
                if group is not None and self.dist:   # keep the ranks' collective sequences aligned while capturing
                    dp.allreduce_sum_(group.grad, self.pg)
            if self.dist:
                dp.average_buffers_(self.bn_buffers, self.pg)
            self._restore(snap)                 # capture executes nothing, but the eager collectives above ran
            self._ahead = None                  # (what the buffers hold was computed with the warm-up's weights)
            self._graph[key] = (graphs, st, cx["out"], nxt)
        graphs, st, res, nxt = self._graph[key]
        ops.copy_multi([(st["images"], images), (st["z"], z)] + [(st["c"][k], v) for k, v in c.items()])
        if nxt is not None:
            ops.copy_multi([(nxt[0], ahead[0]), (nxt[2], ahead[2])] + [(nxt[1][k], v) for k, v in ahead[1].items()])
        if pipe and (self._ahead is None or self._ahead["key"] != self._batch_key(images, z)):
            self._prefetch(st["images"], st["c"], st["z"])      # this batch was not announced: compute its forward now
        self._ahead = None
        pending = None
        for g, group, wait in graphs:
            if wait and pending is not None:
                pending.wait()                  # stream-level: the next graph waits for the all-reduce in flight
                pending = None
            g.replay()
            if group is not None and self.dist:
                pending = dp.allreduce_sum_async_(group.grad, self.pg)
        if self.dist:
            dp.average_buffers_(self.bn_buffers, self.pg)
        if pipe and ahead is not None:          # the replayed prefetch segment left the next batch's forward in the buffers
            self._ahead = {"key": self._batch_key(ahead[0], ahead[2]), "fwd": None}
        return res

    def _replay(self, images, c, z, do_eg):
        key = (tuple(images.shape), do_eg)
        if key not in self._graph:
            st = {"images": images.clone(), "z": z.clone(), "c": {k: v.clone() for k, v in c.items()}}
            snap = self._snapshot()
            self._capture_snapshot = snap
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):          # warm-up outside capture: packs, workspace, plans
                self._iteration(st["images"], st["c"], st["z"], do_eg)
            torch.cuda.current_stream().wait_stream(s)
            self._restore(snap)                 # the warm-up must not count as a training iteration
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                res = self._iteration(st["images"], st["c"], st["z"], do_eg)
            self._graph[key] = (graph, st, res)
        graph, st, res = self._graph[key]
        ops.copy_multi([(st["images"], images), (st["z"], z)] + [(st["c"][k], v) for k, v in c.items()])
        graph.replay()
        return res


class FinetuneStepper:
    """Encoder fine-tuning against a frozen Generator, hand scheduled (reference
    finetune_mnist_bigan.py:64-85, finetune_audio_mnist_bigan.py:62-91, finetune_whale_bigan.py:52-77 with
    ``--metric mse``):

        codes = E(x, a); xr = G(codes, a); loss = mean((x - xr)^2) + mean(codes^2); Adam(E).step()

    E runs forward + full backward, G forward + data gradient only (its weight gradients are never used by the
    reference's optimiser); the two scalar losses and their gradients are O(B*H*W) elementwise work.  ``a`` may be a
    whole batch dict with extra keys (the whale script passes it as is).  ``x`` of shape [B,H,W] (the whale script
    never adds the channel axis) makes ``x - xr`` broadcast to all B*B pairs, exactly as in the reference; [B,1,H,W]
    is the ordinary per-sample error.  ``capture=True`` replays the step from a HIP graph per input shape.
    ``step`` returns {"rec": mse, "latent": mean(codes^2)} as 0-d device tensors (no host sync)."""

    def __init__(self, E, G, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, family=None, capture=False):
        self.E, self.G = E, G
        self.family = family or (SpectFamily(E, G, E) if hasattr(E, "cat_keys") else _mnist_family_eg(E, G))
        self.pE, self.pG = get_plan(E.layers), get_plan(G.layers)
        self.opt_e = FlatGroup(list(E.parameters()), lr, betas, eps)
        self.pE.cache.store.clear()
        self.pE.cache.static = True
        self.capture = capture
        self._graphs = {}

    @torch.no_grad()
    def step(self, x, a):
        a = self.family.used(a)
        if not (self.capture and x.is_cuda):
            return self._step(x, a)
        key = (tuple(x.shape), tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(a.items())), self.E.training,
               self.G.training)
        ent = self._graphs.get(key)
        if ent is None:
            st_x, st_a = x.clone(), {k: v.clone() for k, v in a.items()}
            snap = [t.clone() for t in (self.opt_e.flat, self.opt_e.m, self.opt_e.v, self.opt_e.step_t)]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._step(st_x, st_a)                     # warm-up outside capture: packs, workspace, plans
            torch.cuda.current_stream().wait_stream(side)
            for t, v in zip((self.opt_e.flat, self.opt_e.m, self.opt_e.v, self.opt_e.step_t), snap):
                t.copy_(v)                                 # the warm-up must not count as a training step
            self.opt_e.steps = int(self.opt_e.step_t.item())
            self.pE.cache.refresh()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                res = self._step(st_x, st_a)
            ent = self._graphs[key] = (graph, st_x, st_a, res)
        graph, st_x, st_a, res = ent
        st_x.copy_(x)
        for k, v in a.items():
            st_a[k].copy_(v)
        graph.replay()
        return res

    def _step(self, x, a):
        fam = self.family
        B = x.shape[0]
        H, W = fam.hw
        pairwise = x.dim() == 3 and B > 1          # [B,H,W] - [B,1,H,W] broadcasts to [B,B,H,W] (whale script)
        idx, cont, onehots = fam.conditioning(a)
        n_log = 1 + len(fam.e_tables) + (0 if cont is None else cont.shape[1])
        x0 = ops.assemble_planes(x.reshape(B, H, W).float().contiguous(), idx, [t.detach() for t in fam.e_tables],
                                 cont, B, H, W, (n_log + 3) // 4 * 4)
        codes, sE = chain_forward(self.pE, x0, self.E.training, n_log, True)
        zin = codes.reshape(B, -1)
        gin, g_log = _g_input(fam, zin, onehots, cont)
        xr, sG = chain_forward(self.pG, gin, self.G.training, g_log, True)
        xf, xrf = x.reshape(B, -1).float(), xr.reshape(B, -1)
        if pairwise:
            # mean_{i,j,p} (x_jp - xr_ip)^2 and its gradient 2/(B*P) * (xr_ip - mean_j x_jp)
            xbar = xf.mean(dim=0, keepdim=True)
            rec = xrf.square().mean() - 2.0 * (xrf.mean(dim=0) * xbar[0]).mean() + xf.square().mean()
            g_xr = ((xrf - xbar) * (2.0 / xrf.numel())).reshape(xr.shape).contiguous()
        else:
            diff = xrf - xf
            rec = diff.square().mean()
            g_xr = (diff * (2.0 / diff.numel())).reshape(xr.shape).contiguous()
        latent = zin.square().mean()
        g_gin, _ = chain_backward(self.pG, sG, g_xr, g_log, True, need_params=False)
        g_codes = (g_gin.reshape(B, -1)[:, :zin.shape[1]] + zin * (2.0 / zin.numel())).contiguous()
        dst = self.opt_e.grad_views
        emb = tuple(range(1, 1 + len(fam.e_tables)))       # only these planes of E's input gradient are consumed
        g_x0, _ = chain_backward(self.pE, sE, g_codes.reshape(codes.shape), n_log, bool(emb), True, dst,
                                 gx_planes=emb or None)
        gofs = 0 if (g_x0 is not None and g_x0.shape[-1] == len(emb)) else 1
        for j, t in enumerate(fam.e_tables):
            ops.plane_table_grad(g_x0.contiguous(), gofs + j, x0, 1 + j, idx, j, t.shape[0], out=dst[id(t)])
        self.opt_e.adam()
        self.pE.cache.refresh()
        return {"rec": rec, "latent": latent}


class GeneratorSampler:
    """Generator inference for the ``*_generator_score.py`` loops:

        gen = 0;  for _ in range(mc_rounds): gen = gen + G(randn(B,512,1,1), a);  gen = gen / mc_rounds

    (``audiomnist_generator_score.py:83-88``; ``mnist_generator_score.py:69-74`` is the mc_rounds = 1 case) as ONE
    forward over mc_rounds*B samples -- every layer sees mc_rounds times the rows -- replayed from a HIP graph per
    input shape.  The rounds are summed in the loop's order; the GEMMs pick their tile / split-K by row count, so the
    result equals the loop's to fp32 rounding (~1e-7), not bit for bit."""

    def __init__(self, G, capture=True):
        self.G = G
        self.capture = capture
        self._graphs = {}
        self._versions = None

    def _sync(self):
        """Graph replays run no host code, and a re-pack after a weight update lands in new buffers: drop the graphs
        captured for older parameter versions (inference callers update weights rarely, if ever)."""
        v = tuple(p._version for p in self.G.parameters())
        if v != self._versions:
            self._graphs.clear()
            self._versions = v

    def _forward(self, zs, a):
        R, B = zs.shape[0], zs.shape[1]
        a_rep = {k: v.repeat((R,) + (1,) * (v.dim() - 1)) for k, v in a.items()}
        out = self.G(zs.reshape((R * B,) + tuple(zs.shape[2:])), a_rep)
        out = out.reshape((R, B) + tuple(out.shape[1:]))
        gen = out[0]
        for r in range(1, R):
            gen = gen + out[r]
        return gen / R if R > 1 else gen

    @torch.no_grad()
    def __call__(self, zs, a):
        """zs: [mc_rounds, B, 512, 1, 1] latent draws (or [B, 512, 1, 1]); a: attribute dict as the callers pass it."""
        if zs.dim() == 4:
            zs = zs.unsqueeze(0)
        if not (self.capture and zs.is_cuda):
            return self._forward(zs, a)
        self._sync()
        key = (tuple(zs.shape), tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(a.items())), self.G.training)
        ent = self._graphs.get(key)
        if ent is None:
            st_z, st_a = zs.clone(), {k: v.clone() for k, v in a.items()}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._forward(st_z, st_a)                  # warm-up: weight packs, workspace
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                res = self._forward(st_z, st_a)
            ent = self._graphs[key] = (graph, st_z, st_a, res)
        graph, st_z, st_a, res = ent
        st_z.copy_(zs)
        for k, v in a.items():
            st_a[k].copy_(v)
        graph.replay()
        return res


def _mnist_family_eg(E, G):
    fam = MnistFamily.__new__(MnistFamily)
    fam.e_tables = [E.digit_embedding[0].weight]
    fam.d_tables = []
    fam.g_tables = [G.digit_embedding.weight]
    return fam
