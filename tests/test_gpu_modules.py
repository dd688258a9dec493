"""GPU parity of the drop-in modules and of the ALI iteration against the CPU oracle
(oracle/ali_oracle.py, itself pinned to the reference by tests/test_oracle_golden.py).
Tolerance: BASELINE.json north_star asks 1e-3 relative on losses / reconstructions; the fp32-MFMA
path is held to 2e-4 of each tensor's max-abs (losses 1e-5)."""
import copy

import numpy as np
import pytest
import torch

import ali_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 2e-4


def close(got, ref, rtol=TOL, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


class TieWatch:
    """Counts the LeakyReLU inputs of an oracle forward pass that lie within fp32 summation noise of zero.

    The sign of such a pre-activation depends on the order in which its dot product was accumulated -- on the CPU
    (thread count) as much as on the GPU (tile / split-K choice) -- and it switches the derivative between 1 and the
    negative slope for everything behind it.  A gradient comparison on such a case says nothing, so the module tests
    only run on cases WITHOUT one: ``tie_free`` re-seeds the inputs until the oracle's forward has none, and the
    comparison pass itself fails loudly (``strict``) if one shows up anyway.  Every tensor is then held to the plain
    max-abs bound -- no loosened criterion exists."""

    def __init__(self, *mods, strict=False):
        self.ties = 0
        self.strict = strict
        self.hooks = [m.register_forward_hook(self._hook) for mod in mods for m in mod.modules()
                      if isinstance(m, torch.nn.LeakyReLU)]

    def _hook(self, mod, inp, out):
        x = inp[0].detach().abs()
        self.ties += int((x <= 4e-7 * x.max()).sum())

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for h in self.hooks:
            h.remove()
        if self.strict and exc[0] is None:
            assert self.ties == 0, f"{self.ties} LeakyReLU inputs within fp32 noise of 0: pick another case (tie_free)"


def tie_free(mods, make, run, tries=24):
    """First ``args = make(variant)`` for which the oracle forward ``run(*args)`` has no LeakyReLU input within fp32
    summation noise of zero (about one element per million is: large cases need a few draws)."""
    for v in range(tries):
        args = make(v)
        with TieWatch(*mods) as tw, torch.no_grad():
            run(*args)
        if tw.ties == 0:
            return args
    pytest.fail(f"no tie-free case in {tries} draws")


def case_data(family, B, variant=0):
    """Inputs of a module case; variant 0 is the golden fixture's case (tests/test_oracle_golden.make_module_case)."""
    if family == "mnist":
        xs, a = orc.synth_morphomnist(B, seed=3 + 100 * variant)
        stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
        images, c = orc.mnist_scale_batch(xs, a, stats)
        z = torch.randn(B, 512, 1, 1, generator=torch.Generator().manual_seed(5 + 100 * variant))
        return images, c, z
    return orc.synth_spect_batch(family, B, seed=3 + 100 * variant)


def grad_close(got, ref, rtol=TOL, what=""):
    return close(got, ref, rtol, what)


def to_dev(c):
    return {k: v.cuda() for k, v in c.items()}


def product_models(family, d=64):
    if family == "mnist":
        import image_scms.mnist as pm
        return pm.Encoder(), pm.Generator(), pm.Discriminator()
    import importlib
    pm = importlib.import_module({"audio": "image_scms.audio_mnist", "whale": "image_scms.whalecalls",
                                  "esrf": "image_scms.esrf_acoustic"}[family])
    return pm.Encoder(d), pm.Generator(d), pm.Discriminator(d)


def paired_models(family="mnist", rescale=True, d=64, B=4):
    """oracle modules on CPU and product modules on cuda with identical weights."""
    from test_oracle_golden import make_module_case
    Eo, Go, Do, images, c, z = make_module_case(family, B, d)
    if not rescale:
        torch.manual_seed(11)
        Eo, Go, Do = orc.build_models(family, d)
    E, G, D = product_models(family, d)
    for src, dst in ((Eo, E), (Go, G), (Do, D)):
        dst.load_state_dict(copy.deepcopy(src.state_dict()))
    return (Eo, Go, Do), (E.cuda(), G.cuda(), D.cuda()), images, c, z


def check_param_grads(mod_o, mod_p, what, rtol=TOL):
    go = dict(mod_o.named_parameters())
    for k, p in mod_p.named_parameters():
        ref = go[k].grad if go[k].grad is not None else torch.zeros_like(go[k])
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        grad_close(got, ref, rtol, f"{what}.{k}.grad")


def test_mnist_modules_fwd_bwd_vs_oracle():
    import ali_hip
    (Eo, Go, Do), (E, G, D), _, _, _ = paired_models("mnist")
    gcot = torch.Generator().manual_seed(9)
    B = 4
    # Encoder
    Eo.train(), E.train()
    images, c, _ = tie_free([Eo], lambda v: case_data("mnist", B, v), lambda im, cc, zz: Eo(im, cc))
    cd = to_dev(c)
    with TieWatch(Eo, strict=True):
        exo = Eo(images, c)
    w = torch.randn(exo.shape, generator=gcot)
    (exo * w).sum().backward()
    ex = E(images.cuda(), cd)
    close(ex, exo, what="E.out")
    (ex * w.cuda()).sum().backward()
    check_param_grads(Eo, E, "E")
    # Generator, grads w.r.t. z too
    _, _, z = tie_free([Go], lambda v: case_data("mnist", B, v), lambda im, cc, zz: Go(zz, c))
    zo = z.clone().requires_grad_(True)
    with TieWatch(Go, strict=True):
        gzo = Go(zo, c)
    w = torch.randn(gzo.shape, generator=gcot)
    (gzo * w).sum().backward()
    zp = z.clone().cuda().requires_grad_(True)
    gz = G(zp, cd)
    close(gz, gzo, what="G.out")
    (gz * w.cuda()).sum().backward()
    grad_close(zp.grad, zo.grad, what="G.gz")
    check_param_grads(Go, G, "G")
    # Discriminator: eval mode, then train mode with the oracle's masks replayed
    for mode in ("eval", "train"):
        Do.zero_grad(), D.zero_grad()
        Do.train(mode == "train"), D.train(mode == "train")
        bufs = copy.deepcopy(Do.state_dict())

        def draw(v):
            Do.load_state_dict(bufs)              # a rejected draw must not leave BatchNorm running-stat updates behind
            torch.manual_seed(21 + v)
            return (orc.MaskTape(),)

        def fwd(tape):
            with orc.use_tape(tape):
                Do(images, exo.detach(), c)

        (drawn,) = tie_free([Do], draw, fwd)
        Do.load_state_dict(bufs)
        tape = orc.MaskTape(replay=drawn.masks)
        xo = images.clone().requires_grad_(True)
        zo = exo.detach().clone().requires_grad_(True)
        with orc.use_tape(tape), TieWatch(Do, strict=True):
            dlo = Do(xo, zo, c)
        w = torch.randn(dlo.shape, generator=gcot)
        (dlo * w).sum().backward()
        xp = images.clone().cuda().requires_grad_(True)
        zp = exo.detach().clone().cuda().requires_grad_(True)
        with ali_hip.injected_masks(drawn.masks):
            dl = D(xp, zp, cd)
        close(dl, dlo, what=f"D.{mode}.out")
        (dl * w.cuda()).sum().backward()
        grad_close(zp.grad, zo.grad, what=f"D.{mode}.gz")
        grad_close(xp.grad, xo.grad, what=f"D.{mode}.gx")
        check_param_grads(Do, D, f"D.{mode}")
        for k, v in D.state_dict().items():
            if "running" in k or "num_batches" in k:
                close(v.float(), Do.state_dict()[k].float(), 1e-5, f"D.{mode}.{k}")


@pytest.mark.parametrize("rescale", [True, False])
def test_mnist_ali_steps_vs_oracle(rescale):
    """3 iterations at bs=64 (BASELINE config 1 shape), reference init (losses at ln 2) and O(1) init."""
    import ali_hip
    from image_scms.training_utils import ali_step
    (Eo, Go, Do), (E, G, D), _, _, _ = paired_models("mnist", rescale=rescale)
    x, a = orc.synth_morphomnist(192, seed=1)
    stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
    oe, od = orc.build_optimizers(Eo, Go, Do, "mnist")
    pe = torch.optim.Adam(list(E.parameters()) + list(G.parameters()), lr=1e-4, betas=(0.5, 0.999))
    pd = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
    for m in (Eo, Go, Do, E, G, D):
        m.train()
    torch.manual_seed(123)
    before = {nm: copy.deepcopy(m.state_dict()) for nm, m in (("E", Eo), ("G", Go), ("D", Do))}
    for i in range(3):
        images, c = orc.mnist_scale_batch(x[i * 64:(i + 1) * 64], {k: v[i * 64:(i + 1) * 64] for k, v in a.items()},
                                          stats)
        z = torch.randn(64, 512, 1, 1)
        tape = orc.MaskTape()
        ro = orc.ali_step(Eo, Go, Do, oe, od, images, c, z, tape=tape)
        with ali_hip.injected_masks(tape.masks):
            rp = ali_step(E, G, D, pe, pd, images.cuda(), to_dev(c), z.cuda())
        # iteration 0 sees identical weights: rounding-level agreement.  Later iterations inherit the
        # (sign-like, noise-amplifying) Adam updates of the earlier ones: 1e-3 relative (north_star).
        tol = 1e-5 if i == 0 else 1e-3
        for k in ("loss_eg", "loss_d_real", "loss_d_fake", "dg", "de"):
            assert abs(rp[k].item() - ro[k]) <= tol * max(1.0, abs(ro[k])), (i, k, rp[k].item(), ro[k])
        if i == 0:
            # first Adam step: every weight moves by ~lr*sign(g); compare the *updates*, allowing the rare
            # elements whose gradient is at rounding level (|g| ~ eps) to differ
            for nm, mo, mp in (("E", Eo, E), ("G", Go, G), ("D", Do, D)):
                so = mo.state_dict()
                for k, v in mp.state_dict().items():
                    if not v.is_floating_point() or "running" in k:
                        continue
                    du_o = (so[k] - before[nm][k]).double()
                    du_p = (v.cpu() - before[nm][k]).double()
                    err = (du_p - du_o).abs()
                    if v.numel() >= 10000:          # outlier *fractions* only mean something on big tensors
                        assert (err > 0.05 * 1e-4).double().mean().item() < 2e-3, (nm, k)
                    assert err.mean().item() <= 0.05 * 1e-4, (nm, k, err.mean().item())
    for nm, mo, mp in (("E", Eo, E), ("G", Go, G), ("D", Do, D)):
        so = mo.state_dict()
        for k, v in mp.state_dict().items():
            if "num_batches" in k:
                assert int(v) == int(so[k])
                continue
            if "running" in k:
                close(v.float(), so[k].float(), 1e-3, f"{nm}.{k} after 3 steps")
                continue
            # an Adam update is ~lr*sign-like: an element whose gradient sits at rounding level may move the
            # other way (<= 2*lr per step); everything else must agree to a small fraction of one update
            diff = (v.cpu().double() - so[k].double()).abs()
            assert diff.max().item() <= 3 * 2.2e-4, (nm, k, diff.max().item())
            if v.numel() >= 10000:     # free-running: chaotic on small tensors (see the phase-synchronised test)
                assert diff.mean().item() <= 0.1 * 1e-4, (nm, k, diff.mean().item())
    # reconstructions G(E(x)) within 1e-3 (north_star)
    with torch.no_grad():
        for m in (Eo, Go, E, G):
            m.eval()
        rp_, ro_ = G(E(images.cuda(), to_dev(c)), to_dev(c)).cpu().double(), Go(Eo(images, c), c).double()
        rel = ((rp_ - ro_).norm() / ro_.norm()).item()
        amax = (rp_ - ro_).abs().max().item()
        # images live in [-1, 1]: 1e-3 of the signal, or 1e-3 of the range when G(E(x)) is still ~0 (reference init)
        assert rel <= 1e-3 or amax <= 1e-3, f"G(E(x)) rel L2 {rel:.3e}, max abs {amax:.3e} after 3 optimiser steps"


def test_callers_finetune_and_generator_score(golden_dir):
    """finetune_mnist_bigan.py:64-85 and mnist_generator_score.py:69-74 access patterns against the golden trace."""
    import os
    import image_scms.mnist as pm
    g = np.load(os.path.join(golden_dir, "callers_mnist.npz"), allow_pickle=False)
    torch.manual_seed(31)
    np.random.seed(31)
    E, G = pm.Encoder(), pm.Generator()
    E.apply(pm.init_weights), G.apply(pm.init_weights)
    orc.rescale_for_test_(E, 0.01, bias_seed=7), orc.rescale_for_test_(G, 0.01, bias_seed=8)
    E, G = E.cuda(), G.cuda()
    xs, a = orc.synth_morphomnist(8, seed=4)
    stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
    x, c = orc.mnist_scale_batch(xs, a, stats)
    x, c = x.cuda(), to_dev(c)
    E.train(), G.eval()
    opt = torch.optim.Adam(E.parameters(), lr=1e-4)
    rec, lat = [], []
    for _ in range(2):
        opt.zero_grad()
        codes = E(x, c)
        rl = torch.square(x - G(codes, c)).mean()
        ll = torch.square(codes).mean()
        (rl + ll).backward()
        opt.step()
        rec.append(rl.item()), lat.append(ll.item())
    np.testing.assert_allclose(rec, g["rec"], rtol=1e-4)
    np.testing.assert_allclose(lat, g["lat"], rtol=1e-4)
    with torch.no_grad():
        gen = G(torch.randn(8, 512, 1, 1, generator=torch.Generator().manual_seed(6)).cuda(), c)
    np.testing.assert_allclose(gen.reshape(8, -1)[:, 300:364].cpu().numpy(), g["gen_head"], rtol=1e-3, atol=1e-4)


def test_hip_library_is_loaded_and_mandatory():
    import ali_hip
    lib = ali_hip.load()
    assert lib.ali_version() >= 1
    with open("/proc/self/maps") as f:
        assert "libali_hip.so" in f.read()


def _stepper_setup(rescale=True, capture=False, bs=64, n=3):
    from ali_hip.step import AliStepper
    (Eo, Go, Do), (E, G, D), _, _, _ = paired_models("mnist", rescale=rescale)
    for m in (Eo, Go, Do, E, G, D):
        m.train()
    x, a = orc.synth_morphomnist(n * bs, seed=1)
    stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
    batches = []
    g = torch.Generator().manual_seed(123)
    for i in range(n):
        images, c = orc.mnist_scale_batch(x[i * bs:(i + 1) * bs], {k: v[i * bs:(i + 1) * bs] for k, v in a.items()},
                                          stats)
        batches.append((images, c, torch.randn(bs, 512, 1, 1, generator=g)))
    return (Eo, Go, Do), (E, G, D), AliStepper(E, G, D, capture=capture), batches


def _flat_grads(mods):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).double().cpu()
                      for m in mods for p in m.parameters()])


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


@pytest.mark.parametrize("rescale,bs", [(True, 64), (False, 64), (True, 512), (False, 512)])
def test_hand_scheduled_stepper_vs_oracle(rescale, bs):
    """AliStepper (no autograd, wasted work skipped, flat Adam kernel) vs the reference iteration, phase by phase.
    bs=512 is the benched configuration (BASELINE.json configs[1]): its tile / split-K / wgrad-slab choices and the
    paired 2B = 1024-row Discriminator passes differ from bs=64's.

    Adam's first updates are sign-like (+-lr whatever the gradient's size) and a LeakyReLU whose pre-activation is
    ~1e-7 can take the other branch on the GPU, so a handful of weights legitimately move the other way and the
    *next* phase amplifies that.  The stepper is therefore re-synchronised to the oracle's exact state (weights,
    BN buffers, Adam moments, step counts) before every phase and each phase is held to rounding level:
    gradients rel-L2 <= 2e-3 (allows such flips; typical 1e-6 -- the per-kernel tests hold the tight bar)."""
    import torch.nn as nn
    (Eo, Go, Do), (E, G, D), stepper, batches = _stepper_setup(rescale, bs=bs, n=3 if bs == 64 else 2)
    oe, od = orc.build_optimizers(Eo, Go, Do, "mnist")
    bce = nn.BCEWithLogitsLoss()
    lr = 1e-4

    def check_update(mods_o, mods_p, before, g_ref, what, ties=0):
        """Adam's first steps are sign-like: +-lr whatever the gradient's size.  An element whose gradient is at the
        gradients' noise level (|g| of the order of the GPU-vs-CPU difference: a summation-order effect, or the
        one-sample contribution of a LeakyReLU input at fp32 noise level -- at bs=512 every pass has dozens) may step
        the other way; every element with a gradient clearly above that level must take the oracle's step.  ``ties``
        = LeakyReLU inputs of the oracle's pass within fp32 noise of 0 (TieWatch): each may flip one unit of one
        sample, i.e. perturb the weight row feeding that unit (fan-in <= 4096) by 1/B of a per-sample gradient."""
        wo = torch.cat([p.detach().reshape(-1).double() for m in mods_o for p in m.parameters()])
        wp = torch.cat([p.detach().reshape(-1).double().cpu() for m in mods_p for p in m.parameters()])
        err = ((wp - before) - (wo - before)).abs()
        assert err.max().item() <= 2.2 * lr, (what, err.max().item())
        assert err.mean().item() <= 0.005 * lr, (what, err.mean().item())
        assert (err > 0.05 * lr).double().mean().item() < 3e-3, what
        off = n_big = bad_big = 0
        for m in mods_o:
            for p in m.parameters():
                n = p.numel()
                g = g_ref[off:off + n]
                big = g.abs() > 0.05 * g.pow(2).mean().sqrt()
                n_big += int(big.sum())
                bad_big += int((err[off:off + n][big] > 0.05 * lr).sum())
                off += n
        assert bad_big <= 1e-4 * n_big + 4096 * ties, (what, bad_big, n_big, ties)

    def weights(mods):
        return torch.cat([p.detach().reshape(-1).double() for m in mods for p in m.parameters()])

    for i, (images, c, z) in enumerate(batches):
        B = images.shape[0]
        valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
        tape = orc.MaskTape()
        stepper.load_state(Eo, Go, Do, oe, od)
        # ---- oracle, phase 1 (mnist.py:224-230)
        with orc.use_tape(tape), TieWatch(Eo, Go, Do) as tw:
            w_eg = weights((Eo, Go))
            oe.zero_grad()
            l_eg = (bce(Do(images, Eo(images, c), c), fake) + bce(Do(Go(z, c), z, c), valid)) / 2
            l_eg.backward()
            g_eg = _flat_grads((Eo, Go))
            oe.step()
        from ali_hip import dropout as _dropout
        with _dropout.injected_masks(tape.masks), torch.no_grad():
            cx = stepper._begin(images.cuda(), to_dev(c), z.cuda())
            stepper._phase_eg(cx)
        assert abs(cx["out"]["loss_eg"].item() - l_eg.item()) <= 1e-5 * max(1, abs(l_eg.item()))
        assert _rel(stepper.opt_eg.grad_logical().double().cpu(), g_eg) <= 2e-3, (i, "EG grads")
        check_update((Eo, Go), (E, G), w_eg, g_eg, f"EG update {i}", tw.ties)
        # ---- phase 2 (mnist.py:232-236) from the oracle's post-EG state
        stepper.load_state(Eo, Go, Do, oe, od)
        n0 = len(tape.masks)
        with orc.use_tape(tape), TieWatch(Eo, Do) as tw:
            w_d = weights((Do,))
            od.zero_grad()
            Eo.zero_grad(), Go.zero_grad()
            l_dr = bce(Do(images, Eo(images, c), c), valid)
            l_dr.backward()
            g_d = _flat_grads((Do,))
            od.step()
        with _dropout.injected_masks(tape.masks[n0:]), torch.no_grad():
            stepper._phase_d_real(cx)
        assert abs(cx["out"]["loss_d_real"].item() - l_dr.item()) <= 1e-5 * max(1, abs(l_dr.item()))
        assert _rel(stepper.opt_d.grad_logical().double().cpu(), g_d) <= 2e-3, (i, "D real grads")
        check_update((Do,), (D,), w_d, g_d, f"D real update {i}", tw.ties)
        # ---- phase 3 (mnist.py:237-241)
        stepper.load_state(Eo, Go, Do, oe, od)
        n0 = len(tape.masks)
        with orc.use_tape(tape), TieWatch(Go, Do) as tw:
            w_d = weights((Do,))
            od.zero_grad()
            l_df = bce(Do(Go(z, c), z, c), fake)
            l_df.backward()
            g_d = _flat_grads((Do,))
            od.step()
        with _dropout.injected_masks(tape.masks[n0:]), torch.no_grad():
            stepper._phase_d_fake(cx)
        assert abs(cx["out"]["loss_d_fake"].item() - l_df.item()) <= 1e-5 * max(1, abs(l_df.item()))
        assert _rel(stepper.opt_d.grad_logical().double().cpu(), g_d) <= 2e-3, (i, "D fake grads")
        check_update((Do,), (D,), w_d, g_d, f"D fake update {i}", tw.ties)
        # ---- phase 4 (mnist.py:243-248)
        stepper.load_state(Eo, Go, Do, oe, od)
        n0 = len(tape.masks)
        with orc.use_tape(tape), torch.no_grad():
            dg = Do(Go(z, c), z, c).sigmoid().mean().item()
            de = Do(images, Eo(images, c), c).sigmoid().mean().item()
        with _dropout.injected_masks(tape.masks[n0:]), torch.no_grad():
            stepper._phase_scores(cx)
        assert abs(cx["out"]["dg"].item() - dg) <= 1e-5 and abs(cx["out"]["de"].item() - de) <= 1e-5
        assert len(tape.masks) == 60
        for k, v in D.state_dict().items():
            if "num_batches" in k:
                assert int(v) == int(Do.state_dict()[k]), k
            elif "running" in k:
                close(v.float(), Do.state_dict()[k].float(), 1e-5, k)


def test_stepper_three_iterations_free_running():
    """No re-synchronisation: after 3 full iterations losses and scores stay within the north_star 1e-3."""
    (Eo, Go, Do), (E, G, D), stepper, batches = _stepper_setup(True)
    oe, od = orc.build_optimizers(Eo, Go, Do, "mnist")
    for i, (images, c, z) in enumerate(batches):
        tape = orc.MaskTape()
        ro = orc.ali_step(Eo, Go, Do, oe, od, images, c, z, tape=tape)
        rp = stepper.step(images.cuda(), to_dev(c), z.cuda(), masks=tape.masks)
        for k in ("loss_eg", "loss_d_real", "loss_d_fake", "dg", "de"):
            assert abs(rp[k].item() - ro[k]) <= 1e-3 * max(1.0, abs(ro[k])), (i, k, rp[k].item(), ro[k])


@pytest.mark.parametrize("variant", ["no_join", "own_launches", "no_pairing", "no_pairing_bs512"])
def test_scheduling_variants_of_the_stepper_agree(variant):
    """The launch-saving schedules are pure re-arrangements: (a) D's joint rows written in place by the branch ends
    (no cat / mask pass / slice copies / act' passes) vs the plain three-chain schedule -- bit-identical losses, scores
    and weights; (b) weight gradients issued as one multi-job launch per pass vs one launch per layer -- identical up to
    the summation order of the pixel split (1e-6 of the losses, weights within a sign-flip of rounding-level Adam
    steps); (c) independent chains (E || G, the two branches of the E+G backward pass, D.dx || D.dz backward, E' || D.dx)
    advanced in lock step with their GEMMs issued as multi-job launches (chain.run_parallel / ali_gemm_launch_multi)
    vs every GEMM launched on its own -- bit-identical (every job keeps its own tile, split-K and workspace)."""
    import ali_hip
    from ali_hip import ops
    bs = 512 if variant.endswith("bs512") else 64
    exact = variant in ("no_join", "no_pairing", "no_pairing_bs512")
    ali_hip.manual_seed(5)
    _, (E1, G1, D1), a, batches = _stepper_setup(capture=False, bs=bs)
    ali_hip.manual_seed(5)
    _, (E2, G2, D2), b, _ = _stepper_setup(capture=False, bs=bs)
    assert a._join and a._fold is not None
    if variant == "no_join":
        b._join = False
    old, old_pair = ops.DEFER_WGRAD_LAUNCH, ops.PAIR_GEMMS
    outs = []
    try:
        for images, c, z in batches:
            ops.DEFER_WGRAD_LAUNCH, ops.PAIR_GEMMS = True, True
            r1 = {k: v.item() for k, v in a.step(images.cuda(), to_dev(c), z.cuda()).items()}
            ops.DEFER_WGRAD_LAUNCH = variant != "own_launches"
            ops.PAIR_GEMMS = not variant.startswith("no_pairing")
            r2 = {k: v.item() for k, v in b.step(images.cuda(), to_dev(c), z.cuda()).items()}
            outs.append((r1, r2))
    finally:
        ops.DEFER_WGRAD_LAUNCH, ops.PAIR_GEMMS = old, old_pair
    for i, (r1, r2) in enumerate(outs):
        for k in r1:
            if exact:
                assert r1[k] == r2[k], (i, k, r1[k], r2[k])
            else:   # first iteration: same weights, only the summation order differs; later: sign-like Adam steps of
                # rounding-level gradients have diverged (same bound as the free-running test against the oracle)
                tol = 2e-5 if i == 0 else 1e-3
                assert abs(r1[k] - r2[k]) <= tol * max(1.0, abs(r1[k])), (i, k, r1[k], r2[k])
    if exact:
        assert torch.equal(a.opt_d.flat, b.opt_d.flat) and torch.equal(a.opt_eg.flat, b.opt_eg.flat)
    else:
        steps = len(batches) * 2
        assert (a.opt_d.flat - b.opt_d.flat).abs().max().item() <= steps * 2.2e-4


@pytest.mark.parametrize("bs", [64, 512])
def test_graph_captured_stepper_equals_eager(bs):
    """HIP-graph replay of the iteration == eager launches (device-side Adam step count and dropout counter), bit for
    bit -- also at the benched bs=512, so the eager-vs-oracle checks above carry over to the replayed graph."""
    import ali_hip
    ali_hip.manual_seed(99)
    _, (E1, G1, D1), eager, batches = _stepper_setup(capture=False, bs=bs)
    ali_hip.manual_seed(99)
    _, (E2, G2, D2), graphed, _ = _stepper_setup(capture=True, bs=bs)
    outs = []
    for images, c, z in batches:
        r1 = eager.step(images.cuda(), to_dev(c), z.cuda())
        r2 = graphed.step(images.cuda(), to_dev(c), z.cuda())
        outs.append(({k: v.item() for k, v in r1.items()}, {k: v.item() for k, v in r2.items()}))
    for r1, r2 in outs:
        for k in r1:
            assert r1[k] == r2[k], (k, r1[k], r2[k])
    for (k, v1), (_, v2) in zip(D1.state_dict().items(), D2.state_dict().items()):
        assert torch.equal(v1, v2), k
    assert torch.equal(eager.opt_d.flat, graphed.opt_d.flat) and torch.equal(eager.opt_eg.flat, graphed.opt_eg.flat)
    # Dropout masks differ between iterations (fresh counter every replay)
    assert outs[0][0]["dg"] != outs[1][0]["dg"]


@pytest.mark.parametrize("capture", [False, True])
def test_stepper_with_one_rank_rccl_group(capture):
    """The DP wiring on a 1-rank "nccl" (= RCCL) group: every collective of the schedule is really issued -- the three
    flat-buffer all-reduces (asynchronous, overlapped with the next segment), the BatchNorm buffer average, and with
    ``capture`` the one-HIP-graph-per-segment replay with the collectives launched in between -- and, a 1-rank sum
    being the identity and 1/world = 1, the result must equal the group-less eager stepper bit for bit."""
    import os
    import torch.distributed as dist
    import ali_hip
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        from ali_hip import dp
        from ali_hip.step import AliStepper
        ali_hip.manual_seed(5)
        _, (E1, G1, D1), plain, batches = _stepper_setup()
        ali_hip.manual_seed(5)
        _, (E2, G2, D2), _, _ = _stepper_setup()
        grouped = AliStepper(E2, G2, D2, process_group=dist.group.WORLD, capture=capture)
        assert grouped.dist and grouped.world == 1
        issued = []
        real_async = dp.allreduce_sum_async_

        def counting(flat, group=None):
            issued.append(flat.numel())
            return real_async(flat, group)

        dp.allreduce_sum_async_ = counting
        try:
            for images, c, z in batches:
                r1 = plain.step(images.cuda(), to_dev(c), z.cuda())
                r2 = grouped.step(images.cuda(), to_dev(c), z.cuda())
                assert all(r1[k].item() == r2[k].item() for k in r1)
        finally:
            dp.allreduce_sum_async_ = real_async
        if capture:
            assert any(k[0] == "seg" for k in grouped._graph), "the segmented graph replay did not run"
        # 3 all-reduces per iteration (E+G once, D twice), plus the warm-up iteration of the capture
        assert len(issued) >= 3 * len(batches) and set(issued) == {grouped.opt_eg.n, grouped.opt_d.n}
        assert torch.equal(plain.opt_d.flat, grouped.opt_d.flat) and torch.equal(plain.opt_eg.flat, grouped.opt_eg.flat)
        for (k, v1), (_, v2) in zip(D1.state_dict().items(), D2.state_dict().items()):
            assert torch.equal(v1, v2), k
    finally:
        if created:
            dist.destroy_process_group()


SPECT_CASES = [("audio", 8, 2), ("audio", 64, 2), ("whale", 16, 1), ("esrf", 8, 1)]


@pytest.mark.parametrize("family,d,B", SPECT_CASES)
def test_spect_modules_fwd_bwd_vs_oracle(family, d, B):
    """audio (128x128), whale (256x256), ESRF (512x512) Encoder / Generator / Discriminator forward + backward on the
    HIP kernels vs the oracle (weights of tests/golden/modules_<family>_*.npz's case; inputs re-drawn until the oracle's
    forward has no LeakyReLU input at fp32 noise level, see TieWatch): Linear+Unflatten as a permuted 1x1 GEMM, 5x5
    stride-2 convolutions, 5x5 stride-2 transposed convolutions in 4 sub-pixel phases.  Every output and gradient is
    held to the strict max-abs bound."""
    (Eo, Go, Do), (E, G, D), _, _, _ = paired_models(family, d=d, B=B)
    gcot = torch.Generator().manual_seed(9)
    for m in (Eo, Go, Do, E, G, D):
        m.train()
    images, c, _ = tie_free([Eo], lambda v: case_data(family, B, v), lambda im, cc, zz: Eo(im, cc))
    cd = to_dev(c)
    with TieWatch(Eo, strict=True):
        exo = Eo(images, c)
    w = torch.randn(exo.shape, generator=gcot)
    (exo * w).sum().backward()
    ex = E(images.cuda(), cd)
    close(ex, exo, what="E.out")
    (ex * w.cuda()).sum().backward()
    check_param_grads(Eo, E, "E", rtol=1e-3)
    _, _, z = tie_free([Go], lambda v: case_data(family, B, v), lambda im, cc, zz: Go(zz, c))
    zo = z.clone().requires_grad_(True)
    with TieWatch(Go, strict=True):
        gzo = Go(zo, c)
    w = torch.randn(gzo.shape, generator=gcot)
    (gzo * w).sum().backward()
    zp = z.clone().cuda().requires_grad_(True)
    gz = G(zp, cd)
    close(gz, gzo, what="G.out")
    (gz * w.cuda()).sum().backward()
    grad_close(zp.grad, zo.grad, rtol=1e-3, what="G.gz")
    check_param_grads(Go, G, "G", rtol=1e-3)
    # Discriminator: its own image draw (same attributes), the Encoder's code as z input
    imd, _, _ = tie_free([Do], lambda v: case_data(family, B, v), lambda im, cc, zz: Do(im, exo.detach(), c))
    xo = imd.clone().requires_grad_(True)
    zo = exo.detach().clone().requires_grad_(True)
    with TieWatch(Do, strict=True):
        dlo = Do(xo, zo, c)
    w = torch.randn(dlo.shape, generator=gcot)
    (dlo * w).sum().backward()
    xp = imd.clone().cuda().requires_grad_(True)
    zp = exo.detach().clone().cuda().requires_grad_(True)
    dl = D(xp, zp, cd)
    close(dl, dlo, what="D.out")
    (dl * w.cuda()).sum().backward()
    grad_close(zp.grad, zo.grad, rtol=1e-3, what="D.gz")
    grad_close(xp.grad, xo.grad, rtol=1e-3, what="D.gx")
    check_param_grads(Do, D, "D", rtol=1e-3)


@pytest.mark.parametrize("family,d,B", [("audio", 8, 4), ("whale", 8, 2), ("esrf", 4, 2)])
def test_spect_stepper_iteration_vs_oracle(family, d, B):
    """One hand-scheduled iteration of the spectrogram families (no dropout / BN: deterministic given z) vs the
    oracle's ali_step: losses, scores and the Adam update of every parameter."""
    from ali_hip.step import AliStepper
    (Eo, Go, Do), (E, G, D), images, c, z = paired_models(family, d=d, B=B)
    for m in (Eo, Go, Do, E, G, D):
        m.train()
    oe, od = orc.build_optimizers(Eo, Go, Do, family)
    stepper = AliStepper(E, G, D, betas=(0.5, 0.9))
    before = {nm: copy.deepcopy(m.state_dict()) for nm, m in (("E", Eo), ("G", Go), ("D", Do))}
    ro = orc.ali_step(Eo, Go, Do, oe, od, images, c, z)
    rp = stepper.step(images.cuda(), to_dev(c), z.cuda())
    for k in ("loss_eg", "loss_d_real", "loss_d_fake", "dg", "de"):
        assert abs(rp[k].item() - ro[k]) <= 2e-4 * max(1.0, abs(ro[k])), (k, rp[k].item(), ro[k])
    lr = 1e-4
    for nm, mo, mp in (("E", Eo, E), ("G", Go, G), ("D", Do, D)):
        so = mo.state_dict()
        wo = torch.cat([(so[k] - before[nm][k]).reshape(-1).double() for k in so])
        wp = torch.cat([(v.cpu() - before[nm][k]).reshape(-1).double() for k, v in mp.state_dict().items()])
        err = (wp - wo).abs()
        assert (err > 0.05 * lr).double().mean().item() < 5e-3, nm
        assert err.mean().item() <= 0.02 * lr, (nm, err.mean().item())


@pytest.mark.parametrize("capture", [False, True])
def test_pipelined_last_allreduce_schedule_equals_plain(capture):
    """AliStepper(pipeline_reduce=True): the next iteration's E(x) / G(z) forward passes are computed at the tail of the
    current iteration (where a data-parallel run has its last all-reduce in flight), into persistent buffers, and
    consumed by the next step -- the same arithmetic in another order: bit-identical to the plain schedule, with the
    next batch announced (``ahead``) or not, eager and replayed from the per-segment HIP graphs."""
    import ali_hip
    ali_hip.manual_seed(5)
    _, _, a, batches = _stepper_setup(capture=False, bs=64, n=4)
    ali_hip.manual_seed(5)
    _, _, b, _ = _stepper_setup(capture=capture, bs=64, n=4)
    b.segmented, b.pipeline_reduce = True, True
    dev = [(im.cuda(), to_dev(c), z.cuda()) for im, c, z in batches]
    for i, (images, c, z) in enumerate(dev):
        r1 = {k: v.item() for k, v in a.step(images, c, z).items()}
        nxt = dev[i + 1] if i + 1 < len(dev) and i != 1 else None       # (iteration 2's batch comes unannounced)
        r2 = {k: v.item() for k, v in b.step(images, c, z, ahead=nxt).items()}
        assert r1 == r2, (i, r1, r2)
    assert torch.equal(a.opt_d.flat, b.opt_d.flat) and torch.equal(a.opt_eg.flat, b.opt_eg.flat)
    assert len(b._arena) >= 8


def test_segmented_graph_replay_equals_eager():
    """The data-parallel replay (one HIP graph per segment, collectives in between) on one rank == eager."""
    import ali_hip
    ali_hip.manual_seed(77)
    _, (E1, G1, D1), eager, batches = _stepper_setup(capture=False)
    ali_hip.manual_seed(77)
    _, (E2, G2, D2), seg, _ = _stepper_setup(capture=True)
    seg.segmented = True
    for images, c, z in batches:
        r1 = eager.step(images.cuda(), to_dev(c), z.cuda())
        r2 = seg.step(images.cuda(), to_dev(c), z.cuda())
        for k in r1:
            assert r1[k].item() == r2[k].item(), (k, r1[k].item(), r2[k].item())
    assert torch.equal(eager.opt_d.flat, seg.opt_d.flat) and torch.equal(eager.opt_eg.flat, seg.opt_eg.flat)
    for (k, v1), (_, v2) in zip(D1.state_dict().items(), D2.state_dict().items()):
        assert torch.equal(v1, v2), k


def test_finetune_stepper_matches_reference_trace(golden_dir):
    """FinetuneStepper (E fwd+bwd, G fwd + dgrad only, Adam(E)) reproduces finetune_mnist_bigan.py:64-85 (mse)."""
    import os
    import image_scms.mnist as pm
    from ali_hip.step import FinetuneStepper
    g = np.load(os.path.join(golden_dir, "callers_mnist.npz"), allow_pickle=False)
    torch.manual_seed(31)
    np.random.seed(31)
    E, G = pm.Encoder(), pm.Generator()
    E.apply(pm.init_weights), G.apply(pm.init_weights)
    orc.rescale_for_test_(E, 0.01, bias_seed=7), orc.rescale_for_test_(G, 0.01, bias_seed=8)
    E, G = E.cuda(), G.cuda()
    xs, a = orc.synth_morphomnist(8, seed=4)
    stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
    x, c = orc.mnist_scale_batch(xs, a, stats)
    E.train(), G.eval()
    ft = FinetuneStepper(E, G, lr=1e-4)
    out = [ft.step(x.cuda(), to_dev(c)) for _ in range(2)]
    np.testing.assert_allclose([o["rec"].item() for o in out], g["rec"], rtol=1e-4)
    np.testing.assert_allclose([o["latent"].item() for o in out], g["lat"], rtol=1e-4)


def test_pickled_module_checkpoint_roundtrip(tmp_path):
    """The reference saves whole modules (train_mnist_image_scm.py:61-67) and its callers torch.load them
    (finetune_mnist_bigan.py:60-62): modules that have already run on the HIP path must still pickle."""
    import image_scms.mnist as pm
    from ali_hip.step import AliStepper
    (_, _, _), (E, G, D), images, c, z = paired_models("mnist")
    ex = E(images.cuda(), to_dev(c))                      # builds the kernel plans / packed weights
    stepper = AliStepper(E, G, D)                         # re-points parameters into the flat buffers
    stepper.step(images.cuda(), to_dev(c), z.cuda())
    path = tmp_path / "ckpt.tar"
    torch.save({"E": E, "G": G, "D": D, "E_state_dict": E.state_dict()}, path)
    ck = torch.load(path, map_location="cuda", weights_only=False)
    E2, G2 = ck["E"], ck["G"]
    E.eval(), G.eval(), E2.eval(), G2.eval()
    with torch.no_grad():
        assert torch.equal(G2(E2(images.cuda(), to_dev(c)), to_dev(c)), G(E(images.cuda(), to_dev(c)), to_dev(c)))
    E3 = pm.Encoder()
    E3.load_state_dict(ck["E_state_dict"])
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(E3.state_dict().values(), E.state_dict().values()))


def test_generator_sampler_equals_the_scoring_loop():
    """SURVEY 8f.1, scoring path: mean over mc_rounds of G(z_r, a) as one batched, graph-replayed forward == the
    reference loop (audiomnist_generator_score.py:83-88) to fp32 rounding; weights updated in place are picked up."""
    from ali_hip.step import GeneratorSampler
    _, (E, G, D), images, c, z = paired_models("mnist")
    G.eval()
    cd = to_dev(c)
    R, B = 3, images.shape[0]
    gz = torch.Generator().manual_seed(17)
    sampler = GeneratorSampler(G)
    for trial in range(2):
        zs = torch.randn(R, B, 512, 1, 1, generator=gz).cuda()
        with torch.no_grad():
            gen = 0
            for r in range(R):
                gen = gen + G(zs[r], cd)
            gen = gen / R
        out = sampler(zs, cd)
        close(out, gen, rtol=1e-5, what=f"sampler trial {trial}")
        with torch.no_grad():
            for p in G.parameters():
                p.mul_(1.01)


@pytest.mark.parametrize("capture", [False, True])
def test_stepper_checkpoint_resume_is_exact(capture, tmp_path):
    """save after 2 iterations, resume in a fresh stepper, 2 more == 4 uninterrupted iterations bit for
    bit (weights, BN buffers, both Adam states, dropout stream); the file is the reference's state-dict format."""
    import ali_hip
    from ali_hip.step import AliStepper

    def fresh():
        ali_hip.manual_seed(77)
        _, (E, G, D), st, batches = _stepper_setup(capture=capture)
        return (E, G, D), st, batches

    def digest(st):
        return orc.tensor_digest(torch.cat([t.reshape(-1).float().cpu() for t in st._state_tensors()]))

    (_, _, _), ref, batches = fresh()
    seq = [batches[i % len(batches)] for i in range(4)]
    for images, c, z in seq:
        ref.step(images.cuda(), to_dev(c), z.cuda())
    (E1, G1, D1), first, _ = fresh()
    for images, c, z in seq[:2]:
        first.step(images.cuda(), to_dev(c), z.cuda())
    torch.save(first.state_dict(), tmp_path / "ckpt.tar")
    ck = torch.load(tmp_path / "ckpt.tar")
    assert set(ck["D_state_dict"]) == set(D1.state_dict()) and ck["optimizer_D"]["step"] == 4
    (_, _, _), second, _ = fresh()
    second.load_state_dict(ck)
    for images, c, z in seq[2:]:
        second.step(images.cuda(), to_dev(c), z.cuda())
    assert digest(second) == digest(ref)


def test_mnist_train_entry_point_on_cuda(tmp_path):
    """image_scms.mnist.train(device='cuda') -- the reference's own entry point -- on the stepper: ragged last batch
    (160 = 2*64 + 32: two captured graphs alternate over 2 epochs), d_updates_per_g_update, finite scores, and the
    returned modules / optimisers usable the way train_mnist_image_scm.py:61-67 uses them (pickled to a .tar)."""
    import ali_hip
    import image_scms.mnist as pm
    ali_hip.manual_seed(4)
    torch.manual_seed(4)
    np.random.seed(4)
    x, a = orc.synth_morphomnist(160, seed=3)
    E, G, D, oD, oE = pm.train(x, a, n_epochs=2, device="cuda", save_images_every=None, batch_size=64,
                               d_updates_per_g_update=2)
    assert all(torch.isfinite(p).all() for m in (E, G, D) for p in m.parameters())
    assert int(D.dx[4].num_batches_tracked) > 0
    Eb, Gb, Db = pm.Encoder(), pm.Generator(), pm.Discriminator()
    torch.manual_seed(4)
    moved = sum(float((p.detach().cpu() - q.detach()).abs().sum()) for p, q in zip(E.parameters(), Eb.parameters()))
    assert moved > 0
    torch.save({"E": E, "G": G, "D": D, "optimizer_D": oD, "optimizer_E": oE}, tmp_path / "model.tar")
    back = torch.load(tmp_path / "model.tar", weights_only=False)
    E2, G2 = back["E"].cuda().eval(), back["G"].cuda().eval()
    images, c = pm._scale_batch(x[:8], {k: v[:8] for k, v in a.items()},
                                {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"},
                                "cuda")
    with torch.no_grad():
        rec = G2(E2(images, c), c)
    assert rec.shape == (8, 1, 28, 28) and torch.isfinite(rec).all()
    assert back["optimizer_D"].state_dict()["step"] > 0
