"""GPU parity at the configurations that are benched / named by BASELINE.json but were only covered at reduced size:

* the MorphoMNIST iteration at bs=512 (configs[1]) against the committed reference trajectory
  ``tests/golden/mnist_traj_n1024_bs512.npz`` (reference image_scms/mnist.py:202-248);
* the whale-call and ESRF modules at their real width d=64 (whalecalls.py:230-387, esrf_acoustic.py:134-260);
* ``train_on_stream`` (the loop of audio_mnist.py:372-420 and its copies) against the oracle's iteration.

Tolerances: losses 1e-5 on identical weights, 1e-3 (north_star) once an optimiser step lies in between; module
outputs 2e-4 of the tensor's max-abs."""
import copy
import os

import numpy as np
import pytest
import torch

import ali_oracle as orc
from test_gpu_modules import close, paired_models, product_models, to_dev

pytestmark = pytest.mark.gpu


def test_bs512_stepper_vs_reference_trajectory(golden_dir):
    """The first two iterations of mnist.train at bs=512 -- exactly the benched shape: paired 2B=1024 Discriminator
    passes with per-pass BatchNorm, split-K rules, wgrad slabs -- with the reference's z and Dropout2d masks (taken
    from the oracle, which tests/test_oracle_golden.py shows reproduces this fixture bit for bit).  Losses are held to
    the *fixture's* values: 1e-5 for iteration 0 (identical weights), 1e-3 afterwards (sign-like first Adam steps)."""
    from ali_hip.step import AliStepper
    g = np.load(os.path.join(golden_dir, "mnist_traj_n1024_bs512.npz"), allow_pickle=False)
    x, a = orc.synth_morphomnist(1024, seed=1)
    assert orc.tensor_digest(x) == str(g["x_digest"])
    torch.manual_seed(1)
    E0, G0, D0 = orc.build_models("mnist")           # the weights mnist.train starts from (ctor + init_weights draws)
    init = [copy.deepcopy(m.state_dict()) for m in (E0, G0, D0)]
    torch.manual_seed(1)
    np.random.seed(1)
    rec = []
    Eo, Go, Do, _, _, scores = orc.mnist_train(x, a, n_epochs=1, batch_size=512, record=rec)
    L = g["bce_calls"]
    np.testing.assert_allclose([r["loss_eg"] for r in rec], g["loss_eg"], rtol=2e-5)   # oracle on this host ~ fixture
    E, G, D = product_models("mnist")
    for m, sd in zip((E, G, D), init):
        m.load_state_dict(sd)
        m.cuda().train()
    stepper = AliStepper(E, G, D)
    for i, r in enumerate(rec):
        out = stepper.step(r["images"].cuda(), to_dev(r["c"]), r["z"].cuda(), masks=r["masks"])
        tol = 1e-5 if i == 0 else 1e-3
        for key, ref in (("loss_eg", g["loss_eg"][i]), ("loss_d_real", L[i, 2]), ("loss_d_fake", L[i, 3]),
                         ("dg", r["dg"]), ("de", r["de"])):
            got = out[key].item()
            assert abs(got - ref) <= tol * max(1.0, abs(ref)), (i, key, got, ref)
    # after both iterations: BatchNorm buffers and every weight against the oracle's end state (which
    # tests/test_oracle_golden.py pins to the fixture's per-tensor statistics and weight digest)
    for nm, mod, ref_mod in (("E", E, Eo), ("G", G, Go), ("D", D, Do)):
        so = ref_mod.state_dict()
        for k, v in mod.state_dict().items():
            if "num_batches" in k:
                assert int(v) == int(so[k]) == int(g[f"stats.{nm}.{k}"][0])
            elif "running" in k:
                # running statistics after 12 updates, the last 8 with weights that already took sign-like Adam
                # steps (an element whose gradient is at rounding level may step the other way): measured 1.0e-3 to
                # 3.1e-3 across kernel revisions that only changed summation orders (which elements flip is chaotic)
                close(v.float(), so[k].float(), 1e-2, f"{nm}.{k}")
            else:   # an Adam update is sign-like: |delta| <= ~lr per step whatever the gradient's size, so two runs
                # differ by at most 2*lr per optimiser step where a rounding-level gradient changed sign
                steps = 4 if nm == "D" else 2             # two iterations: E+G step once, D twice per iteration
                diff = (v.cpu().double() - so[k].double()).abs()
                assert diff.max().item() <= steps * 2.2e-4, (nm, k, diff.max().item())
                if v.numel() >= 10000:
                    assert diff.mean().item() <= 0.15 * 1e-4, (nm, k, diff.mean().item())   # measured 0.9-1.0e-5 (share of rounding-level gradients)


@pytest.mark.parametrize("precision", ["f32", "f16"])
@pytest.mark.parametrize("family", ["whale", "esrf"])
def test_full_width_module_forward_vs_oracle(family, precision):
    """Encoder / Generator / Discriminator forward at the reference's width d=64, B=1 (whale 256x256: 56.7 M / 33.6 M /
    56.2 M parameters; ESRF 512x512: 332 M / 56 M / 335 M, layers up to 2048 -> 4096 channels) vs the fp32 oracle.
    fp32 path: 2e-4 of max-abs.  fp16-MFMA path (BASELINE config 5): the north_star's 1e-3 relative (L2) on every
    module output -- eight to thirteen layers of fp16 operand rounding (2^-11 each, fp32 accumulation)."""
    from ali_hip import ops
    torch.manual_seed(13)
    Eo, Go, Do = orc.build_models(family, 64)
    for i, m in enumerate((Eo, Go, Do)):
        orc.rescale_for_test_(m, 0.001, bias_seed=3 + i)
        m.eval()
    images, c, z = orc.synth_spect_batch(family, 1, seed=5)
    with torch.no_grad():
        exo = Eo(images, c)
        gzo = Go(z, c)
        dlo = Do(images, exo, c)
    outs = {}
    for nm, src in (("E", Eo), ("G", Go), ("D", Do)):     # one product module on the card at a time (ESRF: 1.3 GB each)
        idx = "EGD".index(nm)
        mod = product_models(family, 64)[idx]
        mod.load_state_dict(src.state_dict())
        mod = mod.cuda().eval()
        with torch.no_grad(), ops.precision(precision):
            if nm == "E":
                outs[nm] = mod(images.cuda(), to_dev(c)).cpu()
            elif nm == "G":
                outs[nm] = mod(z.cuda(), to_dev(c)).cpu()
            else:
                outs[nm] = mod(images.cuda(), exo.cuda(), to_dev(c)).cpu()
        del mod
        torch.cuda.empty_cache()
    if precision == "f32":
        close(outs["E"], exo, what=f"{family} d=64 E.out")
        close(outs["G"], gzo, what=f"{family} d=64 G.out")
        close(outs["D"], dlo, what=f"{family} d=64 D.out")
        return
    for nm, got, ref in (("E", outs["E"], exo), ("G", outs["G"], gzo)):
        rel = ((got.double() - ref.double()).norm() / ref.double().norm()).item()
        assert 0 < rel <= 1e-3, f"{family} d=64 {nm}.out on fp16 MFMA vs fp32 oracle: rel L2 {rel:.3e}"
    # D's output at B=1 is ONE logit: "relative" to its own (possibly small) value says nothing.  What the north_star
    # bounds is the loss it feeds, softplus(+-logit), whose error is at most the logit's: 1e-3 of max(1, |logit|).
    got, ref = outs["D"].item(), dlo.item()
    assert 0 < abs(got - ref) <= 1e-3 * max(1.0, abs(ref)), f"{family} d=64 D logit on fp16 MFMA {got} vs fp32 oracle {ref}"


# (family, d, B, ALI_TILE_M_SCALE): B * scale = the bench's per-GPU batch (bench.py SPECT: 256 / 128 / 64), so pick_tile
# and wgrad_tile choose for every layer the tile the bench runs it on (64x128 / 128x128 fp32 tiles above 4096 blocks)
BENCH_TILE_CASES = [("audio", 64, 16, 16), ("whale", 64, 4, 32), ("esrf", 64, 2, 32)]


@pytest.mark.parametrize("family,d,B,scale", BENCH_TILE_CASES)
def test_spect_stepper_iteration_on_the_bench_tiles(family, d, B, scale):
    """One hand-scheduled iteration at the reference's width d=64 with O(1)-rescaled weights (the reference init leaves
    every loss at ln 2 whatever the kernels compute) on the tiles of the bench batch: ``ALI_TILE_M_SCALE`` makes the tile
    choice of every GEMM behave as if the batch were ``scale`` times larger, while the oracle only has to run B samples
    (audio_mnist.py:186-198,224-243,384-420; whalecalls.py:462-498; esrf_acoustic.py:144-199,341-377).
    (i) Every GEMM launch of the iteration is audited in situ (tests/launch_audit.py): recomputed with torch's CPU
    convolution from the operands the kernel read, epilogue included, 2e-4 of max-abs -- immune to LeakyReLU sign ties.
    (ii) Losses and scores vs the oracle's ``ali_step`` at 2e-4; the last phase's Discriminator gradients and the Adam
    updates statistically: at this width an iteration has hundreds of LeakyReLU inputs within fp32 noise of 0 (229 in
    the whale case), and one of them taking the other slope moves every gradient below it by ~1e-2 (measured per tensor
    with scratch/dbg_whale_tiles.py: 1e-6 above the flipped unit, 5e-3..1e-2 below; which unit flips depends on the tile)."""
    from ali_hip import ops
    from ali_hip.step import AliStepper
    from launch_audit import LaunchAudit
    (Eo, Go, Do), (E, G, D), images, c, z = paired_models(family, d=d, B=B)
    for m in (Eo, Go, Do, E, G, D):
        m.train()
    oe, od = orc.build_optimizers(Eo, Go, Do, family)
    before = {nm: copy.deepcopy(m.state_dict()) for nm, m in (("E", Eo), ("G", Go), ("D", Do))}
    ro = orc.ali_step(Eo, Go, Do, oe, od, images, c, z)
    with ops.tuning(ALI_TILE_M_SCALE=scale):
        # the knob does what it says: a layer's tile at B with the scale == its tile at the bench batch without
        P = (images.shape[-1] + 2 - 5) // 2 + 1
        P2 = (P + 2 - 5) // 2 + 1
        g_small = ops.geom(B, P, P, d, P2, P2, 2 * d, 5, 5, 2, 1)
        rows_scaled = ops.conv_mtiles(g_small, 0)[1]
        stepper = AliStepper(E, G, D, betas=(0.5, 0.9))
        audit = LaunchAudit()
        with ops.launch_hook(audit):
            rp = stepper.step(images.cuda(), to_dev(c), z.cuda())
        torch.cuda.synchronize()
    g_big = ops.geom(B * scale, P, P, d, P2, P2, 2 * d, 5, 5, 2, 1)
    assert rows_scaled == ops.conv_mtiles(g_big, 0)[1]
    n_conv = sum(1 for m in list(E.modules()) + list(G.modules()) + list(D.modules())
                 if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.Linear)))
    # 2 E + 2 G + 6 D forwards, E / G / 2 x D backward: at least three data-path GEMMs and one weight gradient per layer
    assert audit.checked["fwd"] + audit.checked["bwd_data"] >= 3 * n_conv and audit.checked["wgrad"] >= n_conv, audit.checked
    print(f"launch audit: {audit.checked}, worst {audit.worst:.2e} of max-abs")
    for k in ("loss_eg", "loss_d_real", "loss_d_fake", "dg", "de"):
        assert abs(rp[k].item() - ro[k]) <= 2e-4 * max(1.0, abs(ro[k])), (k, rp[k].item(), ro[k])
    g_o = torch.cat([p.grad.reshape(-1).double() for p in Do.parameters()])
    g_p = stepper.opt_d.grad_logical().double().cpu()
    rel = ((g_p - g_o).norm() / g_o.norm()).item()
    assert rel <= 3e-2, f"{family} D gradients of the last phase: rel L2 {rel:.3e}"
    lr = 1e-4
    for nm, mo, mp in (("E", Eo, E), ("G", Go, G), ("D", Do, D)):
        so = mo.state_dict()
        wo = torch.cat([(so[k] - before[nm][k]).reshape(-1).double() for k in so])
        wp = torch.cat([(v.cpu() - before[nm][k]).reshape(-1).double() for k, v in mp.state_dict().items()])
        err = (wp - wo).abs()
        assert (err > 0.05 * lr).double().mean().item() < 2e-2, (nm, (err > 0.05 * lr).double().mean().item())
        assert err.mean().item() <= 0.05 * lr, (nm, err.mean().item())


@pytest.mark.parametrize("capture", [False, True])
def test_train_on_stream_vs_oracle(capture):
    """``_spect.train_on_stream`` (audio_mnist.py:372-420: batches from a stream, z ~ N(0,1) drawn on the host, ALI
    iteration, epoch scores) for 2 batches on the stepper vs the oracle's ``ali_step`` fed the same draws."""
    from image_scms import _spect
    (Eo, Go, Do), (E, G, D), _, _, _ = paired_models("audio", d=8, B=2)
    for m in (Eo, Go, Do, E, G, D):
        m.train()
    B = 4
    data = [orc.synth_spect_batch("audio", B, seed=40 + i) for i in range(2)]
    keys = list(data[0][1].keys())

    def stream():
        for images, c, _ in data:
            yield dict(audio=images.reshape(B, 128, 128), **c)

    oe, od = orc.build_optimizers(Eo, Go, Do, "audio")
    before = {nm: copy.deepcopy(m.state_dict()) for nm, m in (("E", Eo), ("G", Go), ("D", Do))}
    torch.manual_seed(77)
    dg = de = 0.0
    for images, c, _ in data:
        zm = torch.zeros(B, 512, 1, 1)
        z = torch.normal(zm, zm + 1)                       # the draw train_on_stream makes (audio_mnist.py:386-387)
        r = orc.ali_step(Eo, Go, Do, oe, od, images, c, z)
        dg, de = dg + r["dg"], de + r["de"]
    torch.manual_seed(77)
    *_, scores = _spect.train_on_stream(E, G, D, stream, n_epochs=1, device="cuda", attr_keys=keys, capture=capture)
    assert abs(scores[0][0] - dg / 2) <= 1e-3 and abs(scores[0][1] - de / 2) <= 1e-3, (scores, dg / 2, de / 2)
    lr = 1e-4
    for nm, mo, mp in (("E", Eo, E), ("G", Go, G), ("D", Do, D)):
        so = mo.state_dict()
        wo = torch.cat([(so[k] - before[nm][k]).reshape(-1).double() for k in so])
        wp = torch.cat([(v.cpu() - before[nm][k]).reshape(-1).double() for k, v in mp.state_dict().items()])
        err = (wp - wo).abs()
        # two sign-like Adam steps (E+G) / four (D): elements whose gradient is at rounding level may differ by ~lr each
        assert (err > 0.1 * lr).double().mean().item() < 2e-2, (nm, (err > 0.1 * lr).double().mean().item())
        assert err.mean().item() <= 0.05 * lr, (nm, err.mean().item())


class fp16_operand_emulation:
    """The oracle with the GEMM operands of the layers the HIP path runs on fp16 MFMA (input channel stride % 32 == 0)
    rounded to fp16 -- products and sums stay fp32, exactly what v_mfma_f32_32x32x16_f16 computes.  Gives the gradient
    of the fp16-rounded forward pass: the reference for what the backward kernels must produce."""

    @staticmethod
    def _r(t):
        return t.half().float()

    def __enter__(self):
        import torch.nn as nn
        import torch.nn.functional as F
        r = self._r
        self.saved = (nn.Conv2d._conv_forward, nn.ConvTranspose2d.forward, nn.Linear.forward)
        conv_orig = self.saved[0]

        def conv_fwd(mod, inp, weight, bias):
            first = (mod.in_channels <= 8 and mod.out_channels == 64 and mod.kernel_size == (5, 5)
                     and mod.stride == (2, 2) and (inp.shape[-1] + 2 * mod.padding[1] - 5) // 2 + 1 >= 32)
            if mod.in_channels % 32 == 0 or first:     # (include/ali_hip.h: ali_conv_uses_f16)
                return conv_orig(mod, r(inp), r(weight), bias)
            return conv_orig(mod, inp, weight, bias)

        def convt_fwd(mod, inp, output_size=None):
            f16 = mod.in_channels % 32 == 0 and mod.out_channels > 2      # (<= 2 channels: scatter form on N = taps)
            if mod.in_channels % 32 == 0 and mod.out_channels <= 2:
                f16 = True
            x, w = (r(inp), r(mod.weight)) if f16 else (inp, mod.weight)
            return F.conv_transpose2d(x, w, mod.bias, mod.stride, mod.padding, mod.output_padding, mod.groups,
                                      mod.dilation)

        def lin_fwd(mod, inp):                     # the Generator's Linear: rows padded to a multiple of 32 columns
            return F.linear(r(inp), r(mod.weight), mod.bias)

        nn.Conv2d._conv_forward, nn.ConvTranspose2d.forward, nn.Linear.forward = conv_fwd, convt_fwd, lin_fwd
        return self

    def __exit__(self, *exc):
        import torch.nn as nn
        nn.Conv2d._conv_forward, nn.ConvTranspose2d.forward, nn.Linear.forward = self.saved


@pytest.fixture
def bench_tiles(request):
    """ALI_TILE_M_SCALE for the cases that name one (see BENCH_TILE_CASES)"""
    from ali_hip import ops
    scale = request.getfixturevalue("scale")
    if scale <= 1:
        yield
        return
    with ops.tuning(ALI_TILE_M_SCALE=scale):
        yield
    torch.cuda.synchronize()


@pytest.mark.parametrize("family,d,B,scale", [("esrf", 8, 2, 1), ("audio", 8, 4, 1), ("audio", 64, 16, 16)])
def test_fp16_mfma_stepper_iteration_vs_fp32_oracle(family, d, B, scale, bench_tiles):
    """BASELINE config 5 (esrf_acoustic.py:134-260,333-379 on the fp16-MFMA path, ``AliStepper(precision="f16")``:
    forward and data-gradient GEMMs contract fp16-rounded operands with fp32 accumulation, loss-scaled gradients, fp32
    master weights / weight-gradient accumulation / Adam) against the fp32 CPU oracle: the three losses and the two
    scores within the north_star's 1e-3, reconstructions G(E(x)) within 1e-3 relative, and the Adam step of every
    parameter (sign-like: +-lr) equal except where the gradient is at fp16-rounding level."""
    from ali_hip import ops
    from ali_hip.step import AliStepper
    (Eo, Go, Do), (E, G, D), images, c, z = paired_models(family, d=d, B=B)
    for m in (Eo, Go, Do, E, G, D):
        m.train()
    oe, od = orc.build_optimizers(Eo, Go, Do, family)
    with torch.no_grad():
        rec_o = Go(Eo(images, c), c)
    stepper = AliStepper(E, G, D, betas=(0.5, 0.9), precision="f16")
    assert stepper.loss_scale == 1024.0
    with torch.no_grad(), ops.precision("f16"):
        rec_p = G(E(images.cuda(), to_dev(c)), to_dev(c)).cpu()
    rel = ((rec_p - rec_o).norm() / rec_o.norm()).item()
    assert 0 < rel <= 1e-3, f"G(E(x)) fp16 vs fp32 oracle: rel L2 {rel:.3e}"
    # Phase by phase on the ORACLE's weights (an Adam step is sign-like and amplifies any gradient noise, fp16's
    # included, into the next phase's loss; the phases themselves are what is being checked): mnist.py:224-248
    import torch.nn as nn
    bce = nn.BCEWithLogitsLoss()
    valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
    lr = 1e-4

    def weights(mods):
        return torch.cat([p.detach().reshape(-1).double() for m in mods for p in m.parameters()])

    def check_update(mods_o, mods_p, before, what, group):
        """the phase's gradients (loss scale divided out) against the fp32 oracle's.  fp16 operand rounding (2^-11)
        moves every pre-activation by ~1e-3 of its scale, so the ~1e-3 of the LeakyReLU units that sit that close to 0
        take the other slope (1 <-> 0.2): measured rel-L2 2-3e-2, i.e. sqrt(flipped fraction) -- the exact gradient
        of the fp16-rounded forward pass, not an error of the backward kernels (those are pinned bit for bit on
        fp16-representable data by test_fp16_mfma_gemm_path).  What this guards is the integration: loss scale in,
        loss scale out, every GEMM on the right operands.  The Adam step: bounded by lr per element, close on average."""
        g_o = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).double()
                         for m in mods_o for p in m.parameters()])
        g_p = group.grad_logical().double().cpu() / stepper.loss_scale
        rel = ((g_p - g_o).norm() / g_o.norm()).item()
        assert rel <= 6e-2, (what, rel)
        wp = torch.cat([p.detach().reshape(-1).double().cpu() for m in mods_p for p in m.parameters()])
        err = ((weights(mods_o) - before) - (wp - before)).abs()
        assert err.max().item() <= 2.2 * lr, (what, err.max().item())
        assert err.mean().item() <= 0.1 * lr, (what, err.mean().item())

    sd_eg = [copy.deepcopy(m.state_dict()) for m in (Eo, Go, Do)]
    with torch.no_grad(), ops.precision("f16"):
        stepper.load_state(Eo, Go, Do, oe, od)
        cx = stepper._begin(images.cuda(), to_dev(c), z.cuda())
        w0 = weights((Eo, Go))
        stepper._phase_eg(cx)
    g_eg_hip = stepper.opt_eg.grad_logical().double().cpu() / stepper.loss_scale
    loss_eg_hip = cx["out"]["loss_eg"].item()
    oe.zero_grad()
    l_eg = (bce(Do(images, Eo(images, c), c), fake) + bce(Do(Go(z, c), z, c), valid)) / 2
    l_eg.backward()
    oe.step()
    assert abs(cx["out"]["loss_eg"].item() - l_eg.item()) <= 1e-3 * max(1.0, abs(l_eg.item()))
    check_update((Eo, Go), (E, G), w0, "EG update", stepper.opt_eg)
    # the same E+G gradients against the fp16-operand emulation of the oracle (same weights as the phase above saw)
    Ee, Ge, De = copy.deepcopy(Eo), copy.deepcopy(Go), copy.deepcopy(Do)
    for m, sd in zip((Ee, Ge, De), sd_eg):
        m.load_state_dict(sd)
        m.zero_grad()
    with fp16_operand_emulation():
        l_emu = (bce(De(images, Ee(images, c), c), fake) + bce(De(Ge(z, c), z, c), valid)) / 2
        l_emu.backward()
    g_e = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).double()
                     for m in (Ee, Ge) for p in m.parameters()])
    rel_emu = ((g_eg_hip - g_e).norm() / g_e.norm()).item()
    assert abs(loss_eg_hip - l_emu.item()) <= 2e-4 * max(1.0, abs(l_emu.item())), (loss_eg_hip, l_emu.item())
    # (two fp16 evaluations whose fp32 partial sums are grouped differently -- split-K slabs, 64- vs 32-deep tiles, the
    # CPU's blocking -- differ by 1e-7 before rounding, land on different sides of an fp16 rounding boundary for ~1e-4
    # of the activations and decorrelate from there at the fp16-ulp level, LeakyReLU slopes included: measured 1-2.5e-2,
    # against 2-3e-2 vs the fp32 oracle.  The loss, which averages all of that out, agrees to 2e-4.)
    assert rel_emu <= 4e-2, f"E+G gradients vs fp16-operand emulation: rel L2 {rel_emu:.3e}"
    with torch.no_grad(), ops.precision("f16"):
        stepper.load_state(Eo, Go, Do, oe, od)
        w0 = weights((Do,))
        stepper._phase_d_real(cx)
    od.zero_grad()
    l_dr = bce(Do(images, Eo(images, c), c), valid)
    l_dr.backward()
    od.step()
    assert abs(cx["out"]["loss_d_real"].item() - l_dr.item()) <= 1e-3 * max(1.0, abs(l_dr.item()))
    check_update((Do,), (D,), w0, "D real update", stepper.opt_d)
    with torch.no_grad(), ops.precision("f16"):
        stepper.load_state(Eo, Go, Do, oe, od)
        w0 = weights((Do,))
        stepper._phase_d_fake(cx)
    od.zero_grad()
    l_df = bce(Do(Go(z, c), z, c), fake)
    l_df.backward()
    od.step()
    assert abs(cx["out"]["loss_d_fake"].item() - l_df.item()) <= 1e-3 * max(1.0, abs(l_df.item()))
    check_update((Do,), (D,), w0, "D fake update", stepper.opt_d)
    with torch.no_grad(), ops.precision("f16"):
        stepper.load_state(Eo, Go, Do, oe, od)
        stepper._phase_scores(cx)
    with torch.no_grad():
        dg = Do(Go(z, c), z, c).sigmoid().mean().item()
        de = Do(images, Eo(images, c), c).sigmoid().mean().item()
    assert abs(cx["out"]["dg"].item() - dg) <= 1e-3 and abs(cx["out"]["de"].item() - de) <= 1e-3
    # free-running: a whole iteration through the public entry point stays finite and close (5e-3: Adam amplification),
    # and every fp16-MFMA launch in it equals torch's CPU convolution of the fp16-rounded operands it read (launch audit:
    # 2e-4 of max-abs per launch -- a wrong tap in any layer, which the statistical bounds above could hide, cannot pass)
    from launch_audit import LaunchAudit
    audit = LaunchAudit()
    with ops.launch_hook(audit):
        rp = stepper.step(images.cuda(), to_dev(c), z.cuda())
    assert audit.f16_checked >= 40 and audit.checked["wgrad"] >= 20, (audit.checked, audit.f16_checked)
    print(f"launch audit (fp16 path): {audit.checked}, {audit.f16_checked} on fp16 MFMA, worst {audit.worst:.2e}")
    ro = orc.ali_step(Eo, Go, Do, oe, od, images, c, z)
    for k in ("loss_eg", "loss_d_real", "loss_d_fake", "dg", "de"):
        assert abs(rp[k].item() - ro[k]) <= 5e-3 * max(1.0, abs(ro[k])), (k, rp[k].item(), ro[k])


def test_fp16_weight_twins_follow_the_optimiser():
    """The fp16 twins of the packed weights (AliEpilogue.w16) must be re-rounded after every Adam step, also when the
    re-pack launches are batched (``ops.batched_packs``: the jobs run at the end of the block, the twins after them)
    and when the iteration is replayed from a HIP graph: stale twins train on last iteration's weights."""
    from ali_hip import ops
    from ali_hip.step import AliStepper
    for capture in (False, True):
        (_, _, _), (E, G, D), images, c, z = paired_models("audio", d=8, B=4)
        for m in (E, G, D):
            m.train()
        st = AliStepper(E, G, D, betas=(0.5, 0.9), precision="f16", capture=capture)
        for _ in range(3):
            st.step(images.cuda(), to_dev(c), z.cuda())
        n = 0
        for plan in (st.pE, st.pG, st.pDx, st.pDz, st.pDxz):
            for key, (tag, val, builder, param) in plan.cache.store.items():
                h = ops.shadow16(val)
                if h is not None:
                    n += 1
                    assert torch.equal(h, val.half()), (capture, key)
                    # ... and the pack itself is the current parameter (spot check through the fp32 pack's norm)
                    assert abs(float(val.double().norm()) - float(param.detach().double().norm())) <= 1e-4 * float(
                        param.detach().double().norm()) or "scatter" in str(key), (capture, key)
        assert n >= 6
        # the forward packs are aliases of the flat master weights (FlatGroup.layouts); their fp16 twin is the flat fp16
        # buffer the Adam launch writes alongside -- it must equal half(flat) after every step, eager and replayed
        for grp in (st.opt_eg, st.opt_d):
            assert grp.flat16 is not None and torch.equal(grp.flat16, grp.flat.half()), capture
        with ops.precision("f16"):
            k = 0
            for plan in (st.pE, st.pG, st.pDx, st.pDz, st.pDxz):
                for stg in plan.stages[1:]:
                    if stg.kind in ("conv", "convT") and stg.mod.weight.dim() == 4:
                        cin = stg.mod.weight.shape[1] if stg.kind == "conv" else stg.mod.weight.shape[0]
                        wp = plan.packed(stg, "fwd", cin)
                        h = ops.shadow16(wp)
                        if h is not None and wp.data_ptr() == stg.mod.weight.data_ptr():
                            k += 1
                            assert torch.equal(h, wp.half()), (capture, stg.index)
            assert k >= 8
