"""debug: which elements differ in the first Adam update at bs=512 (stepper vs oracle)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("imagecfgen-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, torch.nn as nn
torch.set_num_threads(8)
import ali_oracle as orc
from test_gpu_modules import _stepper_setup, _flat_grads, to_dev
from ali_hip import dropout as _dropout
for bs in (64, 512):
    (Eo, Go, Do), (E, G, D), stepper, batches = _stepper_setup(True, bs=bs, n=1)
    oe, od = orc.build_optimizers(Eo, Go, Do, "mnist")
    bce = nn.BCEWithLogitsLoss()
    images, c, z = batches[0]
    B = images.shape[0]
    valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
    tape = orc.MaskTape()
    stepper.load_state(Eo, Go, Do, oe, od)
    names = [(f"{nm}.{k}", p.numel()) for nm, m in (("E", Eo), ("G", Go)) for k, p in m.named_parameters()]
    w0 = torch.cat([p.detach().reshape(-1).double() for m in (Eo, Go) for p in m.parameters()])
    with orc.use_tape(tape):
        oe.zero_grad()
        l_eg = (bce(Do(images, Eo(images, c), c), fake) + bce(Do(Go(z, c), z, c), valid)) / 2
        l_eg.backward()
        g_eg = _flat_grads((Eo, Go))
        oe.step()
    with _dropout.injected_masks(tape.masks), torch.no_grad():
        cx = stepper._begin(images.cuda(), to_dev(c), z.cuda())
        stepper._eg_grads(cx)
        gp = stepper.opt_eg.grad.double().cpu().clone()
        stepper._apply_eg()
    wo = torch.cat([p.detach().reshape(-1).double() for m in (Eo, Go) for p in m.parameters()])
    wp = torch.cat([p.detach().reshape(-1).double().cpu() for m in (E, G) for p in m.parameters()])
    err = ((wp - w0) - (wo - w0)).abs()
    bad = err > 0.05 * 1e-4
    rms = g_eg.pow(2).mean().sqrt().item()
    print(f"bs={bs} bad frac {bad.double().mean().item():.3e} rms(g) {rms:.3e} relL2 {((gp-g_eg).norm()/g_eg.norm()).item():.2e}")
    for thr in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
        sel = g_eg.abs() > thr * rms
        print(f"   |g|>{thr:g}*rms: {sel.double().mean().item():.4f} of elements, bad among them {(bad & sel).double().sum().item():.0f}")
    d = (gp - g_eg).abs()
    print("   bad elems: median |g_ref|", g_eg[bad].abs().median().item(), "median |dg|", d[bad].median().item(),
          "max |g_ref| among bad", g_eg[bad].abs().max().item())
    off = 0
    for nm, n in names:
        nb = int(bad[off:off + n].sum())
        if nb:
            seg = g_eg[off:off + n]
            print(f"   {nm:28s} n={n:8d} bad={nb:6d} rms(g)={seg.pow(2).mean().sqrt().item():.2e} max|dg|={d[off:off+n].max().item():.2e}")
        off += n
