import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import ali_oracle as orc
from test_gpu_modules import _stepper_setup, to_dev
(Eo, Go, Do), (E, G, D), stepper, batches = _stepper_setup(True)
oe, od = orc.build_optimizers(Eo, Go, Do, "mnist")
for i, (images, c, z) in enumerate(batches):
    tape = orc.MaskTape()
    w0 = Go.digit_embedding.weight.detach().clone()
    ro = orc.ali_step(Eo, Go, Do, oe, od, images, c, z, tape=tape)
    go = Go.digit_embedding.weight.grad.clone()   # note: polluted by D-step accumulation in the oracle
    rp = stepper.step(images.cuda(), to_dev(c), z.cuda(), masks=tape.masks)
    wo, wp = Go.digit_embedding.weight.detach().double(), G.digit_embedding.weight.detach().cpu().double()
    d = (wo - wp).abs()
    print(i, "upd mean", (wo - w0.double()).abs().mean().item(), "diff mean", d.mean().item(), "max", d.max().item(),
          "frac>1e-6", (d > 1e-6).double().mean().item())
    st = oe.state[Go.digit_embedding.weight]
    mo, vo = st["exp_avg"].double(), st["exp_avg_sq"].double()
    # locate G.digit_embedding in flat group
    off = 0
    for p in stepper.opt_eg.params:
        if p is G.digit_embedding.weight: break
        off += p.numel()
    mp = stepper.opt_eg.m[off:off+2560].cpu().double().view(10,256); vp = stepper.opt_eg.v[off:off+2560].cpu().double().view(10,256)
    print("   m rel", ((mo-mp).norm()/mo.norm()).item(), "v rel", ((vo-vp).norm()/vo.norm()).item(), "|m| mean", mo.abs().mean().item(), "sqrt v mean", vo.sqrt().mean().item())
