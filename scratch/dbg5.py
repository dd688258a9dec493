import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import ali_oracle as orc
from test_gpu_modules import _stepper_setup, to_dev
for rescale in (True, False):
    (Eo, Go, Do), (E, G, D), stepper, batches = _stepper_setup(rescale)
    oe, od = orc.build_optimizers(Eo, Go, Do, "mnist")
    images, c, z = batches[0]
    tape = orc.MaskTape()
    ro = orc.ali_step(Eo, Go, Do, oe, od, images, c, z, tape=tape)
    rp = stepper.step(images.cuda(), to_dev(c), z.cuda(), masks=tape.masks)
    print("rescale", rescale)
    for (k, po), (_, pp) in zip(Do.named_parameters(), D.named_parameters()):
        go, gp = po.grad.double(), pp.grad.cpu().double()
        print(f"D.{k:28s} |g|max {go.abs().max():.3e} mean {go.abs().mean():.3e}  err max {(go-gp).abs().max():.3e} rel {((go-gp).norm()/go.norm()):.3e}")
