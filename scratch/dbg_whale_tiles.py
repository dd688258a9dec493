"""r03: is the whale d=64 B=4 D-gradient deviation (5.8e-3) a tile effect or LeakyReLU ties?  per-tensor rel errors at scale 1 / 32"""
import copy, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("imagecfgen-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import ali_oracle as orc
from test_gpu_modules import paired_models, to_dev, TieWatch
from ali_hip import ops
from ali_hip.step import AliStepper
fam, d, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for scale in (1, int(sys.argv[4])):
    (Eo, Go, Do), (E, G, D), images, c, z = paired_models(fam, d=d, B=B)
    for m in (Eo, Go, Do, E, G, D):
        m.train()
    oe, od = orc.build_optimizers(Eo, Go, Do, fam)
    with TieWatch(Eo, Go, Do) as tw:
        ro = orc.ali_step(Eo, Go, Do, oe, od, images, c, z)
    print("scale", scale, "oracle ties", tw.ties)
    with ops.tuning(ALI_TILE_M_SCALE=scale):
        st = AliStepper(E, G, D, betas=(0.5, 0.9))
        rp = st.step(images.cuda(), to_dev(c), z.cuda())
        torch.cuda.synchronize()
    print({k: (rp[k].item(), ro[k]) for k in ro if k in rp})
    names = [n for n, _ in Do.named_parameters()]
    go = [p.grad.reshape(-1).double() for p in Do.parameters()]
    gp = st.opt_d.grad_logical().double().cpu()
    off = 0
    for n, g in zip(names, go):
        k = g.numel()
        e = (gp[off:off + k] - g).norm() / (g.norm() + 1e-30)
        print(f"  {n:24s} rel {e:.3e}  |g| {g.norm():.3e}")
        off += k
    print("  total", ((gp - torch.cat(go)).norm() / torch.cat(go).norm()).item())
