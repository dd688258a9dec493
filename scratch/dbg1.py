import sys, os, copy, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch, torch.nn as nn
import ali_oracle as orc
from test_gpu_modules import _stepper_setup, to_dev
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read())
except Exception as e: print("no cpu.max", e)
(Eo, Go, Do), (E, G, D), stepper, batches = _stepper_setup(True)
images, c, z = batches[0]
tape = orc.MaskTape()
bce = nn.BCEWithLogitsLoss()
with orc.use_tape(tape):
    d_valid = Do(images, Eo(images, c), c)
    d_fake = Do(Go(z, c), z, c)
    loss = (bce(d_valid, torch.zeros(64,1)) + bce(d_fake, torch.ones(64,1)))/2
    loss.backward()
# product: EG part only via stepper internals
from ali_hip import dropout as _d
masks = tape.masks + [m for m in tape.masks]*2   # enough masks for the rest of the iteration (values irrelevant)
rp = stepper.step(images.cuda(), to_dev(c), z.cuda(), masks=masks)
print("loss_eg", rp["loss_eg"].item(), loss.item())
for nm, mo, mp in (("E", Eo, E), ("G", Go, G)):
    for (k, po), (_, pp) in zip(mo.named_parameters(), mp.named_parameters()):
        go, gp = po.grad.double(), pp.grad.cpu().double()
        print(f"{nm}.{k:28s} |g|max {go.abs().max():.3e} mean {go.abs().mean():.3e}  err max {(go-gp).abs().max():.3e} rel {((go-gp).norm()/go.norm()):.3e}")
