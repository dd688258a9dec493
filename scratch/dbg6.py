import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch, torch.nn as nn
import ali_oracle as orc
from test_gpu_modules import _stepper_setup, to_dev
(Eo, Go, Do), (E, G, D), stepper, batches = _stepper_setup(True)
oe, od = orc.build_optimizers(Eo, Go, Do, "mnist")
images, c, z = batches[0]
tape = orc.MaskTape(); bce = nn.BCEWithLogitsLoss(); B=64
valid, fake = torch.ones(B,1), torch.zeros(B,1)
with orc.use_tape(tape):
    oe.zero_grad()
    l = (bce(Do(images, Eo(images, c), c), fake) + bce(Do(Go(z, c), z, c), valid))/2
    l.backward(); oe.step()
    od.zero_grad()
    l = bce(Do(images, Eo(images, c), c), valid); l.backward()
    ga = torch.cat([p.grad.reshape(-1) for p in Do.parameters()]).clone()
    wEa = torch.cat([p.detach().reshape(-1) for p in list(Eo.parameters())+list(Go.parameters())]).clone()
    od.step()
    od.zero_grad()
    l = bce(Do(Go(z, c), z, c), fake); l.backward()
    gb = torch.cat([p.grad.reshape(-1) for p in Do.parameters()]).clone()
    wDb = torch.cat([p.detach().reshape(-1) for p in Do.parameters()]).clone()
    od.step()
    Do(Go(z,c).detach(), z, c); Do(images, Eo(images,c).detach(), c)
stepper.debug = {}
# capture weights right after EG adam / D-a adam by hooking adam
import ali_hip.step as S
snaps = []
orig = S.FlatGroup.adam
def adam(self, gs=1.0):
    orig(self, gs); snaps.append(self.flat.clone())
S.FlatGroup.adam = adam
rp = stepper.step(images.cuda(), to_dev(c), z.cuda(), masks=tape.masks)
def rel(a, b): a=a.double(); b=b.cpu().double(); return ((a-b).norm()/a.norm()).item(), (a-b).abs().max().item()
print("EG weights after adam: rel, max", rel(wEa, snaps[0]))
print("D grads step a", rel(ga, stepper.debug["d_a"]))
print("D weights after adam a", rel(wDb, snaps[1]))
print("D grads step b", rel(gb, stepper.opt_d.grad))
d = (wEa.double()-snaps[0].cpu().double()).abs()
print("EG weight diffs: frac>1e-5", (d>1e-5).double().mean().item(), "frac>1e-6", (d>1e-6).double().mean().item(), "max", d.max().item())
names = []
off = 0
for nm, mod in (("E", Eo), ("G", Go)):
    for k, p in mod.named_parameters():
        names.append((nm + "." + k, off, p.numel())); off += p.numel()
big = (d > 1e-5).nonzero().reshape(-1)
import collections
cnt = collections.Counter()
for i in big.tolist():
    for nm, o, n in names:
        if o <= i < o + n:
            cnt[nm] += 1
print(cnt)
# gradient values of the oracle at those positions are gone (zeroed); recompute EG grads from m of torch optimizer: exp_avg = (1-b1) g
gE = torch.cat([oe.state[p]["exp_avg"].reshape(-1) for p in list(Eo.parameters())+list(Go.parameters())]) / 0.5
print("oracle |g| at flipped (first-step g = m/0.5... only valid if 1 step): ", gE[big][:12].tolist())
gP = stepper.opt_eg.m.cpu() / 0.5
print("product g there:", gP[big][:12].tolist())
print("typical |g|", gE.abs().mean().item())
