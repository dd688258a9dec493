"""debug: where does the bs=64 EG-gradient discrepancy of E come from (tie or bug)?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("imagecfgen-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, torch.nn as nn
torch.set_num_threads(8)
import ali_oracle as orc
from test_gpu_modules import _stepper_setup, _flat_grads, to_dev, TieWatch
from ali_hip import dropout as _dropout
from ali_hip import step as _step, chain as _chain
bs = 64
(Eo, Go, Do), (E, G, D), stepper, batches = _stepper_setup(True, bs=bs, n=1)
oe, od = orc.build_optimizers(Eo, Go, Do, "mnist")
bce = nn.BCEWithLogitsLoss()
images, c, z = batches[0]
B = bs
valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
tape = orc.MaskTape()
stepper.load_state(Eo, Go, Do, oe, od)
acts = {}
def hook(name):
    def f(mod, inp, out):
        x = inp[0].detach()
        acts.setdefault(name, []).append(x)
    return f
hs = []
for nm, m in (("E", Eo), ("D", Do), ("G", Go)):
    for k, sub in m.named_modules():
        if isinstance(sub, nn.LeakyReLU):
            hs.append(sub.register_forward_hook(hook(f"{nm}.{k}")))
with orc.use_tape(tape):
    oe.zero_grad()
    ex_o = Eo(images, c); ex_o.retain_grad()
    gz_o = Go(z, c); gz_o.retain_grad()
    l_eg = (bce(Do(images, ex_o, c), fake) + bce(Do(gz_o, z, c), valid)) / 2
    l_eg.backward()
for h in hs: h.remove()
for k, lst in acts.items():
    for j, x in enumerate(lst):
        a = x.abs()
        n = int((a <= 4e-7 * a.max()).sum())
        if n:
            idx = (a <= 4e-7 * a.max()).nonzero()[:4].tolist()
            print("tie", k, "call", j, "count", n, "shape", tuple(x.shape), "at", idx, "vals", [float(x[tuple(i)]) for i in idx], "max", float(a.max()))
rec = []
orig = _step.chain_backward
def wrapped(plan, saved, gy, *a, **kw):
    gx, grads = orig(plan, saved, gy, *a, **kw)
    rec.append((plan, None if gx is None else gx.detach().clone()))
    return gx, grads
_step.chain_backward = wrapped
with _dropout.injected_masks(tape.masks), torch.no_grad():
    cx = stepper._begin(images.cuda(), to_dev(c), z.cuda())
    stepper._eg_grads(cx)
_step.chain_backward = orig
names = {id(stepper.pDxz): "dxz", id(stepper.pDz): "dz", id(stepper.pE): "E", id(stepper.pDx): "dx", id(stepper.pG): "G"}
for plan, gx in rec:
    print(names[id(plan)], None if gx is None else tuple(gx.shape))
g_ex = [gx for plan, gx in rec if plan is stepper.pDz][0].reshape(B, -1).cpu().double()
ref = ex_o.grad.reshape(B, -1).double()
print("g_ex relL2", ((g_ex - ref).norm() / ref.norm()).item(), "max", (g_ex - ref).abs().max().item(), "scale", ref.abs().max().item())
bad_rows = ((g_ex - ref).abs().max(dim=1).values > 1e-4 * ref.abs().max()).nonzero().reshape(-1).tolist()
print("rows of g_ex that differ:", bad_rows)
g_gz = [gx for plan, gx in rec if plan is stepper.pDx][0][..., 0].reshape(B, -1).cpu().double()
ref = gz_o.grad.reshape(B, -1).double()
print("g_gz relL2", ((g_gz - ref).norm() / ref.norm()).item())
# per-parameter relL2 of E grads
off = 0
gp = stepper.opt_eg.grad.double().cpu()
for nm, m in (("E", Eo), ("G", Go)):
    for k, p in m.named_parameters():
        n = p.numel()
        r = p.grad.reshape(-1).double() if p.grad is not None else torch.zeros(n, dtype=torch.double)
        d = gp[off:off + n] - r
        print(f"{nm}.{k:28s} relL2 {(d.norm() / (r.norm() + 1e-300)).item():.2e}")
        off += n
