import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch, torch.nn as nn
import ali_oracle as orc
from test_gpu_modules import _stepper_setup, to_dev
from ali_hip.chain import chain_forward
(Eo, Go, Do), (E, G, D), stepper, batches = _stepper_setup(True)
images, c, z = batches[0]
acts = []
hooks = [m.register_forward_hook(lambda m, i, o: acts.append(o.detach())) for m in Go.layers if isinstance(m, (nn.LeakyReLU, nn.Tanh))]
Go(z, c)
idx, cont, onehots = stepper.family.conditioning(to_dev(c))
gin, g_log = stepper._g_input(z.cuda().reshape(64, -1), onehots, cont)
y, saved = chain_forward(stepper.pG, gin, True, g_log, True)
for i, sv in enumerate(saved):
    a = acts[i]; b = sv.y.permute(0, 3, 1, 2).cpu()
    flips = ((a > 0) != (b > 0))
    print(i, "shape", tuple(a.shape), "rel err", ((a.double()-b.double()).norm()/a.double().norm()).item(), "sign flips", int(flips.sum()),
          "values at flips", a[flips][:5].tolist(), b[flips][:5].tolist())
