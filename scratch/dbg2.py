import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), ROOT]
import torch
import ali_hip, image_scms.mnist as pm
from ali_hip.step import AliStepper
from bench import synth_batch
dev = torch.device("cuda")
torch.manual_seed(1)
E, G, D = pm.Encoder(), pm.Generator(), pm.Discriminator()
for m in (E, G, D):
    m.apply(pm.init_weights); m.to(dev).train()
st = AliStepper(E, G, D, capture=True)
b = [synth_batch(512, dev, i) for i in range(2)]
t0 = time.time()
r = st.step(*b[0]); torch.cuda.synchronize(); print("first (capture) step", time.time()-t0, flush=True)
t0 = time.time()
for i in range(10): r = st.step(*b[i % 2])
torch.cuda.synchronize(); dt = (time.time()-t0)/10
print("graph ms/step", dt*1e3, "img/s", 512/dt, {k: v.item() for k, v in r.items()}, flush=True)
