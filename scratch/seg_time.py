"""Single-graph vs segmented (data-parallel style) replay on one GPU: python scratch/seg_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), ROOT]
import torch
import bench
import image_scms.mnist as pm
from ali_hip import ops
from ali_hip.step import AliStepper
dev = torch.device("cuda")
ops.set_workspace_bytes(2 << 30)
for seg in (False, True):
    torch.manual_seed(1)
    E, G, D = pm.Encoder(), pm.Generator(), pm.Discriminator()
    for m in (E, G, D):
        m.apply(pm.init_weights); m.to(dev).train()
    st = AliStepper(E, G, D, capture=True)
    st.segmented = seg
    b = bench.synth_batch(512, dev, 0)
    for _ in range(5): st.step(*b)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): st.step(*b)
    t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print("segmented" if seg else "one graph", f"host issue {1e3*(t1-t0)/50:.3f} ms/iter, wall {1e3*(t2-t0)/50:.3f} ms/iter")
