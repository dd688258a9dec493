"""Per-layer GEMM table of one eager stepper iteration: python scratch/layers.py [mnist|audio|whale|esrf]"""
import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), ROOT]
import torch
import ali_hip
from ali_hip import ops
from ali_hip.step import AliStepper
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "mnist"
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda")
torch.manual_seed(1)
pm = importlib.import_module("image_scms.mnist" if wl == "mnist" else bench.SPECT[wl][3])
E, G, D = pm.Encoder(), pm.Generator(), pm.Discriminator()
for m in (E, G, D):
    m.apply(pm.init_weights); m.to(dev).train()
ops.set_workspace_bytes(2 << 30)
st = AliStepper(E, G, D, capture=False, betas=(0.5, 0.999) if wl == "mnist" else (0.5, 0.9), precision=prec)
b = bench.synth_batch(512, dev, 0) if wl == "mnist" else bench.synth_spect_batch(wl, bench.SPECT[wl][4], dev, 0)
N = 2
for _ in range(2): st.step(*b)
prof = ops.KernelProfile(); ops.set_profile(prof)
for _ in range(N): st.step(*b)
ops.set_profile(None)
tab = prof.by_shape()
rows = sorted(tab.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in tab.values())
print(f"total GEMM ms/step {tot/N:.3f}")
print("kind     B   H   W    C   P   Q    K  R s p | n/step  ms/step   TF/s")
for k, v in rows:
    print(f"{k[0]:8s}{k[1]:4d}{k[2]:4d}{k[3]:4d}{k[4]:5d}{k[5]:4d}{k[6]:4d}{k[7]:5d}{k[8]:3d}{k[9]:2d}{k[10]:2d} | {v['launches']/N:5.1f} {v['ms']/N:8.3f} {v['flops']/(v['ms']*1e-3)/1e12:7.1f}")
