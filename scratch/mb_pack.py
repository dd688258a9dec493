"""r03: data-gradient re-pack ([K][T][C] master -> [C][T][K]) of the large ESRF weights: GB/s of read + written bytes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imagecfgen-pytorch_amd"))
import torch
import ali_hip
from ali_hip import ops
ali_hip.load()


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for K, C, R in ((4096, 2048, 5), (2048, 1024, 5), (1024, 512, 5), (512, 256, 5), (128, 64, 5), (256, 128, 4)):
    T = R * R
    w = torch.randn(K, T, C, device="cuda")                 # master in the forward pack's layout: strides (T*C, C, 1)
    out = torch.zeros(C, T, K, device="cuda")
    for twin in (False, True):
        if twin:
            ops.ensure_shadow16(out)
        ms = timeit(lambda: ops.pack_weights(w.reshape(-1), out, C, T, K, K, 1, C, T * C))
        byts = w.numel() * (8 + (2 if twin else 0))
        print(f"K={K} C={C} T={T} twin={twin}: {ms * 1e3:9.1f} us  {byts / ms / 1e9:7.2f} TB/s")
    ref = w.permute(2, 1, 0).contiguous()
    assert torch.equal(out, ref)
    del w, out, ref
