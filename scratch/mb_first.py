"""r03: the first layer of the ESRF stacks (4 -> 64 channels, 5x5 s2, 512 -> 255, B=64) under different tiles"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imagecfgen-pytorch_amd"))
import torch
import ali_hip
from ali_hip import ops
ali_hip.load()
ops.set_workspace_bytes(2 << 30)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, B, H, C, K in (("esrf", 64, 512, 4, 64), ("whale", 128, 256, 4, 64), ("audio", 256, 128, 8, 64)):
    P = (H + 2 - 5) // 2 + 1
    g = ops.geom(B, H, H, C, P, P, K, 5, 5, 2, 1)
    x = torch.randn(B, H, H, C, device="cuda")
    w = torch.randn(K, 25, C, device="cuda") * 0.1
    y = torch.empty(B, P, P, K, device="cuda")
    bias = torch.randn(K, device="cuda")
    for bm, bn in ((0, 0), (128, 64), (64, 64), (128, 32)):
        with ops.tuning(ALI_BM=bm, ALI_BN=bn):
            for prec in ("f32", "f16"):
                with ops.precision(prec):
                    ms = timeit(lambda: ops.conv_fwd(g, x, w, y, ops.epilogue(bias=bias, act=ops.ACT_LEAKY, slope=0.2)))
                gb = (x.numel() * 4 + y.numel() * (6 if prec == "f16" else 4)) / 1e9
                print(f"{name} tile {bm}x{bn} {prec}: {ms * 1e3:8.1f} us  {gb / ms:7.1f} TB/s... {2.0 * B * P * P * K * C * 25 / ms / 1e9:6.1f} TF/s")
