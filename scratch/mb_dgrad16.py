"""r03: fp16 data-gradient launches of the ESRF stacks with few output channels (N = 64 / 128) under forced tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imagecfgen-pytorch_amd"))
import torch
import ali_hip
from ali_hip import ops
ali_hip.load()
ops.set_workspace_bytes(2 << 30)


def timeit(fn, n=10):
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


# (B, H, C, K, R, stride, pad, P): dx [B,H,H,C] <- dy [B,P,P,K]
cases = [("dgrad 255x64 <- 127x128 p1", 64, 255, 64, 128, 5, 2, 1, 127),
         ("dgrad 127x128 <- 63x256 p1", 64, 127, 128, 256, 5, 2, 1, 63),
         ("dgrad 256x64 <- 128x64 p2 (convT fwd)", 64, 256, 64, 64, 5, 2, 2, 128),
         ("dgrad 128x64 <- 64x128 p2", 64, 128, 64, 128, 5, 2, 2, 64)]
tiles = [(64, 64), (64, 64), (192, 64), (256, 64), (128, 128)]
for name, B, H, C, K, R, st, pad, P in cases:
    g = ops.geom(B, H, H, C, P, P, K, R, R, st, pad)
    dy = torch.randn(B, P, P, K, device="cuda")
    w = torch.randn(C, R * R, K, device="cuda") * 0.02
    dx = torch.empty(B, H, H, C, device="cuda")
    dy._ali16 = dy.half()
    ops.ensure_shadow16(w)
    flops = 2.0 * B * P * P * K * C * R * R
    for bm, bn in tiles:
        if bn > max(C, 64):
            continue
        with ops.precision("f16"), ops.tuning(ALI_BM=bm, ALI_BN=bn):
            ms = timeit(lambda: ops.conv_bwd_data(g, dy, w, dx, ops.epilogue()))
        print(f"{name:40s} tile {bm:3d}x{bn:3d} {ms * 1e3:9.1f} us {flops / ms / 1e9:8.1f} TF/s", flush=True)
    del dy, w, dx
