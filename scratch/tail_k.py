"""Fixed vs per-k cost of the 1x1 tail GEMMs: python scratch/tail_k.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), ROOT]
import torch
import ali_hip
from ali_hip import ops
ops.set_workspace_bytes(1 << 30)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (512, 1024):
    for N in (512, 1024):
        row = []
        for C in (64, 128, 256, 512, 1024, 2048):
            x = torch.randn(M, 1, 1, C, device="cuda"); w = torch.randn(N, 1, C, device="cuda") * 0.03
            y = torch.empty(M, 1, 1, N, device="cuda"); b = torch.randn(N, device="cuda")
            geom = ops.geom(M, 1, 1, C, 1, 1, N, 1, 1, 1, 0)
            us = t(lambda: ops.conv_fwd(geom, x, w, y, ops.epilogue(bias=b, act=ops.ACT_LEAKY, slope=0.1)))
            row.append(f"K={C}: {us:5.1f}")
        print(M, N, " | ".join(row))
