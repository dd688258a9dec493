"""r03: fp16-MFMA gconv loop on plain GEMM shapes vs the 5x5 stride-2 layers of the ESRF stacks (TF/s), twin operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imagecfgen-pytorch_amd"))
import torch
import ali_hip
from ali_hip import ops
ali_hip.load()
ops.set_workspace_bytes(2 << 30)


def timeit(fn, n=20):
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


cases = [("1x1 M=16384 K=4096 N=4096", 16384, 1, 4096, 4096, 1, 1, 0),
         ("1x1 M=65536 K=1024 N=1024", 65536, 1, 1024, 1024, 1, 1, 0),
         ("3x3 s1 p1 64x32x32 C=256 K=256", 64, 32, 256, 256, 3, 1, 1),
         ("esrf 64->128 255->127 5x5 s2", 64, 255, 64, 128, 5, 2, 1),
         ("esrf 128->256 127->63", 64, 127, 128, 256, 5, 2, 1),
         ("esrf 256->512 63->31", 64, 63, 256, 512, 5, 2, 1),
         ("esrf 512->1024 31->15", 64, 31, 512, 1024, 5, 2, 1),
         ("esrf 1024->2048 15->7", 64, 15, 1024, 2048, 5, 2, 1)]
for name, B, H, C, K, R, st, pad in cases:
    P = (H + 2 * pad - R) // st + 1
    g = ops.geom(B, H, H, C, P, P, K, R, R, st, pad)
    x = torch.randn(B, H, H, C, device="cuda")
    w = torch.randn(K, R * R, C, device="cuda") * 0.02
    y = torch.empty(B, P, P, K, device="cuda")
    x._ali16 = x.half()
    ops.ensure_shadow16(w)
    flops = 2.0 * B * P * P * K * C * R * R
    for prec in ("f16", "f32"):
        with ops.precision(prec):
            ms = timeit(lambda: ops.conv_fwd(g, x, w, y, ops.epilogue()))
        print(f"{name:36s} {prec} {ms * 1e3:9.1f} us {flops / ms / 1e9:8.1f} TF/s  tile rows {ops.conv_mtiles(g, 0)[1] if prec == 'f32' else '-'}")
    del x, w, y
