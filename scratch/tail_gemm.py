"""Dense 1x1 'tail' GEMM timings vs ALI_SPLITK: python scratch/tail_gemm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd"), ROOT]
import torch
import ali_hip
from ali_hip import ops
lib = ali_hip.load()
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
for (M, C, K) in ((512, 1024, 1024), (1024, 1024, 1024), (512, 512, 512), (1024, 512, 512)):
    x = torch.randn(M, 1, 1, C, device="cuda"); w = torch.randn(K, 1, C, device="cuda") * 0.03
    y = torch.empty(M, 1, 1, K, device="cuda"); b = torch.randn(K, device="cuda")
    geom = ops.geom(M, 1, 1, C, 1, 1, K, 1, 1, 1, 0)
    row = []
    for S in (0, 1, 2, 4, 8):
        os.environ["ALI_SPLITK"] = str(S); lib.ali_reload_tuning()
        us = t(lambda: ops.conv_fwd(geom, x, w, y, ops.epilogue(bias=b, act=ops.ACT_LEAKY, slope=0.1)))
        row.append(f"S={S}: {us:5.1f}us {2*M*C*K/us/1e6:5.1f}TF")
    print(M, C, K, " | ".join(row))
# floor: an empty-ish kernel
z = torch.zeros(256, device="cuda")
print("tiny kernel (add_) per launch us:", t(lambda: z.add_(1.0)))
