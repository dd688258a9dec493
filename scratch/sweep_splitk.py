"""Sweep the split-K factor of the GEMM kernels over the MorphoMNIST layer shapes (bs=512): us per launch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd")]
import torch
from ali_hip import ops
ops.set_workspace_bytes(2 << 30)

def run(kind, B, H, C, P, K, R, st, pad, iters=20):
    x = torch.randn(B, H, H, C, device="cuda"); y = torch.randn(B, P, P, K, device="cuda")
    g = ops.geom(B, H, H, C, P, P, K, R, R, st, pad)
    if kind == "fwd":
        w = torch.randn(K, R * R, C, device="cuda") * 0.05
        ep = ops.epilogue(bias=torch.randn(K, device="cuda"), act=ops.ACT_LEAKY, slope=0.2)
        f = lambda: ops.conv_fwd(g, x, w, y, ep)
    else:
        w = torch.randn(C, R * R, K, device="cuda") * 0.05
        ep = ops.epilogue()
        f = lambda: ops.conv_bwd_data(g, y, w, x, ep)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

shapes = [("fwd", 512, 8, 128, 3, 256, 4, 2, 0), ("fwd", 512, 24, 32, 11, 64, 4, 2, 0), ("fwd", 512, 1, 1024, 1, 1024, 1, 1, 0),
          ("fwd", 512, 1, 512, 1, 512, 1, 1, 0), ("fwd", 512, 3, 256, 1, 512, 3, 1, 0), ("fwd", 512, 7, 256, 3, 512, 3, 2, 0),
          ("fwd", 512, 14, 64, 7, 128, 4, 2, 1), ("fwd", 512, 7, 128, 3, 256, 4, 2, 1), ("fwd", 512, 3, 512, 1, 800, 3, 1, 0),
          ("bwd", 512, 11, 64, 8, 128, 4, 1, 0), ("bwd", 512, 7, 256, 3, 512, 3, 2, 0), ("bwd", 512, 13, 128, 7, 256, 3, 2, 1),
          ("bwd", 512, 8, 128, 3, 256, 4, 2, 0), ("bwd", 512, 24, 32, 11, 64, 4, 2, 0), ("bwd", 512, 1, 1024, 1, 1024, 1, 1, 0),
          ("bwd", 512, 3, 512, 1, 800, 3, 1, 0), ("bwd", 512, 3, 256, 1, 512, 3, 1, 0), ("bwd", 512, 1, 512, 1, 512, 1, 1, 0)]
Ss = [0, 1, 2, 3, 4, 6, 8, 12, 16]
print("shape".ljust(44), " ".join(f"S={s:<5d}" for s in Ss))
for sh in shapes:
    res = []
    for s in Ss:
        os.environ["ALI_SPLITK"] = str(s)
        res.append(run(*sh))
    fl = 2.0 * sh[1] * sh[4] ** 2 * sh[5] * sh[3] * sh[6] ** 2
    best = min(res)
    print(str(sh).ljust(44), " ".join(f"{r:7.1f}" for r in res), f" | auto {fl/res[0]/1e6:5.1f} TF best {fl/best/1e6:5.1f} TF", flush=True)
