"""r03: micro-benchmark of the direct one-channel kernels at the MNIST bench shapes (us per launch, GB/s of algorithmic bytes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imagecfgen-pytorch_amd"))
import torch
import ali_hip
from ali_hip import ops
ali_hip.load()
dev = torch.device("cuda")
B = 512


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(1)
# G tail forward: ConvTranspose2d(64 -> 1, k4) + Tanh
big = torch.randn(B, 25, 25, 64, device=dev, generator=g)
w = torch.randn(1, 16, 64, device=dev, generator=g)
bias = torch.randn(1, device=dev)
out = torch.empty(B, 28, 28, 1, device=dev)
us = timeit(lambda: ops.tconv1_fwd(big, w, bias, out, B, 25, 25, 64, 4, 4, 0, 1, ops.ACT_TANH, 0.0))
print(f"tconv1_fwd<16,4>   {us:7.1f} us  {big.numel() * 4 / us / 1e3:7.1f} GB/s")
# first-layer data gradient, one plane: big = g_pre [B,24,24,32]
gp = torch.randn(B, 24, 24, 32, device=dev, generator=g)
w5 = torch.randn(1, 25, 32, device=dev, generator=g)
pl = torch.empty(B, 28, 28, 1, device=dev)
us = timeit(lambda: ops.tconv1_fwd(gp, w5, None, pl, B, 24, 24, 32, 5, 5, 0, 1, ops.ACT_NONE, 0.0))
print(f"tconv1_fwd<25,5>   {us:7.1f} us  {gp.numel() * 4 / us / 1e3:7.1f} GB/s")
# G tail data gradient
small = torch.randn(B, 28, 28, device=dev, generator=g)
gbig = torch.empty(B, 25, 25, 64, device=dev)
us = timeit(lambda: ops.tconv1_dgrad(small, 1, w, big, ops.ACT_LEAKY, 0.2, gbig, B, 25, 25, 64, 4, 4, 0))
print(f"tconv1_dgrad<16>   {us:7.1f} us  {2 * big.numel() * 4 / us / 1e3:7.1f} GB/s")
# G tail weight gradient
dw = torch.empty(64, 1, 4, 4, device=dev)
us = timeit(lambda: ops.tconv1_wgrad(big, small, 1, 1, dw, 16, 1, 0, B, 25, 25, 64, 4, 4, 0))
print(f"tconv1_wgrad<1,4>  {us:7.1f} us  {big.numel() * 4 / us / 1e3:7.1f} GB/s  (incl. t1_reduce)")
