"""r03: MNIST Discriminator first conv (5 live planes of 8 -> 32, 5x5, 28 -> 24): per-image kernel, stand-alone timing"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imagecfgen-pytorch_amd"))
import torch
import ali_hip
from ali_hip import ops
ali_hip.load()


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
    return e0.elapsed_time(e1) / n


for B in (256, 512, 1024, 2048):
    g = ops.geom(B, 28, 28, 8, 24, 24, 32, 5, 5, 1, 0)
    x = torch.randn(B, 28, 28, 8, device="cuda")
    x[..., 5:] = 0
    w = torch.randn(32, 25, 8, device="cuda") * 0.1
    y = torch.empty(B, 24, 24, 32, device="cuda")
    bias = torch.randn(32, device="cuda")
    for live in (5, 8):
        ep = ops.epilogue(bias=bias, act=ops.ACT_LEAKY, slope=0.2)
        ep.in_ch_live = live
        ms = timeit(lambda: ops.conv_fwd(g, x, w, y, ep))
        print(f"B {B:5d} live {live}: {ms * 1e3:7.1f} us   {2.0 * B * 576 * 32 * 25 * live / ms / 1e9:6.1f} TF/s (live)")
