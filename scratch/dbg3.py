import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd")]
import torch, torch.nn.functional as F
from ali_hip import ops
torch.manual_seed(0)
def nhwc(t): return t.permute(0,2,3,1).contiguous()
def nchw(t): return t.permute(0,3,1,2).contiguous()
for B in (6, 64):
    Ci, Co, R, H = 64, 1, 4, 25
    w = torch.randn(Ci, Co, R, R)*0.1
    gy = torch.randn(B, Co, 28, 28)
    ref = F.conv2d(gy.double(), w.double())          # convT dgrad == conv with the same weight [Ci][Co] as [K][C]
    T = R*R
    wd = torch.empty(Ci, T, Co, device="cuda")
    ops.pack_weights(w.cuda().contiguous(), wd, Ci, T, Co, Co, Co*T, 1, T)
    dx = torch.empty(B, H, H, Ci, device="cuda")
    g2 = ops.geom(B, 28, 28, Co, H, H, Ci, R, R, 1, 0)
    ops.conv_fwd(g2, nhwc(gy).cuda(), wd, dx, ops.epilogue())
    got = nchw(dx).cpu().double()
    err = (got-ref).abs()
    print("B", B, "rel", (got-ref).norm().item()/ref.norm().item(), "max", err.max().item())
    bad = (err > 1e-4).nonzero()
    print("n bad", len(bad), bad[:10].tolist())
    # per-(h,w) error map
    print((err.amax(dim=(0,1)) > 1e-4).int())
