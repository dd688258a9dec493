import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd")]
import torch
from ali_hip import ops
def bench(B,H,C,K,R,st,pad, iters=30):
    P = (H + 2*pad - R)//st + 1
    x = torch.randn(B,H,H,C, device="cuda"); dy = torch.randn(B,P,P,K, device="cuda")
    dw = torch.empty(K, C, R, R, device="cuda"); db = torch.empty(K, device="cuda")
    g = ops.geom(B,H,H,C,P,P,K,R,R,st,pad)
    f = lambda: ops.conv_bwd_weight(g, x, dy, dw, C, K, C*R*R, R*R, 1, db=db)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/iters
    return us, 2.0*B*P*P*K*C*R*R/us/1e6
for sh in [(512,28,8,32,5,1,0),(1024,28,8,32,5,1,0),(512,24,32,64,4,2,0)]:
    print(sh, "%.1f us %.1f TF" % bench(*sh))
