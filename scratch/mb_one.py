import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd")]
import torch
from ali_hip import ops
B,H,C,K,R,st,pad = [int(v) for v in os.environ.get("SHAPE", "512,11,64,128,4,1,0").split(",")]
P = (H + 2*pad - R)//st + 1
x = torch.randn(B,H,H,C, device="cuda"); w = torch.randn(K,R*R,C, device="cuda")*0.05
y = torch.empty(B,P,P,K, device="cuda"); b = torch.randn(K, device="cuda")
g = ops.geom(B,H,H,C,P,P,K,R,R,st,pad); ep = ops.epilogue(bias=b, act=ops.ACT_LEAKY, slope=0.2)
for _ in range(10): ops.conv_fwd(g,x,w,y,ep)
torch.cuda.synchronize()
