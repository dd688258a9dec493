import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd")]
import torch
from ali_hip import _lib
v = os.environ.get("VARIANT")
if v is not None:
    _lib.LIB_PATH = os.path.join(ROOT, "imagecfgen-pytorch_amd", "ali_hip", f"libali_hip_s{v}.so")
from ali_hip import ops
def bench(B,H,C,K,R,st,pad, iters=30):
    P = (H + 2*pad - R)//st + 1
    x = torch.randn(B,H,H,C, device="cuda"); w = torch.randn(K,R*R,C, device="cuda")*0.05
    y = torch.empty(B,P,P,K, device="cuda"); b = torch.randn(K, device="cuda")
    g = ops.geom(B,H,H,C,P,P,K,R,R,st,pad); ep = ops.epilogue(bias=b, act=ops.ACT_LEAKY, slope=0.2)
    for _ in range(3): ops.conv_fwd(g,x,w,y,ep)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.conv_fwd(g,x,w,y,ep)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/iters
    return us, 2.0*B*P*P*K*C*R*R/us/1e6
shapes = [(512,11,64,128,4,1,0),(512,8,128,256,4,2,0),(512,24,32,64,4,2,0),(512,14,64,128,4,2,1),(512,1,1024,1024,1,1,0),(512,3,256,512,3,1,0),(2048,11,64,128,4,1,0)]
for bm, bn, sk in (("0","0","0"),("128","128","1"),("64","128","1")):
    os.environ["ALI_BM"]=bm; os.environ["ALI_BN"]=bn; os.environ["ALI_SPLITK"]=sk
    res = [bench(*s) for s in shapes]
    print("variant",v,"tile",bm,bn,"splitk",sk," ".join(f"{u:7.1f}us/{t:5.1f}TF" for u,t in res), flush=True)
