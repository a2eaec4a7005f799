"""GPU probe: fused BN(+res)(+ReLU) fwd/bwd at ResNet-50 B=256 shapes — time and effective HBM GB/s vs torch eager."""
import os, sys, time
import torch, torch.nn as nn, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip.bn import bn_act
def ev(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
shapes=[(64,112,False),(64,56,False),(256,56,True),(128,28,False),(512,28,True),(256,14,False),(1024,14,True),(512,7,False),(2048,7,True)]
tot_f=tot_b=tot_rf=tot_rb=0
for c,hw,res in shapes:
    x=torch.randn(B,c,hw,hw,device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r=torch.randn_like(x).requires_grad_(True) if res else None
    dy=torch.randn_like(x)
    bn=nn.BatchNorm2d(c).cuda()
    nbytes=x.numel()*2
    y=bn_act(x,bn,True,r)
    tf=ev(lambda: bn_act(x,bn,True,r))
    def fb():
        y=bn_act(x,bn,True,r); y.backward(dy); 
    tfb=ev(fb); tb=tfb-tf
    def ref_f():
        y=F.batch_norm(x,bn.running_mean,bn.running_var,bn.weight,bn.bias,True,0.1,1e-5)
        if r is not None: y=y+r
        return torch.relu(y)
    trf=ev(ref_f)
    def ref_fb(): ref_f().backward(dy)
    trb=ev(ref_fb)-trf
    pf=(3+(1 if res else 0)); pb=(7+(1 if res else 0))
    print(f"C={c:5d} HW={hw:3d} res={int(res)} tensor={nbytes/1e6:7.1f}MB | ours fwd {tf*1e3:7.1f}us ({pf*nbytes/tf/1e6:6.0f} GB/s eff) bwd {tb*1e3:7.1f}us ({pb*nbytes/tb/1e6:6.0f} GB/s) | torch fwd {trf*1e3:7.1f}us bwd {trb*1e3:7.1f}us", flush=True)
