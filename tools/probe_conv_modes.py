"""GPU probe: dir_conv_fwd / dir_conv_wgrad on every ResNet-50 conv shape (forward shapes, the stride-1 data-gradient
shapes and the weight gradients) at batch B; PROBE_ROWS=1 prints the per-layer table. Run once per kernel configuration:
the experiment switches (DIR_CONV_PF, DIR_CONV_PF_KT, DIR_CONV_NBUF, DIR_CONV_NBUF_KT, DIR_WGRAD_ROUNDS, DIR_BN_CAP) are
read once per process."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip.conv import conv2d_igemm
def ev(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
SH=[(64,64,1,1,56,1),(64,64,3,1,56,3),(64,256,1,1,56,4),(256,64,1,1,56,2),(256,128,1,1,56,1),(128,128,3,2,56,1),(128,512,1,1,28,4),
    (256,512,1,2,56,1),(512,128,1,1,28,3),(128,128,3,1,28,3),(512,256,1,1,28,1),(256,256,3,2,28,1),(256,1024,1,1,14,6),(512,1024,1,2,28,1),
    (1024,256,1,1,14,5),(256,256,3,1,14,5),(1024,512,1,1,14,1),(512,512,3,2,14,1),(512,2048,1,1,7,3),(1024,2048,1,2,14,1),(2048,512,1,1,7,2),(512,512,3,1,7,2)]
tag=" ".join(f"{k}={v}" for k,v in os.environ.items() if k.startswith("DIR_CONV"))
tot_f=tot_d=0
rows=[]
for cin,cout,k,st,h,cnt in SH:
    pad=k//2
    x=torch.randn(B,cin,h,h,device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w=(torch.randn(cout,cin,k,k,device='cuda')*0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    tf=ev(lambda: conv2d_igemm(x,w,st,pad,want_stats=True))
    tot_f+=tf*cnt
    td=0.0
    if st==1:          # data gradient of a stride-1 conv = the same kernel with Cin/Cout swapped at the output resolution
        dy=torch.randn(B,cout,h,h,device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wr=(torch.randn(cin,cout,k,k,device='cuda')*0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        td=ev(lambda: conv2d_igemm(dy,wr,1,pad))
        tot_d+=td*cnt
    rows.append(f"{cin:5d}->{cout:5d} k{k} s{st} H{h:3d} x{cnt} | fwd {tf*1e3:8.1f}us | dgrad {td*1e3:8.1f}us")
from dirhip.conv import conv2d_wgrad
tot_w=0
for idx,(cin,cout,k,st,h,cnt) in enumerate(SH):
    pad=k//2; ho=(h+2*pad-k)//st+1
    x=torch.randn(B,cin,h,h,device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy=torch.randn(B,cout,ho,ho,device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    tw=ev(lambda: conv2d_wgrad(dy,x,k,st,pad))
    tot_w+=tw*cnt
    rows[idx]+=f" | wgrad {tw*1e3:8.1f}us"
print(f"[{tag or 'default'}] B={B} fwd {tot_f:.3f} ms  dgrad(stride-1) {tot_d:.3f} ms  wgrad {tot_w:.3f} ms")
if os.environ.get("PROBE_ROWS"): print("\n".join(rows))
