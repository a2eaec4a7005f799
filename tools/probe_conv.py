"""GPU probe: dir_conv_fwd vs torch F.conv2d (MIOpen) on every ResNet-50 conv shape at batch B (bf16 NHWC)."""
import os, sys
import torch, torch.nn.functional as F
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
# (Cin, Cout, k, stride, Hin, count)
SH=[(64,64,1,1,56,1),(64,64,3,1,56,3),(64,256,1,1,56,4),(256,64,1,1,56,2),(256,128,1,1,56,1),(128,128,3,2,56,1),(128,512,1,1,28,4),
    (256,512,1,2,56,1),(512,128,1,1,28,3),(128,128,3,1,28,3),(512,256,1,1,28,1),(256,256,3,2,28,1),(256,1024,1,1,14,6),(512,1024,1,2,28,1),
    (1024,256,1,1,14,5),(256,256,3,1,14,5),(1024,512,1,1,14,1),(512,512,3,2,14,1),(512,2048,1,1,7,3),(1024,2048,1,2,14,1),(2048,512,1,1,7,2),(512,512,3,1,7,2)]
tot_o=tot_t=0
for cin,cout,k,st,h,cnt in SH:
    pad=k//2
    x=torch.randn(B,cin,h,h,device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w=(torch.randn(cout,cin,k,k,device='cuda')*0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ho=(h+2*pad-k)//st+1
    fl=2.0*B*ho*ho*cout*cin*k*k
    byt=2.0*(B*h*h*cin/(st*st if k==1 else 1)+B*ho*ho*cout+cout*cin*k*k)
    to=ev(lambda: conv2d_igemm(x,w,st,pad,want_stats=True))
    tt=ev(lambda: F.conv2d(x,w,None,st,pad))
    err=(conv2d_igemm(x,w,st,pad).float()-F.conv2d(x,w,None,st,pad).float()).abs().max().item()
    tot_o+=to*cnt; tot_t+=tt*cnt
    print(f"{cin:5d}->{cout:5d} k{k} s{st} H{h:3d} x{cnt} | ours {to*1e3:8.1f}us {fl/to/1e9:7.1f}TF {byt/to/1e6:6.0f}GB/s | miopen {tt*1e3:8.1f}us {fl/tt/1e9:7.1f}TF | x{tt/to:5.2f} maxdiff {err:.3g}",flush=True)
print(f"TOTAL fwd per pass: ours {tot_o:.2f} ms  miopen {tot_t:.2f} ms")
print("---- weight gradient")
from dirhip.conv import conv2d_wgrad
tot_o=tot_t=0
for cin,cout,k,st,h,cnt in SH:
    pad=k//2
    x=torch.randn(B,cin,h,h,device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w=(torch.randn(cout,cin,k,k,device='cuda')*0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ho=(h+2*pad-k)//st+1
    dy=torch.randn(B,cout,ho,ho,device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    fl=2.0*B*ho*ho*cout*cin*k*k
    to=ev(lambda: conv2d_wgrad(dy,x,k,st,pad))
    tt=ev(lambda: torch.ops.aten.convolution_backward(dy,x,w,None,[st,st],[pad,pad],[1,1],False,[0,0],1,[False,True,False]))
    ref=torch.ops.aten.convolution_backward(dy,x,w,None,[st,st],[pad,pad],[1,1],False,[0,0],1,[False,True,False])[1].float()
    mine=conv2d_wgrad(dy,x,k,st,pad)
    err=((mine-ref).abs().max()/ref.abs().max()).item()
    tot_o+=to*cnt; tot_t+=tt*cnt
    print(f"{cin:5d}->{cout:5d} k{k} s{st} H{h:3d} x{cnt} | ours {to*1e3:8.1f}us {fl/to/1e9:7.1f}TF | miopen {tt*1e3:8.1f}us {fl/tt/1e9:7.1f}TF | x{tt/to:5.2f} relerr {err:.2g}",flush=True)
print(f"TOTAL wgrad per pass: ours {tot_o:.2f} ms  miopen {tot_t:.2f} ms")
