import sys, os, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/imbalanced-regression_amd')
from dirhip.conv_f32 import GATHER, TILE, conv2d_f32_fwd
import bench
B=256
for cin,cout,k,st,h,cnt in bench.RESNET50_CONVS:
    pad=k//2; ho=(h+2*pad-k)//st+1
    x=torch.randn(B,cin,h,h,device='cuda').contiguous(memory_format=torch.channels_last)
    w=(torch.randn(cout,cin,k,k,device='cuda')*0.05).contiguous(memory_format=torch.channels_last)
    flop=2.0*B*ho*ho*cout*cin*k*k
    y2=conv2d_f32_fwd(x,w,st,pad,variant=TILE); y3=conv2d_f32_fwd(x,w,st,pad,variant=3)
    eq=torch.equal(y2,y3)
    f2=bench.event_time_ms(lambda i: conv2d_f32_fwd(x,w,st,pad,variant=TILE),3,warm=1)
    f3=bench.event_time_ms(lambda i: conv2d_f32_fwd(x,w,st,pad,variant=3),3,warm=1)
    print(f"{cin:5d}->{cout:5d} k{k} s{st} H{h:3d} x{cnt} tile {f2*1e3:6.0f}us {flop/f2/1e9:5.0f}TF  loader-wave {f3*1e3:6.0f}us {flop/f3/1e9:5.0f}TF equal={eq}", flush=True)
