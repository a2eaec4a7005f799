import sys, os, torch, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/imbalanced-regression_amd')
from dirhip.conv_f32 import GATHER, TILE, conv2d_f32_dgrad, conv2d_f32_fwd, conv2d_f32_wgrad
import bench
B=256
tot={1:[0,0,0],2:[0,0,0]}
for cin,cout,k,st,h,cnt in bench.RESNET50_CONVS:
    pad=k//2; ho=(h+2*pad-k)//st+1
    x=torch.randn(B,cin,h,h,device='cuda').contiguous(memory_format=torch.channels_last)
    w=(torch.randn(cout,cin,k,k,device='cuda')*0.05).contiguous(memory_format=torch.channels_last)
    dy=torch.randn(B,cout,ho,ho,device='cuda').contiguous(memory_format=torch.channels_last)
    flop=2.0*B*ho*ho*cout*cin*k*k
    row=f"{cin:5d}->{cout:5d} k{k} s{st} H{h:3d} x{cnt}"
    for v in (GATHER,TILE):
        f=bench.event_time_ms(lambda i: conv2d_f32_fwd(x,w,st,pad,variant=v),3,warm=1)
        d=bench.event_time_ms(lambda i: conv2d_f32_dgrad(dy,w,(h,h),st,pad,variant=v),3,warm=1)
        g=bench.event_time_ms(lambda i: conv2d_f32_wgrad(dy,x,(k,k),st,pad,variant=v),3,warm=1)
        tot[v][0]+=f*cnt; tot[v][1]+=d*cnt; tot[v][2]+=g*cnt
        row+=f" | v{v} fwd {f*1e3:7.0f}us {flop/f/1e9:5.0f}TF dgrad {d*1e3:7.0f} {flop/d/1e9:5.0f}TF wgrad {g*1e3:7.0f} {flop/g/1e9:5.0f}TF"
    print(row, flush=True)
    del x,w,dy
print("sum ms (fwd,dgrad,wgrad): gather",tot[1],"tile",tot[2])
