"""NYUD2 dense calibration on the NCHW map: i.i.d. random depth (SURVEY's synthetic spec: worst case for the LDS lookups — every lane
of a wavefront a different bin) vs a smooth depth field (what a depth map looks like: neighbouring pixels share or neighbour a bin)."""
import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/imbalanced-regression_amd')
from dirhip import ops
import bench
dev=torch.device('cuda'); g=torch.Generator(device=dev).manual_seed(1)
b,c,h,w,nb=32,128,114,152,93
x=torch.rand(b,c,h,w,device=dev,generator=g); out=torch.empty_like(x)
t1,sc,t2=(torch.rand(nb,c,device=dev,generator=g)+0.5 for _ in range(3))
by=2*x.numel()*4
def smooth_depth():
    low=torch.rand(b,1,8,10,device=dev,generator=g)*9.3+0.7
    d=torch.nn.functional.interpolate(low,size=(h,w),mode='bilinear',align_corners=False)
    return (d+0.02*torch.randn(b,1,h,w,device=dev,generator=g)).clamp(0.7,10.0)
for name,depth in (("iid U(0.7,10)",torch.rand(b,1,h,w,device=dev,generator=g)*9.3+0.7),("smooth field",smooth_depth())):
    bins=ops.bin_scaled(depth.reshape(-1),10.0,7,100)
    us=bench.event_time_ms(lambda i: ops.calibrate_nchw(x,bins,t1,sc,t2,out=out),10)*1e3
    rows=x.permute(0,2,3,1).contiguous().view(-1,c)
    us2=bench.event_time_ms(lambda i: ops.calibrate_fwd_(rows,bins,t1,sc,t2),10)*1e3
    us3=bench.event_time_ms(lambda i: ops.scatter_stats(rows,bins,nb),5)*1e3
    print(f"{name:14s} nchw {us:7.1f} us {by/us/1e6:5.2f} TB/s | rows(NHWC, LDS-staged) {us2:7.1f} us {by/us2/1e6:5.2f} TB/s | scatter_stats {us3:7.1f} us {(rows.numel()*4+rows.shape[0]*4)/us3/1e6:5.2f} TB/s")
us=bench.event_time_ms(lambda i: x.permute(0,2,3,1).contiguous(),5)*1e3
print(f"permute().contiguous() alone {us:7.1f} us")
