import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/imbalanced-regression_amd')
from dirhip import ops, _lib as L
import bench
dev=torch.device('cuda'); g=torch.Generator(device=dev).manual_seed(1)
b,c,nb=65536,2048,100
xs=[torch.randn(b,c,device=dev,generator=g) for _ in range(3)]
bins=torch.randint(0,nb,(b,),device=dev,generator=g,dtype=torch.int32)
m1=torch.randn(nb,c,device=dev,generator=g); sc=torch.rand(nb,c,device=dev,generator=g)+0.5; m2=torch.randn(nb,c,device=dev,generator=g)
def t(fn,n=10): return bench.event_time_ms(fn,n)*1e3
by=2*b*c*4
r=lambda us: f"{us:7.1f} us {by/us/1e6:6.2f} TB/s"
print("row kernel fwd (in place)   ", r(t(lambda i: L.check(L.lib().dir_fds_calibrate_fwd(L.ptr(xs[i%3]),L.DIR_F32,L.ptr(bins),b,c,L.ptr(m1),L.ptr(sc),L.ptr(m2),L.stream_ptr(dev)),"x"))))
print("LDS-staged fwd (in place)   ", r(t(lambda i: ops.calibrate_fwd_lds_(xs[i%3],bins,m1,sc,m2))))
print("bwd (out of place)          ", r(t(lambda i: ops.calibrate_bwd(xs[i%3],bins,sc))))
print("torch mul_ in place         ", r(t(lambda i: xs[i%3].mul_(1.0001))))
o=torch.empty_like(xs[0])
print("torch mul out of place      ", r(t(lambda i: torch.mul(xs[i%3],1.0001,out=o))))
print("torch copy_                 ", r(t(lambda i: o.copy_(xs[i%3]))))
