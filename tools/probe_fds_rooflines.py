import sys, torch, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/imbalanced-regression_amd')
import bench
rows = bench.fds_kernel_rooflines(torch.device('cuda'))
for r in rows: print(f"{r['kernel'][:75]:75s} {r['shape'][:34]:34s} {r['ms']*1e3:9.1f} us  {r['achieved']:8.0f} GB/s  frac {r['frac']:.3f}")
