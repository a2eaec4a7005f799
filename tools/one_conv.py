"""Run one conv shape a few times (for rocprofv3 --pmc). args: cin cout k stride H B iters"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip.conv import conv2d_igemm
cin,cout,k,st,h,B,it=[int(a) for a in sys.argv[1:8]]
x=torch.randn(B,cin,h,h,device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w=(torch.randn(cout,cin,k,k,device='cuda')*0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for _ in range(it): y=conv2d_igemm(x,w,st,k//2,want_stats=True)
torch.cuda.synchronize()
