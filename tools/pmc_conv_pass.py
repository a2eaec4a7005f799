"""One launch of dir_conv_fwd per ResNet-50 conv layer shape (forward + stride-1 dgrad configuration) at batch B, after
one warm-up launch each — meant to run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes).
tools/pmc_conv_parse.py turns the counter CSVs into profiles/r02_conv_pmc_traffic.json."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
from dirhip.conv import conv2d_igemm
from bench import RESNET50_CONVS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for cin, cout, k, st, h, cnt in RESNET50_CONVS:
    pad = k // 2
    ho = (h + 2 * pad - k) // st + 1
    cfgs = [(cin, cout, h, st)] + ([(cout, cin, ho, 1)] if st == 1 else [])
    for ci, co, hh, s_ in cfgs:
        x = torch.randn(B, ci, hh, hh, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, k, k, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        for _ in range(2):
            conv2d_igemm(x, w, s_, pad)
        torch.cuda.synchronize()
        del x, w
