"""GPU probe (not a test): where does a ResNet-50 bf16 channels_last train step spend its time on a fresh
box — first-call (MIOpen kernel JIT / find) cost vs steady state; prints stage timestamps to stderr."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
T0 = time.time()
def log(*a):
    print(f"[{time.time()-T0:7.1f}s]", *a, file=sys.stderr, flush=True)
from dirhip.resnet import resnet50
from dirhip import loss as hl
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = resnet50(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9).to(dev).to(memory_format=torch.channels_last)
model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 100, (B, 1), device=dev).float(); w = torch.ones(B, 1, device=dev)
log("model built")
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out, _ = model(x, y, 0)
    loss = hl.weighted_l1_loss(out, y, w)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
for i in range(3):
    t = time.time(); l = step(); torch.cuda.synchronize(); log(f"step {i}: {time.time()-t:.2f}s loss {l.item():.3f}")
t = time.time()
for i in range(steps): step()
torch.cuda.synchronize(); dt = (time.time()-t)/steps
log(f"steady: {dt*1e3:.1f} ms/step  {B/dt:.0f} img/s  {B*24.287e9/dt/1e12:.1f} TFLOP/s")
with torch.no_grad():
    t = time.time()
    for i in range(steps):
        with torch.autocast("cuda", dtype=torch.bfloat16): model(x, y, 0)
    torch.cuda.synchronize(); dt = (time.time()-t)/steps
log(f"fwd-only (train mode): {dt*1e3:.1f} ms  {B/dt:.0f} img/s")
