"""How long does the host take to ENQUEUE one training step (no device sync inside)?  If this approaches the device
time per step, the path is launch-bound and multi-process scaling suffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_nested4tiny_objects_keypoints_amd import UNet_Nested, FocalLoss_BCE_2d, train_step
B = int(os.environ.get("B", "32")); S = int(os.environ.get("S", "256"))
torch.manual_seed(0)
FS = float(os.environ.get("FS", "1")); FS = int(FS) if FS.is_integer() else FS
CIN, NCLS, DEPTH = int(os.environ.get("CIN", "1")), int(os.environ.get("NCLS", "4")), int(os.environ.get("DEPTH", "4"))
m = UNet_Nested(CIN, NCLS, feature_scale=FS, depth=DEPTH).cuda().train()
if os.environ.get("DTYPE") == "bf16":
    m.set_activation_dtype(torch.bfloat16)
x = torch.randn(B, CIN, S, S, device="cuda"); t = torch.rand(B, NCLS, S, S, device="cuda")
opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True); crit = FocalLoss_BCE_2d(3, size_average=False)
for _ in range(3): train_step(m, opt, crit, x, t)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter(); train_step(m, opt, crit, x, t); host.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 10
print("B=%d S=%d: wall %.2f ms/step, host enqueue median %.2f ms/step (min %.2f)" % (B, S, tot * 1e3, sorted(host)[5] * 1e3, min(host) * 1e3))
