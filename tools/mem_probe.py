"""Peak device memory of one training step (torch.cuda.max_memory_allocated), for the grouped and the per-consumer
input-gradient schedules (run twice: UNETPP_NO_GROUPED_DGRAD unset / =1).  env: B, S, DTYPE, DEPTH, FS, CIN, NCLS."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, engine, train_step

B, S = int(os.environ.get("B", "32")), int(os.environ.get("S", "256"))
FS = float(os.environ.get("FS", "1"))
FS = int(FS) if FS.is_integer() else FS
cin, ncls, depth = int(os.environ.get("CIN", "1")), int(os.environ.get("NCLS", "4")), int(os.environ.get("DEPTH", "4"))
torch.manual_seed(0)
m = UNet_Nested(cin, ncls, feature_scale=FS, depth=depth).cuda().train()
if os.environ.get("DTYPE") == "bf16":
    m.set_activation_dtype(torch.bfloat16)
x, t = torch.randn(B, cin, S, S, device="cuda"), torch.rand(B, ncls, S, S, device="cuda")
opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
crit = FocalLoss_BCE_2d(3, size_average=False)
for _ in range(2):
    train_step(m, opt, crit, x, t)
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
train_step(m, opt, crit, x, t)
torch.cuda.synchronize()
print("grouped_dgrad=%s dtype=%s depth=%d base=%d %dx%d batch %d: peak %.2f GB allocated, %.2f GB reserved"
      % (engine.USE_GROUPED_DGRAD, os.environ.get("DTYPE", "f32"), depth, int(32 / FS), S, S, B,
         torch.cuda.max_memory_allocated() / 1e9, torch.cuda.max_memory_reserved() / 1e9))
