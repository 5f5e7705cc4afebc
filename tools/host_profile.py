"""cProfile of the HOST side of training steps (where do the 4-10 ms of enqueue time per step go?).
env as tools/host_overhead.py (B, S, DTYPE, FS, CIN, NCLS, DEPTH); prints the top functions by own time and by cumulative time."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, train_step

B, S = int(os.environ.get("B", "32")), int(os.environ.get("S", "256"))
FS = float(os.environ.get("FS", "1"))
FS = int(FS) if FS.is_integer() else FS
CIN, NCLS, DEPTH = int(os.environ.get("CIN", "1")), int(os.environ.get("NCLS", "4")), int(os.environ.get("DEPTH", "4"))
torch.manual_seed(0)
m = UNet_Nested(CIN, NCLS, feature_scale=FS, depth=DEPTH).cuda().train()
if os.environ.get("DTYPE") == "bf16":
    m.set_activation_dtype(torch.bfloat16)
x, t = torch.randn(B, CIN, S, S, device="cuda"), torch.rand(B, NCLS, S, S, device="cuda")
opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
crit = FocalLoss_BCE_2d(3, size_average=False)
for _ in range(5):
    train_step(m, opt, crit, x, t)
torch.cuda.synchronize()
N = 10
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    train_step(m, opt, crit, x, t)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
print("total profiled host time per step: %.2f ms (profiler overhead included)" % (1e3 * st.total_tt / N))
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
