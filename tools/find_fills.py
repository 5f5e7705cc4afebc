"""Which host-side calls launch the small fill / add / copy kernels of a train step (GPU box): torch.profiler with
stacks over two steps of the headline workload, grouped by op and the first source line inside this repository."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, train_step  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = UNet_Nested(in_channels=1, n_classes=4, feature_scale=2, depth=4).to(dev).train()
x = torch.randn(8, 1, 256, 256, device=dev)
target = torch.rand(8, 4, 256, 256, device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
for _ in range(3):
    train_step(model, opt, crit, x, target)
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(STEPS):
        train_step(model, opt, crit, x, target)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::add_", "aten::copy_", "aten::mul", "aten::add", "aten::zeros", "aten::ones_like",
        "aten::zeros_like", "aten::full")
count = Counter()
for ev in prof.events():
    if ev.name in want:
        where = "?"
        for fr in ev.stack or []:
            if ROOT in fr and "find_fills" not in fr:
                where = fr.replace(ROOT + "/", "")
                break
        if where == "?" and ev.stack:
            where = ev.stack[0][-90:]
        count[(ev.name, where)] += 1
for (name, where), n in sorted(count.items(), key=lambda kv: -kv[1])[:40]:
    print("%5.1f/step  %-18s %s" % (n / STEPS, name, where))
