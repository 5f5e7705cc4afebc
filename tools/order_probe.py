"""Does a bf16 512x512 step run slower after the fp32 headline model has run in the same process?  Phases: c3, headline, c3,
each `STEPS` timed steps after 10 untimed ones; prints wall ms/step and the host's enqueue time.  Run it plain and
under `rocprofv3 --kernel-trace` (tools/step_gaps.py reads the trace)."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, train_step

STEPS = int(os.environ.get("STEPS", "30"))
crit = FocalLoss_BCE_2d(3, size_average=False)


def phase(name, bf16, size, batch):
    torch.manual_seed(0)
    m = UNet_Nested(1, 4, feature_scale=1).cuda().train()
    if bf16:
        m.set_activation_dtype(torch.bfloat16)
    x, t = torch.randn(batch, 1, size, size, device="cuda"), torch.rand(batch, 4, size, size, device="cuda")
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    for _ in range(10):
        train_step(m, opt, crit, x, t)
    torch.cuda.synchronize()
    host, t0 = [], time.perf_counter()
    for _ in range(STEPS):
        a = time.perf_counter()
        train_step(m, opt, crit, x, t)
        host.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-22s %.3f ms/step (host: enqueue loop %.3f ms/step, median step %.3f, max %.3f); reserved %.2f GB"
          % (name, 1e3 * dt / STEPS, 1e3 * t_enq / STEPS, 1e3 * sorted(host)[STEPS // 2], 1e3 * max(host),
             torch.cuda.memory_reserved() / 1e9), flush=True)
    del m, opt, x, t
    gc.collect()
    if os.environ.get("EMPTY", "1") == "1":
        torch.cuda.empty_cache()


order = os.environ.get("ORDER", "c3,head,c3,c3").split(",")
for k, what in enumerate(order):
    if what == "c3":
        phase("c3 bf16 512 b8 [%d]" % k, True, 512, 8)
    elif what == "c3f":
        phase("c3 fp32 512 b8 [%d]" % k, False, 512, 8)
    elif what == "headbf":
        phase("bf16 256 b32 [%d]" % k, True, 256, 32)
    else:
        phase("headline fp32 [%d]" % k, False, 256, 32)
