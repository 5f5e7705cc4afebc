"""Microbench (GPU box): deep-supervision head forward/backward at BASELINE configs[1] size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_nested4tiny_objects_keypoints_amd import ops
B = int(os.environ.get("B", "32")); REPS = int(os.environ.get("REPS", "5"))
x = torch.randn(B, 256, 256, 32, device="cuda"); w = torch.randn(4, 32, device="cuda") * 0.1; b = torch.zeros(4, device="cuda")
o = torch.empty(B, 4, 256, 256, device="cuda"); go = torch.randn_like(o); dx = torch.empty_like(x)
def t(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / REPS
for p in (0.0, 0.4):
    tf = t(lambda: ops.head_fwd(x, w, b, p, 1, None, o))
    tb = t(lambda: ops.head_bwd(go, o, x, w, p, 1, None, dx, False))
    tb2 = t(lambda: ops.head_bwd(go, o, x, w, p, 1, None, dx, True, gate_x=True))
    print("p=%.1f fwd %.3f ms (%.2f TB/s)  bwd %.3f ms (%.2f TB/s)  bwd acc+gate %.3f ms" % (
        p, tf, (x.numel() + o.numel()) * 4 / tf / 1e9, tb, (2 * x.numel() + 2 * o.numel()) * 4 / tb / 1e9, tb2))
