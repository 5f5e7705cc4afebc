"""Microbench (GPU box): deep-supervision head forward/backward at BASELINE configs[1] size (env B, SIZE, REPS,
DTYPE=bf16 for bf16-stored features)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_nested4tiny_objects_keypoints_amd import ops
B = int(os.environ.get("B", "32")); REPS = int(os.environ.get("REPS", "5"))
S = int(os.environ.get("SIZE", "256")); BF = os.environ.get("DTYPE", "f32") == "bf16"
CH = int(os.environ.get("CH", "32")); NC = int(os.environ.get("NCLS", "4"))   # feature channels, classes (configs[4]: 64, 5)
x = torch.randn(B, S, S, CH, device="cuda"); x = x.to(torch.bfloat16) if BF else x; w = torch.randn(NC, CH, device="cuda") * 0.1; b = torch.zeros(NC, device="cuda")
o = torch.empty(B, NC, S, S, device="cuda"); go = torch.randn_like(o); dx = torch.empty_like(x)
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
        p, tf, (x.numel() * x.element_size() + o.numel() * 4) / tf / 1e9, tb, (2 * x.numel() * x.element_size() + 2 * o.numel() * 4) / tb / 1e9, tb2))
