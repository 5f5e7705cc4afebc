"""In-kernel clock of wgrad_bf16_quad_kernel's tile loop (GPU box; UNETPP_LIB = a -DUNETPP_WQ_EXP_CLOCK build made by
tools/wgrad_quad_ablation.sh).  Runs one wide bf16 weight gradient back to back for SECONDS (default 2) on random
data, then reads the shader cycles and 100 MHz reference ticks the workgroups spent in their tile loops:
clock = cycles / ticks * 100 MHz (MI355X_MICROARCH.md, DVFS item 6).  ZERO=1 repeats it on all-zero operands."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import _lib, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

B, HW, CINS, CO = 8, 256, (64, 64, 64), 64
SECONDS = float(os.environ.get("SECONDS", "2"))


def run(zero):
    mk = (lambda *s: torch.zeros(*s, device="cuda", dtype=torch.bfloat16)) if zero else \
         (lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16))
    xs = [mk(B, HW, HW, c) for c in CINS]
    dy = mk(B, HW, HW, CO)
    ci = sum(CINS)
    dw, db = torch.empty(CO, ci, 3, 3, device="cuda"), torch.empty(CO, device="cuda")
    lib = _lib.lib()
    lib.unetpp_wq_clock_read.restype = C.c_int
    out = (C.c_ulonglong * 2)()

    def once():
        ops.wgrad(B, HW, HW, 9, [V(t) for t in xs], [V(dy)], dw, (1, 9, ci * 9, 0), db)

    once()
    torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < SECONDS:       # warm the chip up under this load
        for _ in range(50):
            once()
        torch.cuda.synchronize()
    lib.unetpp_wq_clock_read(out, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    n = 200
    for _ in range(n):
        once()
    e.record()
    torch.cuda.synchronize()
    lib.unetpp_wq_clock_read(out, 0)
    print("%s operands: tile loops at %.3f GHz (%.3e cycles / %.3e ticks of 10 ns); wgrad + finish %.1f us per call" % (
        "all-zero" if zero else "random  ", out[0] / out[1] * 0.1, out[0], out[1], s.elapsed_time(e) / n * 1e3))


run(False)
if os.environ.get("ZERO", "1") == "1":
    run(True)
