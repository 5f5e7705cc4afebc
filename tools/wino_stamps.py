"""Where a wave of gemm_wino_kernel spends its cycles (GPU box): runs one convolution shape through a profiling build
of the library (gemm_wino.hip compiled with -DUNETPP_WINO_STAMPS, linked as build/stamps/libunetpp_stamps.so -- the
recipe is in tools/README.md) and prints the per-phase s_memtime totals averaged over waves.

  UNETPP_LIB=build/stamps/libunetpp_stamps.so python tools/wino_stamps.py [cin cout hw]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import _lib, engine, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

WGRAD = "--wgrad" in sys.argv  # the weight-gradient kernel (wgrad_wino.hip built with -DUNETPP_WWINO_STAMPS)
args = [v for v in sys.argv[1:] if not v.startswith("--")]
ci, co, hw = (int(v) for v in (args[:3] if len(args) >= 3 else (32, 32, 256)))
B = int(os.environ.get("B", "32"))
REPS = int(os.environ.get("REPS", "10"))
PHASES = ["prologue", "staging store + cursor + load issue", "MFMA first half (s=0)", "(unused)", "MFMA second half (s=1)",
          "barrier after MFMA", "epilogue", "barrier after epilogue + stats"]

if WGRAD:
    PHASES = ["prologue", "DMA issue (waves 0-3)", "MFMA half 0", "DMA issue (waves 4-7)", "MFMA half 1",
              "wait for the DMAs", "barrier", "(unused)"]
lib = _lib.lib()
fn = lib.unetpp_debug_wwino_stamps if WGRAD else lib.unetpp_debug_wino_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
fn.restype = ctypes.c_int
x, y = torch.randn(B, hw, hw, ci, device="cuda"), torch.randn(B, hw, hw, co, device="cuda")
w, bias = torch.randn(co, ci, 3, 3, device="cuda") * 0.05, torch.randn(co, device="cuda")
wp = engine.pack_conv_fwd(w)


dw, db = torch.empty(co, ci, 3, 3, device="cuda"), torch.empty(co, device="cuda")


def run():
    if WGRAD:
        ops.wgrad(B, hw, hw, 9, [V(x)], [V(y)], dw, (1, 9, ci * 9, 0), db)
    else:
        ops.gemm_fwd(B, hw, hw, 9, [V(x)], [V(y)], wp, bias, None)


run()
torch.cuda.synchronize()
fn(None, 1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(REPS):
    run()
e.record()
torch.cuda.synchronize()
out = (ctypes.c_uint64 * 16)()
assert fn(out, 0) == 0
waves = out[8]
us = 1e3 * s.elapsed_time(e) / REPS
total = sum(out[i] for i in range(16) if i != 8)
if not WGRAD:
    PHASES = PHASES + ["(wave count)", "epilogue: geometry + bias", "epilogue: output row 0", "epilogue: output row 1", "", "", "", ""]
    PHASES[6] = "epilogue: zeroing, statistics"
print("%d -> %d channels at %dx%d, batch %d: %.1f us per launch (stamped build), %d waves per launch" %
      (ci, co, hw, hw, B, us, waves // REPS))
print("s_memtime ticks per wave per launch: %.0f (%.2f ticks/ns)" % (total / waves, total / waves / (us * 1e3)))
for i, name in enumerate(PHASES):
    if i == 8 or not name:
        continue
    print("  %-34s %9.0f  %5.1f %%" % (name, out[i] / waves, 100.0 * out[i] / total))
