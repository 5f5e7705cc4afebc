"""GPU-box probe: sample sclk / power with rocm-smi while one wgrad shape runs in a loop (is the MFMA clock power-bound?).
usage: python tools/clock_probe.py <shape substring> [seconds]"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
B, hw = 32, 256
cins, co = [32, 32, 32, 32], 32
ci = sum(cins)
xs = [torch.randn(B, hw, hw, c, device="cuda") for c in cins]
y = torch.randn(B, hw, hw, co, device="cuda")
w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
dw, db = torch.empty_like(w), torch.empty(co, device="cuda")
from unet_nested4tiny_objects_keypoints_amd import engine  # noqa: E402
wp = engine.pack_conv_fwd(w)
bias = torch.randn(co, device="cuda")


def run():
    if what == "wgrad":
        ops.wgrad(B, hw, hw, 9, [V(t) for t in xs], [V(y)], dw, (1, 9, ci * 9, 0), db, target_blocks=256)
    else:
        ops.gemm_fwd(B, hw, hw, 9, [V(t) for t in xs], [V(y, relu=True)], wp, bias)


samples = []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            mclk = re.findall(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
            pw = re.findall(r"Power \(W\): ([\d.]+)", out)
            samples.append((time.time(), sclk[:1], mclk[:1], pw[:1]))
        except Exception as exc:  # noqa: BLE001
            samples.append((time.time(), str(exc)))
        time.sleep(0.1)


run()
torch.cuda.synchronize()
th = threading.Thread(target=poll)
th.start()
time.sleep(0.5)
t0 = time.time()
n = 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.time() - t0 < secs:
    for _ in range(50):
        run()
    n += 50
    torch.cuda.synchronize()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
time.sleep(0.5)
stop = True
th.join()
flops = 2.0 * B * hw * hw * 9 * ci * co
print("%s: %.3f ms/launch  %.1f TF/s over %d launches" % (what, ms, flops / ms / 1e9, n))
for t, *rest in samples:
    print("  t=%.2f %s" % (t - t0, rest))
