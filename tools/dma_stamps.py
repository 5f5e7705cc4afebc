"""Per-phase cycles (s_memtime ticks, 100 MHz) of wave 0 of gemm_bf16_dma_kernel's (unit, chunk) stream, from a profiling build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUNETPP_DMA_STAMPS -I include -I <csrc> -c <csrc>/gemm_bf16_dma.hip -o build/exp/dma_stamps.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_dstamps.so $(ls build/obj/*.o | grep -v gemm_bf16_dma) build/exp/dma_stamps.o
    UNETPP_LIB=$PWD/build/exp/libunetpp_dstamps.so python tools/dma_stamps.py [cin[,cin..] cout hw batch [dgrad]]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import _lib, engine, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

args = sys.argv[1:5] if len(sys.argv) > 4 else ("64,64,64,64,64", "64", "384", "4")
dgrad = len(sys.argv) > 5 and sys.argv[5] == "dgrad"
cins = [int(v) for v in args[0].split(",")]
co, hw, b = int(args[1]), int(args[2]), int(args[3])
xs = [torch.randn(b, hw, hw, c, device="cuda").to(torch.bfloat16) for c in cins]
y = torch.randn(b, hw, hw, co, device="cuda").to(torch.bfloat16)
w = torch.randn(co, sum(cins), 3, 3, device="cuda") * 0.05
bias = torch.randn(co, device="cuda")
wp, wd = engine.pack_conv_fwd(w), engine.pack_conv_dgrad(w)
lib = _lib.lib()
fn = C.CDLL(_lib.LIB_PATH).unetpp_debug_dma_stamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
dxs = [torch.empty_like(t) for t in xs]
if dgrad:
    run = lambda: ops.gemm_fwd(b, hw, hw, 9, [V(y)], [V(t) for t in dxs], wd)  # noqa: E731
else:
    run = lambda: ops.gemm_fwd(b, hw, hw, 9, [V(t) for t in xs], [V(y, relu=True)], wp, bias)  # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
fn(None, 1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
run()
e.record()
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
fn(out, 0)
names = ["cursor advance", "DMA issue + operand requests", "LDS reads + MFMA", "wait for DMA (+ older stores)", "epilogue",
         "barrier"]
chunks, units, wgs = out[8], out[9], out[10]
print("%s %s -> %d at %dx%dx%d: %.1f us; workgroups %d, units %d, chunks %d (ticks of 10 ns)"
      % ("dgrad" if dgrad else "fwd", cins, co, hw, hw, b, 1e3 * s.elapsed_time(e), wgs, units, chunks))
for i, n in enumerate(names):
    print("  %-32s %8.2f us per chunk  %8.2f us per unit" % (n, out[i] / max(1, chunks) / 100.0, out[i] / max(1, units) / 100.0))
print("  %-32s %8.2f us per chunk  %8.2f us per unit" % ("total", sum(out[:8]) / max(1, chunks) / 100.0, sum(out[:8]) / max(1, units) / 100.0))
