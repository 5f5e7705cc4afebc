"""Per-phase cycles of wave 0 of gemm_bf16_kernel's (unit, chunk) stream, from a profiling build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUNETPP_BF16_STAMPS -I include -I <csrc> -c <csrc>/gemm_bf16.hip -o build/exp/gemm_bf16_stamps.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_gstamps.so $(ls build/obj/*.o | grep -v gemm_bf16) build/exp/gemm_bf16_stamps.o
    UNETPP_LIB=$PWD/build/exp/libunetpp_gstamps.so python tools/gbf_stamps.py [cin[,cin..] cout hw batch]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import _lib, engine, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

args = sys.argv[1:5] if len(sys.argv) > 4 else ("32", "32", "512", "8")
cins = [int(v) for v in args[0].split(",")]
co, hw, b = int(args[1]), int(args[2]), int(args[3])
xs = [torch.randn(b, hw, hw, c, device="cuda").to(torch.bfloat16) for c in cins]
y = torch.empty(b, hw, hw, co, device="cuda", dtype=torch.bfloat16)
w = torch.randn(co, sum(cins), 3, 3, device="cuda") * 0.05
bias = torch.randn(co, device="cuda")
wp = engine.pack_conv_fwd(w)
lib = _lib.lib()
fn = lib.unetpp_debug_gbf_stamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
run = lambda: ops.gemm_fwd(b, hw, hw, 9, [V(t) for t in xs], [V(y, relu=True)], wp, bias)  # noqa: E731
run()
torch.cuda.synchronize()
fn(None, 1)
run()
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
fn(out, 0)
names = ["bookkeeping", "cursor + load issue", "LDS reads + MFMA", "barrier (MFMA)", "wait for loads", "epilogue",
         "staging stores", "barrier (staging)"]
chunks, units, wgs = out[8], out[9], out[10]
print("workgroups %d, units %d, chunks %d" % (wgs, units, chunks))
for i, n in enumerate(names):
    print("  %-22s %9.0f cycles per chunk  %9.0f per unit" % (n, out[i] / max(1, chunks), out[i] / max(1, units)))
print("  %-22s %9.0f cycles per chunk  %9.0f per unit" % ("total", sum(out[:8]) / max(1, chunks), sum(out[:8]) / max(1, units)))
