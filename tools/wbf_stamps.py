"""Per-phase s_memtime cycles of wave 0 of wgrad_bf16_kernel's DMA tile loop, from a profiling build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUNETPP_WBF_STAMPS -I include -I <csrc> -c <csrc>/wgrad_bf16.hip -o build/exp/wgrad_bf16_stamps.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_stamps.so $(ls build/obj/*.o | grep -v wgrad_bf16) build/exp/wgrad_bf16_stamps.o
    UNETPP_LIB=$PWD/build/exp/libunetpp_stamps.so python tools/wbf_stamps.py [cin cout hw batch]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import _lib, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

ci, co, hw, b = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 32, 512, 8)))
x = torch.randn(b, hw, hw, ci, device="cuda").to(torch.bfloat16)
dy = torch.randn(b, hw, hw, co, device="cuda").to(torch.bfloat16)
dw, db = torch.empty(co, ci, 3, 3, device="cuda"), torch.empty(co, device="cuda")
lib = _lib.lib()
fn = lib.unetpp_debug_wbf_stamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
run = lambda: ops.wgrad(b, hw, hw, 9, [V(x)], [V(dy)], dw, (1, 9, ci * 9, 0), db)  # noqa: E731
run()
torch.cuda.synchronize()
fn(None, 1)
run()
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
fn(out, 0)
names = ["prologue", "wait DMA", "edge fix-up", "barrier", "DMA issue", "tr reads + MFMA"]
tiles, wgs = out[8], out[9]
print("workgroups %d, tiles %d (%.1f per workgroup)" % (wgs, tiles, tiles / max(1, wgs)))
for i, n in enumerate(names):
    print("  %-18s %10.0f cycles per tile (%10.0f per workgroup)" % (n, out[i] / max(1, tiles), out[i] / max(1, wgs)))
