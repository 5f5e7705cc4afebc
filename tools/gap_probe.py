"""GPU-box probe: dispatch gaps between back-to-back launches of one MFMA kernel (same descriptor, image packed once).
Run under rocprofv3 --kernel-trace and read the gaps with tools/rocprof_kernels.py / the trace."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import _lib, engine  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd._lib import GemmDesc  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

b, hw, ci, co = 32, 256, 32, 32
x = torch.randn(b, hw, hw, ci, device="cuda")
y = torch.empty(b, hw, hw, co, device="cuda")
w = engine.pack_conv_fwd(torch.randn(co, ci, 3, 3, device="cuda") * 0.05).packed()
lib = _lib.lib()
d = GemmDesc()
d.N, d.H, d.W, d.taps, d.n_in, d.n_out = b, hw, hw, 9, 1, 1
V(x).fill(d.inp[0])
V(y).fill(d.out[0])
d.weight = w.data_ptr()
n_img = int(lib.unetpp_gemm_weight_image_floats(C.byref(d)))
img = torch.empty(n_img, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.unetpp_gemm_pack_weight_image(C.byref(d), C.c_void_p(img.data_ptr()), st)
d.weight_image = img.data_ptr()
for _ in range(30):
    lib.unetpp_gemm_fwd(C.byref(d), st)
torch.cuda.synchronize()
print("done")
