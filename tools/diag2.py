import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tests.helpers import rel_err
from unet_nested4tiny_objects_keypoints_amd import engine, ops
from unet_nested4tiny_objects_keypoints_amd.ops import V
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
g = torch.Generator().manual_seed(3)
for (b,h,w,ci,co) in [(4,16,16,32,32),(4,16,16,16,16),(4,32,32,16,16),(2,16,16,64,32)]:
    wt = torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64)
    dy = torch.randn(b, co, h, w, generator=g, dtype=torch.float64)
    gate_in = torch.randn(b, co, h, w, generator=g, dtype=torch.float64)
    gate_out = torch.randn(b, ci, h, w, generator=g, dtype=torch.float64)
    ref = F.conv_transpose2d(dy * (gate_in > 0), wt, padding=1) * (gate_out > 0)
    out = torch.empty(b, h, w, ci, device="cuda")
    ops.gemm_fwd(b, h, w, 9, [V(nhwc(dy.float()), gate=nhwc(gate_in.float()))], [V(out, gate=nhwc(gate_out.float()))], engine.pack_conv_dgrad(wt.float().cuda()))
    print("dgrad gated", (b,h,w,ci,co), rel_err(nchw(out), ref.float()))
for (b,h,w,ci,co) in [(4,32,32,16,8),(4,16,16,32,16)]:
    wt = torch.randn(ci, co, 2, 2, generator=g, dtype=torch.float64)
    dy = torch.randn(b, co, 2*h, 2*w, generator=g, dtype=torch.float64)
    prev = torch.randn(b, ci, h, w, generator=g, dtype=torch.float64)
    ref = prev + F.conv2d(dy, wt, stride=2)
    out = nhwc(prev.float())
    ops.gemm_fwd(b, h, w, 1, engine._phase_views(nhwc(dy.float())), [V(out, accumulate=True)], engine.pack_deconv_dgrad(wt.float().cuda()))
    print("deconv dgrad acc", (b,h,w,ci,co), rel_err(nchw(out), ref.float()))
