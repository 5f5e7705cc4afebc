"""Diagnostic (GPU box): per-parameter gradient error of the HIP path and of the fp32 CPU oracle, both against an fp64
CPU oracle, at a given geometry, all three backward passes with the HIP forward's ReLU gates (tests/helpers.py).
    python tools/diag_grads_shape.py feature_scale batch H W [depth in_channels n_classes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle.step_oracle import focal_bce_2d_oracle  # noqa: E402
from oracle.unet_nested_oracle import UNetNestedOracle  # noqa: E402
from tests.helpers import install_hip_gates, rel_err  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested  # noqa: E402

a = sys.argv[1:]
fs = float(a[0]); fs = int(fs) if fs.is_integer() else fs
b, h, w = int(a[1]), int(a[2]), int(a[3])
depth = int(a[4]) if len(a) > 4 else 4
cin = int(a[5]) if len(a) > 5 else 1
ncls = int(a[6]) if len(a) > 6 else 4
ctor = dict(in_channels=cin, n_classes=ncls, feature_scale=fs, depth=depth)
torch.manual_seed(11)
ref = UNetNestedOracle(**ctor)
state = {k: v.clone() for k, v in ref.state_dict().items()}
x, target = torch.randn(b, cin, h, w), torch.rand(b, ncls, h, w)
if os.environ.get("POISON"):  # fill the caching allocator's pool with NaN: an uninitialised read becomes loud
    junk = [torch.full((1 << 28,), float(os.environ["POISON"]), device="cuda") for _ in range(int(os.environ.get("POISON_GB", "8")))]
    del junk
m = UNet_Nested(**ctor)
m.load_state_dict(state)
m = m.cuda().train()
m.drop_out.eval()
m._debug_keep_saved = True
outs = m(x.cuda())
crit = FocalLoss_BCE_2d(3, size_average=False)
(sum(crit(o, target.cuda()) for o in outs) / len(outs)).backward()
gh = {k: p.grad.cpu() for k, p in m.named_parameters()}
print("non-finite grads:", [k for k, g in gh.items() if not torch.isfinite(g).all()])
print("non-finite outs:", [i for i, o in enumerate(outs) if not torch.isfinite(o).all()])


def run_ref(dtype):
    r = UNetNestedOracle(**ctor)
    r.load_state_dict(state)
    r = r.to(dtype).train()
    r.drop_out.eval()
    gated = install_hip_gates(r, m._debug_saved)
    o = r(x.to(dtype))
    (sum(focal_bce_2d_oracle(t, target.to(dtype)) for t in o) / len(o)).backward()
    print(dtype, "flips", sum(g.flips for g in gated))
    return {k: p.grad for k, p in r.named_parameters()}, [t.detach() for t in o]


g64, o64 = run_ref(torch.float64)
g32, o32 = run_ref(torch.float32)
print("%-36s %10s %10s %10s" % ("param", "hip/ref32", "hip/ref64", "ref32/ref64"))
for k in g64:
    print("%-36s %10.2e %10.2e %10.2e" % (k, rel_err(gh[k], g32[k]), rel_err(gh[k], g64[k]), rel_err(g32[k], g64[k])))
for i in range(len(outs)):
    print("out", i, rel_err(outs[i].detach().cpu(), o64[i]), rel_err(o32[i], o64[i]))
