"""Diagnostic (GPU box): per-parameter gradient error of the HIP path and of the fp32 CPU oracle, both against an fp64 CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import load_golden, sub, rel_err
from oracle.unet_nested_oracle import UNetNestedOracle
from oracle.step_oracle import focal_bce_2d_oracle
from unet_nested4tiny_objects_keypoints_amd import UNet_Nested, FocalLoss_BCE_2d

name = sys.argv[1] if len(sys.argv) > 1 else "c1_fs4_64x64_b4_seed0"
z, ctor = load_golden(name)
state = sub(z, "state0")
x, target = torch.from_numpy(z["x"]), torch.from_numpy(z["target"])

def run_ref(dtype):
    m = UNetNestedOracle(**ctor)
    m.load_state_dict(state)
    m = m.to(dtype).train(); m.drop_out.eval()
    outs = m(x.to(dtype))
    l = sum(focal_bce_2d_oracle(o, target.to(dtype)) for o in outs) / len(outs)
    l.backward()
    return {k: p.grad for k, p in m.named_parameters()}, [o.detach() for o in outs]

g64, o64 = run_ref(torch.float64)
g32, o32 = run_ref(torch.float32)
m = UNet_Nested(**ctor); m.load_state_dict(state); m = m.cuda().train(); m.drop_out.eval()
outs = m(x.cuda())
crit = FocalLoss_BCE_2d(3, size_average=False)
l = sum(crit(o, target.cuda()) for o in outs) / len(outs)
l.backward()
gh = {k: p.grad.cpu() for k, p in m.named_parameters()}
print("%-36s %10s %10s %10s %10s" % ("param", "hip/ref32", "hip/ref64", "ref32/ref64", "gold/ref64"))
gold = sub(z, "grad")
for k in g64:
    print("%-36s %10.2e %10.2e %10.2e %10.2e" % (k, rel_err(gh[k], g32[k]), rel_err(gh[k], g64[k]), rel_err(g32[k], g64[k]), rel_err(gold[k], g64[k])))
for i in range(3):
    print("out", i, rel_err(outs[i].detach().cpu(), o64[i]), rel_err(o32[i], o64[i]))
