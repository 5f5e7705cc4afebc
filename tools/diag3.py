import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import load_golden, sub, rel_err
from oracle.unet_nested_oracle import UNetNestedOracle
from oracle.step_oracle import focal_bce_2d_oracle
from unet_nested4tiny_objects_keypoints_amd import UNet_Nested, FocalLoss_BCE_2d, engine

z, ctor = load_golden("c1_fs4_64x64_b4_seed0")
state = sub(z, "state0")
x, target = torch.from_numpy(z["x"]), torch.from_numpy(z["target"])
ref = UNetNestedOracle(**ctor); ref.load_state_dict(state); ref = ref.double().train(); ref.drop_out.eval()
# capture oracle node outputs X[i][j] with grads
caps = {}
def hook(name):
    def f(mod, inp, out):
        out.retain_grad(); caps[name] = out
    return f
for i in range(4): getattr(ref, "conv%d0" % i).register_forward_hook(hook((i, 0)))
for j in range(1, 4):
    for i in range(4 - j): getattr(ref, "up_concat%d%d" % (i, j)).register_forward_hook(hook((i, j)))
outs = ref(x.double())
l = sum(focal_bce_2d_oracle(o, target.double()) for o in outs) / 3
l.backward()

taken = {}
orig_take = engine._GradBook.take
def take(self, key):
    t = orig_take(self, key)
    taken[key] = t.clone()
    return t
engine._GradBook.take = take
m = UNet_Nested(**ctor); m.load_state_dict(state); m = m.cuda().train(); m.drop_out.eval()
o = m(x.cuda())
crit = FocalLoss_BCE_2d(3, size_average=False)
(sum(crit(q, target.cuda()) for q in o) / 3).backward()
for key in sorted(taken, key=lambda k: (-k[1], -k[0])):
    got = taken[key].permute(0, 3, 1, 2).cpu()
    want = caps[key].grad.float()
    d = (got - want).abs()
    bad = (d > 1e-4 * want.abs().max())
    print(key, "rel_err %.2e" % rel_err(got, want), "bad elems %d / %d" % (int(bad.sum()), bad.numel()),
          "bad-by-channel", bad.sum((0, 2, 3)).tolist()[:16])
    if bad.any():
        idx = bad.nonzero()[:6].tolist()
        print("   first bad idx (n,c,y,x):", idx)

# gate agreement: compare sign patterns of every gated activation (node outputs and mid activations)
mids = {}
def hook_mid(name):
    def f(mod, inp, out): mids[name] = out.detach()
    return f
ref2 = UNetNestedOracle(**ctor); ref2.load_state_dict(state); ref2 = ref2.double().train(); ref2.drop_out.eval()
for i in range(4): getattr(ref2, "conv%d0" % i).conv1.register_forward_hook(hook_mid((i, 0)))
for j in range(1, 4):
    for i in range(4 - j): getattr(ref2, "up_concat%d%d" % (i, j)).conv.conv1.register_forward_hook(hook_mid((i, j)))
ref2(x.double())
saved = {}
orig_fwd = engine.forward_impl
def fwd(model, xx, training, save):
    outs, s = orig_fwd(model, xx, training, save)
    saved["s"] = s
    return outs, s
engine.forward_impl = fwd
m.zero_grad(); o = m(x.cuda())
s = saved["s"]
for key, r in s.pairs.items():
    for nm, mine, want in (("a1", r.a1, mids[key]), ("out", r.out, caps[key].detach())):
        a = mine.permute(0, 3, 1, 2).cpu()
        mism = ((a > 0) != (want > 0))
        if mism.any():
            idx = mism.nonzero().tolist()
            print(key, nm, "gate mismatches:", len(idx), [(i, float(a[tuple(i)]), float(want[tuple(i)])) for i in idx[:4]])
print("gate check done")
