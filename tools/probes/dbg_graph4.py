import copy, sys, os
sys.path.insert(0, os.getcwd())
import torch
from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep, train_step, UNet_Nested
dev = torch.device("cuda:0")
crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
ctor = dict(in_channels=1, n_classes=4, feature_scale=4)
N = 6
mk = lambda ps: torch.optim.Adam(ps, lr=1e-3, fused=True, capturable=True)
torch.manual_seed(81)
a = UNet_Nested(**ctor).to(dev).train(); a.drop_out.p = 0.0
b = copy.deepcopy(a)
oa, ob = mk(a.parameters()), mk(b.parameters())
g = torch.Generator().manual_seed(5)
xs = [torch.randn(2, 1, 64, 64, generator=g).to(dev) for _ in range(N)]
ts = [torch.rand(2, 4, 64, 64, generator=g).to(dev) for _ in range(N)]
step = GraphedTrainStep(a, oa, crit, xs[0], ts[0], capture_optimizer=(os.environ.get("CAPOPT", "1") == "1"))
rec_a, rec_b = [], []
def snap(m, o, outs, loss):
    d = {"loss": loss.detach().clone()}
    for i, t in enumerate(outs): d["out%d" % i] = t.detach().clone()
    for k, p in m.named_parameters():
        d["param/" + k] = p.detach().clone(); d["grad/" + k] = p.grad.detach().clone()
        for sk, sv in o.state[p].items():
            if torch.is_tensor(sv): d["opt/%s/%s" % (k, sk)] = sv.detach().clone()
    for k, v in m.named_buffers(): d["buf/" + k] = v.detach().clone()
    return d
for i in range(N):
    torch.cuda.synchronize()
    outs, loss = step(xs[i], ts[i])
    torch.cuda.synchronize()
    rec_a.append(snap(a, oa, outs, loss))
for i in range(N):
    outs, loss = train_step(b, ob, crit, xs[i], ts[i])
    torch.cuda.synchronize()
    rec_b.append(snap(b, ob, outs, loss))
for i in range(N):
    bad = [k for k in rec_a[i] if not torch.equal(rec_a[i][k], rec_b[i][k])]
    cats = {}
    for k in bad: cats[k.split("/")[0]] = cats.get(k.split("/")[0], 0) + 1
    worst = max([(float((rec_a[i][k].float() - rec_b[i][k].float()).abs().max()), k) for k in bad], default=None)
    print("step", i, "loss graph %.6f eager %.6f" % (float(rec_a[i]["loss"]), float(rec_b[i]["loss"])), "differing:", cats, "worst", worst)
