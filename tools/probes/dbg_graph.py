import copy, sys, os
sys.path.insert(0, os.getcwd())
import torch
from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep, train_step, UNet_Nested
dev = torch.device("cuda:0")
crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
ctor = dict(in_channels=1, n_classes=4, feature_scale=4)
N = 14
mk = lambda ps: torch.optim.Adam(ps, lr=1e-3, fused=True, capturable=True)
def run(variant):
    torch.manual_seed(81)
    a = UNet_Nested(**ctor).to(dev).train(); a.drop_out.p = 0.0
    b = copy.deepcopy(a)
    oa, ob = mk(a.parameters()), mk(b.parameters())
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(2, 1, 64, 64, generator=g).to(dev) for _ in range(N)]
    ts = [torch.rand(2, 4, 64, 64, generator=g).to(dev) for _ in range(N)]
    step = GraphedTrainStep(a, oa, crit, xs[0], ts[0], capture_optimizer=True, restore_state=(os.environ.get("RESTORE", "1") == "1"))
    la = []
    for p in b.parameters():
        p.grad = torch.randn_like(p) * 1e-3
    for i in range(N):
        la.append(float(step(xs[i], ts[i])[1]))
        if variant == "sync+full":
            torch.cuda.synchronize()
        if variant in ("full", "sync+full"):
            train_step(b, ob, crit, xs[i], ts[i])
        elif variant == "fwdbwd":
            outs = b(xs[i]); (sum(crit(o, ts[i]) for o in outs) / 3).backward()
        elif variant == "adam-only":
            ob.step()
        elif variant == "small-alloc":
            junk = [torch.randn(int(n), device=dev) for n in torch.randint(16, 60000, (400,))]
            del junk
        elif variant == "alloc-only":
            junk = [torch.randn(1 << 18, device=dev) for _ in range(40)]
            del junk
        elif variant == "full-sidestream":
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                train_step(b, ob, crit, xs[i], ts[i])
            torch.cuda.current_stream().wait_stream(s)
    return la
ref = run("none")
for v in ("sync+full", "sync+full", "sync+full", "full", "none"):
    la = run(v)
    first = next((i for i in range(N) if la[i] != ref[i]), None)
    print("%-16s first differing loss of the graphed model at step %s" % (v, first), flush=True)
