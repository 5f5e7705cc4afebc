"""Is work of a replayed training-step graph still running after the launch stream says it is done?"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep, UNet_Nested
dev = torch.device("cuda:0")
crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
torch.manual_seed(81)
a = UNet_Nested(in_channels=1, n_classes=4, feature_scale=1).to(dev).train(); a.drop_out.p = 0.0
oa = torch.optim.Adam(a.parameters(), lr=1e-3, fused=True, capturable=True)
x, t = torch.randn(4, 1, 128, 128, device=dev), torch.rand(4, 4, 128, 128, device=dev)
step = GraphedTrainStep(a, oa, crit, x, t, capture_optimizer=(os.environ.get("CAPOPT", "1") == "1"))
params = list(a.parameters())
late = 0
for i in range(20):
    step(x, t)
    if os.environ.get("HOW", "stream") == "device":
        torch.cuda.synchronize()
    else:
        torch.cuda.current_stream().synchronize()
    p1 = [p.detach().clone() for p in params]
    g1 = [p.grad.detach().clone() for p in params]
    torch.cuda.synchronize()
    time.sleep(0.03)
    torch.cuda.synchronize()
    dp = [k for k, (p, q) in enumerate(zip(params, p1)) if not torch.equal(p, q)]
    dg = [k for k, (p, q) in enumerate(zip(params, g1)) if not torch.equal(p.grad, q)]
    if dp or dg:
        late += 1
        if late <= 3:
            print("replay %d: %d parameters and %d gradients changed AFTER the stream synchronize returned (first param idx %s, first grad idx %s of %d)"
                  % (i, len(dp), len(dg), dp[:1], dg[:1], len(params)))
print("replays with late writes: %d / 20" % late)
