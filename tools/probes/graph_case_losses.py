"""Prints the per-iteration losses of tests/test_gpu_graph.py::test_graphed_step_is_bit_identical_to_eager
[bf16-bilinear-sgd-eager] for the graph and the eager runner.  GPUTEST_r04.json recorded 752.95263671875 (graph) against
750.267578125 (eager) at the iteration that failed: the values printed here (bit-identical between the two runners on a
box where the test passes) say WHICH of the two was the wrong one on the driver's box, and at which iteration."""
import copy
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep, UNet_Nested, train_step  # noqa: E402

dev = torch.device("cuda:0")
ctor = dict(in_channels=1, n_classes=4, feature_scale=2, is_deconv=False)
torch.manual_seed(81)
a = UNet_Nested(**ctor).to(dev).train()
a.set_activation_dtype(torch.bfloat16)
a.drop_out.p = 0.0
b = copy.deepcopy(a)
c = copy.deepcopy(a)
crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
mk = lambda m: torch.optim.SGD(m.parameters(), lr=2e-3, momentum=0.9)   # noqa: E731
oa, ob, oc = mk(a), mk(b), mk(c)
g = torch.Generator().manual_seed(5)
xs = [torch.randn(2, 1, 64, 64, generator=g).to(dev) for _ in range(4)]
ts = [torch.rand(2, 4, 64, 64, generator=g).to(dev) for _ in range(4)]
step = GraphedTrainStep(a, oa, crit, xs[0], ts[0], capture_optimizer=False)
for i, (x, t) in enumerate(zip(xs, ts)):
    _, lg = step(x, t)
    _, le = train_step(b, ob, crit, x, t)
    print("iteration %d: graph %r eager %r" % (i + 1, float(lg), float(le.detach())))
# what would the SECOND iteration's loss be if it saw the FIRST iteration's batch (stale static inputs), or stale targets?
train_step(c, oc, crit, xs[0], ts[0])
for name, x, t in (("x1,t1 (both stale)", xs[0], ts[0]), ("x2,t1 (stale target)", xs[1], ts[0]), ("x1,t2 (stale input)", xs[0], ts[1])):
    d = copy.deepcopy(c)
    d.drop_out.p = 0.0
    _, l = train_step(d, mk(d), crit, x, t)
    print("iteration 2 with %s: %r" % (name, float(l.detach())))
