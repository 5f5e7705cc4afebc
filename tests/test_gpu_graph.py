"""The whole training step replayed from a HIP graph (graph.GraphedTrainStep, EXPERIMENTAL) against the eager step
(step.train_step, the reference's loop body trainer/trainer.py:114-136): same kernels, same order, same arguments --
bit-identical parameters, losses and BatchNorm buffers step after step; dropout varies from replay to replay through the
device-side seed word (unetpp_head_fwd / unetpp_head_bwd ``seed_dev``, ABI v8).

Self-comparison (no oracle): tests/conftest.py runs this file LAST, behind every parity test."""
import copy

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              # a gradient accumulator bound to another stream than its producer puts event hops into a capture
              pytest.mark.filterwarnings("error:.*AccumulateGrad node's stream does not match.*")]

BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _make(dev, ctor, bf16, p_drop):
    from unet_nested4tiny_objects_keypoints_amd import UNet_Nested
    torch.manual_seed(81)
    m = UNet_Nested(**ctor).to(dev).train()
    if bf16:
        m.set_activation_dtype(BF)
    m.drop_out.p = p_drop
    return m


def _opt(kind, params):
    if kind == "sgd":
        return torch.optim.SGD(params, lr=2e-3, momentum=0.9)
    if kind == "adam-capturable":
        return torch.optim.Adam(params, lr=1e-3, fused=True, capturable=True)
    return torch.optim.Adam(params, lr=1e-3, fused=True)


@pytest.mark.parametrize("ctor,bf16,opt_kind,capture_opt", [
    (dict(in_channels=1, n_classes=4, feature_scale=4), False, "sgd", True),
    (dict(in_channels=1, n_classes=4, feature_scale=4), False, "adam-capturable", True),
    (dict(in_channels=1, n_classes=4, feature_scale=4), False, "adam", False),                 # eager optimizer on static grads
    (dict(in_channels=3, n_classes=5, feature_scale=4, depth=5), True, "adam-capturable", True),  # configs[4] topology, bf16
    (dict(in_channels=1, n_classes=4, feature_scale=2, is_deconv=False), True, "sgd", False),
], ids=["f32-sgd", "f32-adam-captured", "f32-adam-eager", "bf16-d5-adam-captured", "bf16-bilinear-sgd-eager"])
def test_graphed_step_is_bit_identical_to_eager(dev, ctor, bf16, opt_kind, capture_opt):
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep, train_step
    a = _make(dev, ctor, bf16, 0.0)                      # dropout off: the two paths draw different seeds by design
    b = copy.deepcopy(a)
    before = {k: v.clone() for k, v in a.state_dict().items()}
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    oa, ob = _opt(opt_kind, a.parameters()), _opt(opt_kind, b.parameters())
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(2, ctor["in_channels"], 64, 64, generator=g).to(dev) for _ in range(4)]
    ts = [torch.rand(2, ctor["n_classes"], 64, 64, generator=g).to(dev) for _ in range(4)]
    step = GraphedTrainStep(a, oa, crit, xs[0], ts[0], capture_optimizer=capture_opt, check_topology=True)
    # what was captured is the plain chain graph.py assumes (hipGraphGetNodes / hipGraphGetEdges): kernels only, one root,
    # one leaf, nobody with two successors or two predecessors
    topo = step.topology
    assert topo["chain"] and set(topo["kinds"]) == {"kernel"} and topo["nodes"] > 100, topo
    # the warm-up steps left no trace: parameters and buffers are the initial ones
    for k, v in a.state_dict().items():
        assert torch.equal(v, before[k]), k
    for x, t in zip(xs, ts):
        outs_g, loss_g = step(x, t)
        outs_e, loss_e = train_step(b, ob, crit, x, t)
        assert float(loss_g) == float(loss_e)
        for p, q in zip(outs_g, outs_e):
            assert torch.equal(p, q)
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.equal(p, q), k
            assert p.grad is not None and torch.equal(p.grad, q.grad), k
        for (k, p), (_, q) in zip(a.named_buffers(), b.named_buffers()):
            assert torch.equal(p, q), k


class _IntStepAdamW(torch.optim.Optimizer):
    """An optimizer in the style of the reference's tools/optimizers/adamw.py: ``state['step']`` is a Python int, the
    moments are created on the first step, the update goes through ``p.data`` (test-local; not the reference's code)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self):
        for grp in self.param_groups:
            b1, b2 = grp["betas"]
            for p in grp["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"], st["m"], st["v"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                st["m"].mul_(b1).add_(p.grad, alpha=1 - b1)
                st["v"].mul_(b2).addcmul_(p.grad, p.grad, value=1 - b2)
                size = grp["lr"] * (1 - b2 ** st["step"]) ** 0.5 / (1 - b1 ** st["step"])
                p.data.mul_(1 - grp["lr"] * grp["weight_decay"])
                p.data.addcdiv_(st["m"], st["v"].sqrt().add_(grp["eps"]), value=-size)


def test_warmup_leaves_no_trace_in_an_int_step_optimizer(dev):
    """the warm-up steps must not advance an optimizer that counts its steps in a Python int (the reference's
    tools/optimizers/adamw.py:62,76, adabound.py:78,92): state created by the warm-up is removed, existing state is put
    back by value -- the first replay is step 1 (or step k+1 of a resumed run) exactly as train_step would run it."""
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep, train_step
    ctor = dict(in_channels=1, n_classes=4, feature_scale=4)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(2, 1, 64, 64, generator=g).to(dev) for _ in range(4)]
    ts = [torch.rand(2, 4, 64, 64, generator=g).to(dev) for _ in range(4)]
    for pre_steps in (0, 2):      # fresh optimizer / optimizer that already holds state
        a = _make(dev, ctor, False, 0.0)
        b = copy.deepcopy(a)
        oa, ob = _IntStepAdamW(a.parameters()), _IntStepAdamW(b.parameters())
        for i in range(pre_steps):
            train_step(a, oa, crit, xs[i], ts[i])
            train_step(b, ob, crit, xs[i], ts[i])
        step = GraphedTrainStep(a, oa, crit, xs[0], ts[0], capture_optimizer=False)
        assert len(oa.state) == len(ob.state)
        for p in a.parameters():
            if p in oa.state:
                assert oa.state[p]["step"] == pre_steps
        for x, t in zip(xs[pre_steps:], ts[pre_steps:]):
            _, loss_g = step(x, t)
            _, loss_e = train_step(b, ob, crit, x, t)
            assert float(loss_g) == float(loss_e)
            for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
                assert torch.equal(p, q), k
                assert oa.state[p]["step"] == ob.state[q]["step"]


def test_graphed_step_dropout_varies_and_trains(dev):
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep
    ctor = dict(in_channels=1, n_classes=4, feature_scale=4)
    m = _make(dev, ctor, False, 0.4)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)         # lr 0: only the dropout mask differs between replays
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    x, t = torch.randn(2, 1, 64, 64, device=dev), torch.rand(2, 4, 64, 64, device=dev)
    step = GraphedTrainStep(m, opt, crit, x, t, capture_optimizer=True)
    o1 = [o.clone() for o in step(x, t)[0]]
    g1 = m.final_1.weight.grad.clone()
    o2 = [o.clone() for o in step(x, t)[0]]
    assert not torch.equal(o1[0], o2[0]) and not torch.equal(g1, m.final_1.weight.grad)
    assert getattr(m, "_dropout_seed_dev", None) is None     # eager passes draw their own seeds again
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x), m(x)
    assert all(torch.equal(p, q) for p, q in zip(e1, e2))
    m.train()
    # and it trains: a real learning rate, forty replays
    m2 = _make(dev, ctor, False, 0.4)
    opt2 = torch.optim.Adam(m2.parameters(), lr=2e-3, fused=True, capturable=True)
    step2 = GraphedTrainStep(m2, opt2, crit, x, t, capture_optimizer=True)
    losses = [float(step2(x, t)[1]) for _ in range(40)]
    assert losses[-1] < 0.8 * losses[0], losses


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_seed_dev_adds_to_the_seed(dev, bf16):
    """unetpp_head_fwd / unetpp_head_bwd: seed + *seed_dev, bit for bit the launch with the sum passed by value -- in the
    forward AND in the backward kernel (a captured step regenerates the mask in backward from the same word)."""
    from unet_nested4tiny_objects_keypoints_amd import ops
    torch.manual_seed(3)
    b, h, w, c, ncls = 2, 16, 24, 32, 4
    x = torch.randn(b, h, w, c, device=dev).relu()
    if bf16:
        x = x.to(BF)
    wt, bias = torch.randn(ncls, c, device=dev) * 0.2, torch.randn(ncls, device=dev) * 0.1
    word = torch.tensor([987654321], dtype=torch.int64, device=dev)
    out_a, out_b, out_c = (torch.empty(b, ncls, h, w, device=dev) for _ in range(3))
    ops.head_fwd(x, wt, bias, 0.4, 1000 + 987654321, None, out_a)
    ops.head_fwd(x, wt, bias, 0.4, 1000, None, out_b, seed_dev=word)
    ops.head_fwd(x, wt, bias, 0.4, 1000, None, out_c)
    assert torch.equal(out_a, out_b) and not torch.equal(out_a, out_c)
    d_out = torch.randn(b, ncls, h, w, device=dev)
    dx_a, dx_b = torch.empty_like(x), torch.empty_like(x)
    dw_a, db_a = ops.head_bwd(d_out, out_a, x, wt, 0.4, 1000 + 987654321, None, dx_a, False)
    dw_b, db_b = ops.head_bwd(d_out, out_a, x, wt, 0.4, 1000, None, dx_b, False, seed_dev=word)
    assert torch.equal(dx_a, dx_b) and torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)


def test_graphed_step_refuses_what_it_cannot_capture(dev):
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep
    ctor = dict(in_channels=1, n_classes=4, feature_scale=8)
    m = _make(dev, ctor, False, 0.0)
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    x, t = torch.randn(1, 1, 32, 32, device=dev), torch.rand(1, 4, 32, 32, device=dev)
    with pytest.raises(RuntimeError, match="GPU"):
        GraphedTrainStep(m, opt, crit, x.cpu(), t.cpu())
    m.eval()
    with pytest.raises(RuntimeError, match="TRAINING"):
        GraphedTrainStep(m, opt, crit, x, t)
    m.train()
    step = GraphedTrainStep(m, opt, crit, x, t)
    with pytest.raises(ValueError, match="captured for inputs"):
        step(torch.randn(2, 1, 32, 32, device=dev), t)
    m.eval()
    with pytest.raises(RuntimeError, match="eval"):
        step(x, t)
