"""No kernel of the path reads memory that this pass did not write.

Workspaces (BatchNorm partial rows, weight-gradient slabs, head partials), activations and gradient buffers are
``torch.empty``: whatever the allocator hands out.  Eager, the caching allocator makes that "the bytes of an earlier
tensor"; inside a captured HIP graph (graph.GraphedTrainStep, serving.GraphedForward) it is the graph's private pool, i.e.
the previous REPLAY's bytes -- a kernel that sums a row no workgroup wrote, or a border the staging did not fill, gives
plausible-looking and run-dependent numbers there (the class of failure of GPUTEST_r04).  The test runs the reference's
step body (trainer/trainer.py:114-136) twice from the same state: once as it is, once with every ``torch.empty`` /
``torch.empty_like`` of the package POISONED (NaN for floating storage, 0xFF.. for integers; a second run poisons with a
large finite value, which also survives compare-and-select masking of NaN), and a third and fourth time with every
tensor of the step -- workspaces, activations, gradients, parameters, buffers, inputs, targets, masks -- placed between
two GUARD BANDS of the same filling, so that a read before the first or behind the last element of any tensor lands in
poison instead of in a neighbouring allocation -- outputs, loss, BatchNorm buffers and every gradient must be
bit-identical and finite.  Self-comparison, not parity: it runs after the oracle / golden tests.
"""
import contextlib
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


GUARD_BYTES = 8192   # guard band on either side of a tensor (the widest halo row of the path is 34 pixels x 256 B)


def _fill(t, value):
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(value)
        elif t.dtype == torch.uint8:
            t.fill_(0xFF)
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(-1)
    return t


def guarded(t, value=float("nan"), real_empty=None):
    """A copy of the GPU tensor `t` that sits between two guard bands filled with `value` (same shape, dtype, contiguous,
    256-byte aligned): a kernel that reads before the first or past the last element of the tensor reads the guard."""
    real_empty = real_empty or torch.empty
    g = GUARD_BYTES // t.element_size()
    n = t.numel()
    store = _fill(real_empty(n + 2 * g, dtype=t.dtype, device=t.device), value)
    out = store[g:g + n].view(t.shape)
    out.copy_(t)
    return out


@contextlib.contextmanager
def poisoned_empty(value, guard=False):
    """Every tensor the Python layer allocates uninitialised comes back filled: float('nan') or a large finite value.
    guard=True: it also sits between two guard bands of the same filling (out-of-bounds reads land in them)."""
    real_empty, real_like = torch.empty, torch.empty_like

    def make(shape, dtype, device):
        n = 1
        for d in shape:
            n *= int(d)
        g = GUARD_BYTES // torch.tensor([], dtype=dtype).element_size()
        store = _fill(real_empty(n + 2 * g, dtype=dtype, device=device), value)
        return store[g:g + n].view(tuple(shape))

    def empty(*a, **k):
        t = real_empty(*a, **k)
        if guard and t.is_cuda and t.numel() and t.is_contiguous():
            return make(t.shape, t.dtype, t.device)
        return _fill(t, value)

    def empty_like(src, *a, **k):
        t = real_like(src, *a, **k)
        if guard and t.is_cuda and t.numel() and t.is_contiguous():
            return make(t.shape, t.dtype, t.device)
        return _fill(t, value)

    torch.empty, torch.empty_like = empty, empty_like
    try:
        yield
    finally:
        torch.empty, torch.empty_like = real_empty, real_like


def _state(model, outs, loss):
    d = {"loss": loss.detach().clone()}
    for i, o in enumerate(outs):
        d["out%d" % i] = o.detach().clone()
    for k, p in model.named_parameters():
        d["grad/" + k] = p.grad.detach().clone()
    for k, b in model.named_buffers():
        d["buf/" + k] = b.detach().clone()
    return d


CASES = [
    # (id, constructor, bf16, batch, H, W, model class)
    ("f32-fs4-64", dict(in_channels=1, n_classes=4, feature_scale=4), False, 2, 64, 64),
    ("f32-fs1-96x160", dict(in_channels=1, n_classes=4, feature_scale=1), False, 3, 96, 160),
    ("f32-bilinear-nobn", dict(in_channels=3, n_classes=5, feature_scale=4, is_deconv=False, is_batchnorm=False), False, 2, 48, 80),
    ("f32-d5", dict(in_channels=3, n_classes=5, feature_scale=4, depth=5), False, 2, 64, 96),
    ("f32-d2", dict(in_channels=1, n_classes=4, feature_scale=4, depth=2), False, 4, 64, 64),
    ("bf16-bilinear-fs2-64", dict(in_channels=1, n_classes=4, feature_scale=2, is_deconv=False), True, 2, 64, 64),  # GPUTEST_r04's case
    ("bf16-fs1-128x96", dict(in_channels=1, n_classes=4, feature_scale=1), True, 3, 128, 96),
    ("bf16-d5-base64", dict(in_channels=3, n_classes=5, feature_scale=1, depth=5), True, 1, 96, 48),
    ("bf16-nobn", dict(in_channels=1, n_classes=4, feature_scale=2, is_batchnorm=False), True, 2, 64, 96),
]


@pytest.mark.parametrize("name,ctor,bf16,b,h,w", CASES, ids=[c[0] for c in CASES])
def test_no_launch_reads_unwritten_memory(dev, name, ctor, bf16, b, h, w):
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, train_step
    torch.manual_seed(17)
    base = UNet_Nested(**ctor).to(dev).train()
    if bf16:
        base.set_activation_dtype(BF)
    base.drop_out.p = 0.4
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(b, ctor["in_channels"], h, w, generator=g).to(dev)
    t = torch.rand(b, ctor["n_classes"], h, w, generator=g).to(dev)
    masks = [(torch.rand(b, h, w, base.filters[0], generator=g) >= 0.4).to(torch.uint8).to(dev)   # NHWC keep-masks
             for _ in range(base.depth - 1)]

    def run(poison, guard=False):
        m = copy.deepcopy(base)
        xx, tt, mm = x, t, masks
        if guard:   # parameters, buffers, inputs, targets and masks between guard bands as well
            with torch.no_grad():
                for p in list(m.parameters()) + list(m.buffers()):
                    p.data = guarded(p.data, poison)
            xx, tt, mm = guarded(x, poison), guarded(t, poison), [guarded(k, poison) for k in masks]
        m.dropout_masks = mm                  # explicit masks: the runs draw no seeds
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        ctx = contextlib.nullcontext() if poison is None else poisoned_empty(poison, guard)
        with ctx:
            train_step(m, opt, crit, xx, tt)    # first pass: records the weight-image jobs
            outs, loss = train_step(m, opt, crit, xx, tt)
        torch.cuda.synchronize()
        return _state(m, outs, loss)

    ref = run(None)
    for k, v in ref.items():
        assert torch.isfinite(v.float()).all(), k
    for poison, guard in ((float("nan"), False), (3.0e30, False), (float("nan"), True), (3.0e30, True)):
        got = run(poison, guard)
        bad = [k for k in ref if not torch.equal(ref[k], got[k])]
        assert not bad, "poison %r (guard bands: %s) changes %d tensors, e.g. %s" % (poison, guard, len(bad), bad[:6])


def test_eval_forward_reads_no_unwritten_memory(dev):
    from unet_nested4tiny_objects_keypoints_amd import UNet, UNet_Nested
    torch.manual_seed(3)
    for make, bf16, shape in ((lambda: UNet_Nested(in_channels=1, n_classes=4, feature_scale=2), False, (2, 1, 64, 96)),
                              (lambda: UNet_Nested(in_channels=1, n_classes=4, feature_scale=2), True, (2, 1, 64, 96)),
                              (lambda: UNet(n_classes=5, n_channels=3), False, (1, 3, 40, 56))):
        m = make().to(dev).eval()
        if bf16:
            m.set_activation_dtype(BF)
        x = torch.randn(*shape, device=dev)
        with torch.no_grad():
            ref = m(x)
            ref = ref if isinstance(ref, tuple) else (ref,)
            for poison, guard in ((float("nan"), False), (3.0e30, False), (float("nan"), True)):
                with poisoned_empty(poison, guard):
                    got = m(guarded(x, poison) if guard else x)
                got = got if isinstance(got, tuple) else (got,)
                for p, q in zip(ref, got):
                    assert torch.equal(p, q)


@pytest.mark.parametrize("name,ctor,bf16", [
    ("bf16-bilinear-fs2", dict(in_channels=1, n_classes=4, feature_scale=2, is_deconv=False), True),   # GPUTEST_r04's network
    ("bf16-deconv-fs2", dict(in_channels=1, n_classes=4, feature_scale=2), True),                       # dgrad into 32 channels: one column tile
    ("f32-fs2", dict(in_channels=1, n_classes=4, feature_scale=2), False),
], ids=lambda v: v if isinstance(v, str) else None)
def test_the_step_is_reproducible_run_to_run(dev, name, ctor, bf16):
    """Every tensor of a training step -- all activations kept for backward, BatchNorm coefficients, outputs, loss,
    every gradient -- is bit-identical from run to run, back to back and from an idle device (no floating-point
    atomics on the path; every cross-workgroup sum has a fixed order).  This is the comparison that found the cause of
    GPUTEST_r04: one 256-pixel x 32-column unit of the bilinear path's 1x1 convolution differed in ~1 of 100 steps
    (profiles/r5/gputest_r04_root_cause.txt).  Twelve runs per regime: a guard, not a stress loop."""
    import time

    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested
    torch.manual_seed(81)
    m = UNet_Nested(**ctor).to(dev).train()
    if bf16:
        m.set_activation_dtype(BF)
    m.drop_out.p = 0.0
    m._debug_keep_saved = True
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 64, 64, generator=g).to(dev)
    t = torch.rand(2, 4, 64, 64, generator=g).to(dev)
    running = [b.clone() for b in m.buffers()]

    def one_step():
        for p in m.parameters():
            p.grad = None
        outs = m(x)
        loss = sum(crit(o, t) for o in outs) / len(outs)
        s = m._debug_saved
        got = {}
        for key, r in s.pairs.items():
            for nm in ("y1", "a1", "y2", "out", "pooled", "pool_idx"):
                v = getattr(r, nm)
                if v is not None:
                    got["X%d%d.%s" % (key + (nm,))] = v.clone()
        for key, u in s.ups.items():
            for nm in ("interp", "up"):
                v = getattr(u, nm)
                if v is not None:
                    got["up%d%d.%s" % (key + (nm,))] = v.clone()
        for i, o in enumerate(outs):
            got["out%d" % i] = o.detach().clone()
        got["loss"] = loss.detach().clone()
        loss.backward()
        for k, p in m.named_parameters():
            got["grad/" + k] = p.grad.clone()
        with torch.no_grad():
            for b, v in zip(m.buffers(), running):
                b.copy_(v)
        torch.cuda.synchronize()
        return got

    def bits(v):
        return v.view(torch.int16) if v.dtype == BF else v

    ref = one_step()
    for i in range(24):
        if i >= 12:
            time.sleep(0.01)    # idle start
        got = one_step()
        bad = [k for k in ref if not torch.equal(bits(ref[k]), bits(got[k]))]
        assert not bad, "run %d: %d tensors differ, first %s" % (i, len(bad), bad[:4])


def test_classic_unet_step_reads_no_unwritten_memory_and_is_reproducible(dev):
    """The same two properties for the classic UNet (models/unet.py:8-117; engine_unet.py schedules the same kernels,
    plus floor pooling / zero padding of odd sizes): poisoned + guard-banded allocations change nothing, and eight runs
    of the step give bit-identical outputs, loss and gradients."""
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet
    torch.manual_seed(29)
    base = UNet(n_classes=5, n_channels=3, widths=(16, 32, 64, 128, 128)).to(dev).train()
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, 40, 56, generator=g).to(dev)       # not a multiple of 16: the reference's padding path
    t = torch.rand(2, 5, 40, 56, generator=g).to(dev)

    def run(poison=None, guard=False):
        m = copy.deepcopy(base)
        xx, tt = x, t
        if guard:
            with torch.no_grad():
                for p in list(m.parameters()) + list(m.buffers()):
                    p.data = guarded(p.data, poison)
            xx, tt = guarded(x, poison), guarded(t, poison)
        ctx = contextlib.nullcontext() if poison is None else poisoned_empty(poison, guard)
        with ctx:
            for _ in range(2):
                for p in m.parameters():
                    p.grad = None
                out = m(xx)
                loss = crit(out, tt)
                loss.backward()
        torch.cuda.synchronize()
        d = {"out": out.detach().clone(), "loss": loss.detach().clone()}
        d.update({"grad/" + k: p.grad.clone() for k, p in m.named_parameters()})
        return d

    ref = run()
    for k, v in ref.items():
        assert torch.isfinite(v).all(), k
    for poison, guard in ((float("nan"), False), (3.0e30, True), (float("nan"), True)):
        got = run(poison, guard)
        bad = [k for k in ref if not torch.equal(ref[k], got[k])]
        assert not bad, "poison %r (guard bands: %s): %s" % (poison, guard, bad[:6])
    for i in range(8):
        if i % 2:
            torch.cuda.synchronize()
        got = run()
        bad = [k for k in ref if not torch.equal(ref[k], got[k])]
        assert not bad, "run %d: %s" % (i, bad[:6])
