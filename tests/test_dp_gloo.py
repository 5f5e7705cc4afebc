"""Data-parallel path on CPU: two processes over gloo (world_size 2) drive the GradientAverager exactly as the
engine's backward does (gradient-ready order, node by node) and must end with the average of the two ranks'
gradients in every parameter's ``.grad`` -- set when it was None, accumulated when it was kept (gradient accumulation,
``zero_grad(set_to_none=False)``); plus host-side checks of the ready order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.step_oracle import focal_bce_2d_oracle
from oracle.unet_nested_oracle import UNetNestedOracle
from unet_nested4tiny_objects_keypoints_amd import UNet_Nested, dp

CTOR = dict(in_channels=1, n_classes=4, feature_scale=8)


def _close(a, b):
    return float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-30


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_grads(rank, state):
    """Per-rank gradients from the CPU oracle on the rank's own shard of data (stands in for the HIP backward)."""
    ref = UNetNestedOracle(**CTOR)
    ref.load_state_dict(state)
    ref.train()
    ref.drop_out.eval()
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(2, 1, 16, 16, generator=g)
    t = torch.rand(2, 4, 16, 16, generator=g)
    outs = ref(x)
    (sum(focal_bce_2d_oracle(o, t) for o in outs) / len(outs)).backward()
    return {k: p.grad.clone() for k, p in ref.named_parameters()}


def _worker(rank, world, port, state, bucket_bytes, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(rank)  # different initial weights per rank: the broadcast must fix that
        model = UNet_Nested(**CTOR)
        if rank == 0:
            model.load_state_dict(state)
        avg = dp.make_data_parallel(model, bucket_bytes=bucket_bytes)
        for k, v in model.state_dict().items():
            assert torch.equal(v, state[k]), "broadcast did not replicate %s" % k
        grads = _rank_grads(rank, state)
        named = dict(model.named_parameters())
        names = {id(p): k for k, p in named.items()}
        order = dp.ready_order(model)

        def backward(scale):
            # play the engine's role: write every gradient into its flat-buffer slot, report node by node
            step = 4
            for i in range(0, len(order), step):
                fresh = []
                for p in order[i:i + step]:
                    slot = model._grad_alloc(p)
                    slot.copy_(scale * grads[names[id(p)]])
                    fresh.append((p, slot))
                model._grad_sink(fresh)
            assert model._grad_done() is True  # the averager delivers into p.grad itself

        backward(1.0)
        out = {names[id(p)]: p.grad.clone() for p in order}
        buckets = list(avg.buckets_last_step)
        # gradient accumulation: a second backward without zero_grad must ADD (p.grad used to alias the
        # flat work buffer, so the wgrad kernels overwrote it and autograd then doubled it)
        backward(2.0)
        acc = {names[id(p)]: p.grad.clone() for p in order}
        # zero_grad(set_to_none=False): p.grad stays the same tensor, zeroed in place
        model.zero_grad(set_to_none=False)
        backward(3.0)
        zeroed = {names[id(p)]: p.grad.clone() for p in order}
        # the default zero_grad (set_to_none=True) again
        model.zero_grad()
        backward(0.5)
        fresh2 = {names[id(p)]: p.grad.clone() for p in order}
        # a mixed state: one gradient dropped by hand, the others kept
        order[3].grad = None
        backward(1.0)
        mixed = {names[id(p)]: p.grad.clone() for p in order}
        torch.save({"avg": out, "acc": acc, "zeroed": zeroed, "fresh2": fresh2, "mixed": mixed,
                    "mixed_none": names[id(order[3])], "mine": grads, "buckets": buckets},
                   os.path.join(result_dir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1 << 12, 1 << 30])
def test_gradient_average_world2(tmp_path, bucket_bytes):
    torch.manual_seed(0)
    state = UNetNestedOracle(**CTOR).state_dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, state, bucket_bytes, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    for k in r0["mine"]:
        want = (r0["mine"][k] + r1["mine"][k]) / 2
        assert torch.allclose(r0["avg"][k], want, rtol=1e-6, atol=1e-12), k
        assert torch.equal(r0["avg"][k], r1["avg"][k]), k
        assert _close(r0["acc"][k], 3 * want), k        # 1x then 2x accumulated
        assert _close(r0["zeroed"][k], 3 * want), k     # zeroed in place, then 3x
        assert _close(r0["fresh2"][k], 0.5 * want), k
        f = 1.0 if k == r0["mixed_none"] else 1.5
        assert _close(r0["mixed"][k], f * want), k
    n_buckets = len(r0["buckets"])
    assert n_buckets == (1 if bucket_bytes == 1 << 30 else n_buckets) and n_buckets >= 1
    if bucket_bytes == 1 << 12:
        assert n_buckets > 3  # small buckets: several overlapped all-reduces
    # buckets tile the flat buffer exactly once, in order
    flat = sum(v.numel() for v in r0["mine"].values())
    assert r0["buckets"][0][0] == 0 and r0["buckets"][-1][1] == flat
    assert all(a[1] == b[0] for a, b in zip(r0["buckets"], r0["buckets"][1:]))


@pytest.mark.parametrize("kw", [dict(), dict(depth=5, feature_scale=8), dict(is_deconv=False), dict(is_batchnorm=False),
                                dict(depth=2)])
def test_ready_order_covers_every_parameter_once(kw):
    m = UNet_Nested(**kw)
    order = dp.ready_order(m)
    assert {id(p) for p in order} == {id(p) for p in m.parameters()}
    names = {id(p): k for k, p in m.named_parameters()}
    first = names[id(order[0])]
    last = names[id(order[-1])]
    assert first.startswith("final_%d" % (m.depth - 1))       # heads finish first ...
    assert last.startswith("conv00.conv1")                    # ... the first encoder conv last


# ---------------------------------------------------------------------------------------------------------------------
# bench.py's own multi-process protocol with more than two ranks (the first 8-GPU run must not be the
# first N > 2 run): `python bench.py --gpus 4 --rehearse-cpu` spawns its four ranks itself -- rendezvous on 127.0.0.1,
# broadcast, bucketed all-reduce in the engine's ready order, Adam, max-over-ranks timing, replicas_bit_identical,
# rank 0's JSON line relayed by the parent.  No kernels run (there is no GPU here); the line says so.
def _run_bench(*extra, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rehearse-cpu", "--steps", "3", "--warmup", "1",
                        *extra], capture_output=True, text=True, timeout=600, env=e)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


C4_NET = ["--feature-scale", "0.5", "--depth", "5", "--in-channels", "3", "--n-classes", "5"]   # configs[4]: base 64, D5


@pytest.mark.parametrize("world,extra,grad_bytes", [
    (4, ["--feature-scale", "1"], 8828976),                                                     # configs[2]'s network
    (3, ["--feature-scale", "4", "--depth", "5", "--in-channels", "3", "--n-classes", "5"], None),  # odd world, depth 5
    # the REAL world size of configs[2] and of the data-parallel half of configs[4] (trainer/trainer.py:281-287,336-340
    # runs whatever torch.cuda.device_count() gives: 8 on the target node), the real networks, the real gradient volumes:
    # 8.8 MB in ~0.74 MB buckets and 145 MB (36 167 124 parameters) in ~12 MiB buckets
    (8, ["--feature-scale", "1"], 8828976),
    (8, C4_NET, 36167124 * 4),
], ids=["world4-configs2", "world3-depth5", "world8-configs2", "world8-configs4"])
def test_bench_self_spawn_and_replica_check(world, extra, grad_bytes):
    code, line, err = _run_bench("--gpus", str(world), *extra)
    assert code == 0, err
    cfg = line["config"]
    assert cfg["world_size"] == world and cfg["backend"] == "gloo"
    assert cfg["replicas_bit_identical"] is True
    assert line["value"] is None and "REHEARSAL" in line["metric"]      # cannot be mistaken for a measurement
    assert cfg["buckets_tile_the_gradient"] is True and cfg["bucket_ends_on_reported_groups"] is True, cfg
    assert sum(cfg["bucket_bytes_each"]) == cfg["gradient_bytes"]
    if grad_bytes is not None:
        assert cfg["gradient_bytes"] == grad_bytes                        # 2 207 244 parameters x 4 B
        assert 6 <= cfg["grad_allreduce_buckets"] <= 16, cfg              # auto bucket size: about a dozen per step
        assert cfg["bucket_bytes"] == dp.auto_bucket_bytes(grad_bytes)
        # every bucket but the tail reaches the target size, and none is more than one reported group above it
        assert all(b >= cfg["bucket_bytes"] for b in cfg["bucket_bytes_each"][:-1]), cfg


def test_depth5_bucket_boundaries_follow_the_ready_groups():
    """configs[4]'s network (depth 5, base 64: 145 MB of gradients) in a world of one: the flat vector is laid out in
    dp.ready_groups() order, the engine's reports advance a contiguous frontier, and every all-reduce starts where the
    last one ended and ends on a group boundary -- 12 MiB buckets, about a dozen of them."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        model = UNet_Nested(in_channels=3, n_classes=5, feature_scale=0.5, depth=5)
        avg = dp.make_data_parallel(model, always_reduce=True)
        groups = dp.ready_groups(model)
        assert len(groups) == 1 + 3 * 10 + 2 * 5          # heads; conv2, conv1, up per decoder node; conv2, conv1 per encoder node
        assert avg.flat.numel() == 36167124 and avg.bucket_bytes == dp.auto_bucket_bytes(4 * 36167124)
        off = 0
        for grp in groups:                                # flat layout = report order, group by group
            for p in grp:
                assert avg.offset[id(p)] == off
                off += p.numel()
        for grp in groups:
            model._grad_sink([(p, model._grad_alloc(p).fill_(1.0)) for p in grp])
        assert model._grad_done() is True
        spans = avg.buckets_last_step
        edges, acc = set(), 0
        for grp in groups:
            acc += sum(p.numel() for p in grp)
            edges.add(acc)
        assert spans[0][0] == 0 and spans[-1][1] == avg.flat.numel()
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(e in edges for _, e in spans)
        assert 6 <= len(spans) <= 16, spans
        assert all(e - b >= avg.bucket_elems for b, e in spans[:-1])
        assert all(float(p.grad.min()) == 1.0 and float(p.grad.max()) == 1.0 for p in model.parameters())
    finally:
        dist.destroy_process_group()


def test_bench_self_spawn_reports_a_failing_rank():
    """a rank that dies must turn into a non-zero exit of the parent, not into a hang or a silent success"""
    code, line, err = _run_bench("--gpus", "2", "--depth", "9")          # the model constructor refuses depth 9 on every rank
    assert code != 0 and line is None
    assert "rank exit codes" in err


def test_auto_bucket_bytes_targets_a_dozen_buckets():
    assert dp.auto_bucket_bytes(8828976) == -(-8828976 // 12)                 # configs[1]/[2]: 8.8 MB -> 0.74 MB
    assert 11 << 20 < dp.auto_bucket_bytes(36167124 * 4) < 13 << 20           # configs[4]: 145 MB -> 12 MB
    assert dp.auto_bucket_bytes(1000) == 256 << 10 and dp.auto_bucket_bytes(1 << 40) == 64 << 20


def test_frozen_parameters_get_no_gradient_and_hooks_are_refused():
    """delivery must leave p.grad None for requires_grad=False parameters (an optimizer built over
    model.parameters() would otherwise update them) and must not silently skip registered tensor hooks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        model = UNet_Nested(**CTOR)
        frozen = [model.final_1.weight, model.conv00.conv1[0].weight] if hasattr(model.conv00.conv1, "__getitem__") else \
            [model.final_1.weight, getattr(model.conv00.conv1, "0").weight]
        for p in frozen:
            p.requires_grad_(False)
        dp.make_data_parallel(model)
        order = dp.ready_order(model)

        def backward():
            for p in order:
                model._grad_alloc(p).fill_(1.0)
                model._grad_sink([(p, None)])
            model._grad_done()

        backward()
        assert all(p.grad is None for p in frozen)
        assert all(p.grad is not None and float(p.grad.min()) == 1.0 for p in order if p.requires_grad)
        backward()   # accumulation with frozen parameters present
        assert all(p.grad is None for p in frozen)
        assert all(float(p.grad.min()) == 2.0 for p in order if p.requires_grad)
        order[5].register_hook(lambda g: g)
        with pytest.raises(RuntimeError, match="hook"):
            backward()
    finally:
        dist.destroy_process_group()
