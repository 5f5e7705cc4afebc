"""Data-parallel training step on the GPU: two ranks (both on cuda:0 of the 1-GPU test box, gloo transport -- the
driver's multi-GPU runs use RCCL) run the HIP model with the bucketed gradient averager hooked into backward.
After one step both ranks must hold the SAME parameters, equal to a single-process step on the averaged
gradient of the two shards (computed with the CPU oracle)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

CTOR = dict(in_channels=1, n_classes=4, feature_scale=8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, state, out_dir):
    import torch.distributed as dist

    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, dp, train_step
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.manual_seed(50 + rank)              # ranks start from different weights: the broadcast must fix it
        m = UNet_Nested(**CTOR)
        if rank == 0:
            m.load_state_dict(state)
        m = m.to(dev).train()
        m.drop_out.eval()
        avg = dp.make_data_parallel(m, bucket_bytes=16 << 10)
        g = torch.Generator().manual_seed(200 + rank)
        x = torch.randn(2, 1, 32, 32, generator=g).to(dev)
        t = torch.rand(2, 4, 32, 32, generator=g).to(dev)
        opt = torch.optim.SGD(m.parameters(), lr=0.05)
        train_step(m, opt, FocalLoss_BCE_2d(gamma=3, size_average=False), x, t)
        torch.cuda.synchronize()
        torch.save({"params": {k: p.detach().cpu() for k, p in m.named_parameters()},
                    "buckets": len(avg.buckets_last_step)}, os.path.join(out_dir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_matches_averaged_oracle(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    from oracle.step_oracle import focal_bce_2d_oracle
    from oracle.unet_nested_oracle import UNetNestedOracle
    from tests.helpers import is_pre_bn_bias
    torch.manual_seed(7)
    ref = UNetNestedOracle(**CTOR)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    mp.spawn(_worker, args=(2, _free_port(), state, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["buckets"] >= 2
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k   # replicas stay bit-identical
    # expected: SGD step on the mean of the two shards' gradients (BatchNorm statistics per shard, as under the
    # reference's nn.DataParallel).  The per-shard gradients come from the SAME HIP path run single-process here
    # (its parity with the oracle is test_gpu_model's job), so the comparison is exact up to the order of one
    # addition and is not disturbed by ReLU gate flips (tests/helpers.py).
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    grads = []
    for rank in range(2):
        m = UNet_Nested(**CTOR)
        m.load_state_dict(state)
        m = m.cuda().train()
        m.drop_out.eval()
        g = torch.Generator().manual_seed(200 + rank)
        x = torch.randn(2, 1, 32, 32, generator=g).cuda()
        t = torch.rand(2, 4, 32, 32, generator=g).cuda()
        outs = m(x)
        (sum(crit(o, t) for o in outs) / len(outs)).backward()
        grads.append({k: p.grad.cpu() for k, p in m.named_parameters()})
    for k, p0 in state.items():
        if k not in grads[0]:
            continue
        want = p0 - 0.05 * (grads[0][k] + grads[1][k]) / 2
        got = r0["params"][k]
        scale = float((0.05 * (grads[0][k] + grads[1][k]) / 2).abs().max()) + 1e-12
        # one fp32 ulp of an O(1) parameter is 1.2e-7: (p - lr*g) is rounded once on each side
        assert float((got - want).abs().max()) < 1e-5 * scale + 2.5e-7, k
    # and the oracle agrees on the size of the step (coarse: gate flips allowed)
    ref.train()
    ref.drop_out.eval()
    g = torch.Generator().manual_seed(200)
    x = torch.randn(2, 1, 32, 32, generator=g)
    t = torch.rand(2, 4, 32, 32, generator=g)
    outs = ref(x)
    (sum(focal_bce_2d_oracle(o, t) for o in outs) / len(outs)).backward()
    k = "final_3.weight"
    assert float((ref.final_3.weight.grad - grads[0][k]).abs().max()) < 1e-4 * float(grads[0][k].abs().max())
    assert not is_pre_bn_bias(k, CTOR)


def _rccl_worker(rank, port, state, out_dir):
    import torch.distributed as dist

    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, dp, train_step
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)  # nccl = RCCL on ROCm
    try:
        res = {}
        for tag, hook in (("plain", False), ("rccl", True)):
            m = UNet_Nested(**CTOR)
            m.load_state_dict(state)
            m = m.to(dev).train()
            m.drop_out.eval()
            buckets = 0
            if hook:
                avg = dp.make_data_parallel(m, bucket_bytes=16 << 10, always_reduce=True)
            g = torch.Generator().manual_seed(300)
            x = torch.randn(2, 1, 32, 32, generator=g).to(dev)
            t = torch.rand(2, 4, 32, 32, generator=g).to(dev)
            opt = torch.optim.SGD(m.parameters(), lr=0.05)
            train_step(m, opt, FocalLoss_BCE_2d(gamma=3, size_average=False), x, t)
            torch.cuda.synchronize()
            if hook:
                buckets = len(avg.buckets_last_step)
            res[tag] = {k: p.detach().cpu() for k, p in m.named_parameters()}
            res[tag + "_buckets"] = buckets
            # gradient accumulation / zero_grad(set_to_none=False) through the real backward (p.grad must
            # never alias the flat work buffer the next backward writes into)
            crit = FocalLoss_BCE_2d(gamma=3, size_average=False)

            def bwd(scale):
                outs = m(x)
                (scale * sum(crit(o, t) for o in outs) / len(outs)).backward()

            m.zero_grad()
            bwd(1.0)
            bwd(2.0)
            res[tag + "_acc"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
            m.zero_grad(set_to_none=False)
            bwd(1.0)
            torch.cuda.synchronize()
            res[tag + "_zeroed"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        torch.save(res, os.path.join(out_dir, "rccl.pt"))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_collective_path_world_of_one(tmp_path):
    """The bucketed all-reduces on the side stream through the real RCCL backend (a world of one on the 1-GPU box:
    the reduction is the identity, so the step must equal the unhooked step bit for bit)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    from oracle.unet_nested_oracle import UNetNestedOracle
    torch.manual_seed(9)
    state = {k: v.clone() for k, v in UNetNestedOracle(**CTOR).state_dict().items()}
    mp.spawn(_rccl_worker, args=(_free_port(), state, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "rccl.pt")
    assert r["rccl_buckets"] >= 2
    for k in r["plain"]:
        assert torch.equal(r["plain"][k], r["rccl"][k]), k
        assert torch.equal(r["plain_acc"][k], r["rccl_acc"][k]), k
        assert torch.equal(r["plain_zeroed"][k], r["rccl_zeroed"][k]), k
    k = "final_3.weight"
    assert float((r["rccl_acc"][k] - 3 * r["rccl_zeroed"][k]).abs().max()) < 1e-5 * float(r["rccl_acc"][k].abs().max())


def test_reserved_cus_knob_changes_grids_not_results():
    """unetpp_set_reserved_cus (ABI v8): with 64 of the CUs left to a concurrent collective the
    persistent grids and weight-gradient splits shrink; a train step gives the same outputs and gradients up to the
    partition of fp32 sums (BatchNorm partial rows per workgroup, weight-gradient slabs)."""
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    assert lib.unetpp_set_reserved_cus(-1) == 0
    torch.manual_seed(5)
    m = UNet_Nested(in_channels=1, n_classes=4, feature_scale=2).to(dev).train()
    m.drop_out.eval()
    x, t = torch.randn(4, 1, 128, 128, device=dev), torch.rand(4, 4, 128, 128, device=dev)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)

    def run():
        m.zero_grad()
        outs = m(x)
        (sum(crit(o, t) for o in outs) / len(outs)).backward()
        return [o.detach().clone() for o in outs], {k: p.grad.clone() for k, p in m.named_parameters()}

    o0, g0 = run()
    try:
        assert lib.unetpp_set_reserved_cus(64) == 64
        assert lib.unetpp_set_reserved_cus(100000) >= 64      # clamped to CUs - 8
        assert lib.unetpp_set_reserved_cus(64) == 64
        o1, g1 = run()
    finally:
        assert lib.unetpp_set_reserved_cus(0) == 0
    for a, b in zip(o0, o1):
        assert float((a - b).abs().max()) <= 2e-6
    from tests.helpers import is_pre_bn_bias
    for k in g0:
        if is_pre_bn_bias(k, {}):   # analytically zero (the batch mean removes them): rounding noise on both sides
            continue
        scale = float(g0[k].abs().max())
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-5 * scale + 1e-12, k
