#!/usr/bin/env python3
"""Benchmark of the hot path: images/sec of one full training step of UNet_Nested on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
UNet_Nested(in_channels=1, n_classes=4, feature_scale=1) = depth 4, base width 32, 256x256, batch 32 PER GPU,
fp32, synthetic data (seed 0: x ~ N(0,1), target ~ U(0,1)), weights from the module's own kaiming init.
One "step" is the reference's loop body (trainer/trainer.py:114-136): zero_grad + forward (dropout active)
+ FocalLoss_BCE_2d on the 3 heads + mean + backward + Adam step (+ gradient all-reduce when N > 1).
Weak scaling: the per-GPU batch is fixed; `value` is global images/sec.

`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself: the parent never touches the
GPU, spawns N children of this script with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, relays rank 0's JSON line and exits
non-zero if any rank failed (the reference's multi-GPU entry is in-process nn.DataParallel: one command, N GPUs,
trainer/trainer.py:281-287,336-340).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      the dominant kernel (most device time) timed live with HIP events on the launch stream over the
                timed steps.  `achieved` = multiply-adds the matrix pipe EXECUTES (x2) / summed launch duration,
                `frac` = achieved / dense fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md) -- a fraction of a
                hardware limit, <= 1.  For the Winograd kernels the executed work is 16/36 of the algorithmic
                2*9*Cin*Cout per pixel; the algorithmic rate is reported beside it (`achieved_algorithmic`).
                `traffic` (HBM bytes per launch from PMC counters) cannot be collected from inside the process: the
                default single-GPU run of the headline workload therefore starts, BEFORE it touches the GPU itself,
                three short children of this script per configuration under `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE,
                SQ/GRBM: separate passes, 3 train steps each; ~3 s per pass) and summarises them with tools/pmc_traffic.py (`traffic_source`: "live");
                without rocprofv3, for other workloads or with --no-live-pmc the numbers are read from
                profiles/pmc_hbm_traffic_latest.json when that file was made from THIS build (build hash).  Beside
                it `traffic_algorithmic` (every operand read/written once) and their ratio.
  roofline_x00  the X_0,0 conv block forward (1.2457 GFLOP and 42.2 MB algorithmic per image):
                frac = max(compute floor, HBM floor) / measured block time.
  other_configs the default run (the headline workload) then times BASELINE configs[3] (512x512, bf16 storage, batch 8) and
                configs[4] (depth 5, base 64, 3 -> 5 channels, 384x384, bf16 storage, batch 4) for >= 20 steps each with
                the same bracket, and attaches value / ms_per_step / roofline / top kernels per configuration.
  cpu_baseline  the CPU oracle (oracle/, a PyTorch restatement of the reference proven equal to it on golden
                fixtures) timed on this host's cores on a bounded sample of the same workload (batch 4).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_TBS = 8.0
X00_GFLOP_PER_IMG = 1.2457
X00_MB_PER_IMG = 42.2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (configs[1]: 32)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--feature-scale", type=float, default=1)
    ap.add_argument("--depth", type=int, default=4, help="resolution levels (reference: 4; configs[4]: 5)")
    ap.add_argument("--in-channels", type=int, default=1)
    ap.add_argument("--n-classes", type=int, default=4)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="activation storage: f32 (configs[1], reference numerics) or bf16 with fp32 accumulation (configs[3]/[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-oracle steps (BASELINE.md section 4: >= 3)")
    ap.add_argument("--prewarm", type=int, default=10, help="untimed steps before the W warm-up steps (see main)")
    ap.add_argument("--no-launch-timing", action="store_true", help="skip per-launch HIP events (roofline = null)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the bf16 configurations (BASELINE configs[3], configs[4]) the default run times after the "
                         "headline and attaches under `other_configs`")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the rocprofv3 --pmc passes the default single-GPU headline run starts before its own "
                         "measurement (roofline.traffic / mfma_busy_pmc then come from profiles/pmc_hbm_traffic_latest.json)")
    ap.add_argument("--other-steps", type=int, default=30, help="timed steps of each `other_configs` entry (>= 20)")
    ap.add_argument("--graphed", action="store_true",
                    help="EXPERIMENTAL: replay the step from a HIP graph (GraphedTrainStep; single GPU) -- for the configuration "
                         "named on the command line and for `other_configs` (launch events then move behind the timed "
                         "region).  No default line uses it")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="multi-process plumbing only, on the CPU over gloo, with synthetic gradients and NO kernels: "
                         "self-spawn, rendezvous, broadcast, bucketed all-reduce, optimizer, max-over-ranks timing, "
                         "replica check (tests/test_dp_gloo.py); the line it prints is labelled a rehearsal")
    return ap.parse_args()


def usable_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota, not the host's core count
    (a 1-GPU box exposes 256 logical CPUs but grants a share of them; 256 threads on a 16-CPU quota thrash)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(n, int(os.environ.get("UNETPP_CPU_THREADS", "16")))


def config_key(args):
    """names a benchmark configuration in profiles/pmc_hbm_traffic_latest.json (tools/pmc_traffic.py)"""
    fs = int(args.feature_scale) if float(args.feature_scale).is_integer() else args.feature_scale
    return "%s_d%d_fs%s_s%d_b%d_i%d_c%d" % (args.dtype, args.depth, fs, args.size, args.batch, args.in_channels, args.n_classes)


_LIVE_PMC = {}   # config key -> summary made by live_pmc() in this run


def live_pmc(args):
    """PMC passes of THIS run (single GPU, before this process touches the GPU: the children are separate programs
    started under the profiler, `python3 bench.py ...` directly behind `--`).  Three passes as MI355X_MICROARCH.md
    prescribes for the TCC counters (FETCH_SIZE and WRITE_SIZE apart) plus one SQ/GRBM pass for the matrix-pipe
    occupancy, each over 3 train steps of the configuration; tools/pmc_traffic.py applies the guide's unit and gfx950
    corrections.  Any failure (no rocprofv3, a pass that times out or leaves no counter file) returns None and the
    committed summary is used instead -- the measurement itself never depends on the profiler."""
    import contextlib
    import io
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="unetpp_pmc_", dir="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--prewarm", "0",
             "--no-cpu-baseline", "--no-launch-timing", "--no-other-configs", "--no-live-pmc",
             "--dtype", args.dtype, "--size", str(args.size), "--batch", str(args.batch), "--depth", str(args.depth),
             "--feature-scale", str(args.feature_scale), "--in-channels", str(args.in_channels),
             "--n-classes", str(args.n_classes)]
    passes = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"]))
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.time()
    try:
        for name, counters in passes:
            out = os.path.join(tmp, name)
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", out, "-o", "p", "--"] + child
            with open(os.path.join(tmp, name + ".log"), "w") as log:
                # own session: a pass that runs into the timeout is ended as a GROUP (profiler AND the benchmark child
                # under it), so nothing of it is left on the GPU when the measurement proper starts
                proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
                try:
                    rc = proc.wait(timeout=float(os.environ.get("UNETPP_BENCH_PMC_TIMEOUT", "240")))
                except subprocess.TimeoutExpired:
                    import signal
                    try:
                        os.killpg(proc.pid, signal.SIGKILL)
                    except OSError:
                        pass
                    proc.wait()
                    return None
            if rc != 0 or not os.path.exists(os.path.join(out, "p_counter_collection.csv")):
                return None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_traffic
        summary = os.path.join(tmp, "summary.json")
        with contextlib.redirect_stdout(io.StringIO()):
            pmc_traffic.main(os.path.join(tmp, "fetch"), os.path.join(tmp, "write"), summary, config_key(args),
                             os.path.join(tmp, "sq"))
        with open(summary) as f:
            doc = json.load(f)["configs"][config_key(args)]
        doc["seconds"] = round(time.time() - t0, 1)
        return doc
    except Exception as exc:   # the profiler is an extra: never the reason for a missing bench line
        print("live PMC passes failed (%s: %s): using the committed summary" % (type(exc).__name__, exc), file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_counters(kernel_label, build_hash, cfg_key):
    """(HBM bytes per launch, matrix-pipe occupancy, source) of the dominant kernel: from the rocprofv3 --pmc passes this
    run made itself (live_pmc), else from the committed PMC summary of THIS configuration (tools/pmc_traffic.py made it
    from the same three passes of this command).  A summary stamped with another build's hash is refused: it describes
    other kernels."""
    doc, source = _LIVE_PMC.get(cfg_key), "live: rocprofv3 --pmc passes started by this run (3 train steps each)"
    if doc is None:
        path, source = os.path.join(ROOT, "profiles", "pmc_hbm_traffic_latest.json"), "profiles/pmc_hbm_traffic_latest.json (same build hash)"
        try:
            with open(path) as f:
                doc = json.load(f)["configs"][cfg_key]
        except (OSError, ValueError, KeyError):
            return None, None, None
    rows = doc.get("kernels", [])
    if doc.get("build_hash") != build_hash:
        return None, None, None
    stem = kernel_label.rstrip(">")  # "gemm_fast_kernel<9" matches "gemm_fast_kernel<9, 5, 1>"
    hits = [r for r in rows if r["kernel"].startswith(stem)]
    n = sum(r["launches"] for r in hits)
    if n == 0:
        return None, None, None
    traffic = round(sum((r["fetch_bytes_x2_per_launch"] + r["write_bytes_per_launch"]) * r["launches"] for r in hits) / n)
    busy = [(r["sq"]["mfma_busy"] * r["sq"]["avg_us"] * r["sq"]["dispatches"], r["sq"]["avg_us"] * r["sq"]["dispatches"])
            for r in hits if "sq" in r and "mfma_busy" in r["sq"]]
    mfma_busy = round(sum(b for b, _ in busy) / sum(t for _, t in busy), 4) if busy else None   # time-weighted
    return traffic, mfma_busy, source


def cpu_baseline(args, n_cls):
    """The oracle's train step (trainer/trainer.py:114-136 restated) on the host cores, batch 4."""
    import torch

    from oracle.step_oracle import focal_bce_2d_oracle
    from oracle.unet_nested_oracle import UNetNestedOracle
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    fs = int(args.feature_scale) if float(args.feature_scale).is_integer() else args.feature_scale
    model = UNetNestedOracle(in_channels=args.in_channels, n_classes=n_cls, feature_scale=fs, depth=args.depth).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    b = 4
    x = torch.randn(b, args.in_channels, args.size, args.size)
    target = torch.rand(b, n_cls, args.size, args.size)

    def step():
        opt.zero_grad()
        outs = model(x)
        loss = sum(focal_bce_2d_oracle(o, target) for o in outs) / len(outs)
        loss.backward()
        opt.step()

    step()  # warm-up
    t0 = time.perf_counter()
    for _ in range(max(3, args.cpu_steps)):
        step()
    dt = time.perf_counter() - t0
    model.eval()
    with torch.no_grad():
        model(x)
        t1 = time.perf_counter()
        model(x)
        fwd = time.perf_counter() - t1
    return {
        "value": round(b * max(3, args.cpu_steps) / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
        "torch_threads": torch.get_num_threads(),
        "fwd_ms_per_img": round(1e3 * fwd / b, 3),
        "sample": "%d timed train steps (+1 warm-up) of the CPU oracle at batch %d, %dx%d, base width %d, fp32, Adam"
                  % (max(3, args.cpu_steps), b, args.size, args.size, int(32 / args.feature_scale)),
    }


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) and relay rank 0's line.
    Runs before anything in this process touches the GPU (no torch import at all) and never execs."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(codes):
        print("bench.py: rank exit codes %s" % codes, file=sys.stderr)
        raise SystemExit(1)


def replicas_bit_identical(model, dist):
    """every rank must hold bit-identical parameters after the run (DataParallel keeps ONE copy): MAX == MIN over ranks
    of the parameter bits"""
    import torch
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).view(torch.int32)
    hi, lo = flat.clone(), flat.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return bool(torch.equal(hi, lo))


def rehearse_cpu(args):
    """`--rehearse-cpu`: the N-rank protocol of this benchmark without a GPU and without a single kernel -- every rank
    plays the engine's role with per-rank pseudo-gradients written into the averager's flat buffer node by node (the
    order and granularity of engine.backward_impl), then steps Adam.  What it proves: N processes rendezvous, start
    identical, exchange exactly the buckets the real run would, stay bit-identical, and rank 0's line reaches stdout."""
    import torch
    import torch.distributed as dist

    from unet_nested4tiny_objects_keypoints_amd import UNet_Nested, dp
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fs = int(args.feature_scale) if float(args.feature_scale).is_integer() else args.feature_scale
    torch.manual_seed(rank)  # ranks start different: the broadcast has to fix that
    model = UNet_Nested(in_channels=args.in_channels, n_classes=args.n_classes, feature_scale=fs, depth=args.depth)
    averager = dp.make_data_parallel(model)
    groups = dp.ready_groups(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(1000 + rank)

    def step():
        opt.zero_grad()
        for grp in groups:  # the engine reports a convolution's (and its BatchNorm's) gradients together
            fresh = []
            for p in grp:
                slot = model._grad_alloc(p)
                slot.copy_(torch.randn(p.shape, generator=gen))
                fresh.append((p, slot))
            model._grad_sink(fresh)
        assert model._grad_done() is True
        opt.step()

    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    same = replicas_bit_identical(model, dist)
    # the all-reduces of a step must tile the flat gradient vector, each ending where a reported group ends (a bucket
    # never cuts a convolution's gradients in two: the engine reports them together)
    edges, acc = set(), 0
    for grp in groups:
        acc += sum(p.numel() for p in grp)
        edges.add(acc)
    spans = list(averager.buckets_last_step)
    tiled = bool(spans) and spans[0][0] == 0 and spans[-1][1] == averager.flat.numel() and \
        all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    on_edges = all(e in edges for _, e in spans)
    if rank == 0:
        print(json.dumps({
            "metric": "REHEARSAL of the multi-process protocol (no kernels, CPU, gloo) -- not a measurement",
            "value": None, "unit": None, "n_gpus": 0, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * float(t.item()) / max(1, args.steps), 3), "data": "synthetic gradients",
            "config": {"world_size": world, "backend": dist.get_backend(), "gradient_bytes": 4 * averager.flat.numel(),
                       "bucket_bytes": averager.bucket_bytes, "grad_allreduce_buckets": len(averager.buckets_last_step),
                       "bucket_bytes_each": [4 * (e - b) for b, e in spans], "buckets_tile_the_gradient": tiled,
                       "bucket_ends_on_reported_groups": on_edges, "reported_groups": len(groups),
                       "replicas_bit_identical": same}}))
    dist.barrier()
    dist.destroy_process_group()


# BASELINE.json configs[3] / configs[4] as single-GPU geometries (per-GPU batch stated in the workload string): timed by
# the default run after the headline and attached under `other_configs`
OTHER_CONFIGS = (
    ("configs[3]: deep supervision, 512x512, bf16",
     dict(dtype="bf16", size=512, batch=8, feature_scale=1, depth=4, in_channels=1, n_classes=4)),
    ("configs[4]: 3-channel 384x384, 5 key-point maps, depth 5, base 64, bf16",
     dict(dtype="bf16", size=384, batch=4, feature_scale=0.5, depth=5, in_channels=3, n_classes=5)),
)


# Per-GPU batches beside the continuity entries above (BASELINE names no batch for these two configurations): short
# untimed-kernel runs (no launch events) whose value / ms_per_step / peak memory go under `batch_sweep` of the entry
BATCH_SWEEP = {0: (16,), 1: (8, 16)}
HBM_GB = 288.0


def is_headline(args):
    """the default workload (configs[1]); runs with other shapes asked for on the command line stay single-config"""
    return (args.dtype, args.size, args.batch, float(args.feature_scale), args.depth, args.in_channels, args.n_classes) == \
        ("f32", 256, 32, 1.0, 4, 1, 4)


def workload_name(args):
    return ("UNet_Nested(in=%d,n_classes=%d,base=%d,depth=%d) %dx%d train step, batch %d/GPU, "
            "FocalLoss_BCE_2d on %d heads, Adam, dropout p=0.4 active, %s"
            % (args.in_channels, args.n_classes, int(32 / args.feature_scale), args.depth, args.size, args.size,
               args.batch, args.depth - 1,
               "fp32" if args.dtype == "f32" else "bf16 activation storage / fp32 accumulation and parameters"))


def other_entry(name, o, r, world):
    """one `other_configs` element: the same keys as the headline, trimmed to what a reader needs per configuration"""
    roof = r["roofline"]
    if roof is not None:
        roof = {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_algorithmic",
                                      "traffic_over_algorithmic", "mfma_busy_pmc", "traffic_source", "avg_launch_ms", "launches_per_step",
                                      "floor_hbm_ms", "floor_mfma_ms", "instrumented_steps_after_timed_region") if k in roof}
    kern = r["kernels"]
    if kern is not None:  # the five kernels with the most device time
        kern = dict(sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"])[:5])
    return {"config": name, "workload": workload_name(o), "dtype": o.dtype, "value": round(r["value"], 2),
            "peak_mem_gb": r["peak_mem_gb"], "unit": "images/sec", "n_gpus": world, "steps": o.steps, "warmup": o.warmup, "prewarm_steps": o.prewarm,
            "step_runner": r["step_runner"], "ms_per_step": round(r["ms_per_step"], 3), "host_enqueue_ms_median": r["host_enqueue_ms"], "host_enqueue_ms_min": r["host_enqueue_ms_min"], "fwd_ms_per_img": round(r["fwd_ms_per_img"], 4),
            "per_gpu_batch": o.batch, "global_batch": world * o.batch, "roofline": roof, "kernels": kern,
            "replicas_bit_identical": r["replicas_identical"]}


def measure(args, ctx):
    """Times `args.steps` train steps of ONE configuration (after prewarm + warmup untimed ones) under the contract's
    barrier / synchronize bracket and returns rank 0's result pieces (None on the other ranks)."""
    torch, dist, dev = ctx["torch"], ctx["dist"], ctx["dev"]
    world, rank, distributed = ctx["world"], ctx["rank"], ctx["distributed"]
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, dp, ops, train_step

    n_cls = args.n_classes
    fs = int(args.feature_scale) if float(args.feature_scale).is_integer() else args.feature_scale
    torch.cuda.reset_peak_memory_stats(dev)
    torch.manual_seed(0)
    model = UNet_Nested(in_channels=args.in_channels, n_classes=n_cls, feature_scale=fs, depth=args.depth).to(dev).train()
    if args.dtype == "bf16":
        model.set_activation_dtype(torch.bfloat16)
    averager = None
    if distributed:
        averager = dp.make_data_parallel(model)
    elif os.environ.get("UNETPP_BENCH_FORCE_DP") == "1":
        # diagnostic (never set by the driver): the data-parallel machinery in a world of one -- flat gradient buffer,
        # bucketed RCCL all-reduce on the side stream, delivery into p.grad -- to price what it costs beside the kernels
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        averager = dp.make_data_parallel(model, always_reduce=True)
    torch.manual_seed(1000 + rank)  # every rank its own shard of synthetic data
    x = torch.randn(args.batch, args.in_channels, args.size, args.size, device=dev)
    target = torch.rand(args.batch, n_cls, args.size, args.size, device=dev)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    # Opt-in (`--graphed`, single GPU): the same step replayed from a HIP graph (graph.GraphedTrainStep, experimental: the
    # launches, their order and their arguments are train_step's; the dropout seed varies through a device word).  Every
    # default line -- headline and `other_configs` -- comes from the eager train_step.
    graphed = (getattr(args, "graphed", False) and not distributed and averager is None
               and os.environ.get("UNETPP_BENCH_NO_GRAPH") != "1")
    try:  # same Adam, one fused device kernel: 0.13 ms of host time per step instead of 2.6 ms (foreach, 90 tensors)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=graphed)
    except (TypeError, RuntimeError):
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    run_step = train_step
    if graphed:
        from unet_nested4tiny_objects_keypoints_amd import GraphedTrainStep
        replay = GraphedTrainStep(model, opt, crit, x, target, capture_optimizer=True, restore_state=False)
        run_step = lambda m, o, c, xx, tt: replay(xx, tt)   # noqa: E731

    # W untimed warm-up steps as the contract says, after `--prewarm` more untimed steps (printed as `prewarm_steps`):
    # the first ~10 steps of a process run up to 10 % slower (allocator growth, code-object loading, clock ramp),
    # whatever W the caller picks.
    for _ in range(args.prewarm + args.warmup):
        run_step(model, opt, crit, x, target)

    # Per-launch HIP events (for the roofline object) on every 10th timed step only: an event pair around each of the
    # 77 MFMA launches of a step costs the stream ~6 us each (0.9 ms per instrumented step; every 4th step still took
    # 0.4-1.2 % off the headline value, measured against --no-launch-timing).
    timer = None
    if rank == 0 and not args.no_launch_timing:
        timer = ops.LaunchTimer()
    sample_every, sampled_steps = 10, 0
    # Headline: the launch events sit INSIDE the timed region (steps 0, 10, ...), as the roofline contract asks.  The
    # `other_configs` entries take them on `instrumented_steps_after` extra steps right behind their timed region instead
    # (every rank runs those steps: they carry the gradient all-reduce): in a process that has already timed another
    # configuration an instrumented step of these short-kernel bf16 steps costs ~15 ms of stream time instead of ~1
    # (903 vs 1083 img/s at configs[3] with identical per-kernel times, profiles/r4/launch_event_cost.txt).
    events_inside = not (getattr(args, "events_after", False) or getattr(args, "graphed", False))
    host_ms = []  # host time to ENQUEUE a step (no sync inside): when it nears ms_per_step the run is launch-bound
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # launches on steps 0, 10, ...; named regions (the X_0,0 block) on steps 1, 5, 9, ... WITHOUT per-launch events
        # inside them (two event pairs inside the block would add ~12 us to its ~420)
        sample = timer is not None and events_inside and i % sample_every == 0
        sample_regions = timer is not None and events_inside and i % 4 == 1
        if timer is not None:
            timer.want_launches, timer.want_regions = sample, sample_regions
        ops.set_timer(timer if (sample or sample_regions) else None)
        sampled_steps += int(sample)
        h0 = time.perf_counter()
        run_step(model, opt, crit, x, target)
        host_ms.append(1e3 * (time.perf_counter() - h0))
    ops.set_timer(None)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.set_timer(None)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if not events_inside:
        for _ in range(2):
            if timer is not None:
                timer.want_launches, timer.want_regions = True, False
            ops.set_timer(timer)
            sampled_steps += int(timer is not None)
            train_step(model, opt, crit, x, target)
        ops.set_timer(None)
        torch.cuda.synchronize()

    replicas_identical = replicas_bit_identical(model, dist) if distributed else None

    # eval-mode forward latency (the reference's "high-speed inference" claim; BASELINE metric part 2)
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            model(x)
        torch.cuda.synchronize()
        fwd_ms_per_img = 1e3 * (time.perf_counter() - t1) / (reps * args.batch)
    model.train()

    peak_mem_gb = torch.cuda.max_memory_allocated(dev) / 1e9
    dp_info = None if averager is None else {"grad_allreduce_buckets": len(averager.buckets_last_step),
                                             "grad_bucket_bytes": averager.bucket_bytes,
                                             "gradient_bytes": 4 * averager.flat.numel()}
    del opt, x, target, model, averager  # the next configuration needs the memory
    if os.environ.get("UNETPP_BENCH_KEEP_CACHE") != "1":
        torch.cuda.empty_cache()
    if rank != 0:
        return None

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.batch * args.steps / elapsed
    roofline, roofline_x00, kernels = None, None, None
    if timer is not None:
        launches, regions = timer.summary()
        kernels = {k: {"launches_per_step": v["launches"] / sampled_steps, "ms_per_step": round(v["ms"] / sampled_steps, 3),
                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                       "algorithmic_gb_per_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in launches.items()}
        dom = max(launches.items(), key=lambda kv: kv[1]["ms"])
        alg = dom[1]["flops"] / (dom[1]["ms"] * 1e-3) / 1e12
        # what the matrix pipe executes: Winograd F(2x2,3x3) runs 16 multiply-adds per 36 algorithmic ones
        wino = dom[0].startswith("gemm_wino") or dom[0].startswith("wgrad_wino")
        executed = alg / 2.25 if wino else alg
        from unet_nested4tiny_objects_keypoints_amd import _lib as _l
        traffic, mfma_busy, pmc_source = pmc_counters(dom[0], _l.source_hash(), config_key(args))
        traffic_alg = dom[1]["bytes"] / dom[1]["launches"]
        roofline = {"kernel": dom[0], "bound": "mfma", "achieved": round(executed, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(executed / PEAK_F32_MFMA_TFLOPS, 4),
                    "achieved_note": "multiply-adds executed on the matrix pipe x2 / launch time (HIP events)",
                    "achieved_algorithmic": round(alg, 2),
                    "algorithm": ("winograd F(2x2,3x3), fp32: 16 MFMA multiply-adds per 36 algorithmic ones"
                                  if wino else "direct summation, fp32 MFMA"),
                    "traffic": traffic,
                    "traffic_algorithmic": round(traffic_alg),
                    "traffic_over_algorithmic": None if traffic is None else round(traffic / traffic_alg, 3),
                    "mfma_busy_pmc": mfma_busy,
                    "traffic_source": pmc_source,
                    "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes "
                                    "of this command: started by this run itself (traffic_source live), else read from "
                                    "profiles/pmc_hbm_traffic_latest.json -- null when that file is from another build); "
                                    "traffic_algorithmic = 4 B (bf16 storage: 2 B) x (Cin + Cout) x pixels, plus the "
                                    "accumulated outputs and ReLU gates an input-gradient launch has to read; mfma_busy_pmc = "
                                    "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) from a third pass",
                    "launches_per_step": dom[1]["launches"] / sampled_steps,
                    "timed_steps_with_launch_events": sampled_steps if events_inside else 0,
                    "instrumented_steps_after_timed_region": 0 if events_inside else sampled_steps,
                    "avg_launch_ms": round(dom[1]["ms"] / dom[1]["launches"], 4),
                    "flop_per_launch_avg": dom[1]["flops"] / dom[1]["launches"]}
        if args.dtype == "bf16":
            # bf16 storage: the dense bf16 MFMA ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B) lies above most layers of this
            # network, so the kernel is priced against BOTH limits and `bound` names the one that governs its launches
            peak_bf = 2500.0
            t_mfma = dom[1]["flops"] / (peak_bf * 1e12)
            t_hbm = dom[1]["bytes"] / (PEAK_HBM_TBS * 1e12)
            gbs = dom[1]["bytes"] / (dom[1]["ms"] * 1e-3) / 1e9
            hbm_bound = t_hbm >= t_mfma
            roofline.update({
                "bound": "hbm" if hbm_bound else "mfma",
                "achieved": round(gbs, 1) if hbm_bound else round(alg, 2),
                "peak": PEAK_HBM_TBS * 1e3 if hbm_bound else peak_bf,
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round(max(t_hbm, t_mfma) / (dom[1]["ms"] * 1e-3), 4),
                "achieved_note": "algorithmic bytes (2 B x (Cin + Cout) x pixels) or FLOPs of the launches / their HIP-event "
                                 "time; frac = max(HBM floor, bf16 MFMA floor) / measured time",
                "achieved_algorithmic": round(alg, 2), "algorithmic_gb_per_s": round(gbs, 1),
                "algorithm": "direct implicit GEMM, v_mfma_f32_32x32x16_bf16, bf16 storage, fp32 accumulation",
                "floor_hbm_ms": round(1e3 * t_hbm / sampled_steps, 3), "floor_mfma_ms": round(1e3 * t_mfma / sampled_steps, 3)})
        if "X00.fwd" in regions and args.size == 256 and fs == 1 and args.in_channels == 1 and args.dtype == "f32":
            t_img_us = 1e3 * regions["X00.fwd"]["ms"] / regions["X00.fwd"]["count"] / args.batch
            floor_c = X00_GFLOP_PER_IMG * 1e9 / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e6
            floor_h = X00_MB_PER_IMG * 1e6 / (PEAK_HBM_TBS * 1e12) * 1e6
            roofline_x00 = {"block": "X_0,0 forward (conv-BN-ReLU x2 + pool, train mode)",
                            "us_per_img": round(t_img_us, 2), "floor_compute_us": round(floor_c, 2),
                            "floor_hbm_us": round(floor_h, 2), "bound": "mfma" if floor_c >= floor_h else "hbm",
                            "frac": round(max(floor_c, floor_h) / t_img_us, 4)}
    return {"peak_mem_gb": round(peak_mem_gb, 2), "step_runner": "hip_graph" if graphed else "eager", "host_enqueue_ms": round(sorted(host_ms)[len(host_ms) // 2], 3), "host_enqueue_ms_min": round(min(host_ms), 3), "value": value, "ms_per_step": ms_per_step, "fwd_ms_per_img": fwd_ms_per_img, "roofline": roofline,
            "roofline_x00": roofline_x00, "kernels": kernels, "dp": dp_info,
            "replicas_identical": replicas_identical, "n_cls": n_cls, "fs": fs}


def sweep_and_graph_check(index, o, entry_, ctx):
    """Single-GPU extras of an `other_configs` entry: (1) the same step at larger per-GPU batches (BATCH_SWEEP), 12 timed steps
    each without launch events, with the peak device memory, and the largest batch the 288 GB would hold by that memory's
    slope; (2) for configs[3], whose host enqueue time sits beside its step time, the same step replayed from a HIP graph:
    when the replay is more than 2 % faster the eager line is host-bound and says so."""
    out = {}
    rows = [{"per_gpu_batch": o.batch, "value": entry_["value"], "ms_per_step": entry_["ms_per_step"], "peak_mem_gb": entry_["peak_mem_gb"]}]
    for b in BATCH_SWEEP.get(index, ()):
        q = argparse.Namespace(**dict(vars(o), batch=b, steps=12, warmup=4, prewarm=4, no_launch_timing=True, events_after=True, graphed=False))
        try:
            r = measure(q, ctx)
            rows.append({"per_gpu_batch": b, "value": round(r["value"], 2), "ms_per_step": round(r["ms_per_step"], 3),
                         "peak_mem_gb": r["peak_mem_gb"]})
        except Exception as exc:   # (out of memory ends the sweep, not the line)
            rows.append({"per_gpu_batch": b, "error": "%s: %s" % (type(exc).__name__, str(exc)[:120])})
            break
    ok = [r_ for r_ in rows if "error" not in r_]
    out["batch_sweep"] = rows
    if len(ok) >= 2:
        slope = (ok[-1]["peak_mem_gb"] - ok[0]["peak_mem_gb"]) / (ok[-1]["per_gpu_batch"] - ok[0]["per_gpu_batch"])
        fixed = ok[0]["peak_mem_gb"] - slope * ok[0]["per_gpu_batch"]
        if slope > 0:
            out["largest_batch_that_fits_estimate"] = int((0.92 * HBM_GB - fixed) / slope)
            out["largest_batch_note"] = "from the measured peak device memory per image (%.2f GB) against 92 %% of %d GB; measured batches: %s" % (
                slope, HBM_GB, [r_["per_gpu_batch"] for r_ in ok])
        out["best_batch"] = max(ok, key=lambda r_: r_["value"])["per_gpu_batch"]
    if index == 0 and not getattr(o, "graphed", False):
        q = argparse.Namespace(**dict(vars(o), steps=20, warmup=5, prewarm=5, no_launch_timing=True, events_after=True, graphed=True))
        try:
            r = measure(q, ctx)
            out["graphed_ms_per_step"] = round(r["ms_per_step"], 3)
            out["host_bound"] = bool(r["ms_per_step"] < 0.98 * entry_["ms_per_step"])
        except Exception as exc:
            out["graphed_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:120])
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    if args.rehearse_cpu:
        return rehearse_cpu(args)
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(
        k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)   # this process is itself being profiled: no nesting
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and is_headline(args) and not args.no_live_pmc
            and not args.no_launch_timing and not under_profiler and os.environ.get("UNETPP_BENCH_LIVE_PMC", "1") != "0"):
        # children under rocprofv3, finished before this process initialises the GPU; the two other configurations too
        todo = [args] + ([] if args.no_other_configs else [argparse.Namespace(**dict(vars(args), **over)) for _, over in OTHER_CONFIGS])
        for cfg in todo:
            doc = live_pmc(cfg)
            if doc is None:
                break   # no profiler here (or it failed once): the committed summary serves all of them
            _LIVE_PMC[config_key(cfg)] = doc
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world and rank == 0 and distributed:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal knobs for a 1-GPU box (never set by the driver): every rank on cuda:0 and gloo instead of RCCL,
    # which exercises the whole multi-process path (broadcast, bucketed all-reduce on the side stream, barriers).
    one_dev = os.environ.get("UNETPP_BENCH_SINGLE_DEVICE") == "1"
    backend = os.environ.get("UNETPP_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import contextlib

    import __graft_entry__ as entry
    if rank == 0:
        with contextlib.redirect_stdout(sys.stderr):   # stdout carries the ONE JSON line and nothing else
            entry.build()
    if distributed:
        dist.barrier()

    ctx = {"torch": torch, "dist": dist, "dev": dev, "world": world, "rank": rank, "distributed": distributed}
    others_first = os.environ.get("UNETPP_BENCH_OTHERS_FIRST") == "1"   # diagnostic (never set by the driver): order effects
    res = None if others_first else measure(args, ctx)
    # BASELINE configs[3] and configs[4] (bf16 storage) in the same run, attached to the same line: every rank takes part
    # (they are data-parallel steps like the headline), a failure there must not cost the headline line
    others = None
    if not args.no_other_configs and is_headline(args):
        others = []
        for name, over in OTHER_CONFIGS:
            o = argparse.Namespace(**dict(vars(args), **over))
            # eager train_step, like the headline (round 5: the graph replay buys no device time -- DESIGN.md section 8 --
            # and is experimental; `--graphed` opts in)
            o.steps, o.warmup, o.prewarm, o.events_after, o.graphed = max(20, args.other_steps), 10, 15, True, bool(args.graphed)
            try:
                r = measure(o, ctx)
                entry_ = None if r is None else other_entry(name, o, r, world)
            except Exception as exc:
                if distributed:
                    raise  # ranks would fall out of step: better no line than a hung job
                entry_ = {"config": name, "error": "%s: %s" % (type(exc).__name__, exc)}
            if entry_ is not None and "error" not in entry_ and not distributed and os.environ.get("UNETPP_BENCH_NO_SWEEP") != "1":
                entry_.update(sweep_and_graph_check(OTHER_CONFIGS.index((name, over)), o, entry_, ctx))
            if entry_ is not None:
                others.append(entry_)
    if others_first:
        res = measure(args, ctx)
    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    value, ms_per_step, fwd_ms_per_img = res["value"], res["ms_per_step"], res["fwd_ms_per_img"]
    roofline, roofline_x00, kernels = res["roofline"], res["roofline_x00"], res["kernels"]
    dp_info, replicas_identical, n_cls = res["dp"] or {}, res["replicas_identical"], res["n_cls"]
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, n_cls)
    line = {
        "metric": "images/sec (train step) UNet++ L=4 256x256",
        "value": round(value, 2),
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "prewarm_steps": args.prewarm,
        "ms_per_step": round(ms_per_step, 3),
        "host_enqueue_ms_median": res["host_enqueue_ms"],
        # wall time of the call that ENQUEUES a step.  Once the host is a step ahead the HIP queue pushes back and the call
        # waits for the device, so the median tracks the DEVICE time of a step (it can exceed ms_per_step: instrumented and
        # short calls skew the mean the other way); the minimum is what the host needs when nothing blocks it.  Host-bound
        # would be min ~ ms_per_step; the direct check is other_configs[0].host_bound (HIP-graph replay vs eager).
        "host_enqueue_ms_min": res["host_enqueue_ms_min"],
        "step_runner": res["step_runner"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": workload_name(args),
                   "global_batch": world * args.batch, "per_gpu_batch": args.batch,
                   "parallelism": "dp%d" % world if distributed else "single",
                   "world_size": dist.get_world_size() if distributed else 1,
                   "backend": dist.get_backend() if distributed else None,
                   "grad_allreduce_buckets": dp_info.get("grad_allreduce_buckets"),
                   "grad_bucket_bytes": dp_info.get("grad_bucket_bytes"),
                   "gradient_bytes": dp_info.get("gradient_bytes"),
                   "replicas_bit_identical": replicas_identical},
        "fwd_ms_per_img": round(fwd_ms_per_img, 4),
        "roofline": roofline,
        "roofline_x00": roofline_x00,
        "kernels": kernels,
        "cpu_baseline": cpu,
        "other_configs": others,
    }
    if cpu is not None:
        line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
    print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
