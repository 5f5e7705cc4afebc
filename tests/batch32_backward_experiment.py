"""One-off evidence run (not collected by pytest: the float64 oracle's backward at this size takes minutes): the
BENCHMARKED configuration -- BASELINE configs[1], base 32, 256 x 256, batch 32 -- through forward, focal loss and the whole
backward pass of the HIP model against the pinned CPU oracle in float64 whose backward takes the HIP forward's ReLU gates
and pool winners (tests/helpers.install_hip_gates), exactly what tests/test_gpu_model.py::test_train_step_vs_oracle does
at batch 4.  Every parameter gradient must meet the suite's tolerance (tests/helpers.assert_grads_close, TOL = 1e-4).

    python tests/batch32_backward_experiment.py [batch] [c2|c3|c5]   -> gpurun_out/batch32_backward_vs_float64[_c3|_c5].json

c3 / c5: the same in fp32 at the geometry AND batch of BASELINE configs[3] (512 x 512, batch 8) / configs[4] (depth 5, base
64, 3 -> 5 channels, 384 x 384, batch 4) -- the suite runs those geometries at batch 1.

models/unet.py:255-300, trainer/trainer.py:114-136."""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def heartbeat(t0, stop):
    while not stop.wait(45.0):
        print("[%4.0f s] the float64 oracle is still running" % (time.time() - t0), flush=True)


def main():
    from oracle.step_oracle import focal_bce_2d_oracle
    from oracle.unet_nested_oracle import UNetNestedOracle
    from tests.helpers import check_flips, install_hip_gates, is_pre_bn_bias, rel_err
    from tests.test_gpu_model import TOL, _hip_model, _loss
    cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
    ctor, h, w, b0 = {"c2": (dict(in_channels=1, n_classes=4, feature_scale=1), 256, 256, 32),
                      "c3": (dict(in_channels=1, n_classes=4, feature_scale=1), 512, 512, 8),
                      "c5": (dict(in_channels=3, n_classes=5, feature_scale=0.5, depth=5), 384, 384, 4)}[cfg]
    b = int(sys.argv[1]) if len(sys.argv) > 1 and int(sys.argv[1]) > 0 else b0
    dev = torch.device("cuda", 0)
    t0 = time.time()
    stop = threading.Event()
    threading.Thread(target=heartbeat, args=(t0, stop), daemon=True).start()
    torch.manual_seed(17)
    ref = UNetNestedOracle(**ctor).train()
    ref.drop_out.eval()
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    m = _hip_model(ctor, state, dev).train()
    m.drop_out.eval()
    m._debug_keep_saved = True
    x, target = torch.randn(b, ctor["in_channels"], h, w), torch.rand(b, ctor["n_classes"], h, w)
    outs = m(x.to(dev))
    loss = _loss(outs, target.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    print("[%4.0f s] HIP forward + backward done, loss %.6f" % (time.time() - t0, float(loss)), flush=True)
    ref = ref.double()
    gated = install_hip_gates(ref, m._debug_saved)
    ro = ref(x.double())
    rl = sum(focal_bce_2d_oracle(o, target.double()) for o in ro) / len(ro)
    print("[%4.0f s] oracle forward done, loss %.6f" % (time.time() - t0, float(rl)), flush=True)
    rl.backward()
    print("[%4.0f s] oracle backward done" % (time.time() - t0), flush=True)
    stop.set()
    flips = check_flips(gated, "batch %d" % b)
    out_err = max(rel_err(o.detach().cpu(), r.detach()) for o, r in zip(outs, ro))
    rows, worst = {}, ("", 0.0)
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if is_pre_bn_bias(k, ctor):
            continue   # analytically zero gradient (a bias in front of BatchNorm): compared absolutely in the suite
        g, r = p.grad.double().cpu(), q.grad.double()
        e = float((g - r).abs().max() / r.abs().max())
        l2 = float((g - r).norm() / r.norm())
        rows[k] = {"max_rel": e, "l2_rel": l2}
        if e > worst[1]:
            worst = (k, e)
    res = {"configuration": "%s geometry: %s, %dx%d, batch %d, fp32, train mode (dropout off), FocalLoss_BCE_2d on every head" % (
               {"c2": "configs[1]", "c3": "configs[3]", "c5": "configs[4]"}[cfg], sorted(ctor.items()), h, w, b),
           "oracle": "UNetNestedOracle in float64, backward with the HIP forward's ReLU gates / pool winners",
           "relu_gate_or_pool_flips_vs_the_oracles_own": int(flips), "outputs_max_rel": out_err,
           "loss_hip": float(loss), "loss_oracle": float(rl), "loss_rel": abs(float(loss) - float(rl)) / abs(float(rl)),
           "parameters_compared": len(rows), "worst_gradient": {"name": worst[0], "max_rel": worst[1]},
           "worst_gradient_l2": max(v["l2_rel"] for v in rows.values()), "tolerance": TOL,
           "within_tolerance": bool(worst[1] < TOL and out_err < TOL), "seconds": round(time.time() - t0, 1)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "batch32_backward_vs_float64%s.json" % ("" if cfg == "c2" else "_" + cfg)), "w") as f:
        json.dump({"summary": res, "per_parameter": rows}, f, indent=1)
    print(json.dumps(res))
    if not res["within_tolerance"]:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
