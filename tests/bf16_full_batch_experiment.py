"""One-off evidence run (not collected by pytest): tests/test_gpu_bf16.py::test_bf16_train_step_vs_oracles -- every bound
of the stated bf16 tolerance, including every parameter gradient against the pinned fp32 oracle with the HIP forward's
routing -- at the BENCHMARKED batches of BASELINE configs[3] (512 x 512, batch 8) and configs[4] (depth 5, base 64, 3 -> 5
channels, 384 x 384, batch 4); the suite runs those geometries at batch 1 (three CPU passes of the oracle per case: minutes
at these sizes).  The test's record lines go to gpurun_out/bf16_vs_oracle.jsonl.

    python tests/bf16_full_batch_experiment.py [c3|c5 ...]"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = {"c3": (dict(in_channels=1, n_classes=4, feature_scale=1), 8, 512, 512),
         "c5": (dict(in_channels=3, n_classes=5, feature_scale=0.5, depth=5), 4, 384, 384)}


def main():
    from tests.test_gpu_bf16 import test_bf16_train_step_vs_oracles as run
    dev = torch.device("cuda", 0)
    t0 = time.time()
    stop = threading.Event()

    def beat():
        while not stop.wait(45.0):
            print("[%4.0f s] the CPU oracles are still running" % (time.time() - t0), flush=True)
    threading.Thread(target=beat, daemon=True).start()
    for name in (sys.argv[1:] or ["c3", "c5"]):
        ctor, b, h, w = CASES[name]
        run(dev, ctor, b, h, w)   # raises AssertionError when a bound is exceeded
        print("[%4.0f s] %s at batch %d: every bf16 bound of the suite holds" % (time.time() - t0, name, b), flush=True)
    stop.set()


if __name__ == "__main__":
    main()
