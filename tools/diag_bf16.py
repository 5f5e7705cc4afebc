"""Diagnostic (GPU box): per-parameter gradient error of the bf16-storage HIP path against the fp32 CPU oracle.
    python tools/diag_bf16.py feature_scale batch H W [depth in_channels n_classes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests.test_gpu_bf16 import _bf16_vs_oracle  # noqa: E402

a = sys.argv[1:]
fs = float(a[0]); fs = int(fs) if fs.is_integer() else fs
b, h, w = int(a[1]), int(a[2]), int(a[3])
depth = int(a[4]) if len(a) > 4 else 4
cin = int(a[5]) if len(a) > 5 else 1
ncls = int(a[6]) if len(a) > 6 else 4
res = _bf16_vs_oracle(torch.device("cuda:0"), dict(in_channels=cin, n_classes=ncls, feature_scale=fs, depth=depth), b, h, w, 51, probe=bool(int(os.environ.get("PROBE", "0"))))
print({k: v for k, v in res.items() if k not in ("grad_l2", "sim_grad_l2", "cos")})
for k in res["grad_l2"]:
    print("%-36s vs fp32 %.4f  vs bf16-sim %.4f  cos(sim) %.5f" % (k, res["grad_l2"][k], res["sim_grad_l2"][k], res["cos"][k]))
