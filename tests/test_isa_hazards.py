"""Static ISA check (CPU, needs hipcc only): inline-asm packed adds are invisible to hipcc's hazard recogniser, so the
two Winograd kernels are compiled to assembly and every inline-asm VALU write is checked against the C operands of the
MFMAs issued in the 7 wait states before it (tools/check_asm_mfma_hazard.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("src", ["wgrad_wino.hip", "gemm_wino.hip"])
def test_inline_asm_valu_writes_keep_clear_of_mfma_c_operands(tmp_path, src):
    from unet_nested4tiny_objects_keypoints_amd import _lib
    import check_asm_mfma_hazard

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / (src + ".s"))
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", _lib.INCLUDE, "-I", _lib.CSRC,
           *_lib.EXTRA_FLAGS.get(src, ()), "--offload-device-only", "-S", os.path.join(_lib.CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert check_asm_mfma_hazard.main(out, "_kernel") == 0
