"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU, and exports
exactly the entry points include/unetpp_hip.h declares; the ctypes mirrors have the C structs' layout; the
product refuses to run without the library or on CPU tensors (no fallback)."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "unetpp_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(unetpp_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as entry
    entry.build()
    from unet_nested4tiny_objects_keypoints_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(built_lib):
    names = _declared_functions()
    assert len(names) >= 25
    handle = ctypes.CDLL(built_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    # and the Python signature table covers exactly the header
    assert sorted(built_lib.SIGNATURES) == names


def test_exported_symbols_are_plain_c(built_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert set(_declared_functions()) <= exported
    assert built_lib.lib().unetpp_abi_version() == built_lib.ABI_VERSION == 11
    assert built_lib.lib().unetpp_build_arch() == b"gfx950"


def test_struct_layout_matches_header(built_lib, tmp_path):
    """sizeof/offsetof from a C compile of the header vs the ctypes mirrors."""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "unetpp_hip.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(unetpp_view), offsetof(unetpp_view, gate),'
                   'offsetof(unetpp_view, gate_sum), sizeof(unetpp_gemm_desc), offsetof(unetpp_gemm_desc, out),'
                   'offsetof(unetpp_gemm_desc, weight_image), sizeof(unetpp_wgrad_desc), offsetof(unetpp_wgrad_desc, dy),'
                   'offsetof(unetpp_wgrad_desc, slabs), sizeof(unetpp_weight_src), offsetof(unetpp_weight_src, k_inner),'
                   'sizeof(unetpp_pack_job), offsetof(unetpp_pack_job, image), offsetof(unetpp_pack_job, out_len),'
                   'sizeof(unetpp_bn_fused), offsetof(unetpp_bn_fused, count), offsetof(unetpp_bn_fused, momentum),'
                   'offsetof(unetpp_gemm_desc, bn));return 0;}')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    L = built_lib
    want = [ctypes.sizeof(L.View), L.View.gate.offset, L.View.gate_sum.offset, ctypes.sizeof(L.GemmDesc),
            L.GemmDesc.out.offset, L.GemmDesc.weight_image.offset, ctypes.sizeof(L.WgradDesc), L.WgradDesc.dy.offset,
            L.WgradDesc.slabs.offset, ctypes.sizeof(L.WeightSrc), L.WeightSrc.k_inner.offset, ctypes.sizeof(L.PackJob),
            L.PackJob.image.offset, L.PackJob.out_len.offset, ctypes.sizeof(L.BnFused), L.BnFused.count.offset,
            L.BnFused.momentum.offset, L.GemmDesc.bn.offset]
    assert got == want


def test_argument_validation_without_gpu(built_lib):
    """Entry points that do not touch the device: size queries and argument checks (status codes, no throw)."""
    lib = built_lib.lib()
    assert lib.unetpp_gemm_pixel_blocks(32, 256, 256) == 32 * 32 * 8
    assert lib.unetpp_gemm_pixel_blocks(0, 256, 256) == 0
    assert lib.unetpp_gemm_stats_rows(32, 256, 256) == 32 * 32 * 8 and lib.unetpp_gemm_stats_rows(1, 64, 64) == 2048
    assert lib.unetpp_wgrad_max_split(1, 8, 8) == 1
    wd = built_lib.WgradDesc()   # which kernel a weight-gradient descriptor gets is decided on the host
    wd.N, wd.H, wd.W, wd.taps, wd.n_x, wd.n_dy = 2, 32, 32, 9, 2, 1
    for v, c in ((wd.x[0], 64), (wd.x[1], 128), (wd.dy[0], 64)):
        v.C = v.c_len = c
    assert lib.unetpp_wgrad_pairs_per_workgroup(ctypes.byref(wd)) == 1     # fp32: one tile pair per workgroup
    wd.flags = built_lib.GEMM_BF16
    assert lib.unetpp_wgrad_pairs_per_workgroup(ctypes.byref(wd)) == 4     # bf16, every view a multiple of 64 wide
    wd.x[1].C = wd.x[1].c_len = 96
    assert lib.unetpp_wgrad_pairs_per_workgroup(ctypes.byref(wd)) == 1
    assert lib.unetpp_wgrad_pairs_per_workgroup(None) == 0
    assert lib.unetpp_head_bwd_blocks(100) == 2
    assert lib.unetpp_bn_bwd_blocks(32 * 256 * 256, 32) % 8 == 0
    d = built_lib.GemmDesc()
    assert lib.unetpp_gemm_fwd(ctypes.byref(d), None) == -1          # N = 0
    assert lib.unetpp_gemm_weight_image_floats(ctypes.byref(d)) == 0
    d.N, d.H, d.W, d.taps, d.n_in, d.n_out = 1, 8, 8, 5, 1, 1
    assert lib.unetpp_gemm_fwd(ctypes.byref(d), None) == -1          # taps must be 1 or 9
    w = built_lib.WgradDesc()
    assert lib.unetpp_wgrad(ctypes.byref(w), None) == -1
    assert lib.unetpp_pack_weight(None, None, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, None) == -1
    assert lib.unetpp_head_fwd(None, None, None, 1, 8, 8, 32, 4, 0.4, 0, None, None, None, None) == -1


def test_no_cpu_fallback():
    from unet_nested4tiny_objects_keypoints_amd import UNet_Nested, ops
    m = UNet_Nested(in_channels=1, feature_scale=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 1, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.nchw_to_nhwc(torch.randn(1, 3, 8, 8))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from unet_nested4tiny_objects_keypoints_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(RuntimeError, match="not built"):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "unet_nested4tiny_objects_keypoints_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn


def test_no_launch_path_reads_the_environment(built_lib):
    """VERDICT r4 item 7 / ADVICE r3: dispatcher switches are a process-wide table (unetpp_debug_set) that is filled from
    UNETPP_* variables once; no source file but the table's owner may call getenv, and there only inside the
    once-initialiser.  The launch-to-finalize hand-over of the BatchNorm row count travels by value (no thread_local)."""
    csrc = os.path.join(ROOT, "unet_nested4tiny_objects_keypoints_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, fn)).read()
        code = re.sub(r"//[^\n]*", "", text)     # comments may talk about it
        if fn == "gemm_pix.hip":
            body = code[code.index("void opts_from_environment()"):]
            body = body[:body.index("bool opt_is_set(")]
            assert code.count("getenv(") == body.count("getenv(") == 2, "getenv outside the once-initialiser"
            assert "std::call_once" in body
        else:
            assert "getenv(" not in code, fn
        assert "g_bn_rows" not in code and "note_bn_rows" not in code, fn
    lib = built_lib.lib()
    import ctypes
    v = ctypes.c_int64(-1)
    assert lib.unetpp_debug_get(b"BF16_DMA_FORM", ctypes.byref(v)) == 0 and v.value == -1   # unset: value untouched
    assert lib.unetpp_debug_set(b"BF16_DMA_FORM", 8, 1) == 0
    assert lib.unetpp_debug_get(b"BF16_DMA_FORM", ctypes.byref(v)) == 1 and v.value == 8
    with built_lib.debug_switch("BF16_DMA_FORM", 4):     # a scoped override puts back what it found
        assert lib.unetpp_debug_get(b"BF16_DMA_FORM", ctypes.byref(v)) == 1 and v.value == 4
    assert lib.unetpp_debug_get(b"BF16_DMA_FORM", ctypes.byref(v)) == 1 and v.value == 8
    assert lib.unetpp_debug_set(b"BF16_DMA_FORM", 0, 0) == 0
    assert lib.unetpp_debug_get(b"BF16_DMA_FORM", ctypes.byref(v)) == 0
    assert lib.unetpp_debug_get(b"NO_SUCH_SWITCH", ctypes.byref(v)) < 0
    assert lib.unetpp_debug_set(b"NO_SUCH_SWITCH", 1, 1) < 0
    assert lib.unetpp_debug_set(None, 1, 1) < 0


def test_clean_build_from_sources(tmp_path):
    """`build()` returns early when the in-tree .so is newer than its sources, so the shipped binary is
    what normally runs.  Force a from-scratch hipcc build of every source into a temporary directory and check that the
    result is a complete library: same ABI version, gfx950, every symbol of the argtypes table."""
    import ctypes as C

    from unet_nested4tiny_objects_keypoints_amd import _lib
    out = _lib.build_library(force=True, out_path=str(tmp_path / "libunetpp_clean.so"), obj_dir=str(tmp_path / "obj"))
    handle = C.CDLL(out)
    handle.unetpp_abi_version.restype = C.c_int
    handle.unetpp_build_arch.restype = C.c_char_p
    assert handle.unetpp_abi_version() == _lib.ABI_VERSION
    assert handle.unetpp_build_arch() == b"gfx950"
    for name in _lib.SIGNATURES:
        getattr(handle, name)  # AttributeError if a source file dropped out of the build
