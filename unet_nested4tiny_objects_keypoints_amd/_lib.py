"""Loader for the C-ABI HIP library (include/unetpp_hip.h) -- ctypes, no torch types cross the ABI.

The library is built in-tree by ``build_library()`` (hipcc --offload-arch=gfx950) as
``unet_nested4tiny_objects_keypoints_amd/libunetpp_hip.so``.  There is no CPU fallback: if the
shared object is missing or does not load, ``lib()`` raises and every op of the package fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_PKG_DIR)
LIB_PATH = os.environ.get("UNETPP_LIB", os.path.join(_PKG_DIR, "libunetpp_hip.so"))  # override: kernel A/B runs
CSRC = os.path.join(_PKG_DIR, "csrc")
INCLUDE = os.path.join(_REPO, "include")
SOURCES = ("gemm_pix.hip", "gemm_fast.hip", "gemm_pw.hip", "gemm_pw_bf16.hip", "gemm_wino.hip", "gemm_bf16.hip", "gemm_bf16_dma.hip", "wgrad_bf16.hip", "pointwise_bf16.hip", "keypoints.hip", "weight_image.hip", "wgrad.hip", "wgrad_fast.hip", "wgrad_dma.hip", "wgrad_pw.hip", "wgrad_wino.hip", "first_layer.hip", "pointwise.hip", "caller.hip")
HEADERS = ("common.h", "gemm_units.h", "wgrad_reduce.h", "lds_asm.h", "bf16_common.h", "dropout.h", "bn_fused.h",
           "wino_experiments.h", "dma_experiments.h")
MAX_VIEWS = 8
ABI_VERSION = 11
# packed-f32 VALU (SLP-vectorised add pairs) costs issue slots beside MFMAs: keep the Winograd transforms scalar
EXTRA_FLAGS = {"gemm_wino.hip": ("-fno-slp-vectorize",), "wgrad_wino.hip": ("-fno-slp-vectorize",)}
GEMM_DIRECT = 1  # unetpp_gemm_desc.flags: direct summation only (no Winograd)
GEMM_BF16 = 2    # unetpp_gemm_desc / unetpp_wgrad_desc flags: bf16 storage of every activation view


class View(C.Structure):
    """mirror of struct unetpp_view"""
    _fields_ = [
        ("ptr", C.c_void_p),
        ("C", C.c_int32), ("c_off", C.c_int32), ("c_len", C.c_int32),
        ("Hs", C.c_int32), ("Ws", C.c_int32),
        ("sy", C.c_int32), ("sx", C.c_int32), ("oy", C.c_int32), ("ox", C.c_int32),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("gate", C.c_void_p),
        ("relu", C.c_int32), ("accumulate", C.c_int32), ("gate_sum", C.c_int32), ("reserved", C.c_int32),
    ]


class BnFused(C.Structure):
    """mirror of struct unetpp_bn_fused"""
    _fields_ = [
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
        ("mean", C.c_void_p), ("invstd", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("count", C.c_int64), ("eps", C.c_float), ("momentum", C.c_float),
    ]


class GemmDesc(C.Structure):
    """mirror of struct unetpp_gemm_desc"""
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("taps", C.c_int32), ("n_in", C.c_int32), ("n_out", C.c_int32),
        ("flags", C.c_int32), ("reserved", C.c_int32),
        ("inp", View * MAX_VIEWS), ("out", View * MAX_VIEWS),
        ("weight", C.c_void_p), ("bias", C.c_void_p), ("stats_partial", C.c_void_p),
        ("weight_image", C.c_void_p),
        ("bn", BnFused),
    ]


class WeightSrc(C.Structure):
    """mirror of struct unetpp_weight_src"""
    _fields_ = [
        ("src", C.c_void_p),
        ("s_t", C.c_int64), ("s_k", C.c_int64), ("s_ko", C.c_int64), ("s_n", C.c_int64), ("s_no", C.c_int64),
        ("k_inner", C.c_int32), ("n_inner", C.c_int32), ("flip", C.c_int32), ("reserved", C.c_int32),
    ]


class PackJob(C.Structure):
    """mirror of struct unetpp_pack_job"""
    _fields_ = [
        ("src", WeightSrc),
        ("image", C.c_void_p),
        ("taps", C.c_int32), ("flags", C.c_int32), ("n_in", C.c_int32), ("n_out", C.c_int32),
        ("in_len", C.c_int32 * MAX_VIEWS), ("out_len", C.c_int32 * MAX_VIEWS),
    ]


class CopyJob(C.Structure):
    """mirror of struct unetpp_copy_job"""
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p),
        ("n_outer", C.c_int64), ("n_inner", C.c_int64), ("src_stride", C.c_int64), ("dst_stride", C.c_int64),
    ]


MAX_HEADS = 8


class FocalHeads(C.Structure):
    """mirror of struct unetpp_focal_heads"""
    _fields_ = [
        ("pred", C.c_void_p * MAX_HEADS), ("grad", C.c_void_p * MAX_HEADS),
        ("n_heads", C.c_int32), ("reserved", C.c_int32),
    ]


class WgradDesc(C.Structure):
    """mirror of struct unetpp_wgrad_desc"""
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("taps", C.c_int32), ("n_x", C.c_int32), ("n_dy", C.c_int32),
        ("x", View * MAX_VIEWS), ("dy", View * MAX_VIEWS),
        ("n_split", C.c_int32), ("flags", C.c_int32),
        ("slabs", C.c_void_p),
    ]


_P, _I32, _I64, _F, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); kept in step with include/unetpp_hip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "unetpp_abi_version": (C.c_int, []),
    "unetpp_build_arch": (C.c_char_p, []),
    "unetpp_last_kernel_name": (C.c_char_p, []),
    "unetpp_set_reserved_cus": (_I32, [_I32]),
    "unetpp_debug_set": (C.c_int, [C.c_char_p, _I64, _I32]),
    "unetpp_debug_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "unetpp_usable_cus": (_I32, [C.POINTER(C.c_int32)]),
    "unetpp_gemm_pixel_blocks": (_I64, [_I32, _I32, _I32]),
    "unetpp_gemm_stats_rows": (_I64, [_I32, _I32, _I32]),
    "unetpp_gemm_fwd": (C.c_int, [C.POINTER(GemmDesc), _P]),
    "unetpp_gemm_weight_image_floats": (_I64, [C.POINTER(GemmDesc)]),
    "unetpp_gemm_pack_weight_image": (C.c_int, [C.POINTER(GemmDesc), _P, _P]),
    "unetpp_gemm_pack_weight_image_from": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(WeightSrc), _P, _P]),
    "unetpp_gemm_pack_weight_images": (C.c_int, [_P, _I32, _I64, _P]),
    "unetpp_wgrad_max_split": (_I32, [_I32, _I32, _I32]),
    "unetpp_wgrad_slab_planes": (_I32, [C.POINTER(WgradDesc)]),
    "unetpp_wgrad_pairs_per_workgroup": (_I32, [C.POINTER(WgradDesc)]),
    "unetpp_wgrad": (C.c_int, [C.POINTER(WgradDesc), _P]),
    "unetpp_wgrad_finish": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _P, _I64, _I64, _I64, _I64, _P, _P]),
    "unetpp_pack_weight": (C.c_int, [_P, _P, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _P]),
    "unetpp_bn_finalize": (C.c_int, [_P, _I64, _I32, _I64, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    "unetpp_bn_eval_coeffs": (C.c_int, [_P, _P, _P, _P, _F, _I32, _P, _P, _P]),
    "unetpp_affine_relu_pool": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "unetpp_maxpool_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "unetpp_bn_bwd_blocks": (_I64, [_I64, _I32]),
    "unetpp_bn_bwd_reduce": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P]),
    "unetpp_bn_bwd_finalize": (C.c_int, [_P, _I64, _I32, _P, _P, _P]),
    "unetpp_bn_bwd_apply": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P]),
    "unetpp_bn_bwd_pool_ok": (C.c_int, [_I32, _I32, _I32, _I32]),
    "unetpp_bn_bwd_reduce_pool": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "unetpp_bn_bwd_apply_pool": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "unetpp_head_fwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _U64, _P, _P, _P, _P]),
    "unetpp_head_bwd_blocks": (_I64, [_I64]),
    "unetpp_head_bwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _U64, _P, _P, _P, _I32, _I32, _P, _P]),
    "unetpp_sum_partials": (C.c_int, [_P, _I64, _I64, _P, _P]),
    "unetpp_focal_bce_blocks": (_I64, [_I64]),
    "unetpp_focal_bce": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _P]),
    "unetpp_focal_bce_heads": (C.c_int, [C.POINTER(FocalHeads), _P, _I64, _I64, _F, _P, _P, _P]),
    "unetpp_copy_jobs": (C.c_int, [_P, _I32, _I64, _P]),
    "unetpp_heatmap_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "unetpp_create_heatmap": (C.c_int, [_P, _I32, _I32, _I32, _I32, _F, _P, _P, _P]),
    "unetpp_bilinear2x_fwd": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P]),
    "unetpp_bilinear2x_bwd": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "unetpp_nchw_to_nhwc": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P]),
    "unetpp_nhwc_to_nchw": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P]),
    # bf16-storage companions (pointwise_bf16.hip)
    "unetpp_affine_relu_pool_bf16": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "unetpp_bn_bwd_blocks_bf16": (_I64, [_I64, _I32]),
    "unetpp_bn_bwd_reduce_bf16": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "unetpp_bn_bwd_apply_bf16": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "unetpp_head_fwd_bf16": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _U64, _P, _P, _P, _P]),
    "unetpp_first_layer_dgrad_bf16": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "unetpp_maxpool_bwd_bf16": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "unetpp_bilinear2x_fwd_bf16": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P]),
    "unetpp_bilinear2x_bwd_bf16": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _I32, _P, _P]),
    # heat-map side of validation (keypoints.hip)
    "unetpp_heatmap_pattern_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "unetpp_heatmap_pattern": (C.c_int, [_P, _I32, _I32, _P, _P, _I32, _I32, _I32, _F, _P, _P, _P]),
    "unetpp_keypoints_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "unetpp_keypoints_extract": (C.c_int, [_I32, _P, _I32, _I32, _I32, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "unetpp_head_bwd_bf16": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _U64, _P, _P, _P, _I32, _I32, _P, _P]),
}

_LIB = None


def source_hash() -> str:
    """Hash of every source the library is built from: stamps PMC summaries so that bench.py can tell a profile of
    another build from one of this build (tools/pmc_traffic.py, bench.py roofline.traffic)."""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(INCLUDE, "unetpp_hip.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_library(force: bool = False, verbose: bool = False, out_path: str = None, obj_dir: str = None) -> str:
    """Compile the HIP sources for gfx950 into the in-tree shared object (cross-compiles without a GPU).
    out_path / obj_dir: build somewhere else (tests/test_abi.py forces a from-scratch build into a temporary directory)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "unetpp_hip.h")]
    lib_path = out_path or LIB_PATH
    if not force and os.path.exists(lib_path) and all(os.path.getmtime(lib_path) >= os.path.getmtime(d) for d in deps):
        return lib_path
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC]
    obj_dir = obj_dir or os.path.join(_REPO, "build", "obj")
    os.makedirs(obj_dir, exist_ok=True)
    jobs = []
    for src in SOURCES:  # one object per source, compiled concurrently; EXTRA_FLAGS are per kernel file
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = common + list(EXTRA_FLAGS.get(src, ())) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((obj, cmd, subprocess.Popen(cmd)))
    for obj, cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    # link inside the object directory and move the result: the offload bundler drops its temporaries beside the
    # output file, and an interrupted link once left 32 of them in the package directory
    staged = os.path.join(obj_dir, os.path.basename(lib_path) + ".link")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", staged] + [j[0] for j in jobs]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    os.replace(staged, lib_path)
    if out_path is None:
        global _LIB
        _LIB = None
    return lib_path


def lib():
    """The loaded library with argtypes set.  Raises if the HIP library is not built -- by design."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "unetpp HIP library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for this path)." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.unetpp_abi_version() != ABI_VERSION:
            raise RuntimeError("unetpp HIP library ABI mismatch")
        _LIB = handle
    return _LIB


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError("%s failed with status %d (%s)" % (
            what, status, {-1: "invalid argument", -2: "kernel launch error"}.get(status, "unknown")))


class debug_switch:
    """with debug_switch("BF16_DMA_FORM", 8): ...  -- a dispatcher switch of the library (unetpp_debug_set) for the
    duration of a block; tests use it to hold two kernels against each other inside one process, tools for A/B runs.
    On exit the switch goes back to what it was on entry (unetpp_debug_get): unset, or the value an enclosing block or the
    UNETPP_<name> environment variable had given it.  Process-wide: not for concurrent use from several threads."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name.encode(), int(value)
        self._before = None

    def __enter__(self):
        old = C.c_int64(0)
        state = lib().unetpp_debug_get(self.name, C.byref(old))
        if state < 0:
            check(state, "unetpp_debug_get")
        self._before = (state == 1, int(old.value))
        check(lib().unetpp_debug_set(self.name, self.value, 1), "unetpp_debug_set")
        return self

    def __exit__(self, *exc):
        was_set, value = self._before
        check(lib().unetpp_debug_set(self.name, value if was_set else 0, 1 if was_set else 0), "unetpp_debug_set")
        return False
