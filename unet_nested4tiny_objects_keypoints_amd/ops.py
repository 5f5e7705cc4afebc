"""Tensor-facing wrappers over the C ABI (include/unetpp_hip.h).

Every function takes CUDA(HIP) fp32 tensors owned by PyTorch, hands raw pointers, sizes and the
current HIP stream to the library, and returns immediately (asynchronous on that stream).  PyTorch
here is plumbing only: device memory and the stream.  Activations are NHWC ``[N, H, W, C]``.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import MAX_VIEWS, GemmDesc, PackJob, View, WeightSrc, WgradDesc, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class LaunchTimer:
    """Optional per-launch timing with HIP events on the launching stream (used by bench.py for the
    roofline line).  Records (kind, flops, bytes, start_event, end_event) for every MFMA-kernel launch
    and (name, start, end) for named regions; read the times after a device synchronise.  An event pair costs the
    stream about 6 us, so a caller that wants honest region times samples launches and regions on DIFFERENT steps
    (want_launches / want_regions)."""

    def __init__(self):
        self.launches = []
        self.regions = []
        self.want_launches = True
        self.want_regions = True

    def _pair(self):
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def summary(self):
        out = {}
        for kind, flops, nbytes, s, e in self.launches:
            d = out.setdefault(kind, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["bytes"] += nbytes
        reg = {}
        for name, s, e in self.regions:
            d = reg.setdefault(name, {"count": 0, "ms": 0.0})
            d["count"] += 1
            d["ms"] += s.elapsed_time(e)
        return out, reg


_TIMER: Optional[LaunchTimer] = None


def set_timer(t: Optional[LaunchTimer]) -> None:
    global _TIMER
    _TIMER = t


class region:
    """with ops.region("X00.fwd"): ...  -- a named span on the current stream when a LaunchTimer is set."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = _TIMER is not None and _TIMER.want_regions
        if self.on:
            self.s, self.e = _TIMER._pair()
            self.s.record()
        return self

    def __exit__(self, *exc):
        if self.on and _TIMER is not None:
            self.e.record()
            _TIMER.regions.append((self.name, self.s, self.e))
        return False


def _timed_call(kind, flops, fn, nbytes=0.0):
    """kind = label of the launch, or None to ask the library which kernel it picked (unetpp_last_kernel_name).
    flops / nbytes: ALGORITHMIC work of the launch (every operand element read or written once)."""
    if _TIMER is None or not _TIMER.want_launches:
        return fn()
    s, e = _TIMER._pair()
    s.record()
    r = fn()
    e.record()
    if kind is None:
        kind = _lib.lib().unetpp_last_kernel_name().decode()
    _TIMER.launches.append((kind, flops, nbytes, s, e))
    return r


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need(t: torch.Tensor, what: str, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: this path has no CPU fallback" % what)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (what, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % what)
    return t


_ACT_DTYPES = (torch.float32, torch.bfloat16)  # storage types of activations (bf16: UNETPP_GEMM_BF16 launches)


def _storage_flag(outs, ins):
    """UNETPP_GEMM_BF16 when the launch's activations are bf16.  All views of a launch share the storage type, except
    that the 1..4-channel network input of the first convolution stays fp32."""
    bf = outs[0].t.dtype == torch.bfloat16
    for v in outs:
        if (v.t.dtype == torch.bfloat16) != bf:
            raise TypeError("mixed fp32 / bf16 output views in one launch")
    for v in ins:
        if (v.t.dtype == torch.bfloat16) != bf and not (bf and len(ins) == 1 and v.t.shape[3] <= 4):
            raise TypeError("mixed fp32 / bf16 views in one launch")
    return _lib.GEMM_BF16 if bf else 0


@dataclass
class V:
    """A channel slice of an NHWC tensor (fp32 or bf16 storage), optionally on a strided pixel grid (struct unetpp_view)."""
    t: torch.Tensor
    c_off: int = 0
    c_len: Optional[int] = None
    sy: int = 1
    sx: int = 1
    oy: int = 0
    ox: int = 0
    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    gate: Optional[torch.Tensor] = None
    relu: bool = False
    accumulate: bool = False
    gate_sum: bool = False

    def fill(self, dst: View) -> int:
        t = _need(self.t, "view tensor", self.t.dtype if self.t.dtype in _ACT_DTYPES else torch.float32)
        if t.dim() != 4:
            raise ValueError("view tensor must be NHWC [N,H,W,C]")
        c = t.shape[3]
        n = c - self.c_off if self.c_len is None else self.c_len
        dst.ptr = t.data_ptr()
        dst.C, dst.c_off, dst.c_len = c, self.c_off, n
        dst.Hs, dst.Ws = t.shape[1], t.shape[2]
        dst.sy, dst.sx, dst.oy, dst.ox = self.sy, self.sx, self.oy, self.ox
        dst.scale = None if self.scale is None else _need(self.scale, "scale").data_ptr()
        dst.shift = None if self.shift is None else _need(self.shift, "shift").data_ptr()
        if self.gate is not None:
            g = _need(self.gate, "gate", t.dtype)
            if g.shape != t.shape:
                raise ValueError("gate must have the geometry of the gated tensor")
            dst.gate = g.data_ptr()
        else:
            dst.gate = None
        dst.relu = int(self.relu)
        dst.accumulate = int(self.accumulate)
        dst.gate_sum = int(self.gate_sum)
        return n


USE_FAST_GEMM = True  # tests flip this to exercise the generic kernel on the same shapes
# workgroups of the first layer's weight gradient (A/B knob).  Round 5: 1024 (four per CU) instead of 2048 -- every
# workgroup ends in a 28-row cross-thread sum that costs more than a patch of its arithmetic: 84 -> 69 us (fp32 headline),
# 106 -> 80 us (configs[4], 3 channels); 768 / 512 / 256 are slower again (75-145 us)
_SMALL_WGRAD_BLOCKS = int(os.environ.get("UNETPP_SMALL_WGRAD_BLOCKS", "1024"))
USE_WINOGRAD = os.environ.get("UNETPP_NO_WINOGRAD") is None  # 3x3 fast path: Winograd F(2x2,3x3) unless disabled


@dataclass
class WSrc:
    """A GEMM weight [taps][K][N] given by a parameter in its torch layout (struct unetpp_weight_src): element
    (tap, k, n) sits at tap' * s_t + (k % k_inner) * s_k + (k // k_inner) * s_ko + (n % n_inner) * s_n +
    (n // n_inner) * s_no, tap' = taps-1-tap when flip.  The fast kernels build their LDS image straight from it;
    `packed()` materialises the [taps][K][N] operand for the generic kernel."""
    t: torch.Tensor
    taps: int
    k: int
    n: int
    s_t: int
    s_k: int
    s_n: int
    s_ko: int = 0
    s_no: int = 0
    k_inner: int = 0
    n_inner: int = 0
    flip: bool = False
    pack: Optional[callable] = None  # () -> packed tensor

    def numel(self) -> int:
        return self.taps * self.k * self.n

    @property
    def device(self):
        return self.t.device

    def fill(self, dst: WeightSrc) -> None:
        dst.src = _need(self.t, "weight").data_ptr()
        dst.s_t, dst.s_k, dst.s_ko, dst.s_n, dst.s_no = self.s_t, self.s_k, self.s_ko, self.s_n, self.s_no
        dst.k_inner, dst.n_inner, dst.flip = self.k_inner, self.n_inner, int(self.flip)

    def packed(self) -> torch.Tensor:
        if self.pack is None:
            raise ValueError("this weight source has no packed form")
        return self.pack()


class PackPlan:
    """All weight images of a forward (or backward) pass in ONE launch (unetpp_gemm_pack_weight_images).

    The first time a launch is seen (engine pass `phase`, weight source, channel structure) its image is packed by
    itself into a persistent buffer and a job is recorded; from then on `begin(phase)` packs every recorded image of
    that pass with a single kernel before the pass starts and the launches just pick their buffer.

    The images are rebuilt on EVERY pass: nothing observable from Python says that a parameter's storage kept its
    contents (the reference's own optimizers update through ``p.data`` -- tools/optimizers/adamw.py:95-98,
    sgdw.py:108, adabound.py:120 -- which leaves ``p._version`` untouched, as do ``init.*_(m.weight.data)``,
    ``dist.broadcast(p.data)`` or EMA/clipping through ``.data``).  Skipping is an explicit opt-in for serving:
    ``frozen = True`` (UNet_Nested.freeze_weight_images) keeps the images until ``invalidate()``, which the module
    calls from ``load_state_dict``, ``_apply`` (.to()/.float()/.cuda()) and ``train()``.  Entries keep their source
    tensor alive and are dropped when unused for a few passes."""

    class _Entry:
        __slots__ = ("image", "n_img", "job", "src", "phase", "fresh", "used")

    class _Copy:  # a group of strided copies into one destination buffer (unetpp_copy_jobs), redone with the images
        __slots__ = ("dst", "jobs", "srcs", "srcsig", "phase", "fresh", "used", "elems")

    def __init__(self):
        self.entries = {}
        self.copies = {}     # key -> _Copy
        self._copy_tables = {}  # phase -> (device table, keys, max elements)
        self.copy_launches = 0  # batched copy launches issued (tests)
        self._tables = {}    # phase -> (device table, host array, signatures, max floats)
        self._packed = set()  # phases whose images are current (only consulted while frozen)
        self.frozen = False
        self.phase = None
        self.pass_id = 0
        self.launches = 0     # batched pack launches issued (tests)
        self.pinned = 0       # > 0 while a captured HIP graph holds raw pointers into this plan (serving.GraphedForward)
        self._retired = []    # job tables replaced while pinned: a captured pack launch may still read them

    def pin(self) -> None:
        """A captured graph bakes in the addresses of the weight images and of the device job table of its pack launch:
        while pinned, entries are not evicted and a replaced job table is kept alive instead of freed."""
        self.pinned += 1

    def unpin(self) -> None:
        self.pinned = max(0, self.pinned - 1)
        if not self.pinned:
            self._retired.clear()

    def invalidate(self) -> None:
        """The parameters may have changed: the next pass of every phase rebuilds its images."""
        self._packed.clear()

    def __deepcopy__(self, memo):  # copies / pickles of a model start with an empty plan
        return PackPlan()

    def __reduce__(self):
        return (PackPlan, ())

    def begin(self, phase: str) -> None:
        self.phase = phase
        self.pass_id += 1
        if not self.pinned:
            stale = [k for k, e in self.entries.items() if self.pass_id - e.used > 16]
            for k in stale:
                del self.entries[k]
            for k in [k for k, c in self.copies.items() if self.pass_id - c.used > 16]:
                del self.copies[k]
        self._begin_copies(phase)   # (first: weight images of this pass may be packed FROM a copied buffer)
        sigs = tuple(k for k, e in self.entries.items() if e.phase == phase)
        if not sigs:
            return
        ents = [self.entries[k] for k in sigs]
        tab = self._tables.get(phase)
        if tab is None or tab[2] != sigs:
            arr = (PackJob * len(ents))(*[e.job for e in ents])
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(ents[0].image.device)
            if self.pinned and phase in self._tables:
                self._retired.append(self._tables[phase])
            tab = (dev, arr, sigs, max(e.n_img for e in ents))
            self._tables[phase] = tab
            self._packed.discard(phase)
        if not (self.frozen and phase in self._packed):
            check(_lib.lib().unetpp_gemm_pack_weight_images(_ptr(tab[0]), len(ents), tab[3], _stream()),
                  "unetpp_gemm_pack_weight_images")
            self.launches += 1
            self._packed.add(phase)
        for e in ents:
            e.fresh = self.pass_id

    def _begin_copies(self, phase: str) -> None:
        keys = tuple(k for k, c in self.copies.items() if c.phase == phase)
        if not keys:
            return
        cps = [self.copies[k] for k in keys]
        tab = self._copy_tables.get(phase)
        if tab is None or tab[1] != keys:
            jobs = [j for c in cps for j in c.jobs]
            arr = (_lib.CopyJob * len(jobs))(*jobs)
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(cps[0].dst.device)
            if self.pinned and phase in self._copy_tables:
                self._retired.append(self._copy_tables[phase])
            tab = (dev, keys, max(c.elems for c in cps), len(jobs))
            self._copy_tables[phase] = tab
            self._packed.discard(phase)
        if not (self.frozen and phase in self._packed):
            check(_lib.lib().unetpp_copy_jobs(_ptr(tab[0]), tab[3], tab[2], _stream()), "unetpp_copy_jobs")
            self.copy_launches += 1
        for c in cps:
            c.fresh = self.pass_id

    def copy_for(self, key, phase: str, srcsig, make):
        """A destination buffer filled by strided copies that ``begin(phase)`` redoes in ONE launch from the second pass on.
        -> (dst, ready): ready = the buffer already holds this pass's contents.  The first time `key` is seen -- or when
        `srcsig` (whatever identifies the sources: their addresses) changed -- ``make()`` -> (dst tensor, source tensors
        kept alive, [(src tensor, src offset, dst offset, n_outer, n_inner, src_stride, dst_stride) in floats]) builds the
        job list; the caller fills the buffer itself for that pass (ready = False)."""
        c = self.copies.get(key)
        if c is None or c.srcsig != srcsig or c.phase != phase:
            dst, srcs, items = make()
            c = PackPlan._Copy()
            c.dst, c.srcs, c.srcsig, c.phase, c.fresh = dst, srcs, srcsig, phase, -1
            c.jobs, c.elems = [], 1
            for (src, s_off, d_off, n_outer, n_inner, s_stride, d_stride) in items:
                j = _lib.CopyJob()
                j.src, j.dst = src.data_ptr() + 4 * s_off, dst.data_ptr() + 4 * d_off
                j.n_outer, j.n_inner, j.src_stride, j.dst_stride = n_outer, n_inner, s_stride, d_stride
                if n_outer * n_inner >= 2 ** 31:
                    raise ValueError("copy job too large")
                c.jobs.append(j)
                c.elems = max(c.elems, n_outer * n_inner)
            self.copies[key] = c
            old_tab = self._copy_tables.pop(phase, None)   # same keys, other jobs: the table is rebuilt by the next begin()
            if old_tab is not None and self.pinned:
                self._retired.append(old_tab)
        c.used = self.pass_id
        return c.dst, c.fresh == self.pass_id

    def knows_copy(self, key, phase: str, srcsig) -> bool:
        """begin(phase) will fill this buffer (callers that run BEFORE begin: the grouped input-gradient weights)"""
        c = self.copies.get(key)
        return c is not None and c.srcsig == srcsig and c.phase == phase

    def image_for(self, sig, n_img: int, weight: "WSrc", d: GemmDesc):
        """-> (image tensor, already packed for this pass)"""
        e = self.entries.get(sig)
        if e is None:
            e = PackPlan._Entry()
            e.image = torch.empty(n_img, dtype=torch.float32, device=weight.device)
            e.n_img, e.src, e.phase, e.fresh = n_img, weight.t, self.phase, -1
            job = PackJob()
            weight.fill(job.src)
            job.image = e.image.data_ptr()
            job.taps, job.flags, job.n_in, job.n_out = d.taps, d.flags, d.n_in, d.n_out
            for i in range(d.n_in):
                job.in_len[i] = d.inp[i].c_len
            for i in range(d.n_out):
                job.out_len[i] = d.out[i].c_len
            e.job = job
            self.entries[sig] = e
        e.used = self.pass_id
        return e.image, e.fresh == self.pass_id


_PLAN: Optional[PackPlan] = None


def set_pack_plan(plan: Optional[PackPlan]) -> None:
    global _PLAN
    _PLAN = plan


def current_plan() -> Optional[PackPlan]:
    """the plan of the engine pass that is running, if any"""
    return _PLAN if (_PLAN is not None and _PLAN.phase is not None) else None


def gemm_pixel_blocks(n: int, h: int, w: int) -> int:
    return int(_lib.lib().unetpp_gemm_pixel_blocks(n, h, w))


@dataclass
class BatchNormFinish:
    """BatchNorm finalize attached to the convolution call that takes the statistics (struct unetpp_bn_fused): the call
    leaves mean / invstd / scale / shift and the updated running statistics behind; the persistent kernels write one
    row of sums per workgroup and the library finishes those few rows itself."""
    gamma: torch.Tensor
    beta: torch.Tensor
    running_mean: Optional[torch.Tensor]
    running_var: Optional[torch.Tensor]
    eps: float
    momentum: float
    count: int

    def outputs(self, c: int):
        dev = self.gamma.device
        self.mean, self.invstd, self.scale, self.shift = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(4))
        return self.mean, self.invstd, self.scale, self.shift


def gemm_stats_rows(n: int, h: int, w: int) -> int:
    return int(_lib.lib().unetpp_gemm_stats_rows(n, h, w))


def gemm_fwd(n: int, h: int, w: int, taps: int, ins: Sequence[V], outs: Sequence[V], weight: torch.Tensor,
             bias: Optional[torch.Tensor] = None, stats_partial: Optional[torch.Tensor] = None,
             direct: bool = False, bn: Optional[BatchNormFinish] = None) -> None:
    """direct=True forbids the Winograd form of the 3x3 fast path (bit-exact direct summation).
    bn: finalize the BatchNorm statistics inside this call (stats_partial then is workspace of gemm_stats_rows() rows)."""
    if len(ins) > MAX_VIEWS or len(outs) > MAX_VIEWS:
        raise ValueError("too many views")
    d = GemmDesc()
    d.N, d.H, d.W, d.taps = n, h, w, taps
    d.flags = (_lib.GEMM_DIRECT if (direct or not USE_WINOGRAD) else 0) | _storage_flag(outs, ins)
    d.n_in, d.n_out = len(ins), len(outs)
    k = sum(v.fill(d.inp[i]) for i, v in enumerate(ins))
    nc = sum(v.fill(d.out[i]) for i, v in enumerate(outs))
    from_src = isinstance(weight, WSrc)  # parameter in torch layout, or the packed [taps][K][N] operand
    if not from_src:
        _need(weight, "packed weight")
    if weight.numel() != taps * k * nc or (from_src and (weight.taps, weight.k, weight.n) != (taps, k, nc)):
        raise ValueError("weight has %d elements, expected %d*%d*%d" % (weight.numel(), taps, k, nc))
    if bias is not None and _need(bias, "bias").numel() != nc:
        raise ValueError("bias length mismatch")
    rows = gemm_pixel_blocks(n, h, w) if bn is None else gemm_stats_rows(n, h, w)
    if stats_partial is not None and _need(stats_partial, "stats").numel() != rows * nc * 2:
        raise ValueError("stats_partial size mismatch")
    if bn is not None:
        if stats_partial is None or len(outs) != 1:
            raise ValueError("a fused BatchNorm finalize needs stats_partial and a single output view")
        mean, invstd, scale, shift = bn.outputs(nc)
        for t, what in ((bn.gamma, "gamma"), (bn.beta, "beta")):
            if _need(t, what).numel() != nc:
                raise ValueError("BatchNorm %s length mismatch" % what)
        d.bn.gamma, d.bn.beta = bn.gamma.data_ptr(), bn.beta.data_ptr()
        d.bn.running_mean = None if bn.running_mean is None else _need(bn.running_mean, "running_mean").data_ptr()
        d.bn.running_var = None if bn.running_var is None else _need(bn.running_var, "running_var").data_ptr()
        d.bn.mean, d.bn.invstd, d.bn.scale, d.bn.shift = mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr()
        d.bn.count, d.bn.eps, d.bn.momentum = int(bn.count), float(bn.eps), float(bn.momentum)

    d.weight = None if from_src else weight.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    d.stats_partial = None if stats_partial is None else stats_partial.data_ptr()
    d.weight_image = None
    if USE_FAST_GEMM:
        lib = _lib.lib()
        n_img = int(lib.unetpp_gemm_weight_image_floats(C.byref(d)))
        if n_img > 0:  # aligned views: a fast kernel applies; its weight image is built in one launch
            ready = False
            if from_src and _PLAN is not None and _PLAN.phase is not None:
                sig = (weight.t.data_ptr(), weight.s_t, weight.s_k, weight.s_ko, weight.s_n, weight.s_no, weight.k_inner,
                       weight.n_inner, weight.flip, taps, d.flags, tuple(v.c_len for v in d.inp[:d.n_in]),
                       tuple(v.c_len for v in d.out[:d.n_out]))
                image, ready = _PLAN.image_for(sig, n_img, weight, d)
            else:
                image = torch.empty(n_img, dtype=torch.float32, device=weight.device)
            if ready:
                pass  # packed by PackPlan.begin() together with the other images of this pass
            elif from_src:
                ws = WeightSrc()
                weight.fill(ws)
                check(lib.unetpp_gemm_pack_weight_image_from(C.byref(d), C.byref(ws), _ptr(image), _stream()),
                      "unetpp_gemm_pack_weight_image_from")
            else:
                check(lib.unetpp_gemm_pack_weight_image(C.byref(d), _ptr(image), _stream()),
                      "unetpp_gemm_pack_weight_image")
            d.weight_image = image.data_ptr()
    if d.weight_image is None and from_src:  # generic kernel: it reads the packed operand
        packed = weight.packed()
        d.weight = packed.data_ptr()
    # operands the launch has to read besides its inputs: accumulated outputs and ReLU gates of the output views
    acc = sum(v.c_len for v in d.out[:d.n_out] if v.accumulate) + sum(v.c_len for v in d.out[:d.n_out] if v.gate)
    _timed_call(None, 2.0 * n * h * w * taps * k * nc,
                lambda: check(_lib.lib().unetpp_gemm_fwd(C.byref(d), _stream()), "unetpp_gemm_fwd"),
                (2.0 if d.flags & _lib.GEMM_BF16 else 4.0) * n * h * w * (k + nc + acc))


def first_layer_dgrad_bf16(dy: torch.Tensor, weight: torch.Tensor, dx: torch.Tensor) -> None:
    """dx fp32 [N,H,W,Cin<=4] = input gradient of the first 3x3 convolution from a bf16 dy [N,H,W,Cout]"""
    n, h, w, co = dy.shape
    ci = dx.shape[3]
    _need(dy, "dy", torch.bfloat16)
    _need(weight, "weight")
    _need(dx, "dx")
    if tuple(weight.shape) != (co, ci, 3, 3) or tuple(dx.shape[:3]) != (n, h, w):
        raise ValueError("shape mismatch")
    check(_lib.lib().unetpp_first_layer_dgrad_bf16(_ptr(dy), _ptr(weight), n, h, w, ci, co, _ptr(dx), _stream()),
          "unetpp_first_layer_dgrad_bf16")


def pack_weight(dst: torch.Tensor, src: torch.Tensor, t: int, k: int, n: int, dstr, sstr, flip: bool = False) -> None:
    _need(dst, "pack dst")
    _need(src, "pack src")
    check(_lib.lib().unetpp_pack_weight(_ptr(dst), _ptr(src), t, k, n, dstr[0], dstr[1], dstr[2],
                                        sstr[0], sstr[1], sstr[2], int(flip), _stream()), "unetpp_pack_weight")


def wgrad(n: int, h: int, w: int, taps: int, xs: Sequence[V], dys: Sequence[V], dw: Optional[torch.Tensor],
          dw_strides, db: Optional[torch.Tensor], n_inner: Optional[int] = None, target_blocks: int = 256,
          direct: bool = False) -> None:
    """dw / db are written (not accumulated).  dw_strides = (d_t, d_k, d_n, d_o) into the torch-layout gradient.
    direct=True forbids the Winograd form of the 3x3 kernel."""
    lib = _lib.lib()
    d = WgradDesc()
    d.N, d.H, d.W, d.taps = n, h, w, taps
    d.n_x, d.n_dy = len(xs), len(dys)
    k = sum(v.fill(d.x[i]) for i, v in enumerate(xs))
    nc = sum(v.fill(d.dy[i]) for i, v in enumerate(dys))
    pairs = sum((v.c_len + 31) // 32 for v in d.x[:len(xs)]) * sum((v.c_len + 31) // 32 for v in d.dy[:len(dys)])
    if len(xs) == 1 and xs[0].t.shape[3] <= 4:
        target_blocks = _SMALL_WGRAD_BLOCKS  # the 1..4-channel first layer: a VALU kernel with an expensive per-workgroup sum
    d.flags = (_lib.GEMM_DIRECT if (direct or not USE_WINOGRAD) else 0) | _storage_flag(dys, xs)
    wg_pairs = int(lib.unetpp_wgrad_pairs_per_workgroup(C.byref(d)))
    phys = C.c_int32(0)
    usable = int(lib.unetpp_usable_cus(C.byref(phys)))   # data parallel: CUs not left to the collective (all unless asked)
    if 0 < usable < phys.value:
        target_blocks = max(8, target_blocks * usable // phys.value)
    if dys[0].t.dtype == torch.bfloat16 and target_blocks == 256 and wg_pairs == 1:
        target_blocks = 512  # the bf16 pair kernel runs two 4-wave workgroups per CU (the quad kernel one of 8 waves)
    split = max(1, min(int(lib.unetpp_wgrad_max_split(n, h, w)), target_blocks // max(1, pairs // max(1, wg_pairs))))
    d.n_split = split
    planes = int(lib.unetpp_wgrad_slab_planes(C.byref(d)))  # taps, or 16 for the Winograd kernel
    slabs = torch.empty(split * (planes * k + 1) * nc, dtype=torch.float32, device=xs[0].t.device)
    d.slabs = slabs.data_ptr()
    _timed_call(None, 2.0 * n * h * w * taps * k * nc,
                lambda: check(lib.unetpp_wgrad(C.byref(d), _stream()), "unetpp_wgrad"),
                (2.0 if d.flags & _lib.GEMM_BF16 else 4.0) * n * h * w * (k + nc))
    if n_inner is None:
        n_inner = nc
    check(lib.unetpp_wgrad_finish(_ptr(slabs), split, planes, k, nc, n_inner, _ptr(dw), dw_strides[0], dw_strides[1],
                                  dw_strides[2], dw_strides[3], _ptr(db), _stream()), "unetpp_wgrad_finish")


def bn_finalize(partial, n_blocks, c, count, gamma, beta, eps, momentum, running_mean, running_var):
    dev = partial.device
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(4))
    check(_lib.lib().unetpp_bn_finalize(_ptr(partial), n_blocks, c, count, _ptr(gamma), _ptr(beta), eps, momentum,
                                        _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(invstd), _ptr(scale),
                                        _ptr(shift), _stream()), "unetpp_bn_finalize")
    return mean, invstd, scale, shift


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps):
    c = gamma.numel()
    scale, shift = (torch.empty(c, dtype=torch.float32, device=gamma.device) for _ in range(2))
    check(_lib.lib().unetpp_bn_eval_coeffs(_ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), eps, c,
                                           _ptr(scale), _ptr(shift), _stream()), "unetpp_bn_eval_coeffs")
    return scale, shift


def _is_bf16(t):
    return t.dtype == torch.bfloat16


def affine_relu_pool(y, scale, shift, relu, act, pooled, pool_idx):
    n, h, w, c = y.shape
    if _is_bf16(y):
        check(_lib.lib().unetpp_affine_relu_pool_bf16(_ptr(y), _ptr(scale), _ptr(shift), int(relu), n, h, w, c, _ptr(act),
                                                      _ptr(pooled), _ptr(pool_idx), _stream()),
              "unetpp_affine_relu_pool_bf16")
        return
    check(_lib.lib().unetpp_affine_relu_pool(_ptr(y), _ptr(scale), _ptr(shift), int(relu), n, h, w, c, _ptr(act),
                                             _ptr(pooled), _ptr(pool_idx), _stream()), "unetpp_affine_relu_pool")


def maxpool_bwd(d_pooled, pool_idx, d_act, gate=None):
    """d_act[window argmax] += d_pooled; bf16 storage: optionally followed by d_act *= (gate > 0) (the node's ReLU mask)"""
    n, h, w, c = d_act.shape
    if _is_bf16(d_act):
        check(_lib.lib().unetpp_maxpool_bwd_bf16(_ptr(d_pooled), _ptr(pool_idx), n, h, w, c, _ptr(d_act), _ptr(gate), _stream()),
              "unetpp_maxpool_bwd_bf16")
        return
    if gate is not None:
        raise ValueError("the fused ReLU mask is a bf16-storage option")
    check(_lib.lib().unetpp_maxpool_bwd(_ptr(d_pooled), _ptr(pool_idx), n, h, w, c, _ptr(d_act), _stream()),
          "unetpp_maxpool_bwd")


def _bn_backward_bf16(d_act, y, scale, shift, mean, invstd, gamma, dy_out, dgamma, dbeta, pool):
    lib = _lib.lib()
    n, h, w, c = y.shape
    blocks = int(lib.unetpp_bn_bwd_blocks_bf16(n * h * w, c))
    if blocks < 1:
        raise ValueError("bf16 BatchNorm backward needs C = 8 * 2^k channels, got %d" % c)
    partial = torch.empty(blocks * c * 2, dtype=torch.float32, device=y.device)
    dp, pi = (None, None) if pool is None else (_need(pool[0], "d_pooled", torch.bfloat16), _need(pool[1], "pool_idx", torch.uint8))
    st = _stream()
    check(lib.unetpp_bn_bwd_reduce_bf16(_ptr(d_act), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), _ptr(dp),
                                        _ptr(pi), n, h, w, c, _ptr(partial), st), "unetpp_bn_bwd_reduce_bf16")
    check(lib.unetpp_bn_bwd_finalize(_ptr(partial), blocks, c, _ptr(dgamma), _ptr(dbeta), st), "unetpp_bn_bwd_finalize")
    check(lib.unetpp_bn_bwd_apply_bf16(_ptr(d_act), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                       _ptr(dgamma), _ptr(dbeta), _ptr(dp), _ptr(pi), n, h, w, c, _ptr(dy_out), st),
          "unetpp_bn_bwd_apply_bf16")
    return dgamma, dbeta


def bn_backward(d_act, y, scale, shift, mean, invstd, gamma, dy_out, dgamma=None, dbeta=None, pool=None):
    """Training-mode BatchNorm+ReLU backward.  Returns (dgamma, dbeta); dy_out may alias d_act.
    pool = (d_pooled, pool_idx): the gradient of the 2x2 max-pool of this activation, routed to the argmax while
    d_act is read (both passes) when the shape allows, else by a maxpool_bwd pass into d_act first."""
    lib = _lib.lib()
    n, h, w, c = y.shape
    pixels = n * h * w
    if dgamma is None:
        dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
    if dbeta is None:
        dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
    if _is_bf16(y):
        return _bn_backward_bf16(d_act, y, scale, shift, mean, invstd, gamma, dy_out, dgamma, dbeta, pool)
    if pool is not None and not lib.unetpp_bn_bwd_pool_ok(n, h, w, c):
        maxpool_bwd(pool[0], pool[1], d_act)
        pool = None
    blocks = int(lib.unetpp_bn_bwd_blocks(pixels, c))
    partial = torch.empty(blocks * c * 2, dtype=torch.float32, device=y.device)
    st = _stream()
    if pool is None:
        check(lib.unetpp_bn_bwd_reduce(_ptr(d_act), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), pixels,
                                       c, _ptr(partial), st), "unetpp_bn_bwd_reduce")
    else:
        d_pooled, pool_idx = _need(pool[0], "d_pooled"), _need(pool[1], "pool_idx", torch.uint8)
        check(lib.unetpp_bn_bwd_reduce_pool(_ptr(d_act), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd),
                                            _ptr(d_pooled), _ptr(pool_idx), n, h, w, c, _ptr(partial), st),
              "unetpp_bn_bwd_reduce_pool")
    check(lib.unetpp_bn_bwd_finalize(_ptr(partial), blocks, c, _ptr(dgamma), _ptr(dbeta), st), "unetpp_bn_bwd_finalize")
    if pool is None:
        check(lib.unetpp_bn_bwd_apply(_ptr(d_act), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd),
                                      _ptr(gamma), _ptr(dgamma), _ptr(dbeta), pixels, c, _ptr(dy_out), st),
              "unetpp_bn_bwd_apply")
    else:
        check(lib.unetpp_bn_bwd_apply_pool(_ptr(d_act), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd),
                                           _ptr(gamma), _ptr(dgamma), _ptr(dbeta), _ptr(pool[0]), _ptr(pool[1]), n, h,
                                           w, c, _ptr(dy_out), st), "unetpp_bn_bwd_apply_pool")
    return dgamma, dbeta


def head_fwd(x, weight, bias, p_drop, seed, mask, out_nchw, seed_dev=None):
    """seed_dev: optional one-element int64 device tensor added to `seed` when the kernel runs (graph-captured steps)"""
    n, h, w, c = x.shape
    n_cls = weight.shape[0]
    fn = _lib.lib().unetpp_head_fwd_bf16 if _is_bf16(x) else _lib.lib().unetpp_head_fwd
    check(fn(_ptr(x), _ptr(weight), _ptr(bias), n, h, w, c, n_cls, float(p_drop), C.c_uint64(seed), _ptr(mask),
             _ptr(seed_dev), _ptr(out_nchw), _stream()), "unetpp_head_fwd")


def head_bwd(d_out, out, x, weight, p_drop, seed, mask, dx, accumulate, gate_x=False, seed_dev=None):
    """Returns (dW [n_cls, C, 1, 1], db [n_cls]); dx is written or accumulated in place."""
    lib = _lib.lib()
    n, h, w, c = x.shape
    n_cls = weight.shape[0]
    blocks = int(lib.unetpp_head_bwd_blocks(n * h * w))
    ln = n_cls * c + n_cls
    partial = torch.empty(blocks * ln, dtype=torch.float32, device=x.device)
    sums = torch.empty(ln, dtype=torch.float32, device=x.device)
    st = _stream()
    fn = lib.unetpp_head_bwd_bf16 if _is_bf16(x) else lib.unetpp_head_bwd
    check(fn(_ptr(d_out), _ptr(out), _ptr(x), _ptr(weight), n, h, w, c, n_cls, float(p_drop), C.c_uint64(seed), _ptr(mask),
             _ptr(seed_dev), _ptr(dx), int(accumulate), int(gate_x), _ptr(partial), st), "unetpp_head_bwd")
    check(lib.unetpp_sum_partials(_ptr(partial), blocks, ln, _ptr(sums), st), "unetpp_sum_partials")
    return sums[:n_cls * c].view(n_cls, c, 1, 1), sums[n_cls * c:]


def bilinear2x_fwd(x, y):
    n, h, w, c = x.shape
    if _is_bf16(x):
        check(_lib.lib().unetpp_bilinear2x_fwd_bf16(_ptr(x), n, h, w, c, _ptr(y), _stream()), "unetpp_bilinear2x_fwd_bf16")
        return
    check(_lib.lib().unetpp_bilinear2x_fwd(_ptr(x), n, h, w, c, _ptr(y), _stream()), "unetpp_bilinear2x_fwd")


def bilinear2x_bwd(dy, dx, accumulate, gate=None):
    """dx (+)= stencil^T(dy); bf16 storage: optionally followed by dx *= (gate > 0) (the node's ReLU mask)"""
    n, h, w, c = dx.shape
    if _is_bf16(dx):
        check(_lib.lib().unetpp_bilinear2x_bwd_bf16(_ptr(dy), n, h, w, c, _ptr(dx), int(accumulate), _ptr(gate), _stream()),
              "unetpp_bilinear2x_bwd_bf16")
        return
    if gate is not None:
        raise ValueError("the fused ReLU mask is a bf16-storage option")
    check(_lib.lib().unetpp_bilinear2x_bwd(_ptr(dy), n, h, w, c, _ptr(dx), int(accumulate), _stream()),
          "unetpp_bilinear2x_bwd")


def nchw_to_nhwc(src):
    n, c, h, w = src.shape
    _need(src, "input")
    if c == 1:
        return src.view(n, h, w, 1)  # same bytes
    dst = torch.empty((n, h, w, c), dtype=torch.float32, device=src.device)
    check(_lib.lib().unetpp_nchw_to_nhwc(_ptr(src), n, c, h, w, _ptr(dst), _stream()), "unetpp_nchw_to_nhwc")
    return dst


def nhwc_to_nchw(src):
    n, h, w, c = src.shape
    if c == 1:
        return src.view(n, 1, h, w)
    dst = torch.empty((n, c, h, w), dtype=torch.float32, device=src.device)
    check(_lib.lib().unetpp_nhwc_to_nchw(_ptr(src), n, c, h, w, _ptr(dst), _stream()), "unetpp_nhwc_to_nchw")
    return dst


def focal_bce(pred: torch.Tensor, target: torch.Tensor, rows: int, gamma: float, want_grad: bool = True):
    """FocalLoss_BCE_2d value (0-dim tensor) and d loss / d pred in one pass over (pred, target)."""
    lib = _lib.lib()
    _need(pred, "pred")
    _need(target, "target")
    if pred.shape != target.shape:
        raise ValueError("pred and target must have the same shape")
    n = pred.numel()
    blocks = int(lib.unetpp_focal_bce_blocks(n))
    partial = torch.empty(blocks, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    check(lib.unetpp_focal_bce(_ptr(pred), _ptr(target), n, rows, float(gamma), _ptr(grad), _ptr(partial), _ptr(loss),
                               _stream()), "unetpp_focal_bce")
    return loss.reshape(()), grad


def focal_bce_heads(preds, target: torch.Tensor, rows: int, gamma: float, want_grad: bool = True):
    """The trainer's loop over the deep-supervision heads (criterion on every head, mean over heads) in one launch:
    -> (losses [1 + heads]: the mean, then every head's FocalLoss_BCE_2d value; d mean / d pred per head or None).
    Bit for bit what the loop computes with focal_bce and tensor arithmetic (tests/test_gpu_caller.py)."""
    lib = _lib.lib()
    if not 1 <= len(preds) <= _lib.MAX_HEADS:
        raise ValueError("1 to %d heads" % _lib.MAX_HEADS)
    _need(target, "target")
    for p in preds:
        _need(p, "pred")
        if p.shape != target.shape:
            raise ValueError("pred and target must have the same shape")
    n = target.numel()
    blocks = int(lib.unetpp_focal_bce_blocks(n))
    partial = torch.empty(len(preds) * blocks, dtype=torch.float32, device=target.device)
    grads = [torch.empty_like(p) for p in preds] if want_grad else None
    loss = torch.empty(1 + len(preds), dtype=torch.float32, device=target.device)
    hd = _lib.FocalHeads()
    hd.n_heads = len(preds)
    for i, p in enumerate(preds):
        hd.pred[i] = p.data_ptr()
        hd.grad[i] = grads[i].data_ptr() if want_grad else None
    check(lib.unetpp_focal_bce_heads(C.byref(hd), _ptr(target), n, rows, float(gamma), _ptr(partial), _ptr(loss),
                                     _stream()), "unetpp_focal_bce_heads")
    return loss, grads


def create_heatmap(points: torch.Tensor, height: int, width: int, radius: float = 3.0) -> torch.Tensor:
    """points [N, P, 2] (x, y) fp32 on the GPU -> target heat maps [N, 4, H, W] (tools/misc/helper.py:87-172)."""
    lib = _lib.lib()
    _need(points, "points")
    if points.dim() != 3 or points.shape[2] != 2 or points.shape[1] < 6:
        raise ValueError("points must be [N, P >= 6, 2]")
    n, p = points.shape[0], points.shape[1]
    out = torch.empty(n, 4, height, width, dtype=torch.float32, device=points.device)
    ws = torch.empty(int(lib.unetpp_heatmap_workspace_bytes(n, height, width)), dtype=torch.uint8, device=points.device)
    check(lib.unetpp_create_heatmap(_ptr(points), n, p, height, width, float(radius), _ptr(out), _ptr(ws), _stream()),
          "unetpp_create_heatmap")
    return out


def heatmap_pattern(points: torch.Tensor, pattern, height: int, width: int, radius: float = 3.0) -> torch.Tensor:
    """points [N, P, 2] (x, y) fp32 on the GPU, pattern = list of lists of key-point indices -> [N, len(pattern), H, W]
    (tools/misc/heatmap.py:203-230)."""
    lib = _lib.lib()
    _need(points, "points")
    if points.dim() != 3 or points.shape[2] != 2:
        raise ValueError("points must be [N, P, 2]")
    n, p = points.shape[0], points.shape[1]
    flat = [i for hmap in pattern for i in hmap]
    if not flat or min(flat) < 0 or max(flat) >= p or any(len(h) == 0 for h in pattern):
        raise ValueError("pattern indexes key points 0..%d, every map needs at least one" % (p - 1))
    begins = [0]
    for hmap in pattern:
        begins.append(begins[-1] + len(hmap))
    mp = torch.tensor(flat, dtype=torch.int32, device=points.device)
    mb = torch.tensor(begins, dtype=torch.int32, device=points.device)
    out = torch.empty(n, len(pattern), height, width, dtype=torch.float32, device=points.device)
    ws = torch.empty(int(lib.unetpp_heatmap_pattern_workspace_bytes(n, len(pattern), height, width)), dtype=torch.uint8,
                     device=points.device)
    check(lib.unetpp_heatmap_pattern(_ptr(points), n, p, _ptr(mp), _ptr(mb), len(pattern), height, width, float(radius),
                                     _ptr(out), _ptr(ws), _stream()), "unetpp_heatmap_pattern")
    return out


def _keypoints_stages(heat, thr, num, max_regions, segmentation, ws, changed, points, counts, select=True):
    """the staged C-ABI calls of one extraction (include/unetpp_hip.h: unetpp_keypoints_extract)"""
    lib = _lib.lib()
    maps, h, w = heat.shape
    st = _stream()
    args = (_ptr(heat), maps, h, w, _ptr(thr), num, max_regions)

    def stage(k, what, sweeps=1):
        check(lib.unetpp_keypoints_extract(k, *args, sweeps, _ptr(ws), _ptr(changed), _ptr(points), _ptr(counts), st), what)

    def until_stable(k, what, sweeps):
        # A sweep moves a label / a distance at least one pixel along its path, and a path through a mask is at most
        # h * w / 2 pixels long (a one-pixel serpentine): that many sweeps bound the loop.  Every batch ends in a blocking
        # read-back, so the batches double (normally 2-3 of them suffice).
        done, limit = 0, h * w // 2 + sweeps
        while done < limit:
            changed.zero_()
            stage(k, what, sweeps)
            done += sweeps
            if int(changed.item()) == 0:
                return
            sweeps = min(2 * sweeps, 256)
        raise RuntimeError("keypoints_extract: %s did not converge" % what)

    stage(0, "keypoints mask")
    if segmentation == "watershed":
        stage(3, "keypoints distance init")
        until_stable(4, "keypoints distance sweeps", 8)
        stage(5, "keypoints cores")
    until_stable(1, "keypoints merge", 4)
    if segmentation == "watershed":
        stage(6, "keypoints markers")
        until_stable(7, "keypoints flood", 4)
        stage(8, "keypoints regions")
    if select:
        stage(2, "keypoints select")


def keypoints_regions(heat: torch.Tensor, threshold: float = 0.5, segmentation: str = "watershed"):
    """heat [maps, H, W] fp32 on the GPU -> (labels int32 [maps, H, W]: the raster index of the region's first (core) pixel,
    -1 outside every region; distance int32 [maps, H, W]: the 3x3 chamfer distance in 16-bit fixed point, zeros for
    "components").  The region step of the extraction alone (region_segment_, tools/misc/heatmap.py:100-144, returns
    one boolean mask per region: ``labels == root`` here)."""
    lib = _lib.lib()
    _need(heat, "heat")
    if heat.dim() != 3:
        raise ValueError("heat must be [maps, H, W]")
    if segmentation not in ("watershed", "components"):
        raise ValueError("segmentation must be 'watershed' or 'components'")
    maps, h, w = heat.shape
    dev, hw = heat.device, h * w
    ws = torch.zeros(int(lib.unetpp_keypoints_workspace_bytes(maps, h, w, 1)), dtype=torch.uint8, device=dev)
    changed = torch.zeros(1, dtype=torch.int32, device=dev)
    points = torch.empty(maps, 1, 2, dtype=torch.float32, device=dev)
    counts = torch.empty(maps, dtype=torch.int32, device=dev)
    thr = torch.full((maps,), float(threshold), dtype=torch.float32, device=dev)
    _keypoints_stages(heat, thr, 1, 1, segmentation, ws, changed, points, counts, select=False)
    ints = ws[maps * hw * 8 + maps * 16:].view(torch.int32)   # workspace: best u64 [maps*hw], cand u64 [maps*max_regions*2], label, dist, marker
    return ints[:maps * hw].view(maps, h, w).clone(), ints[maps * hw:2 * maps * hw].view(maps, h, w).clone()


def keypoints_extract(heat: torch.Tensor, num: int, threshold: float = 0.5, max_regions: int = 4096,
                      segmentation: str = "watershed"):
    """heat [maps, H, W] fp32 on the GPU -> (points [maps, num, 2] as (x, y), -1 padded; counts [maps] regions found).
    tools/misc/heatmap.py:148-200; a map without any region is retried once at 0.9 * threshold (heatmap.py:176-198).
    segmentation="watershed": the reference's region step (region_segment_, heatmap.py:100-144) -- chamfer distance
    cores grown back through the binary mask, touching blobs split, coreless blobs dropped; "components": a region is
    an 8-connected component of the mask (the round-2 stand-in).  max_regions sizes the fast ranking buffer only: maps
    with more regions are selected exactly over all of them (no failure mode the reference does not have)."""
    lib = _lib.lib()
    _need(heat, "heat")
    if heat.dim() != 3:
        raise ValueError("heat must be [maps, H, W]")
    if segmentation not in ("watershed", "components"):
        raise ValueError("segmentation must be 'watershed' or 'components'")
    maps, h, w = heat.shape
    dev = heat.device
    ws = torch.empty(int(lib.unetpp_keypoints_workspace_bytes(maps, h, w, max_regions)), dtype=torch.uint8, device=dev)
    changed = torch.zeros(1, dtype=torch.int32, device=dev)
    points = torch.empty(maps, num, 2, dtype=torch.float32, device=dev)
    counts = torch.empty(maps, dtype=torch.int32, device=dev)
    thr = torch.full((maps,), float(threshold), dtype=torch.float32, device=dev)

    def run():
        _keypoints_stages(heat, thr, num, max_regions, segmentation, ws, changed, points, counts)

    run()
    # one small read-back per call: are there maps without any region?  (the sweeps above already read a flag back
    # every few iterations; everything else -- mask, labels, peaks, the exact top-`num` selection over ALL regions --
    # stays on the device, and there is no limit on the number of regions: kp_select_kernel)
    if bool((counts == 0).any()):  # retry the empty maps at 0.9 * threshold; the others keep their result
        keep_points, keep_counts = points.clone(), counts.clone()
        empty = (counts == 0)
        thr = torch.where(empty, thr * 0.9, thr)
        run()
        points = torch.where(empty.view(-1, 1, 1), points, keep_points)
        counts = torch.where(empty, counts, keep_counts)
    return points, counts
