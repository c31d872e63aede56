"""Tensor-level wrappers over the C-ABI (include/scrabble_hip.h).

torch is used for device memory and the current HIP stream only; every arithmetic op below is a
hand-written gfx950 kernel in libscrabble_hip.so.  All tensors are fp32, contiguous, NHWC, on the
GPU.  Nothing here falls back to torch math: a missing library or a failed launch raises."""
from __future__ import annotations

import os as _os
from typing import Optional, Tuple

import torch

from ._lib import call, lib

RELU_IN, ACCUM, RELU_OUT, TANH_OUT = 1, 2, 4, 8
MMA_BF16 = 256
BN_EPS = 1e-3


class KernelTimer:
    """HIP-event timing of kernel families on the stream the kernels are launched on (bench.py's
    roofline leg).  `flops` is the algorithmic 2*M*N*K of the launch."""

    def __init__(self, only=None):
        self.records = {}
        self.shapes = {}
        self.nbytes = {}        # algorithmic bytes (operands read once + result written once) per family
        self.only = only        # None = every family; else the set of family-name prefixes to time
        self.roof = {}          # family -> "mfma" | "hbm" | "valu"

    def wants(self, family):
        return self.only is None or any(family.startswith(p) for p in self.only)

    class _Region:
        def __init__(self, timer, family, flops, tag=None, nbytes=0.0):
            self.t, self.family, self.flops, self.tag, self.nbytes = timer, family, flops, tag, nbytes

        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())

        def __exit__(self, *exc):
            self.b.record(torch.cuda.current_stream())
            self.t.records.setdefault(self.family, []).append((self.a, self.b, self.flops))
            self.t.nbytes[self.family] = self.t.nbytes.get(self.family, 0.0) + self.nbytes
            if self.tag is not None:
                self.t.shapes.setdefault((self.family, self.tag), []).append((self.a, self.b, self.flops))

    def region(self, family, flops, tag=None, nbytes=0.0):
        return KernelTimer._Region(self, family, flops, tag, nbytes)

    def by_shape(self):
        """[(family, tag, launches, ms, tflops)] sorted by time, after a device synchronize."""
        torch.cuda.synchronize()
        rows = []
        for (fam, tag), recs in self.shapes.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in recs)
            fl = sum(f for _, _, f in recs)
            rows.append((fam, tag, len(recs), ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0))
        return sorted(rows, key=lambda r: -r[3])

    def summary(self):
        """{family: {launches, ms, tflops}} after a device synchronize."""
        torch.cuda.synchronize()
        out = {}
        for fam, recs in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in recs)
            fl = sum(f for _, _, f in recs)
            out[fam] = {"launches": len(recs), "ms": ms, "flops": fl, "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                        "bytes": self.nbytes.get(fam, 0.0)}
        return out


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _Null()
PROFILER: Optional[KernelTimer] = None


def _timed(family: str, flops: float, thin: bool, tag=None, tensors=()):
    if PROFILER is None:
        return _NULL
    fam = family + ("_thin" if thin else "")
    if not PROFILER.wants(fam):
        return _NULL
    PROFILER.roof[fam] = "hbm" if thin else "mfma"
    return PROFILER.region(fam, flops, tag, float(sum(t.numel() * t.element_size() for t in tensors if t is not None)))


def _hbm(family: str, *tensors, flops: float = 0.0, roof: str = "hbm"):
    """Timed region of a memory-bound kernel family: algorithmic bytes = every operand read once + every result
    written once (bench.py reports GB/s against the 8 TB/s HBM peak)."""
    if PROFILER is None or not PROFILER.wants(family):
        return _NULL
    PROFILER.roof[family] = roof
    return PROFILER.region(family, flops, None, float(sum(t.numel() * t.element_size() for t in tensors if t is not None)))


_DEV_INDEX = None


def _stream() -> int:
    """Raw handle of torch's current HIP stream.  (torch.cuda.current_stream() builds a Stream object and resolves the
    device through several Python layers: ~30 us per call, i.e. 15-20 ms of host time per train_step.)"""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEV_INDEX)


# ---- side stream for the weight-gradient launches -------------------------------------------------------------------
# Weight gradients are LEAVES of a backward sweep: nothing reads dW before the sweep's all-reduce / optimizer update, while the
# data-grad chain is sequentially dependent.  In fp32 mode they are queued on a second HIP stream (ordered after the kernel
# that produced their gradient operand by an event), so the tail of a data-grad launch -- the last, partly filled round of
# workgroups over the 256 CUs, a third of the time of a 4x20 layer at the 8-way shard batch -- overlaps weight-grad
# workgroups instead of idling (the fp32 kernels keep 2-3 workgroups per CU resident, so two launches share a CU).  Off when
# kernel timing is active (HIP-event brackets would measure overlapped launches), in deterministic mode, and in bf16 / fp8
# modes (operand copies made on one stream are read on the other), and for launches of more than SIDE_MAX_BATCH samples: measured
# on MI355X (profiles/r03_side_stream.txt), per-GPU batch 16: 52.98 -> 52.04 ms / step (+1.8 %); batch 128: 346.1 -> 349.8 ms
# (-1 %: the grids fill many rounds of the chip, two resident kernels only compete for L2).
# Round 3, after the Winograd path: OFF by default.  The shard-size step shrank to ~32 ms of GPU work and the 13-37 stream waits a step
# needs cost the host 0.3-1 ms EACH (hipStreamWaitEvent, box-dependent: profiles/r03_host_profile_bs16.txt, r03_streams_bs16.txt) -- on a
# slow host the step became host-bound (43.8 ms with the side streams against 40.1 ms on one stream) for a gain of 1.8 % on a fast one.
# The network stream below (a handful of joins per step, +3.4 %) stays on.  SG_SIDE_WGRAD=1 turns the side streams back on.
SIDE_WGRAD = _os.environ.get("SG_SIDE_WGRAD", "0") == "1"
SIDE_MAX_BATCH = int(_os.environ.get("SG_SIDE_MAX_BATCH", "96"))
_SIDE = {}             # raw handle of the stream that is "main" for a sweep -> {"stream": its side stream, "dirty": bool}
CAPTURING = False      # graph_step.GraphedStep sets this while a step is captured into a HIP graph (single stream)


# Round 4: the network stream in bf16 / fp8 mode too (c3: 1 774 -> 1 836 img/s).  What crosses streams there are the operand copies
# ("twins") -- made and read inside one network's passes, i.e. on one stream; tensors shared by two networks (x_f, the real images)
# enter each network's pass through its own concatenated batch, so no copy made on one stream is read on the other before a join --
# and the fp8 amax slots, pooled per stream (_amax_slot).  tests/test_train_loop_gpu.py::test_network_and_side_streams_do_not_change_
# the_step[bf16] holds the two-stream step to the single-stream one.  SG_STREAMS_LOWP=0: fp32 mode only, as in round 3.
STREAMS_LOWP = _os.environ.get("SG_STREAMS_LOWP", "1") == "1"


def _streams_ok() -> bool:
    return PROFILER is None and not DETERMINISTIC and (CONV_DTYPE == "f32" or STREAMS_LOWP) and not CAPTURING


def side_enabled() -> bool:
    return SIDE_WGRAD and _streams_ok() and CONV_DTYPE == "f32"        # (operand copies made for a weight-grad would be read on the sweep's stream)


def _side_of_current():
    key = torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    e = _SIDE.get(key)
    if e is None:
        e = {"stream": torch.cuda.Stream(), "dirty": False}
        _SIDE[key] = e
    return e


class side_stream:
    """with ops.side_stream(t1, t2, ...): the launches inside run on the side stream OF THE CURRENT STREAM, after everything queued
    so far on the current stream (in particular the producers of t1, t2, ... -- tensors the side launches read: their memory is
    not handed out again before the side stream is done with them).  A no-op context when side_enabled() is False."""

    def __init__(self, *inputs):
        self.on = side_enabled() and inputs[0].shape[0] <= SIDE_MAX_BATCH
        self.inputs = inputs

    def __enter__(self):
        if not self.on:
            return self
        e = _side_of_current()
        side = e["stream"]
        self.main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(self.main)
        side.wait_event(ev)
        for t in self.inputs:
            if t is not None:
                t.record_stream(side)
        torch.cuda.set_stream(side)
        e["dirty"] = True
        return self

    def __exit__(self, *exc):
        if self.on:
            torch.cuda.set_stream(self.main)
        return False


def side_join() -> None:
    """The current stream waits for everything queued on ITS side stream (called before gradients are exchanged / applied)."""
    e = _SIDE.get(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
    if e is not None and e["dirty"]:
        ev = torch.cuda.Event()
        ev.record(e["stream"])
        torch.cuda.current_stream().wait_event(ev)
        e["dirty"] = False


# ---- a second stream for a whole network's passes ------------------------------------------------------------------------
# At the data-parallel shard batch a conv launch fills the 256 CUs for 1-3 rounds of workgroups; the last, partly filled round
# costs the fp32 kernels 11-17 % against the bs-128 step (111 vs 133 TF/s).  The discriminator and the style promoter are two
# networks of the same size whose passes depend on each other nowhere between the generator's forward and the loss head, and
# again between the loss head and the generator's backward: train_step queues S's fused forward (and later its backward sweep)
# on this stream and D's / R's on the launch stream, so the tails of one network's launches are filled by the other's.
# Same switch and conditions as the weight-grad side stream (fp32 mode, no kernel timing, not deterministic, small launches).
NET_STREAM = _os.environ.get("SG_NET_STREAM", "1") == "1"
_NET = {"stream": None}


# Round 4: at EVERY batch size.  At the headline batch the launches fill the chip many times over, but a network pass alternates
# matrix-bound launches (the grouped Winograd products) with HBM-bound ones (the transform sweeps at 7 TB/s, pools, BN); with S on its
# own stream the HBM-bound launches of one network run under the matrix-bound ones of the other: 148.5 -> 140.4 ms / step at bs 128 on
# the same box (profiles/r04_streams_bs128.txt; the weight-grad side streams on top of it: 143.5 ms -- they stay off).
NET_MAX_BATCH = int(_os.environ.get("SG_NET_MAX_BATCH", str(1 << 30)))


def net_stream_enabled(batch: int) -> bool:
    return NET_STREAM and _streams_ok() and batch <= NET_MAX_BATCH


class net_stream:
    """with ops.net_stream(inputs...) as ns: launches run on the network stream after everything queued so far on the launch
    stream; ns.join(outputs...) afterwards makes the launch stream wait for them (and tells the allocator that the outputs,
    allocated on the network stream, are read on the launch stream)."""

    def __init__(self, *inputs):
        self.inputs = inputs

    def __enter__(self):
        if _NET["stream"] is None:
            _NET["stream"] = torch.cuda.Stream()
        ns = _NET["stream"]
        self.main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(self.main)
        ns.wait_event(ev)
        for t in self.inputs:
            if t is not None:
                t.record_stream(ns)
        torch.cuda.set_stream(ns)
        return self

    def __exit__(self, *exc):
        self.done = torch.cuda.Event()
        self.done.record(_NET["stream"])
        torch.cuda.set_stream(self.main)
        return False

    def join(self, *outputs):
        self.main.wait_event(self.done)
        for t in outputs:
            if t is not None:
                t.record_stream(self.main)


_GHOSTS = set()        # storage pointers of the never-written fp32 handles behind operand-only results (bf16 / fp8 modes), this step


def _p(t: Optional[torch.Tensor]):
    """Raw device pointer of an fp32 tensor for the C-ABI.  A never-written handle of an operand-only result (conv2d_fwd(want16=
    'only'), avgpool2_bwd_operands) has no fp32 contents: every consumer must read its bf16 / fp8 copy, so handing its fp32
    pointer to a kernel is a routing bug (e.g. the mode flags changed between forward and backward) and fails here, loudly,
    instead of computing on uninitialised memory."""
    if t is None:
        return None
    if _GHOSTS and t.untyped_storage().data_ptr() in _GHOSTS:
        raise RuntimeError("fp32 contents of an operand-only result requested (shape %s): its consumers must read the bf16 / fp8 copy; "
                           "conv dtype / deterministic mode changed between the forward and the backward pass?" % (tuple(t.shape),))
    return t.data_ptr()


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise ValueError("expected a contiguous fp32 CUDA tensor, got %s %s contiguous=%s"
                             % (t.device, t.dtype, t.is_contiguous()))


def _flags(relu_in=False, accum=False, relu_out=False, tanh_out=False) -> int:
    return (RELU_IN if relu_in else 0) | (ACCUM if accum else 0) | (RELU_OUT if relu_out else 0) | (TANH_OUT if tanh_out else 0)


def empty(*shape, like: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    return torch.empty(shape, device=like.device, dtype=dtype)


# ---------------------------------------------------------------- convolutions
# Matrix-core operand type of the 3x3 / 1x1 convolutions: "f32" (v_mfma_f32_32x32x2_f32, the parity mode) or "bf16"
# (v_mfma_f32_32x32x16_bf16 with fp32 accumulation: BASELINE config c3).  Tensors in HBM are fp32 either way.
CONV_DTYPE = "f32"
# fp32 mode: let the data-grad launches of the >= 128-channel layers read a transposed filter copy (straight [K,N] loader
# instead of the transposing one).  Measured on MI355X: no gain (129.7 vs 130.0 TF/s in-step), so it is off by default.
TRANSPOSED_DGRAD_FILTERS = _os.environ.get("SG_DGRAD_WT", "0") == "1"
_PACK_CACHE = {}


def set_conv_dtype(dtype: str) -> None:
    """'f32' (parity mode), 'bf16' (config c3) or 'fp8' (config c5, first slice: fp8 e4m3 operands for the forward and
    data-grad launches of the >= 128-channel 3x3 / 1x1 convolutions outside the recognizer, bf16 everywhere else)."""
    global CONV_DTYPE
    if dtype not in ("f32", "bf16", "fp8"):
        raise ValueError("conv dtype must be 'f32', 'bf16' or 'fp8'")
    CONV_DTYPE = dtype
    _PACK_CACHE.clear()
    _TWINS.clear()
    _GHOSTS.clear()


_FP8_BLOCK = [0]


class bf16_only:
    """Context: no fp8 launches inside (the recognizer of config c5 stays bf16)."""

    def __enter__(self):
        _FP8_BLOCK[0] += 1

    def __exit__(self, *exc):
        _FP8_BLOCK[0] -= 1
        return False


def _low() -> bool:
    return CONV_DTYPE in ("bf16", "fp8")


def _fp8_ok(K: int, N: int, kh: int, kw: int, same: bool) -> bool:
    return USE_V2 and CONV_DTYPE == "fp8" and not _FP8_BLOCK[0] and K % 128 == 0 and N % 256 == 0 and (same or (kh == 1 and kw == 1))


def fp8_of(t: torch.Tensor, relu: bool = False):
    """-> (fp8 e4m3 copy of relu?(t) * 448 / amax as a uint8 tensor, amax device scalar); one conversion per (t, relu)."""
    key = (t.untyped_storage().data_ptr(), "fp8", bool(relu), t.storage_offset(), t.numel())
    e = _TWINS.get(key)
    if e is None:
        _chk(t)
        out = torch.empty(t.shape, device=t.device, dtype=torch.uint8)
        if _is_ghost(t):                         # an operand-only conv result: quantise its bf16 values, scale = the recorded
            t16 = _twin_get(t)                   # amax of the whole tensor (a batch slice shares it)
            amax = _TWINS[(t.untyped_storage().data_ptr(), "amax")][1][0:1]
            with _hbm("cvt_fp8", t16, out):
                call("sg_cvt_fp8_bf16", t16.data_ptr(), out.data_ptr(), t.numel(), int(relu), _p(amax), _stream())
        else:
            rec = _amax_get(t)                   # the producing conv's epilogue already took max |t|
            amax = rec[0:1] if rec is not None else torch.zeros(1, device=t.device)
            with _hbm("cvt_fp8", t if rec is None else None, t, out):
                if rec is None:
                    call("sg_amax_f32", _p(t), t.numel(), _p(amax), _stream())
                call("sg_cvt_fp8", _p(t), out.data_ptr(), t.numel(), int(relu), _p(amax), _stream())
        e = (t, out, amax)
        _TWINS[key] = e
    return e[1], e[2]


def _fp8_wgrad_ok(Cin: int, Cout: int, kh: int, kw: int, same: bool) -> bool:
    """config c5: the weight-grad launches that run on fp8 operands (e4m3 activations x e5m2 gradients): the layers whose
    forward / data-grad launches are fp8 and whose channel counts fill the kernel's 256 x 256 tile."""
    return USE_V2 and CONV_DTYPE == "fp8" and not _FP8_BLOCK[0] and Cin % 256 == 0 and Cout % 256 == 0 and (same or (kh == 1 and kw == 1))


def grad_operand_fp8(dy: torch.Tensor, sample_scale, want_colsum: bool):
    """The fp8 weight-grad operand of a gradient tensor: (e5m2 copy of sample_scale[b] * dy[b] as uint8, its amax device scalar,
    fp32 column sums or None).  ONE amax sweep (max |dy| and max |scale dy| together) and ONE convert sweep yield that copy, the
    column sums (= the bias gradient) and -- when the gradient has none yet -- the e4m3 copy the data-grad launch of the same
    gradient reads (registered under fp8_of's key); kept for the step."""
    key = (dy.untyped_storage().data_ptr(), "g8", 0 if sample_scale is None else sample_scale.data_ptr(), dy.storage_offset(), dy.numel())
    e = _TWINS.get(key)
    if e is None or (want_colsum and e[4] is None):
        _chk(dy, sample_scale)
        C = dy.shape[-1]
        M = dy.numel() // C
        k4 = (dy.untyped_storage().data_ptr(), "fp8", False, dy.storage_offset(), dy.numel())
        have4 = k4 in _TWINS
        rec = _amax_get(dy, sample_scale, need_scaled=True)
        amax2 = rec if rec is not None else torch.zeros(2, device=dy.device)
        out5 = torch.empty(dy.shape, device=dy.device, dtype=torch.uint8)
        out4 = None if have4 else torch.empty(dy.shape, device=dy.device, dtype=torch.uint8)
        colsum = torch.zeros(C, device=dy.device, dtype=torch.float32)
        rows = (M // sample_scale.numel()) if sample_scale is not None else 1
        with _hbm("cvt_fp8", dy if rec is None else None, dy, out5, out4):
            if rec is None:
                call("sg_amax2_f32", _p(dy), dy.numel(), _p(sample_scale), rows * C, _p(amax2), _stream())
            call("sg_cvt_fp8_grad", _p(dy), out5.data_ptr(), None if out4 is None else out4.data_ptr(), M, C, _p(sample_scale), rows,
                 _p(amax2), _p(colsum), _stream())
        if out4 is not None:
            _TWINS[k4] = (dy, out4, amax2[0:1])
        e = (dy, sample_scale, out5, amax2[1:2], colsum, amax2)
        _TWINS[key] = e
    return e[2], e[3], e[4]


def _amax2_of(dy: torch.Tensor, sample_scale) -> torch.Tensor:
    """Device pair {max |dy|, max |sample_scale dy|}: from the gradient's fp8 operand entry when it has one, else one read sweep."""
    sptr = 0 if sample_scale is None else sample_scale.data_ptr()
    e = _TWINS.get((dy.untyped_storage().data_ptr(), "g8", sptr, dy.storage_offset(), dy.numel()))
    if e is not None:
        return e[5]
    rec = _amax_get(dy, sample_scale, need_scaled=True)
    if rec is not None:
        return rec
    key = (dy.untyped_storage().data_ptr(), "a2", sptr, dy.storage_offset(), dy.numel())
    e = _TWINS.get(key)
    if e is None:
        _chk(dy, sample_scale)
        amax2 = torch.zeros(2, device=dy.device)
        rows = (dy.numel() // sample_scale.numel()) if sample_scale is not None else 4
        with _hbm("cvt_fp8", dy):
            call("sg_amax2_f32", _p(dy), dy.numel(), _p(sample_scale), rows, _p(amax2), _stream())
        e = (dy, amax2)
        _TWINS[key] = e
    return e[1]


def _grad_colsum(dy: torch.Tensor, sample_scale) -> torch.Tensor:
    """fp32 column sums of sample_scale[b] * dy[b] (a bias gradient), from whichever operand entry of the gradient holds them."""
    sptr = 0 if sample_scale is None else sample_scale.data_ptr()
    base = (dy.untyped_storage().data_ptr(), dy.storage_offset(), dy.numel())
    e = _TWINS.get((base[0], "g8", sptr, base[1], base[2]))
    if e is not None and e[4] is not None:
        return e[4]
    return grad_operand(dy, sample_scale, True)[1]


GHOST_NAN = _os.environ.get("SG_GHOST_NAN", "0") == "1"      # tests: poison the never-written fp32 tensors behind operand-only results
OPERAND_ONLY_MIN_TILES = 512     # conv1 -> conv2 chains keep the intermediate as bf16 only when the launch fills >= 2 rounds of 256-pixel
                                 # tiles (smaller launches want the reduction-split tail, which meets in the fp32 result)


def _is_ghost(t: torch.Tensor) -> bool:
    return (t.untyped_storage().data_ptr(), "ghost") in _TWINS


def operand_only_ok(B: int, H: int, W: int, C: int) -> bool:
    """bf16 / fp8 modes: may the result [B,H,W,C] of a conv that feeds ONLY a C -> C 3x3 conv (and its backward launches) exist as
    a bf16 operand copy alone?  Yes when every consumer reads operand copies: the second-generation kernels take C -> C 3x3 in all
    three directions, the launch is large enough to do without the reduction-split tail, and the deterministic mode (which sends
    the 64 -> 64 weight-grad to the first-generation kernel, an fp32 reader) is off."""
    if not (USE_V2 and _low() and not DETERMINISTIC and C % 8 == 0):
        return False
    if not (_v2_ok(C, C, 3, 3, True) and (C % 256 == 0 or C == 64)):
        return False
    return -(-(B * H * W) // 256) * -(-C // 256) >= OPERAND_ONLY_MIN_TILES


def avgpool2_bwd_operands(dout: torch.Tensor, wscale=None, want_dw: bool = True) -> torch.Tensor:
    """The gradient of a ResNetBlockDown's conv2 output, d_c2 = avgpool2_bwd(dout) [B,2Ho,2Wo,C], for its two consumers -- conv2's
    weight-grad and data-grad.  bf16 / fp8 modes with a conv2 (C -> C, 3x3) that runs on the second-generation kernels: ONE kernel
    reads dout and writes the operand copies those launches read (plain + per-sample-scaled bf16, or e4m3 + e5m2 with their amax
    scalars); the returned fp32 tensor is a NEVER-WRITTEN handle that carries the copies through the twin registry (no fp32
    d_c2 in HBM: 2-4 B/element written instead of 4 written + 8 re-read by amax / conversion sweeps).  The bias gradient of
    conv2 (= column sums of d_c2 = column sums of dout) comes from dout's own operand entry.  Otherwise: plain sg_avgpool2_bwd."""
    B, Ho, Wo, C = dout.shape
    H, W = 2 * Ho, 2 * Wo
    use8 = _fp8_wgrad_ok(C, C, 3, 3, True) and _fp8_ok(C, C, 3, 3, True)
    use16 = (not use8) and _v2_ok(C, C, 3, 3, True) and (C % 256 == 0 or (C == 64 and not DETERMINISTIC))
    if not (use8 or use16) or C % 8:
        return avgpool2_bwd(dout)
    _chk(dout, wscale)
    ghost = torch.empty(B, H, W, C, device=dout.device, dtype=torch.float32)
    if GHOST_NAN:
        ghost.fill_(float("nan"))
    gp, n = ghost.untyped_storage().data_ptr(), ghost.numel()
    _GHOSTS.add(gp)
    sptr = 0 if wscale is None else wscale.data_ptr()
    colsum = _grad_colsum(dout, wscale) if want_dw else None
    if use16:
        plain = torch.empty(ghost.shape, device=dout.device, dtype=torch.bfloat16)
        scaled = torch.empty(ghost.shape, device=dout.device, dtype=torch.bfloat16) if (want_dw and wscale is not None) else None
        with _hbm("pool", dout, plain, scaled):
            call("sg_avgpool2_bwd_bf16", _p(dout), plain.data_ptr(), None if scaled is None else scaled.data_ptr(), _p(wscale), B, H, W, C, _stream())
        _twin_put(ghost, plain)
        if want_dw:
            _TWINS[(gp, sptr, 0, n)] = (ghost, wscale, plain if scaled is None else scaled, colsum)
    else:
        a2 = _amax2_of(dout, wscale)
        out4 = torch.empty(ghost.shape, device=dout.device, dtype=torch.uint8)
        out5 = torch.empty(ghost.shape, device=dout.device, dtype=torch.uint8) if want_dw else None
        amax_dx = torch.empty(2, device=dout.device)
        with _hbm("pool", dout, out4, out5):
            call("sg_avgpool2_bwd_fp8", _p(dout), out4.data_ptr(), None if out5 is None else out5.data_ptr(), _p(wscale), _p(a2), _p(amax_dx),
                 B, H, W, C, _stream())
        _TWINS[(gp, "fp8", False, 0, n)] = (ghost, out4, amax_dx[0:1])
        if want_dw:
            _TWINS[(gp, "g8", sptr, 0, n)] = (ghost, wscale, out5, amax_dx[1:2], colsum, amax_dx)
    return ghost



def packed_filter_fp8(w: torch.Tensor, kind: str):
    """-> (fp8 copy [tap][N][K] of a Conv2D filter scaled by 448 / amax, amax device scalar); made once per optimizer step."""
    key = (w.data_ptr(), tuple(w.shape), kind + "8", w._version)
    hit = _PACK_CACHE.get(key)
    if hit is not None:
        return hit[1], hit[2]
    kh, kw, Cin, Cout = w.shape
    amax = torch.zeros(1, device=w.device)
    out = torch.empty(w.numel(), device=w.device, dtype=torch.uint8)
    call("sg_amax_f32", _p(w), w.numel(), _p(amax), _stream())
    if kind == "fwd":
        call("sg_pack_filter_fp8", _p(w), out.data_ptr(), _p(amax), kh * kw, Cin, Cout, 1, _stream())
    else:
        call("sg_pack_filter_fp8", _p(w), out.data_ptr(), _p(amax), kh * kw, Cout, Cin, 0, _stream())
    _PACK_CACHE[key] = (w, out, amax)
    return out, amax


DETERMINISTIC = False


def set_deterministic(on: bool) -> None:
    """on: every convolution launch -- first- and second-generation kernels, fp32 / bf16 / fp8 -- runs with ONE adder per output
    address: forward / data-grad launches are not cut along the reduction (a sample's activations no longer depend on the
    batch it is launched in), weight-grad launches run one pixel chunk per (tap, tile) and the fused / separate bias-gradient
    column sums one workgroup, so dW and db have a fixed summation order (bitwise reproducible run to run).  Still
    float-atomic in this mode: BatchNorm's per-sample dgamma / dbeta partials, the filter bank's dz, the attention key sweep
    at small batch, spectral-norm's power iteration.  off: the default CU-quantum tail split and pixel chunking."""
    global DETERMINISTIC
    DETERMINISTIC = bool(on)
    lib().sg_set_deterministic(1 if on else 0)


def weights_changed() -> None:
    """Called by the optimizers (and anything else that rewrites parameters through raw pointers): the packed bf16
    filter copies are stale."""
    _PACK_CACHE.clear()


def packed_filter(w: torch.Tensor, kind: str) -> torch.Tensor:
    """bf16 copy [tap][N][K] of a Conv2D filter [kh,kw,Cin,Cout] for the forward ('fwd': K = Cin, N = Cout) or the
    data-grad ('bwd': K = Cout, N = Cin) launch; made once per optimizer step."""
    # The entry keeps a reference to `w`: while it lives, the filter's memory cannot be freed and handed to another
    # parameter, so (address, shape, version) identifies the filter contents (torch in-place writes bump the version,
    # the optimizer kernels -- raw-pointer writes -- clear the cache).
    key = (w.data_ptr(), tuple(w.shape), kind, w._version)
    hit = _PACK_CACHE.get(key)
    if hit is not None:
        return hit[1]
    if len(_PACK_CACHE) > 512:
        _PACK_CACHE.clear()
    kh, kw, Cin, Cout = w.shape
    if kind in ("wino_fwd2", "wino_bwd2", "wino_fwd4", "wino_bwd4"):   # Winograd-domain filters U [(tile+2)^2][N][K] (conv_winograd.hip)
        tile = int(kind[-1])
        out = torch.empty((tile + 2) ** 2, Cout * Cin, device=w.device, dtype=torch.float32)
        if kind.startswith("wino_fwd"):
            call("sg_wino_filter", _p(packed_filter(w, "fwd_f32t")), _p(out), Cout, Cin, 0, tile, _stream())
        else:
            call("sg_wino_filter", _p(w), _p(out), Cin, Cout, 1, tile, _stream())
        _PACK_CACHE[key] = (w, out)
        return out
    out = None if kind in ("bwd_f32", "fwd_f32t") else torch.empty(w.numel(), device=w.device, dtype=torch.bfloat16)
    if kind in ("bwd_f32", "fwd_f32t"):   # fp32 [kh,kw,Cout,Cin]: per tap [N = Cout][K = Cin] for the second-generation forward
        out = torch.empty(kh, kw, Cout, Cin, device=w.device, dtype=torch.float32)   # launch (and the old transposed data-grad option)
        call("sg_transpose_filter", _p(w), _p(out), kh * kw, Cin, Cout, _stream())
    elif kind == "t_fwd":            # Conv2DTranspose filter w [kh,kw,Co,Ci]: each tap already is [N = Co][K = Ci] -> convert only
        call("sg_pack_filter_bf16", _p(w), out.data_ptr(), kh * kw, w.shape[3], w.shape[2], 0, _stream())
    elif kind == "t_bwd":            # its data-grad reduces over Co: per tap [K = Co][N = Ci] -> [N = Ci][K = Co]
        call("sg_pack_filter_bf16", _p(w), out.data_ptr(), kh * kw, w.shape[2], w.shape[3], 1, _stream())
    elif kind == "fwd":
        call("sg_pack_filter_bf16", _p(w), out.data_ptr(), kh * kw, Cin, Cout, 1, _stream())
    else:
        call("sg_pack_filter_bf16", _p(w), out.data_ptr(), kh * kw, Cout, Cin, 0, _stream())
    _PACK_CACHE[key] = (w, out)
    return out


def _bf16_ok(K: int, N: int) -> bool:
    return _low() and K % 8 == 0 and N > 32 and K > 1


# ---- bf16 twins (second-generation bf16 path, conv_bf16v2.hip) ---------------------------------------------------
# In bf16 mode the large convolutions read their activation operand as a bf16 NHWC tensor.  A "twin" is the bf16 copy of
# an fp32 activation: written by the producing conv's epilogue (y16 / dx16) or by sg_cvt_bf16 on first use, and found
# again through this registry (keyed by the fp32 tensor's storage; a batch slice of a registered tensor maps to the
# same slice of its twin).  An entry holds a reference to the fp32 tensor, so its storage cannot be recycled while the
# entry lives; kernels that write INTO an existing tensor drop its entry (_touch).  new_step() empties the registry.
USE_V2 = _os.environ.get("SG_BF16_V2", "1") == "1"
_TWINS = {}


_AMAX_POOL = {}        # raw stream handle -> {"buf", "next"}: a pool is zeroed and handed out on ONE stream (its slots are written by that
                       # stream's conv epilogues and read by that stream's conversion kernels, or behind a join)


def _amax_slot(like: torch.Tensor) -> torch.Tensor:
    """Two zeroed floats (max |t|, max |scale t|) from a per-step, per-stream pool: ONE memset launch per 1024 convolutions."""
    pool = _AMAX_POOL.setdefault(_stream(), {"buf": None, "next": 0})
    if pool["buf"] is None or pool["next"] + 2 > pool["buf"].numel() or pool["buf"].device != like.device:
        pool["buf"] = torch.zeros(2048, device=like.device)
        pool["next"] = 0
    s = pool["buf"][pool["next"]:pool["next"] + 2]
    pool["next"] += 2
    return s


def _want_amax() -> bool:
    """config c5: conv epilogues record the amax of their result for the fp8 launch that reads it next."""
    return CONV_DTYPE == "fp8" and not _FP8_BLOCK[0]


def _amax_put(t: torch.Tensor, slot: torch.Tensor, scale) -> None:
    if t.storage_offset() == 0 and t.numel() * 4 == t.untyped_storage().nbytes():
        _TWINS[(t.untyped_storage().data_ptr(), "amax")] = (t, slot, 0 if scale is None else scale.data_ptr())


def _amax_get(t: torch.Tensor, scale=None, need_scaled: bool = False):
    """The producer-recorded amax pair of a WHOLE tensor, or None.  need_scaled: element 1 must belong to `scale`."""
    e = _TWINS.get((t.untyped_storage().data_ptr(), "amax"))
    if e is None or t.storage_offset() != 0 or t.numel() != e[0].numel():
        return None
    if need_scaled and e[2] != (0 if scale is None else scale.data_ptr()):
        return None
    return e[1]


def new_step() -> None:
    _TWINS.clear()
    _GHOSTS.clear()
    _AMAX_POOL.clear()
    _WINO_V["map"].clear()
    _WINO_V["on"] = WINO_KEEP_V


def end_step() -> None:
    """The step's backward sweeps are queued: the kept Winograd transforms are released (and launches outside a step keep none)."""
    _WINO_V["map"].clear()
    _WINO_V["on"] = False


def _touch(t) -> None:
    if t is not None and _WINO_V["map"]:
        _WINO_V["map"].pop(t.untyped_storage().data_ptr(), None)
    if t is not None and _TWINS:
        k = t.untyped_storage().data_ptr()
        _GHOSTS.discard(k)
        _TWINS.pop(k, None)
        for kk in [kk for kk in _TWINS if isinstance(kk, tuple) and kk[0] == k]:
            del _TWINS[kk]


def bf16_scaled(t: torch.Tensor, rowscale: torch.Tensor) -> torch.Tensor:
    """bf16 copy of rowscale[b] * t[b] (the weight-grad operand of a shared backward sweep); one conversion per (t, rowscale)."""
    key = (t.untyped_storage().data_ptr(), rowscale.data_ptr(), t.storage_offset(), t.numel())
    e = _TWINS.get(key)
    if e is None:
        e = (t, rowscale, cvt_bf16(t, rowscale=rowscale), None)
        _TWINS[key] = e
    return e[2]


def _twin_put(t: torch.Tensor, t16: torch.Tensor) -> None:
    if t.storage_offset() != 0 or t.numel() * 4 != t.untyped_storage().nbytes():
        return
    if len(_TWINS) > 512:
        _TWINS.clear()
        _GHOSTS.clear()
    _TWINS[t.untyped_storage().data_ptr()] = (t, t16.view(-1))


def _twin_get(t: torch.Tensor):
    e = _TWINS.get(t.untyped_storage().data_ptr())
    if e is None:
        return None
    off = t.storage_offset()
    return e[1][off:off + t.numel()].view(t.shape)


def cvt_bf16(t: torch.Tensor, relu=False, rowscale=None) -> torch.Tensor:
    """fp32 -> bf16 copy (sg_cvt_bf16); rowscale [B] multiplies sample b first."""
    _chk(t, rowscale)
    out = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16)
    n = t.numel()
    assert n % 8 == 0
    with _hbm("cvt_bf16", t, out):
        call("sg_cvt_bf16", _p(t), out.data_ptr(), n, int(relu), _p(rowscale), (n // rowscale.numel()) if rowscale is not None else 8, _stream())
    return out


def cvt_bf16_bias(t: torch.Tensor, rowscale, db: torch.Tensor, want_plain: bool = False):
    """bf16 copy of (rowscale[b] *) t [B,H,W,C] and db [C] += its fp32 column sums, in one sweep (sg_cvt_bf16_bias);
    want_plain: -> (scaled copy, unscaled copy), both written in that sweep."""
    _chk(t, rowscale, db)
    out = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16)
    plain = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16) if want_plain else None
    C = t.shape[-1]
    M = t.numel() // C
    with _hbm("cvt_bf16", t, out, plain):
        call("sg_cvt_bf16_bias", _p(t), out.data_ptr(), None if plain is None else plain.data_ptr(), M, C, _p(rowscale),
             (M // rowscale.numel()) if rowscale is not None else 1, _p(db), _stream())
    return (out, plain) if want_plain else out


def grad_operand(dy: torch.Tensor, sample_scale, want_colsum: bool):
    """The bf16 weight-grad operand of a gradient tensor: (bf16 copy of sample_scale[b] * dy[b], fp32 column sums of it or
    None).  Whatever is missing -- the scaled copy, the column sums (= the bias gradient of every conv that produced dy's
    forward tensor: conv2 and the 1x1 shortcut of a ResNetBlockDown share them) and, when the gradient has no plain twin yet,
    that twin too (the data-grad launch of the same gradient wants it) -- is made in ONE sweep over dy and kept for the step."""
    key = (dy.untyped_storage().data_ptr(), 0 if sample_scale is None else sample_scale.data_ptr(), dy.storage_offset(), dy.numel())
    e = _TWINS.get(key)
    plain = _twin_get(dy)
    if e is None or (want_colsum and e[3] is None):
        whole = dy.storage_offset() == 0 and dy.numel() * 4 == dy.untyped_storage().nbytes()
        colsum = torch.zeros(dy.shape[-1], device=dy.device, dtype=torch.float32)
        if sample_scale is None:
            t16 = cvt_bf16_bias(dy, None, colsum)
            if plain is None:
                _twin_put(dy, t16)
        elif plain is None and whole:
            t16, p16 = cvt_bf16_bias(dy, sample_scale, colsum, want_plain=True)
            _twin_put(dy, p16)
        else:
            t16 = cvt_bf16_bias(dy, sample_scale, colsum)
        e = (dy, sample_scale, t16, colsum)
        _TWINS[key] = e
    return e[2], e[3]


def bf16_of(t: torch.Tensor) -> torch.Tensor:
    """The bf16 twin of an fp32 activation (made on first use)."""
    t16 = _twin_get(t)
    if t16 is None:
        t16 = cvt_bf16(t)
        _twin_put(t, t16)
    return t16


USE_F32_V2 = _os.environ.get("SG_F32_V2", "1") == "1"


F32_V2_MIN_TILES = 512      # (tests set 0: every eligible shape through the second-generation kernel)


def _f32v2_ok(K: int, N: int, kh: int, kw: int, same: bool, pixels: int) -> bool:
    """fp32 mode: the DMA-fed second-generation kernel (256-pixel tiles, one workgroup per CU) takes the stride-1 convs on its
    tile grid that fill at least two rounds of the 256 CUs and reduce over >= 1024 terms; measured per layer at bs 128
    (profiles/r02_shapes_bs128.txt): +5..10 % on the data-grads (no ReLU in the loop), level on the forward convs, behind the
    first-generation kernel (128 x 128 tiles, two workgroups per CU) on small grids and short reductions."""
    if not (USE_F32_V2 and CONV_DTYPE == "f32" and K % 32 == 0 and N % 64 == 0 and (same or (kh == 1 and kw == 1))):
        return False
    tiles = -(-pixels // 256) * -(-N // 256)
    return tiles >= F32_V2_MIN_TILES and (F32_V2_MIN_TILES == 0 or kh * kw * K >= 1024)


# ---- Winograd-domain fp32 3x3 convolutions (conv_winograd.hip) --------------------------------------------------------
USE_WINOGRAD = _os.environ.get("SG_WINOGRAD", "1") == "1"
# F(4x4, 3x3) -- 36 products per 4x4 outputs, 2.25 per output against 4 for F(2x2, 3x3) and 9 for the direct form -- wherever H and W
# are multiples of 4 (every layer of the fixed-width step); F(2x2, 3x3) on the remaining even shapes (odd word lengths of the
# bucketed widths: W = 2 L on the 4-row layers).  SG_WINO_TILE=2 keeps F(2x2) everywhere.
WINO_TILE = int(_os.environ.get("SG_WINO_TILE", "4"))
# Where the reduced product count pays for the transform sweeps (profiles/r03_probe_winograd.txt, direct / Winograd time per launch):
# F(2x2): 1.6-2.0x over >= 512 channels, 1.3-1.5x at 256, 1.1-1.3x at 128 -> 256, 0.9x at 128 -> 128 and below;
# F(4x4): 2.7-3.3x over >= 512 channels, 2.0-2.5x at 256, 1.4-1.9x at 128, 1.2x (forward) / 1.7x (weight-grad) at 64 -> 512.
WINO_MIN_C = {2: int(_os.environ.get("SG_WINO_MIN_C", "128")), 4: int(_os.environ.get("SG_WINO4_MIN_C", "64"))}       # both channel counts at least this ...
WINO_MIN_KN = {2: int(_os.environ.get("SG_WINO_MIN_KN", "32768")), 4: int(_os.environ.get("SG_WINO4_MIN_KN", "16384"))}   # ... and their product at least this
WINO4_WGRAD_MIN_TILES = 128     # weight-grad: below this many 4x4 tiles the 36 reductions are too short (B = 16 on the 4x20 layers) -> F(2x2)
_WINO_WS = {}              # raw stream handle -> scratch buffer (V and Mt of the launch in flight on that stream)
# The forward launch's transformed input V is exactly what the weight gradient of the same convolution needs: inside a train_step
# (new_step() ... end_step()) forward launches keep V in a tensor of their own (2.25x the layer input for F(4x4)) and the weight-grad
# finds it again through the input's storage -- a batch slice of the input maps to a row range of every plane -- instead of
# transforming relu(x) a second time.
WINO_KEEP_V = _os.environ.get("SG_WINO_KEEP_V", "1") == "1"
_WINO_V = {"on": False, "map": {}}      # map: storage pointer of x -> (x, relu_in, tile, V [P * Tp * K], Tp)


def _wino_v_get(x: torch.Tensor, relu_in: bool, tile: int):
    """-> (pointer of the first row of x's tiles in plane 0, rows between planes) or None"""
    e = _WINO_V["map"].get(x.untyped_storage().data_ptr())
    if e is None or e[1] != relu_in or e[2] != tile or e[0].shape[1:] != x.shape[1:]:
        return None
    per = x.shape[1] * x.shape[2] * x.shape[3]
    off = x.storage_offset() - e[0].storage_offset()
    if off < 0 or off % per or off // per + x.shape[0] > e[0].shape[0]:
        return None
    t0 = (off // per) * (x.shape[1] // tile) * (x.shape[2] // tile)
    if e[5] != _stream():                   # read on another stream than the one it was made on (weight-grads on the sweep's side stream)
        e[3].record_stream(torch.cuda.current_stream())
    return e[3].data_ptr() + 4 * t0 * x.shape[3], e[4]


def _wino_tile(H: int, W: int) -> int:
    return 4 if (WINO_TILE == 4 and H % 4 == 0 and W % 4 == 0) else 2


# Few-tile launches: every one of the (tile + 2)^2 frequency planes is padded to 128 rows, the direct form pads its B H W pixels ONCE.  At
# T = 18 tiles (the recognizer's 8 x 12 map at B = 3) F(4x4) multiplies 36 x 128 rows where the direct kernel multiplies 9 x 384: more
# products, not fewer.  The Winograd path is taken only where its padded row count is below WINO_ROW_GAIN x the direct form's (its
# products run at ~0.75 of the direct loop's rate per row: short reductions); 0 switches the criterion off (the kernels' own tests).
WINO_ROW_GAIN = float(_os.environ.get("SG_WINO_ROW_GAIN", "0.75"))
# output channels of a Winograd-domain forward / data-grad launch: a multiple of 64 (round 4: 128 x 64 tiles of the grouped product for the
# 64-filter layers -- the 512 -> 64 data-grads of the D-shaped trunks ran direct at 131 TF/s, 4x the products; SG_WINO_N_MULT=128: as in round 3)
WINO_N_MULT = int(_os.environ.get("SG_WINO_N_MULT", "64"))


def _wino_rows_ok(B, H: int, W: int, tile: int) -> bool:
    if B is None or WINO_ROW_GAIN <= 0 or DETERMINISTIC:       # (deterministic mode: a sample's result must not depend on the batch it rides in)
        return True
    T = B * (H // tile) * (W // tile)
    return (tile + 2) ** 2 * (-(-T // 128) * 128) <= WINO_ROW_GAIN * 9 * (-(-(B * H * W) // 128) * 128)


def _wino_ok(K: int, N: int, kh: int, kw: int, same: bool, H: int, W: int, B=None) -> bool:
    if not (USE_WINOGRAD and CONV_DTYPE == "f32" and kh == 3 and kw == 3 and same and H % 2 == 0 and W % 2 == 0 and K % 32 == 0 and N % WINO_N_MULT == 0):
        return False
    t = _wino_tile(H, W)
    return min(K, N) >= WINO_MIN_C[t] and K * N >= WINO_MIN_KN[t] and _wino_rows_ok(B, H, W, t)


def _wino_workspace(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    """Scratch for one Winograd-domain convolution, one buffer per stream (launches of one stream run in order, so the next
    convolution may overwrite it), grown to the largest request."""
    key = _stream()
    buf = _WINO_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        _WINO_WS.pop(key, None)
        buf = None
        buf = torch.empty(nbytes, device=like.device, dtype=torch.uint8)
        _WINO_WS[key] = buf
    return buf


def _wino_conv(a, w, out, bias, bias2, mask, K: int, N: int, relu_in: bool, flags: int, tag, ups: bool = False) -> None:
    """out [B,H,W,N] = conv3x3_same(a [B,H,W,K]) through input transform -> grouped products -> output transform (tag[0] = "wino_fwd":
    forward of the filter w; "wino_dgrad": its data-grad).  ups (data-grad, F(4x4)): a is the half-resolution gradient [B,H/2,W/2,K] and
    stands for 0.25 * upsample2x2(a) (UPS2_IN)."""
    B, H, W, _ = a.shape
    if ups:
        H, W = 2 * H, 2 * W
        flags |= UPS2_IN
    tile = _wino_tile(H, W)
    P = (tile + 2) ** 2
    fwd = tag[0] == "wino_fwd"
    u = packed_filter(w, ("wino_fwd%d" if fwd else "wino_bwd%d") % tile)
    T = B * (H // tile) * (W // tile)
    Tp = -(-T // 128) * 128                 # = sg_wino_plane_rows(B, H, W, tile)
    nbytes = 4 * P * Tp * (K + N)           # = sg_wino_workspace_bytes(B, H, W, K, N, tile)
    s = _stream()
    keep = (fwd and _WINO_V["on"] and a.storage_offset() == 0 and _wino_wgrad_ok(K, N, 3, 3, True, H, W)
            and (tile == 2 or T >= WINO4_WGRAD_MIN_TILES))
    if keep:                                # V in a tensor of its own (the weight gradient reads it again), Mt in the stream's scratch
        Vt = torch.empty(P * Tp * K, device=a.device, dtype=torch.float32)
        if len(_WINO_V["map"]) > 256:
            _WINO_V["map"].clear()
        _WINO_V["map"][a.untyped_storage().data_ptr()] = (a, bool(relu_in), tile, Vt, Tp, s)
        V = Vt.data_ptr()
        Mt = _wino_workspace(4 * P * Tp * N, a).data_ptr()
        if PROFILER is None:
            call("sg_wino_input", _p(a), V, B, H, W, K, int(relu_in), tile, s)
            call("sg_wino_gemm", V, _p(u), Mt, B, H, W, K, N, tile, s)
            call("sg_wino_output", Mt, _p(out), _p(bias), _p(bias2), None, B, H, W, N, flags, tile, s)
            return
    else:
        ws = _wino_workspace(nbytes, a)
        V = ws.data_ptr()
        Mt = V + 4 * P * Tp * K
    if PROFILER is None:                    # one call for the three launches (the host queues a shard-size step in half its GPU time)
        if fwd:
            call("sg_conv2d_fwd_wino", _p(a), _p(u), _p(bias), _p(bias2), _p(out), B, H, W, K, N, flags | (1 if relu_in else 0), tile, V, nbytes, s)
        else:
            call("sg_conv2d_bwd_data_wino", _p(a), _p(u), _p(mask), _p(out), B, H, W, N, K, flags, tile, V, nbytes, s)
        return
    with _hbm("wino_transform", a, flops=0.0) as _:
        if PROFILER is not None and PROFILER.wants("wino_transform"):
            PROFILER.nbytes["wino_transform"] = PROFILER.nbytes.get("wino_transform", 0.0) + 4.0 * P * T * K
        if ups:
            call("sg_wino_input_ups", _p(a), V, B, H, W, K, tile, s)
        else:
            call("sg_wino_input", _p(a), V, B, H, W, K, int(relu_in), tile, s)
    with _timed("igemm_wino", 2.0 * P * Tp * K * N, False, tag):        # (executed FLOPs: the grouped products, pad rows included)
        if PROFILER is not None and PROFILER.wants("igemm_wino"):
            PROFILER.nbytes["igemm_wino"] = PROFILER.nbytes.get("igemm_wino", 0.0) + 4.0 * P * (Tp * K + N * K + Tp * N)
        call("sg_wino_gemm", V, _p(u), Mt, B, H, W, K, N, tile, s)
    with _hbm("wino_transform", out, mask, flops=0.0):
        if PROFILER is not None and PROFILER.wants("wino_transform"):
            PROFILER.nbytes["wino_transform"] = PROFILER.nbytes.get("wino_transform", 0.0) + 4.0 * P * T * N
        call("sg_wino_output", Mt, _p(out), _p(bias), _p(bias2), _p(mask), B, H, W, N, flags & ~UPS2_IN, tile, s)


def _wino_wgrad_ok(Cin: int, Cout: int, kh: int, kw: int, same: bool, H: int, W: int) -> bool:
    """Weight gradient in the Winograd domain: (tile + 2)^2 [Cin x tiles] x [tiles x Cout] products.  Not in deterministic mode (one
    pixel chunk per product would leave a few hundred workgroups for the whole launch: the direct kernel's single-chunk form is used)."""
    if not (USE_WINOGRAD and CONV_DTYPE == "f32" and not DETERMINISTIC and kh == 3 and kw == 3 and same and H % 2 == 0 and W % 2 == 0
            and Cin % 32 == 0 and Cout % 64 == 0):
        return False
    t = _wino_tile(H, W)
    return min(Cin, Cout) >= WINO_MIN_C[t] and Cin * Cout >= (min(WINO_MIN_KN[4], 4096) if t == 4 else WINO_MIN_KN[2])      # (F(4x4): 64 -> 64 included)


def _wino_wgrad(x, dy, dw, db, sample_scale, relu_in: bool, ups: bool = False) -> None:
    """ups (F(4x4) only): dy is the half-resolution gradient [B,H/2,W/2,Cout] standing for 0.25 * upsample2x2(dy)."""
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    tile = _wino_tile(H, W)
    if tile == 4 and B * (H // 4) * (W // 4) < WINO4_WGRAD_MIN_TILES:
        tile = 2
    assert not ups or tile == 4
    P = (tile + 2) ** 2
    T = B * (H // tile) * (W // tile)
    Tp = -(-T // 128) * 128
    nbytes = 4 * (P * (Tp * (Cin + Cout) + Cin * Cout) + 64 * Cout)          # = sg_wino_wgrad_workspace_bytes(B, H, W, Cin, Cout, tile)
    ws = _wino_workspace(nbytes, x)
    V = ws.data_ptr()
    Qt = V + 4 * P * Tp * Cin
    dU = Qt + 4 * P * Tp * Cout
    s = _stream()
    kept = _wino_v_get(x, bool(relu_in), tile)          # the forward launch's transform of this very tensor (or of the batch it is a slice of)
    v_rows = 0
    if kept is not None:
        V, v_rows = kept
    elif PROFILER is None:
        call("sg_conv2d_bwd_weight_wino", _p(x), _p(dy), _p(dw), _p(db), _p(sample_scale), B, H, W, Cin, Cout, (1 if relu_in else 0) | (UPS2_IN if ups else 0),
             tile, V, nbytes, s)
        return
    with _hbm("wino_transform", x if kept is None else None, dy):
        if PROFILER is not None and PROFILER.wants("wino_transform"):
            PROFILER.nbytes["wino_transform"] = PROFILER.nbytes.get("wino_transform", 0.0) + 4.0 * P * T * ((Cin if kept is None else 0) + Cout)
        if kept is None:
            call("sg_wino_input", _p(x), V, B, H, W, Cin, int(relu_in), tile, s)
        call("sg_wino_grad_input_ups" if ups else "sg_wino_grad_input", _p(dy), Qt, _p(sample_scale), _p(db), dU + 4 * P * Cin * Cout, B, H, W, Cout, tile, s)
    with _timed("wgrad_wino", 2.0 * P * T * Cin * Cout, False, ("wino_wgrad", B, H, W, Cin, Cout, 3)):
        if PROFILER is not None and PROFILER.wants("wgrad_wino"):
            PROFILER.nbytes["wgrad_wino"] = PROFILER.nbytes.get("wgrad_wino", 0.0) + 4.0 * P * (T * (Cin + Cout) + Cin * Cout)
        call("sg_wino_wgrad_gemm", V, Qt, dU, B, H, W, Cin, Cout, tile, v_rows, s)
    with _hbm("wino_transform", dw, dw):
        if PROFILER is not None and PROFILER.wants("wino_transform"):
            PROFILER.nbytes["wino_transform"] = PROFILER.nbytes.get("wino_transform", 0.0) + 4.0 * P * Cin * Cout
        call("sg_wino_filter_grad", dU, _p(dw), Cin, Cout, tile, s)


def _v2_ok(K: int, N: int, kh: int, kw: int, same: bool) -> bool:
    return USE_V2 and _low() and K % 64 == 0 and N % 64 == 0 and (same or (kh == 1 and kw == 1))


POOL2_OUT, UPS2_IN = 16, 64
FUSE_POOL = _os.environ.get("SG_FUSE_POOL", "1") == "1"


def conv2d_avgpool_fwd(x, w, bias=None, relu_in=False):
    """avg_pool2x2(conv3x3_same(relu?(x)) + bias) -> [B, H/2, W/2, Cout]: the conv2 -> tf.nn.pool(AVG) pair of a ResNetBlockDown
    (resnet_ops.py:102-106).  Where the convolution runs in the Winograd domain with 4 x 4 tiles the output transform writes the 2 x 2
    means of its tile directly (SG_POOL2_OUT): the full-resolution tensor -- which nothing else reads, forward or backward -- is never
    written or re-read.  Elsewhere: the convolution, then sg_avgpool2_add_fwd."""
    _chk(x, w, bias)
    B, H, W, Cin = x.shape
    kh, kw, wc, Cout = w.shape
    if FUSE_POOL and _wino_ok(Cin, Cout, kh, kw, True, H, W, B) and _wino_tile(H, W) == 4:
        out = empty(B, H // 2, W // 2, Cout, like=x)
        _wino_conv(x, w, out, bias, None, None, Cin, Cout, relu_in, POOL2_OUT, ("wino_fwd", B, H, W, Cin, Cout, kh))
        return out
    return avgpool2_add_fwd(conv2d_fwd(x, w, bias, relu_in=relu_in))


def pooled_grad_foldable(B: int, H: int, W: int, Cin: int, Cout: int, want_dw: bool) -> bool:
    """May the backward of `conv3x3 -> avg_pool2x2` skip the full-resolution gradient?  Yes when the convolution's data-grad (and, if
    wanted, its weight-grad) run in the Winograd domain with 4 x 4 tiles: both gradient transforms then read the pooled gradient."""
    if not (FUSE_POOL and CONV_DTYPE == "f32"):
        return False
    if not (_wino_ok(Cout, Cin, 3, 3, True, H, W, B) and _wino_tile(H, W) == 4):
        return False
    if want_dw and not (_wino_wgrad_ok(Cin, Cout, 3, 3, True, H, W) and B * (H // 4) * (W // 4) >= WINO4_WGRAD_MIN_TILES):
        return False
    return True


def conv2d_avgpool_bwd(x, dout, w, mask, dw=None, db=None, sample_scale=None, relu_in=True):
    """Backward of conv2d_avgpool_fwd: dout [B,H/2,W/2,Cout] is the gradient of the POOLED output; d_c = 0.25 * upsample2x2(dout) is the
    gradient of the convolution's output and is never materialised (UPS2_IN).  dw / db (optional) += the convolution's weight / bias
    gradient (x = its input, relu_in as in the forward, sample_scale as in conv2d_bwd_weight); returns dx = data-grad(d_c), zeroed
    where mask <= 0.  Callers check pooled_grad_foldable first."""
    _chk(x, dout, w, mask, dw, db, sample_scale)
    B, H, W, Cin = x.shape
    Cout = dout.shape[3]
    assert dout.shape[1] * 2 == H and dout.shape[2] * 2 == W and w.shape[2] == Cin and w.shape[3] == Cout
    if dw is not None:
        _wino_wgrad(x, dout, dw, db, sample_scale, relu_in, ups=True)
    dx = empty(B, H, W, Cin, like=x)
    _wino_conv(dout, w, dx, None, None, mask, Cout, Cin, False, 0, ("wino_dgrad", B, H, W, Cin, Cout, 3), ups=True)
    return dx


def conv2d_fwd(x, w, bias=None, bias2=None, same=True, relu_in=False, relu_out=False, tanh_out=False,
               out=None, accum=False, want16=False):
    """want16 (bf16 / fp8 modes): also keep a bf16 twin of the result for the next conv (written by the kernel's epilogue).
    want16 = "only": the result feeds nothing but conv launches that read operand copies (operand_only_ok) -- the kernel writes
    the bf16 copy ALONE and the returned fp32 tensor is a never-written handle (configs c3 / c5: the activation between the two
    chained convs of a ResNetBlockDown is stored as bf16; 2 B/element written instead of 6)."""
    _chk(x, w, bias, bias2, out)
    B, H, W, Cin = x.shape
    kh, kw, wc, Cout = w.shape
    assert wc == Cin, (w.shape, x.shape)
    Ho, Wo = (H, W) if same else (H - kh + 1, W - kw + 1)
    _touch(out)
    if out is None:
        out = empty(B, Ho, Wo, Cout, like=x)
    use8 = _fp8_ok(Cin, Cout, kh, kw, same) and not tanh_out
    only = want16 == "only" and not accum and (use8 or _v2_ok(Cin, Cout, kh, kw, same)) and not tanh_out and operand_only_ok(B, Ho, Wo, Cout)
    if only:
        if GHOST_NAN:
            out.fill_(float("nan"))
        _TWINS[(out.untyped_storage().data_ptr(), "ghost")] = (out,)
        _GHOSTS.add(out.untyped_storage().data_ptr())
    if _wino_ok(Cin, Cout, kh, kw, same, H, W, B) and not tanh_out:
        _wino_conv(x, w, out, bias, bias2, None, Cin, Cout, relu_in, _flags(False, accum, relu_out),
                   ("wino_fwd", B, Ho, Wo, Cin, Cout, kh))
        return out
    with _timed("igemm_fp8" if use8 else "igemm", 2.0 * B * Ho * Wo * kh * kw * Cin * Cout, Cin == 1 or Cout == 1,
                ("fwd", B, Ho, Wo, Cin, Cout, kh), (x, w, out)):
        if use8:
            x8, ax = fp8_of(x, relu_in)                       # the operand ReLU is folded into the conversion
            w8, aw = packed_filter_fp8(w, "fwd")
            y16 = torch.empty(out.shape, device=out.device, dtype=torch.bfloat16) if want16 else None
            am = _amax_slot(out)
            call("sg_conv2d_fwd_fp8", x8.data_ptr(), _p(ax), w8.data_ptr(), _p(aw), _p(bias), _p(bias2), None if only else _p(out),
                 None if y16 is None else y16.data_ptr(), B, H, W, Cin, Cout, kh, kw, int(same), _flags(False, accum, relu_out), _p(am), _stream())
            _amax_put(out, am, None)
            if y16 is not None:
                _twin_put(out, y16)
        elif _v2_ok(Cin, Cout, kh, kw, same) and not tanh_out:
            x16 = bf16_of(x)
            y16 = torch.empty(out.shape, device=out.device, dtype=torch.bfloat16) if want16 else None
            am = _amax_slot(out) if _want_amax() else None
            call("sg_conv2d_fwd_bf16v2", x16.data_ptr(), packed_filter(w, "fwd").data_ptr(), _p(bias), _p(bias2), None if only else _p(out),
                 None if y16 is None else y16.data_ptr(), B, H, W, Cin, Cout, kh, kw, int(same), _flags(relu_in, accum, relu_out), _p(am), _stream())
            if am is not None:
                _amax_put(out, am, None)
            if y16 is not None:
                _twin_put(out, y16)
        elif _f32v2_ok(Cin, Cout, kh, kw, same, out.shape[0] * out.shape[1] * out.shape[2]) and not tanh_out:
            call("sg_conv2d_fwd_v2", _p(x), _p(packed_filter(w, "fwd_f32t")), _p(bias), _p(bias2), _p(out), B, H, W, Cin, Cout, kh, kw,
                 int(same), _flags(relu_in, accum, relu_out), _stream())
        elif _bf16_ok(Cin, Cout) and not tanh_out:
            call("sg_conv2d_fwd_bf16", _p(x), packed_filter(w, "fwd").data_ptr(), _p(bias), _p(bias2), _p(out), B, H, W, Cin, Cout,
                 kh, kw, int(same), _flags(relu_in, accum, relu_out), _stream())
        else:
            call("sg_conv2d_fwd", _p(x), _p(w), _p(bias), _p(bias2), _p(out), B, H, W, Cin, Cout, kh, kw, int(same),
                 _flags(relu_in, accum, relu_out, tanh_out), _stream())
    return out


def conv2d_bwd_data(dy, w, in_hw: Tuple[int, int], mask=None, same=True, out=None, accum=False, want16=False, amax_scale=None):
    """amax_scale [B] (optional, config c5): the per-sample factors the result will carry as a weight-grad operand -- the epilogue
    records max |dx| and max |amax_scale[b] dx| for the fp8 conversion of dx (no amax sweep)."""
    _chk(dy, w, mask, out, amax_scale)
    B = dy.shape[0]
    H, W = in_hw
    kh, kw, Cin, Cout = w.shape
    assert dy.shape[3] == Cout
    _touch(out)
    if out is None:
        out = empty(B, H, W, Cin, like=dy)
    use8 = _fp8_ok(Cout, Cin, kh, kw, same)
    if _wino_ok(Cout, Cin, kh, kw, same, H, W, dy.shape[0]) and tuple(dy.shape[1:3]) == (H, W):
        _wino_conv(dy, w, out, None, None, mask, Cout, Cin, False, _flags(accum=accum),
                   ("wino_dgrad", B, H, W, Cin, Cout, kh))
        return out
    with _timed("igemm_fp8" if use8 else "igemm", 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * kh * kw * Cin * Cout, Cin == 1 or Cout == 1,
                ("dgrad", B, H, W, Cin, Cout, kh), (dy, w, out, mask)):
        if use8:
            dy8, ady = fp8_of(dy)
            w8, aw = packed_filter_fp8(w, "bwd")
            dx16 = torch.empty(out.shape, device=out.device, dtype=torch.bfloat16) if want16 else None
            m16 = None if mask is None else _twin_get(mask)
            am = _amax_slot(out)
            call("sg_conv2d_bwd_data_fp8", dy8.data_ptr(), _p(ady), w8.data_ptr(), _p(aw), None if m16 is not None else _p(mask),
                 None if m16 is None else m16.data_ptr(), _p(out), None if dx16 is None else dx16.data_ptr(), B, H, W, Cin, Cout,
                 kh, kw, int(same), _flags(accum=accum), _p(am), _p(amax_scale), _stream())
            _amax_put(out, am, amax_scale)
            if dx16 is not None:
                _twin_put(out, dx16)
        elif _v2_ok(Cout, Cin, kh, kw, same):
            dy16 = bf16_of(dy)
            dx16 = torch.empty(out.shape, device=out.device, dtype=torch.bfloat16) if want16 else None
            m16 = None if mask is None else _twin_get(mask)          # the ReLU mask as bf16 when a twin exists (half the bytes)
            am = _amax_slot(out) if _want_amax() else None
            call("sg_conv2d_bwd_data_bf16v2", dy16.data_ptr(), packed_filter(w, "bwd").data_ptr(), None if m16 is not None else _p(mask),
                 None if m16 is None else m16.data_ptr(), _p(out),
                 None if dx16 is None else dx16.data_ptr(), B, H, W, Cin, Cout, kh, kw, int(same), _flags(accum=accum), _p(am),
                 _p(amax_scale), _stream())
            if am is not None:
                _amax_put(out, am, amax_scale)
            if dx16 is not None:
                _twin_put(out, dx16)
        elif _f32v2_ok(Cout, Cin, kh, kw, same, B * H * W):
            call("sg_conv2d_bwd_data_v2", _p(dy), _p(w), _p(mask), _p(out), B, H, W, Cin, Cout, kh, kw, int(same), _flags(accum=accum), _stream())
        elif _bf16_ok(Cout, Cin):
            call("sg_conv2d_bwd_data_bf16", _p(dy), packed_filter(w, "bwd").data_ptr(), _p(mask), _p(out), B, H, W, Cin, Cout,
                 kh, kw, int(same), _flags(accum=accum), _stream())
        elif TRANSPOSED_DGRAD_FILTERS and Cin >= 128 and Cout >= 128 and Cin % 4 == 0:
            # a transposed filter copy (made once per optimizer step) lets the launch use the straight [K,N] filter loader
            call("sg_conv2d_bwd_data_wt", _p(dy), _p(packed_filter(w, "bwd_f32")), _p(mask), _p(out), B, H, W, Cin, Cout,
                 kh, kw, int(same), _flags(accum=accum), _stream())
        else:
            call("sg_conv2d_bwd_data", _p(dy), _p(w), _p(mask), _p(out), B, H, W, Cin, Cout, kh, kw, int(same),
                 _flags(accum=accum), _stream())
    return out


def conv2d_bwd_weight(x, dy, dw, same=True, relu_in=False, db=None, sample_scale=None):
    """dw += weight-grad; db (optional, Cout > 1) += bias-grad in the same sweep over dy; sample_scale [B] (optional)
    weights each sample's contribution."""
    _chk(x, dy, dw, db, sample_scale)
    B, H, W, Cin = x.shape
    kh, kw, wc, Cout = dw.shape
    assert wc == Cin and dy.shape[3] == Cout
    if _wino_wgrad_ok(Cin, Cout, kh, kw, same, H, W):
        _wino_wgrad(x, dy, dw, db, sample_scale, relu_in)
        return
    if _fp8_wgrad_ok(Cin, Cout, kh, kw, same):
        # config c5: e4m3 activations (the copy the forward launch read: ReLU folded into the conversion) x e5m2 gradients
        x8, ax = fp8_of(x, relu_in)
        dy8, ady, colsum = grad_operand_fp8(dy, sample_scale, db is not None)
        if db is not None:
            add(db, colsum, out=db)
        with _timed("wgrad_fp8", 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * kh * kw * Cin * Cout, False, ("wgrad", B, H, W, Cin, Cout, kh)):
            call("sg_conv2d_bwd_weight_fp8", x8.data_ptr(), _p(ax), dy8.data_ptr(), _p(ady), _p(dw), B, H, W, Cin, Cout, kh, kw, int(same), _stream())
        return
    if USE_V2 and _low() and (same or kh * kw == 1) and ((Cin % 64 == 0 and Cout % 256 == 0) or (Cin == 64 and Cout == 64 and not DETERMINISTIC)):
        # second-generation path: bf16 operands by DMA; the per-sample factors are folded into dy's bf16 copy, whose
        # conversion sweep also yields the bias gradient (fp32 column sums) and, if missing, the plain twin for the data-grad
        x16 = bf16_of(x)
        if db is None and sample_scale is None:
            dy16 = bf16_of(dy)
        elif db is None:
            dy16 = bf16_scaled(dy, sample_scale)
        else:
            dy16, colsum = grad_operand(dy, sample_scale, True)
            add(db, colsum, out=db)
        with _timed("wgrad", 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * kh * kw * Cin * Cout, False, ("wgrad", B, H, W, Cin, Cout, kh)):
            call("sg_conv2d_bwd_weight_bf16v2", x16.data_ptr(), dy16.data_ptr(), _p(dw), B, H, W, Cin, Cout, kh, kw, int(same),
                 _flags(relu_in), _stream())
        return
    with _timed("wgrad", 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * kh * kw * Cin * Cout, Cin == 1 or Cout == 1,
                ("wgrad", B, H, W, Cin, Cout, kh), (x, dy)):
        call("sg_conv2d_bwd_weight", _p(x), _p(dy), _p(dw), _p(db), _p(sample_scale), B, H, W, Cin, Cout, kh, kw, int(same),
             _flags(relu_in) | (MMA_BF16 if _low() else 0), _stream())


def conv2d_transpose_fwd(x, w, bias=None, bias2=None, stride=(2, 2), out=None, accum=False):
    _chk(x, w, bias, bias2, out)
    _touch(out)
    B, H, W, Cin = x.shape
    kh, kw, Cout, wc = w.shape
    assert wc == Cin
    sh, sw = stride
    if out is None:
        out = empty(B, sh * H, sw * W, Cout, like=x)
    with _timed("igemm", 2.0 * B * H * W * kh * kw * Cin * Cout, False, ("convT_fwd", B, H, W, Cin, Cout, kh), (x, w, out)):
        # bf16 mode: every parity class needs a tap (kernel >= stride), otherwise the fp32 entry point writes the bias-only classes
        if _bf16_ok(Cin, Cout) and kh >= sh and kw >= sw:
            call("sg_conv2d_transpose_fwd_bf16", _p(x), packed_filter(w, "t_fwd").data_ptr(), _p(bias), _p(bias2), _p(out), B, H, W,
                 Cin, Cout, kh, kw, sh, sw, _flags(accum=accum), _stream())
        else:
            call("sg_conv2d_transpose_fwd", _p(x), _p(w), _p(bias), _p(bias2), _p(out), B, H, W, Cin, Cout, kh, kw, sh, sw,
                 _flags(accum=accum), _stream())
    return out


def conv2d_transpose_bwd_data(dy, w, stride=(2, 2), mask=None, out=None, accum=False):
    _chk(dy, w, mask, out)
    _touch(out)
    kh, kw, Cout, Cin = w.shape
    sh, sw = stride
    B, Hs, Ws, _ = dy.shape
    H, W = Hs // sh, Ws // sw
    if out is None:
        out = empty(B, H, W, Cin, like=dy)
    with _timed("igemm", 2.0 * B * H * W * kh * kw * Cin * Cout, False, ("convT_dgrad", B, H, W, Cin, Cout, kh), (dy, w, out, mask)):
        if _bf16_ok(Cout, Cin):
            call("sg_conv2d_transpose_bwd_data_bf16", _p(dy), packed_filter(w, "t_bwd").data_ptr(), _p(mask), _p(out), B, H, W,
                 Cin, Cout, kh, kw, sh, sw, _flags(accum=accum), _stream())
        else:
            call("sg_conv2d_transpose_bwd_data", _p(dy), _p(w), _p(mask), _p(out), B, H, W, Cin, Cout, kh, kw, sh, sw,
                 _flags(accum=accum), _stream())
    return out


def conv2d_transpose_bwd_weight(x, dy, dw, stride=(2, 2)):
    _chk(x, dy, dw)
    B, H, W, Cin = x.shape
    kh, kw, Cout, wc = dw.shape
    assert wc == Cin
    sh, sw = stride
    if USE_V2 and _low() and Cin % 64 == 0 and Cout % 64 == 0 and H * W >= 64 and (Cin % 256 == 0 or not DETERMINISTIC):
        x16, dy16 = bf16_of(x), bf16_of(dy)
        with _timed("wgrad", 2.0 * B * H * W * kh * kw * Cin * Cout, False, ("convT_wgrad", B, H, W, Cin, Cout, kh)):
            call("sg_conv2d_transpose_bwd_weight_bf16v2", x16.data_ptr(), dy16.data_ptr(), _p(dw), B, H, W, Cin, Cout, kh, kw, sh, sw, _stream())
        return
    with _timed("wgrad", 2.0 * B * H * W * kh * kw * Cin * Cout, False, ("convT_wgrad", B, H, W, Cin, Cout, kh)):
        call("sg_conv2d_transpose_bwd_weight", _p(x), _p(dy), _p(dw), B, H, W, Cin, Cout, kh, kw, sh, sw, 0, _stream())


def bias_grad(dy, db):
    _chk(dy, db)
    N = dy.shape[-1]
    with _hbm("bias_grad", dy):
        call("sg_bias_grad", _p(dy), _p(db), dy.numel() // N, N, _stream())


# ---------------------------------------------------------------- pooling / elementwise
def avgpool2_add_fwd(a, b=None):
    _chk(a, b)
    B, H, W, C = a.shape
    out = empty(B, H // 2, W // 2, C, like=a)
    with _hbm("pool", a, b, out):
        call("sg_avgpool2_add_fwd", _p(a), _p(b), _p(out), B, H, W, C, _stream())
    return out


def avgpool2_bwd(dout):
    _chk(dout)
    B, Ho, Wo, C = dout.shape
    dx = empty(B, 2 * Ho, 2 * Wo, C, like=dout)
    with _hbm("pool", dout, dx):
        call("sg_avgpool2_bwd", _p(dout), _p(dx), B, 2 * Ho, 2 * Wo, C, _stream())
    return dx


def add(a, b, out=None):
    _chk(a, b, out)
    _touch(out)
    assert a.shape == b.shape
    if out is None:
        out = torch.empty_like(a)
    with _hbm("elementwise", a, b, out):
        call("sg_add", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def normalize_u8(u8: torch.Tensor) -> torch.Tensor:
    """uint8 pixels on the device -> fp32 (x - 127.5) / 127.5 (data_utils.py:82), same shape."""
    assert u8.is_cuda and u8.dtype == torch.uint8 and u8.is_contiguous() and u8.numel() % 16 == 0
    out = torch.empty(u8.shape, device=u8.device, dtype=torch.float32)
    with _hbm("elementwise", u8, out):
        call("sg_normalize_u8", u8.data_ptr(), _p(out), u8.numel(), _stream())
    return out


def bias_add(y, bias):
    """y[..., c] += bias[c] in place."""
    _chk(y, bias)
    _touch(y)
    C = y.shape[-1]
    with _hbm("elementwise", y, y):
        call("sg_bias_add", _p(y), _p(bias), y.numel() // C, C, _stream())
    return y


def relu_mask(dy, ref, out=None):
    _chk(dy, ref, out)
    _touch(out)
    if out is None:
        out = torch.empty_like(dy)
    with _hbm("elementwise", dy, ref, out):
        call("sg_relu_mask", _p(dy), _p(ref), _p(out), dy.numel(), _stream())
    return out


def tanh_bwd(y, dy):
    _chk(y, dy)
    dx = torch.empty_like(dy)
    with _hbm("elementwise", y, dy, dx):
        call("sg_tanh_bwd", _p(y), _p(dy), _p(dx), dy.numel(), _stream())
    return dx


def maxpool_fwd(x, ph, pw):
    _chk(x)
    B, H, W, C = x.shape
    y = empty(B, H // ph, W // pw, C, like=x)
    idx = empty(B, H // ph, W // pw, C, like=x, dtype=torch.uint8)
    with _hbm("pool", x, y, idx):
        call("sg_maxpool_fwd", _p(x), _p(y), idx.data_ptr(), B, H, W, C, ph, pw, _stream())
    return y, idx


def maxpool_bwd(dy, idx, ph, pw, out=None, accum=False):
    _chk(dy, out)
    _touch(out)
    B, Ho, Wo, C = dy.shape
    if out is None:
        out = empty(B, Ho * ph, Wo * pw, C, like=dy)
    with _hbm("pool", dy, idx, out):
        call("sg_maxpool_bwd", _p(dy), idx.data_ptr(), _p(out), B, Ho * ph, Wo * pw, C, ph, pw, int(accum), _stream())
    return out


def gap_fwd(x, relu=True):
    _chk(x)
    B, H, W, C = x.shape
    out = empty(B, C, like=x)
    with _hbm("pool", x, out):
        call("sg_gap_fwd", _p(x), _p(out), B, H * W, C, int(relu), _stream())
    return out


def gap_bwd(dout, x, relu=True):
    _chk(dout, x)
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    with _hbm("pool", dout, x, dx):
        call("sg_gap_bwd", _p(dout), _p(x), _p(dx), B, H * W, C, int(relu), _stream())
    return dx


def scale_add(o, x, sigma, out=None):
    _chk(o, x, sigma, out)
    _touch(out)
    if out is None:
        out = torch.empty_like(x)
    with _hbm("elementwise", o, x, out):
        call("sg_scale_add", _p(o), _p(x), _p(sigma), _p(out), x.numel(), _stream())
    return out


def scale(a, s):
    _chk(a, s)
    out = torch.empty_like(a)
    with _hbm("elementwise", a, out):
        call("sg_scale", _p(a), _p(s), _p(out), a.numel(), _stream())
    return out


def dot_accum(a, b, out):
    _chk(a, b, out)
    with _hbm("elementwise", a, b):
        call("sg_dot_accum", _p(a), _p(b), _p(out), a.numel(), _stream())


def rowscale(x, s):
    _chk(x, s)
    out = torch.empty_like(x)
    rows = s.numel()
    with _hbm("elementwise", x, out):
        call("sg_rowscale", _p(x), _p(s), _p(out), rows, x.numel() // rows, _stream())
    return out


# ---------------------------------------------------------------- dense
def gemm(A, B, M, N, K, lda, ldb, transA=False, transB=False, bias=None, out=None, ldc=None, alpha=1.0, beta=0.0,
         A_off=0, out_off=0):
    """C = alpha*op(A)*op(B) + beta*C (+bias).  A_off / out_off are element offsets into A / out, which with
    lda / ldc address a column block of a wider row-major matrix (the z chunks of the generator)."""
    _chk(A, B, bias, out)
    _touch(out)
    if out is None:
        out = empty(M, N, like=A)
    ldc = N if ldc is None else ldc
    call("sg_gemm", A.data_ptr() + 4 * A_off, _p(B), out.data_ptr() + 4 * out_off, _p(bias), M, N, K, lda, ldb, ldc,
         int(transA), int(transB), float(alpha), float(beta), _stream())
    return out


def dense_fwd(x2d, w, bias=None):
    """x2d [M,K] @ w [K,N] (+bias)."""
    M, K = x2d.shape
    N = w.shape[1]
    return gemm(x2d, w, M, N, K, K, N, bias=bias)


def dense_bwd_input(dy2d, w):
    """dy [M,N] @ w^T [N,K] -> [M,K]."""
    M, N = dy2d.shape
    K = w.shape[0]
    return gemm(dy2d, w, M, K, N, N, N, transB=True)


def dense_bwd_weight(x2d, dy2d, dw):
    """dw [K,N] += x^T [K,M] @ dy [M,N]."""
    M, K = x2d.shape
    N = dy2d.shape[1]
    gemm(x2d, dy2d, K, N, M, K, N, transA=True, out=dw, beta=1.0)


# ---------------------------------------------------------------- batch norm
def bn_stats_sums(x):
    """fp64 [2C]: per-channel sum and sum of squares over all rows of x[..., C]."""
    _chk(x)
    C = x.shape[-1]
    M = x.numel() // C
    ws = empty(lib().sg_bn_stats_workspace_floats(M, C), like=x)
    sums = empty(2 * C, like=x, dtype=torch.float64)
    with _hbm("bn_stats", x):
        call("sg_bn_stats_sums", _p(x), M, C, _p(ws), sums.data_ptr(), _stream())
    return sums


def bn_stats_finalize(sums, count, like):
    C = sums.numel() // 2
    mean, var = empty(C, like=like), empty(C, like=like)
    call("sg_bn_stats_finalize", sums.data_ptr(), float(count), _p(mean), _p(var), C, _stream())
    return mean, var


def bn_apply(x, mean, var, gamma, beta, per_sample: bool, relu: bool, eps=BN_EPS):
    _chk(x, mean, var, gamma, beta)
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    with _hbm("bn_apply", x, y):
        call("sg_bn_apply", _p(x), _p(mean), _p(var), _p(gamma), _p(beta), C if per_sample else 0, _p(y), B, H * W, C, eps, int(relu), _stream())
    return y


def bn_bwd_reduce(dy, y, x, mean, var, gamma, per_sample: bool, relu: bool, eps=BN_EPS, dgamma_c=None, dbeta_c=None):
    """-> dgamma [B,C], dbeta [B,C] (per-sample sums), chan fp64 [4C]; dgamma_c/dbeta_c [C] += sums over b."""
    _chk(dy, y, x, mean, var, gamma, dgamma_c, dbeta_c)
    B, H, W, C = x.shape
    dgamma = torch.zeros(B, C, device=x.device, dtype=torch.float32)
    dbeta = torch.zeros(B, C, device=x.device, dtype=torch.float32)
    chan = empty(4 * C, like=x, dtype=torch.float64)
    with _hbm("bn_bwd_reduce", dy, y, x):
        call("sg_bn_bwd_reduce", _p(dy), _p(y), _p(x), _p(mean), _p(var), _p(gamma), C if per_sample else 0, _p(dgamma), _p(dbeta),
             chan.data_ptr(), _p(dgamma_c), _p(dbeta_c), B, H * W, C, eps, int(relu), _stream())
    return dgamma, dbeta, chan


def bn_bwd_apply(dy, y, x, mean, var, gamma, per_sample: bool, chan, count, relu: bool, use_stats: bool, eps=BN_EPS):
    _chk(dy, y, x, mean, var, gamma)
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    with _hbm("bn_bwd_apply", dy, y, x, dx):
        call("sg_bn_bwd_apply", _p(dy), _p(y), _p(x), _p(mean), _p(var), _p(gamma), C if per_sample else 0,
             None if chan is None else chan.data_ptr(), float(count), _p(dx), B, H * W, C, eps, int(relu), int(use_stats), _stream())
    return dx


def bn_update_moving(mm, mv, mean, var, count, momentum=0.99):
    _chk(mm, mv, mean, var)
    call("sg_bn_update_moving", _p(mm), _p(mv), _p(mean), _p(var), float(count), momentum, mm.numel(), _stream())


# ---------------------------------------------------------------- filter bank
def filterbank_fwd(z, y, table):
    _chk(z, table)
    B, L = y.shape
    assert y.dtype == torch.int32 and y.is_contiguous() and z.shape[1] == 128
    assert table.shape[1] == 32 and table.shape[2] == 8192
    seed = empty(B, 4, 4 * L, 512, like=z)
    with _hbm("filterbank_fwd", seed, table):
        call("sg_filterbank_fwd", _p(z), y.data_ptr(), _p(table), _p(seed), B, L, table.shape[0], _stream())
    return seed


def filterbank_bwd(z, y, table, dseed, dtable, dz):
    _chk(z, table, dseed, dtable, dz)
    B, L = y.shape
    with _hbm("filterbank_bwd", dseed, table, dtable):
        call("sg_filterbank_bwd", _p(z), y.data_ptr(), _p(table), _p(dseed), _p(dtable), _p(dz), B, L, table.shape[0], _stream())


# ---------------------------------------------------------------- attention
def attention_fwd(theta, phi, g):
    _chk(theta, phi, g)
    B, Nq, dk = theta.shape
    Nk, dv = g.shape[1], g.shape[2]
    out = empty(B, Nq, dv, like=theta)
    lse = empty(B, Nq, like=theta)
    with _hbm("attention_fwd", theta, phi, g, out, flops=2.0 * B * Nq * Nk * (dk + dv), roof="mfma_f32"):
        call("sg_attention_fwd", _p(theta), _p(phi), _p(g), _p(out), _p(lse), B, Nq, Nk, dk, dv, _stream())
    return out, lse


def attention_bwd(theta, phi, g, out, lse, dout):
    _chk(theta, phi, g, out, lse, dout)
    B, Nq, dk = theta.shape
    Nk, dv = g.shape[1], g.shape[2]
    dtheta, dphi, dg = torch.empty_like(theta), torch.empty_like(phi), torch.empty_like(g)
    delta = empty(B, Nq, like=theta)
    with _hbm("attention_bwd", theta, phi, g, out, dout, dtheta, dphi, dg, flops=2.0 * B * Nq * Nk * (3 * dk + 2 * dv + dv), roof="mfma_f32"):
        call("sg_attention_bwd", _p(theta), _p(phi), _p(g), _p(out), _p(lse), _p(dout), _p(dtheta), _p(dphi), _p(dg), _p(delta),
             B, Nq, Nk, dk, dv, _stream())
    return dtheta, dphi, dg


# ---------------------------------------------------------------- CTC
def softmax_ctc(logits, labels, input_length, label_length, need_grad=True):
    _chk(logits)
    B, T, C = logits.shape
    assert labels.dtype == torch.int32 and labels.is_contiguous() and labels.shape[0] == B
    if not (0 < int(label_length) <= labels.shape[1]) or not (0 < int(input_length) <= T):
        raise ValueError("CTC lengths out of range: input_length %d (T = %d), label_length %d (labels %s)"
                         % (input_length, T, label_length, tuple(labels.shape)))
    loss = empty(B, like=logits)
    dlogits = torch.empty_like(logits) if need_grad else None
    call("sg_softmax_ctc", _p(logits), labels.data_ptr(), labels.shape[1], _p(loss), _p(dlogits), B, T, C, int(input_length),
         int(label_length), _stream())
    return loss, dlogits


# ---------------------------------------------------------------- loss head
def loss_sums(d_r, d_f, s_my, s_f, s_r, r_f, r_r, mode: int):
    _chk(d_r, d_f, s_my, s_f, s_r, r_f, r_r)
    sums = empty(12, like=d_r, dtype=torch.float64)
    call("sg_loss_sums", _p(d_r), _p(d_f), _p(s_my), _p(s_f), _p(s_r), _p(r_f), _p(r_r), d_r.numel(), mode, sums.data_ptr(), _stream())
    return sums


def loss_grads(d_r, d_f, s_my, s_f, s_r, r_f, mode: int, balance: bool, alpha: float, sums):
    B = d_r.numel()
    scalars = empty(16, like=d_r)
    outs = [empty(B, like=d_r) for _ in range(7)]
    shD, shS = empty(3, B, like=d_r), empty(3, B, like=d_r)
    call("sg_loss_grads", _p(d_r), _p(d_f), _p(s_my), _p(s_f), _p(s_r), _p(r_f), B, mode, int(balance), float(alpha), sums.data_ptr(),
         _p(scalars), *[_p(o) for o in outs], _p(shD), _p(shS), _stream())
    return scalars, outs + [shD, shS]


# ---------------------------------------------------------------- optimizers / spectral norm
def adam_update(p, g, m, v, lr_t, beta_1, beta_2, eps=1e-7):
    """lr_t: the bias-corrected step size as a float, or a 1-element device tensor holding it (graph-captured steps)."""
    _chk(p, g, m, v)
    with _hbm("adam", p, p, g, m, m, v, v):
        if torch.is_tensor(lr_t):
            call("sg_adam_update_dlr", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(lr_t), float(beta_1), float(beta_2), float(eps), _stream())
        else:
            call("sg_adam_update", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr_t), float(beta_1), float(beta_2), float(eps), _stream())
    weights_changed()


def rmsprop_update(p, g, ms, lr, rho=0.9, eps=1e-7):
    _chk(p, g, ms)
    with _hbm("adam", p, p, g, ms, ms):
        call("sg_rmsprop_update", _p(p), _p(g), _p(ms), p.numel(), float(lr), float(rho), float(eps), _stream())
    weights_changed()


def spectral_norm(w, u, power_iteration=1):
    _chk(w, u)
    N = w.shape[-1]
    K = w.numel() // N
    out = torch.empty_like(w)
    ws = empty(lib().sg_spectral_norm_workspace_floats(K, N), like=w)
    call("sg_spectral_norm", _p(w), _p(u), _p(out), _p(ws), K, N, int(power_iteration), _stream())
    return out


def spectral_norm_bwd(w, u, g, dw):
    """dw += gradient w.r.t. w of spectral_norm(w, u, 1) given g = gradient w.r.t. the normalised weight."""
    _chk(w, u, g, dw)
    N = w.shape[-1]
    K = w.numel() // N
    ws = empty(lib().sg_spectral_norm_bwd_workspace_floats(K, N), like=w)
    call("sg_spectral_norm_bwd", _p(w), _p(u), _p(g), _p(dw), _p(ws), K, N, _stream())


def loss_terms(d_r, d_f, s_my, s_f, s_r, mode: int):
    _chk(d_r, d_f, s_my, s_f, s_r)
    B = d_r.numel()
    out = empty(7, B, like=d_r)
    call("sg_loss_terms", _p(d_r), _p(d_f), _p(s_my), _p(s_f), _p(s_r), B, mode, _p(out), _stream())
    return out


# ---------------------------------------------------------------- make_my_recognizer extras
def leaky_relu_fwd(x, alpha=0.01):
    _chk(x)
    y = torch.empty_like(x)
    call("sg_leaky_relu_fwd", _p(x), _p(y), x.numel(), float(alpha), _stream())
    return y


def leaky_relu_bwd(dy, x, alpha=0.01):
    _chk(dy, x)
    dx = torch.empty_like(dy)
    call("sg_leaky_relu_bwd", _p(dy), _p(x), _p(dx), dy.numel(), float(alpha), _stream())
    return dx


def mul_mask(x, mask, rows_per_mask=1):
    """x [rows, cols] (any leading shape) times mask[rows / rows_per_mask, cols]."""
    _chk(x, mask)
    cols = x.shape[-1]
    out = torch.empty_like(x)
    call("sg_mul_mask", _p(x), _p(mask), _p(out), x.numel() // cols, cols, int(rows_per_mask), _stream())
    return out


def lstm_cell_fwd(z, z_off, ldz, c_prev, c_out, h_out, h_off, ldh, h_copy, hc_off, ldc, B, H):
    call("sg_lstm_cell_fwd", z.data_ptr() + 4 * z_off, ldz, _p(c_prev), _p(c_out), h_out.data_ptr() + 4 * h_off, ldh,
         None if h_copy is None else h_copy.data_ptr() + 4 * hc_off, ldc, B, H, _stream())


def lstm_cell_bwd(gates, g_off, ldz, c_prev, c_t, dh_a, a_off, lda, dh_b, dc_next, dc_prev, B, H):
    call("sg_lstm_cell_bwd", gates.data_ptr() + 4 * g_off, ldz, _p(c_prev), _p(c_t), dh_a.data_ptr() + 4 * a_off, lda, _p(dh_b),
         _p(dc_next), _p(dc_prev), B, H, _stream())
