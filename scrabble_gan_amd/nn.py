"""Host-side building blocks: flat parameter stores and the explicit forward/backward of every
block of the ScrabbleGAN nets, expressed over the C-ABI kernels in `ops`.

There is no autograd: each `*_fwd` returns (output, ctx) and the matching `*_bwd` consumes ctx,
accumulates weight gradients into the store's flat gradient buffer (the kernels' += contract) and
returns the input gradient.  That keeps the four "tapes" of the reference train_step
(/root/reference/src/bigacgan/data_utils.py:398-468) as four explicit backward sweeps over saved
contexts, lets D/S passes on different batches share one gradient buffer, and makes the flat
buffers directly all-reducible (RCCL) and updatable (one fused Adam launch per network).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops


# --------------------------------------------------------------------------------------------
# initialisers (Keras semantics; host-side, run once)
# --------------------------------------------------------------------------------------------
FAST_INIT = False          # tests that overwrite every weight right after construction skip the QR factorisations
_ORTHO_CACHE: Dict = {}     # (shape, generator state) -> (kernel, generator state afterwards): identical re-creations are free


def orthogonal(shape: Sequence[int], gen: torch.Generator) -> torch.Tensor:
    """tf.initializers.orthogonal(gain=1): QR of a N(0,1) matrix flattened to [prod(shape[:-1]), shape[-1]]."""
    rows, cols = int(math.prod(shape[:-1])), int(shape[-1])
    if FAST_INIT:
        return (torch.randn(tuple(shape), generator=gen) / math.sqrt(max(rows, cols))).contiguous()
    big = rows * cols >= 1 << 18
    if big:
        key = (tuple(shape), hash(gen.get_state().numpy().tobytes()))
        hit = _ORTHO_CACHE.get(key)
        if hit is not None:
            gen.set_state(hit[1])
            return hit[0].clone()
    a = torch.randn((cols, rows) if rows < cols else (rows, cols), generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if rows < cols:
        q = q.t()
    out = q.reshape(tuple(shape)).float().contiguous()
    if big:
        if len(_ORTHO_CACHE) > 64:
            _ORTHO_CACHE.clear()
        _ORTHO_CACHE[key] = (out.clone(), gen.get_state())
    return out


def glorot_uniform(shape: Sequence[int], gen: torch.Generator) -> torch.Tensor:
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(math.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(tuple(shape), generator=gen, dtype=torch.float64) * 2 - 1) * lim).float()


def zeros(shape, gen=None):
    return torch.zeros(tuple(shape))


def ones(shape, gen=None):
    return torch.ones(tuple(shape))


# --------------------------------------------------------------------------------------------
# flat parameter store
# --------------------------------------------------------------------------------------------
class ParamStore:
    """All trainable variables of one network in ONE flat fp32 buffer (16-byte aligned slices),
    with a same-shaped flat gradient buffer; non-trainable state (BN moving statistics) in a
    second flat buffer.  `p[name]` / `g[name]` are views."""

    def __init__(self, specs: List[Tuple[str, Tuple[int, ...], Callable, bool]], device, gen: torch.Generator):
        self.device = device
        self.names: List[str] = []
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        self.trainable: Dict[str, bool] = {}
        offs = {True: 0, False: 0}
        self._off: Dict[str, int] = {}
        host = {}
        for name, shape, init, trainable in specs:
            assert name not in self.shapes, name
            self.names.append(name)
            self.shapes[name] = tuple(shape)
            self.trainable[name] = trainable
            self._off[name] = offs[trainable]
            n = int(math.prod(shape)) if len(shape) else 1
            offs[trainable] += (n + 3) // 4 * 4
            host[name] = init(shape, gen).reshape(-1).float()
        self.flat = torch.zeros(max(offs[True], 4), device=device)
        self.grad = torch.zeros_like(self.flat)
        self.state = torch.zeros(max(offs[False], 4), device=device)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        for name in self.names:
            n = host[name].numel()
            o = self._off[name]
            buf = self.flat if self.trainable[name] else self.state
            view = buf[o:o + n].view(self.shapes[name])
            view.copy_(host[name].view(self.shapes[name]))
            self.p[name] = view
            if self.trainable[name]:
                self.g[name] = self.grad[o:o + n].view(self.shapes[name])

    def zero_grad(self):
        self.grad.zero_()

    def trainable_names(self) -> List[str]:
        return [n for n in self.names if self.trainable[n]]

    def load(self, weights: Dict[str, torch.Tensor]):
        for k, v in weights.items():
            self.p[k].copy_(v.reshape(self.shapes[k]).to(self.device, torch.float32))

    def export(self) -> Dict[str, torch.Tensor]:
        return {k: self.p[k].detach().cpu().clone() for k in self.names}

    def num_params(self) -> int:
        return sum(int(math.prod(self.shapes[n])) if self.shapes[n] else 1 for n in self.trainable_names())


class AppliedSNStore:
    """kernel_reg = 'applied' (the evident intent of scrabble_gan.gin:22 with arch_ops.py:98-126): one
    forward pass of a network convolves with w~ = spectral_norm(w, u) for every weight the reference registers a
    kernel_regularizer on (all Conv2D / Conv2DTranspose / Dense kernels of G, D and S; not biases, BatchNorm affines, the
    filter bank or the recognizer), u ~ N(0,1) drawn afresh for the pass (arch_ops.py:110).  This object stands in for the
    ParamStore during that pass and its backward sweeps: `p[name]` is w~ for those weights, `g[name]` a shadow gradient
    (d/dw~); fold() pushes the shadow gradients through the power iteration (sg_spectral_norm_bwd) into the real store."""

    def __init__(self, store: "ParamStore", names, gen: torch.Generator):
        self.base = store
        self.device = store.device
        self._off = store._off
        self.shapes = store.shapes
        self.p = dict(store.p)
        self.g = dict(store.g)
        self.sn = {}
        for n in names:
            w = store.p[n]
            u = torch.randn(w.shape[-1], generator=gen).to(store.device)        # [1, N] of arch_ops.py:110
            self.p[n] = ops.spectral_norm(w, u).view(w.shape)
            self.g[n] = torch.zeros_like(w)
            self.sn[n] = u

    def fold(self):
        """real gradient += d w~/d w applied to the shadow gradient; the shadow gradients are then cleared."""
        for n, u in self.sn.items():
            ops.spectral_norm_bwd(self.base.p[n], u, self.g[n], self.base.g[n])
            self.g[n].zero_()


def sn_names(store: "ParamStore"):
    """The weights that carry kernel_regularizer=k_reg in the reference: every trainable kernel with >= 2 dims except the
    filter bank (arch_ops.py:85-87: add_weight without regularizer) -- resnet_ops.py:18,24,57,66,71,99,104,111;
    net_architecture.py:254,286,345,404; arch_ops.py:40,46,57,65 (the NonLocalBlock kernels in 'persistent' mode)."""
    return [n for n in store.trainable_names() if len(store.shapes[n]) >= 2 and n != "filter_bank"]


class Reducer:
    """Cross-rank sum hook (data parallel).  The default is the identity (one process).

    sync_bn: BatchNorm statistics of the generator are summed over all ranks (SyncBN: the step then equals the
    single-device step on the global batch).  The 7 forward + 7 backward statistic exchanges of a step cannot
    be merged into fewer collectives: each BatchNorm's input depends on the previous one's output, so the exchanges are
    sequentially dependent (each is a <= 8 KB latency-bound all-reduce, ~14 x 20-30 us against a 55 ms step at the 8-way
    shard size).  sync_bn=False (configure(sync_bn=False)) normalises with per-rank statistics instead: no BN collective
    at all, not equivalent to the single-device step."""

    world_size = 1
    sync_bn = True

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def all_reduce_sum_async(self, t: torch.Tensor):
        """Start a SUM all-reduce of `t` that may overlap later kernels; returns a handle for wait()."""
        return None

    def wait(self, handle) -> None:
        return None


LOCAL = Reducer()


# --------------------------------------------------------------------------------------------
# ResNetBlockDown (/root/reference/src/bigacgan/resnet_ops.py:84-120)
# --------------------------------------------------------------------------------------------
def block_down_specs(pre: str, cin: int, cout: int):
    return [(pre + ".conv1.w", (3, 3, cin, cout), orthogonal, True), (pre + ".conv1.b", (cout,), zeros, True),
            (pre + ".conv2.w", (3, 3, cout, cout), orthogonal, True), (pre + ".conv2.b", (cout,), zeros, True),
            (pre + ".short.w", (1, 1, cin, cout), orthogonal, True), (pre + ".short.b", (cout,), zeros, True)]


POOL_FIRST_SHORTCUT = True


def block_down_fwd(x, S: ParamStore, pre: str, is_last: bool):
    """The shortcut branch of a pooling block is avg_pool(conv1x1(x) + b) (resnet_ops.py:109-113).  A 1x1 convolution and a
    2x2 mean are both linear and act on different axes, so it equals conv1x1(avg_pool(x)) + b: the SAME sums in another
    order, on a quarter of the pixels (4x fewer FLOPs for the forward, data-grad and weight-grad launches of every pooled
    shortcut; they are the least efficient MFMA launches of the step: 64 / 512 reduction terms).  POOL_FIRST_SHORTCUT = False
    keeps the reference's order; both are parity-tested against the oracle (which pools last)."""
    p = S.p
    # (c1 feeds only conv2 and conv2's backward launches: in bf16 / fp8 modes it may exist as a bf16 operand copy alone)
    c1 = ops.conv2d_fwd(x, p[pre + ".conv1.w"], p[pre + ".conv1.b"], relu_in=True, want16="only")        # :97-99
    xp = None
    if is_last:
        out = ops.conv2d_fwd(x, p[pre + ".short.w"], p[pre + ".short.b"])                     # :109-111
        ops.conv2d_fwd(c1, p[pre + ".conv2.w"], p[pre + ".conv2.b"], relu_in=True, out=out, accum=True)   # :102-104,114
    elif POOL_FIRST_SHORTCUT and x.shape[-1] % 4 == 0:
        xp = ops.avgpool2_add_fwd(x)                                                          # pooled block input
        out = ops.conv2d_avgpool_fwd(c1, p[pre + ".conv2.w"], p[pre + ".conv2.b"], relu_in=True)   # :102-106 (pooled in the conv's epilogue where it can be)
        ops.conv2d_fwd(xp, p[pre + ".short.w"], p[pre + ".short.b"], out=out, accum=True, want16=True)     # :109-114, pooled first
        # (want16: in bf16 mode the launch that completes the block's output also writes its bf16 twin -- the next block's conv
        #  operand -- instead of a conversion sweep over the finished tensor)
    else:
        c2 = ops.conv2d_fwd(c1, p[pre + ".conv2.w"], p[pre + ".conv2.b"], relu_in=True)
        s = ops.conv2d_fwd(x, p[pre + ".short.w"], p[pre + ".short.b"])
        out = ops.avgpool2_add_fwd(c2, s)                                                     # :105-106,112-114
    return out, (x, c1, xp)


def block_down_bwd(ctx, dout, S: ParamStore, pre: str, is_last: bool, want_dx: bool, want_dw: bool, wscale=None):
    """`wscale` [B] (optional) weights each sample's contribution to the WEIGHT gradients only (shared backward sweep)."""
    x, c1, xp = ctx
    p, g = S.p, S.g
    H, W = x.shape[1], x.shape[2]
    pooled_short = xp is not None
    # Small launches with the side stream on: the block's three weight-grad launches go to the side stream TOGETHER, behind ONE
    # event wait placed after conv2's data-grad (hipStreamWaitEvent costs the host 0.29 ms per call on this stack -- 37 of them were
    # 11 of the 24 ms it takes to queue a shard-size step, tools/host_profile.py); the side stream still runs beside the block's
    # remaining data-grads and the next block's.
    one_join = want_dw and ops.side_enabled() and dout.shape[0] <= ops.SIDE_MAX_BATCH
    if want_dw and pooled_short and not one_join:     # the shortcut saw avg_pool(x): its weight / bias gradients come from the pooled grid
        with ops.side_stream(xp, dout, wscale):
            ops.conv2d_bwd_weight(xp, dout, g[pre + ".short.w"], db=g[pre + ".short.b"], sample_scale=wscale)
    cin, cout = x.shape[-1], c1.shape[-1]
    fold = (pooled_short and not is_last and not one_join and not ops.side_enabled()
            and ops.pooled_grad_foldable(x.shape[0], H, W, cout, cout, want_dw))
    if fold:
        # (round 4) conv2's two backward launches read the POOLED gradient: d_c2 = 0.25 * upsample(dout) is folded into their Winograd
        # gradient transforms and never written (ops.conv2d_avgpool_bwd); conv2 is cout -> cout, its input c1 is also its ReLU mask
        d_c1 = ops.conv2d_avgpool_bwd(c1, dout, p[pre + ".conv2.w"], c1, dw=g[pre + ".conv2.w"] if want_dw else None,
                                      db=g[pre + ".conv2.b"] if want_dw else None, sample_scale=wscale, relu_in=True)
        d_c2 = None
    elif is_last:
        d_c2 = dout                  # gradient of conv2's output (and of the 1x1 output when it pools last)
    elif pooled_short:               # only conv2's launches read it: low-precision modes write their operand copies directly
        d_c2 = ops.avgpool2_bwd_operands(dout, wscale, want_dw)
    else:
        d_c2 = ops.avgpool2_bwd(dout)
    if want_dw and not one_join and not fold:          # (weight gradients are leaves of the sweep: side stream, see ops.side_stream)
        with ops.side_stream(c1, x, d_c2, wscale):
            ops.conv2d_bwd_weight(c1, d_c2, g[pre + ".conv2.w"], relu_in=True, db=g[pre + ".conv2.b"], sample_scale=wscale)
            if not pooled_short:
                ops.conv2d_bwd_weight(x, d_c2, g[pre + ".short.w"], db=g[pre + ".short.b"], sample_scale=wscale)
    # (bf16 twin of d_c1 from the epilogue, unless both of its consumers -- conv1's weight-grad and data-grad -- read fp8 copies)
    if not fold:
        d_c1 = ops.conv2d_bwd_data(d_c2, p[pre + ".conv2.w"], (H, W), mask=c1, want16=not ops._fp8_wgrad_ok(cin, cout, 3, 3, True),
                                   amax_scale=wscale)
    if one_join:
        with ops.side_stream(x, c1, xp, dout, d_c2, d_c1, wscale):
            if pooled_short:
                ops.conv2d_bwd_weight(xp, dout, g[pre + ".short.w"], db=g[pre + ".short.b"], sample_scale=wscale)
            ops.conv2d_bwd_weight(c1, d_c2, g[pre + ".conv2.w"], relu_in=True, db=g[pre + ".conv2.b"], sample_scale=wscale)
            if not pooled_short:
                ops.conv2d_bwd_weight(x, d_c2, g[pre + ".short.w"], db=g[pre + ".short.b"], sample_scale=wscale)
            ops.conv2d_bwd_weight(x, d_c1, g[pre + ".conv1.w"], relu_in=True, db=g[pre + ".conv1.b"], sample_scale=wscale)
    elif want_dw:
        with ops.side_stream(x, d_c1, wscale):
            ops.conv2d_bwd_weight(x, d_c1, g[pre + ".conv1.w"], relu_in=True, db=g[pre + ".conv1.b"], sample_scale=wscale)
    if not want_dx:
        return None
    if pooled_short:
        dx = ops.avgpool2_bwd(ops.conv2d_bwd_data(dout, p[pre + ".short.w"], (H // 2, W // 2)))
    else:
        dx = ops.conv2d_bwd_data(d_c2, p[pre + ".short.w"], (H, W))
    ops.conv2d_bwd_data(d_c1, p[pre + ".conv1.w"], (H, W), mask=x, out=dx, accum=True, amax_scale=wscale)    # (dx = the block below's dout)
    return dx


# --------------------------------------------------------------------------------------------
# NonLocalBlock (/root/reference/src/bigacgan/arch_ops.py:5-72) as a pure function of its kernels
# --------------------------------------------------------------------------------------------
def nonlocal_weights(C: int, gen: torch.Generator, device) -> Dict[str, torch.Tensor]:
    """The four orthogonal 1x1 kernels NonLocalBlock.call builds (arch_ops.py:38-65)."""
    return {"theta": orthogonal((1, 1, C, C // 8), gen).to(device), "phi": orthogonal((1, 1, C, C // 8), gen).to(device),
            "g": orthogonal((1, 1, C, C // 2), gen).to(device), "o": orthogonal((1, 1, C // 2, C), gen).to(device)}


_NL_RING = 4
_NL_RINGS: Dict = {}


def nonlocal_weights_batch(requests, device) -> List[Dict[str, torch.Tensor]]:
    """Draw the kernels of several NonLocalBlock calls at once: QR on the host, ONE pinned staging buffer and
    ONE asynchronous H2D copy, so the per-call re-draw of the reference (fact 3) costs no mid-step host sync.
    `requests` = [(C, generator), ...]."""
    host, shapes = [], []
    # The QRs are 64x8 .. 64x32: with a many-thread intra-op pool each costs ~1 ms of fork/join (6.6 ms per kernel set
    # at 8 threads vs 0.3 ms at one), which would make the HOST the bottleneck of a small-batch step.
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for C, gen in requests:
            for k, shp in (("theta", (1, 1, C, C // 8)), ("phi", (1, 1, C, C // 8)), ("g", (1, 1, C, C // 2)), ("o", (1, 1, C // 2, C))):
                host.append(orthogonal(shp, gen).reshape(-1))
                shapes.append((k, shp))
    finally:
        torch.set_num_threads(nthreads)
    flat = torch.cat(host)
    if device.type == "cuda":
        # A ring of persistent (pinned staging buffer, device buffer, event) slots per payload size.  Round 4: the former
        # `torch.empty(pin_memory=True)` + `.to(device, non_blocking=True)` pair cost the HOST 20-30 ms per step at the shard batch
        # (tools/host_profile.py, profiles/r04_host_profile_bs16.txt: the copy waited behind the whole queued step) -- the eager
        # shard step ran 37.2 ms where its graph replay, which uploads through persistent buffers, ran 33.8 ms.  A slot is reused
        # _NL_RING steps later; by then its copy has long completed (the event wait below is a formality) and every kernel that read
        # the device buffer was queued before the copy that overwrites it (one launch stream; the network stream joins it each step).
        key = (flat.numel(), device.index)
        ring = _NL_RINGS.get(key)
        if ring is None:
            ring = _NL_RINGS[key] = {"next": 0, "slots": [
                {"pinned": torch.empty(flat.numel(), dtype=torch.float32, pin_memory=True),
                 "dev": torch.empty(flat.numel(), dtype=torch.float32, device=device), "event": None} for _ in range(_NL_RING)]}
        slot = ring["slots"][ring["next"] % _NL_RING]
        ring["next"] += 1
        if slot["event"] is not None:
            slot["event"].synchronize()
        slot["pinned"].copy_(flat)
        slot["dev"].copy_(slot["pinned"], non_blocking=True)
        if slot["event"] is None:
            slot["event"] = torch.cuda.Event()
        slot["event"].record(torch.cuda.current_stream(device))
        dev = slot["dev"]
    else:
        dev = flat.to(device)
    out, off, idx = [], 0, 0
    for _ in requests:
        d = {}
        for _k in range(4):
            k, shp = shapes[idx]
            n = int(math.prod(shp))
            d[k] = dev[off:off + n].view(shp)
            off += n
            idx += 1
        out.append(d)
    return out


def nonlocal_fwd(x, nlw: Dict[str, torch.Tensor], sigma: torch.Tensor, out=None):
    B, H, W, C = x.shape
    w_t, w_p, w_g, w_o = (nlw[k].view(1, 1, *nlw[k].shape[-2:]) for k in ("theta", "phi", "g", "o"))
    theta = ops.conv2d_fwd(x, w_t)                                         # :38-41
    phi, i_phi = ops.maxpool_fwd(ops.conv2d_fwd(x, w_p), 2, 2)             # :44-48
    gg, i_g = ops.maxpool_fwd(ops.conv2d_fwd(x, w_g), 2, 2)                # :55-59
    Nq, Nk = H * W, (H // 2) * (W // 2)
    dk, dv = w_t.shape[-1], w_g.shape[-1]     # C/8 and C/2 (zero-padded to the kernel's 8 / 32 for make_my_discriminator's C = 32)
    o, lse = ops.attention_fwd(theta.view(B, Nq, dk), phi.view(B, Nk, dk), gg.view(B, Nk, dv))   # :51-52,61
    oc = ops.conv2d_fwd(o.view(B, H, W, dv), w_o)                          # :62-65
    out = ops.scale_add(oc, x, sigma.view(1), out=out)                     # :67
    return out, (x, theta, phi, i_phi, gg, i_g, o, lse, oc, (w_t, w_p, w_g, w_o))


def nonlocal_bwd(ctx, dout, sigma: torch.Tensor, dsigma: torch.Tensor, dnlw: Optional[Dict[str, torch.Tensor]] = None, out=None,
                 wscale=None):
    x, theta, phi, i_phi, gg, i_g, o, lse, oc, (w_t, w_p, w_g, w_o) = ctx
    B, H, W, C = x.shape
    Nq, Nk = H * W, (H // 2) * (W // 2)
    dk, dv = w_t.shape[-1], w_g.shape[-1]
    ops.dot_accum(dout if wscale is None else ops.rowscale(dout, wscale), oc, dsigma.view(1))
    d_oc = ops.scale(dout, sigma.view(1))
    d_o = ops.conv2d_bwd_data(d_oc, w_o, (H, W))
    dth, dph, dg = ops.attention_bwd(theta.view(B, Nq, dk), phi.view(B, Nk, dk), gg.view(B, Nk, dv),
                                     o.view(B, Nq, dv), lse, d_o.view(B, Nq, dv))
    dth = dth.view(B, H, W, dk)
    dph_f = ops.maxpool_bwd(dph.view(B, H // 2, W // 2, dk), i_phi, 2, 2)
    dg_f = ops.maxpool_bwd(dg.view(B, H // 2, W // 2, dv), i_g, 2, 2)
    if dnlw is not None:     # 'persistent' mode: the 1x1 kernels are trainable
        ops.conv2d_bwd_weight(o.view(B, H, W, dv), d_oc, dnlw["o"].view(1, 1, dv, C), sample_scale=wscale)
        ops.conv2d_bwd_weight(x, dth, dnlw["theta"].view(1, 1, C, dk), sample_scale=wscale)
        ops.conv2d_bwd_weight(x, dph_f, dnlw["phi"].view(1, 1, C, dk), sample_scale=wscale)
        ops.conv2d_bwd_weight(x, dg_f, dnlw["g"].view(1, 1, C, dv), sample_scale=wscale)
    dx = ops.conv2d_bwd_data(dth, w_t, (H, W))
    ops.conv2d_bwd_data(dph_f, w_p, (H, W), out=dx, accum=True)
    ops.conv2d_bwd_data(dg_f, w_g, (H, W), out=dx, accum=True)
    return ops.add(dx, dout, out=dx if out is None else out)


# --------------------------------------------------------------------------------------------
# batch norm with batch statistics (+ per-sample affine = ConditionalBatchNorm, resnet_ops.py:5-33)
# --------------------------------------------------------------------------------------------
def bn_train_fwd(x, gamma, beta, per_sample: bool, relu: bool, reducer: Reducer = LOCAL):
    B, H, W, C = x.shape
    sums = ops.bn_stats_sums(x)
    sync = reducer.world_size > 1 and reducer.sync_bn
    if sync:
        sums = reducer.all_reduce_sum(sums)                      # SyncBN: (sum x, sum x^2) over all ranks
    count = B * H * W * (reducer.world_size if sync else 1)
    mean, var = ops.bn_stats_finalize(sums, count, x)
    y = ops.bn_apply(x, mean, var, gamma, beta, per_sample, relu)
    return y, (x, y, mean, var, gamma, count)


def bn_train_bwd(ctx, dy, per_sample: bool, relu: bool, reducer: Reducer = LOCAL, dgamma_c=None, dbeta_c=None):
    """-> dx, dgamma [B,C], dbeta [B,C] (per-sample sums), chan (fp64 [4C]); dgamma_c/dbeta_c [C] += per-channel grads."""
    x, y, mean, var, gamma, count = ctx
    dgamma, dbeta, chan = ops.bn_bwd_reduce(dy, y, x, mean, var, gamma, per_sample, relu, dgamma_c=dgamma_c, dbeta_c=dbeta_c)
    if reducer.world_size > 1 and reducer.sync_bn:
        C = x.shape[-1]
        red = reducer.all_reduce_sum(chan[:2 * C].clone())
        chan = torch.cat([red, chan[2 * C:]])
    dx = ops.bn_bwd_apply(dy, y, x, mean, var, gamma, per_sample, chan, count, relu, True)
    return dx, dgamma, dbeta, chan


# --------------------------------------------------------------------------------------------
# ResNetBlockUp (/root/reference/src/bigacgan/resnet_ops.py:36-81)
# --------------------------------------------------------------------------------------------
def block_up_specs(pre: str, cin: int, cout: int):
    return [(pre + ".cbn1.gamma.w", (32, cin), orthogonal, True), (pre + ".cbn1.beta.w", (32, cin), orthogonal, True),
            (pre + ".convT.w", (3, 3, cout, cin), orthogonal, True), (pre + ".convT.b", (cout,), zeros, True),
            (pre + ".cbn2.gamma.w", (32, cout), orthogonal, True), (pre + ".cbn2.beta.w", (32, cout), orthogonal, True),
            (pre + ".conv.w", (3, 3, cout, cout), orthogonal, True), (pre + ".conv.b", (cout,), zeros, True),
            (pre + ".short.w", (1, 1, cout, cin), orthogonal, True), (pre + ".short.b", (cout,), zeros, True),
            (pre + ".cbn1.mm", (cin,), zeros, False), (pre + ".cbn1.mv", (cin,), ones, False),
            (pre + ".cbn2.mm", (cout,), zeros, False), (pre + ".cbn2.mv", (cout,), ones, False)]


def _cbn_fwd(x, z, zi: int, S: ParamStore, pre: str, reducer: Reducer, update_moving: bool):
    """z [B,128]; this block's conditioning chunk is z[:, 32*zi : 32*zi+32] (net_architecture.py:260-262)."""
    B, C = x.shape[0], x.shape[-1]
    gamma = ops.gemm(z, S.p[pre + ".gamma.w"], B, C, 32, 128, C, A_off=32 * zi)       # resnet_ops.py:18-21
    beta = ops.gemm(z, S.p[pre + ".beta.w"], B, C, 32, 128, C, A_off=32 * zi)         # :24-27
    y, ctx = bn_train_fwd(x, gamma, beta, True, True, reducer)
    if update_moving:
        ops.bn_update_moving(S.p[pre + ".mm"], S.p[pre + ".mv"], ctx[2], ctx[3], ctx[5])
    return y, ctx


def _cbn_bwd(ctx, dy, z, dz, zi: int, S: ParamStore, pre: str, reducer: Reducer):
    dx, dgamma, dbeta, _ = bn_train_bwd(ctx, dy, True, True, reducer)
    B, C = dgamma.shape
    # Dense(gamma), Dense(beta) backward: weight grads [32,C] += z_i^T dgamma ; dz_i += dgamma Wg^T + dbeta Wb^T
    ops.gemm(z, dgamma, 32, C, B, 128, C, transA=True, out=S.g[pre + ".gamma.w"], beta=1.0, A_off=32 * zi)
    ops.gemm(z, dbeta, 32, C, B, 128, C, transA=True, out=S.g[pre + ".beta.w"], beta=1.0, A_off=32 * zi)
    ops.gemm(dgamma, S.p[pre + ".gamma.w"], B, 32, C, C, C, transB=True, out=dz, ldc=128, beta=1.0, out_off=32 * zi)
    ops.gemm(dbeta, S.p[pre + ".beta.w"], B, 32, C, C, C, transB=True, out=dz, ldc=128, beta=1.0, out_off=32 * zi)
    return dx


def block_up_fwd(x, z, zi: int, S: ParamStore, pre: str, is_last: bool, reducer: Reducer = LOCAL, update_moving=True):
    p = S.p
    stride = (2, 1) if is_last else (2, 2)                                                    # :54
    y1, c1 = _cbn_fwd(x, z, zi, S, pre + ".cbn1", reducer, update_moving)                     # :50-51
    t = ops.conv2d_transpose_fwd(y1, p[pre + ".convT.w"], p[pre + ".convT.b"], stride=stride)   # :57-59
    y2, c2 = _cbn_fwd(t, z, zi, S, pre + ".cbn2", reducer, update_moving)                     # :62-63
    out = ops.conv2d_fwd(y2, p[pre + ".conv.w"], p[pre + ".conv.b"])                          # :65-66
    ops.conv2d_transpose_fwd(x, p[pre + ".short.w"], p[pre + ".short.b"], stride=stride, out=out, accum=True)   # :69-73
    return out, (x, c1, c2, stride)


def block_up_bwd(ctx, dout, z, dz, zi: int, S: ParamStore, pre: str, reducer: Reducer = LOCAL):
    x, c1, c2, stride = ctx
    p, g = S.p, S.g
    y1, y2 = c1[1], c2[1]
    ops.bias_grad(dout, g[pre + ".short.b"])
    ops.conv2d_bwd_weight(y2, dout, g[pre + ".conv.w"], db=g[pre + ".conv.b"])
    dy2 = ops.conv2d_bwd_data(dout, p[pre + ".conv.w"], (y2.shape[1], y2.shape[2]))
    dt = _cbn_bwd(c2, dy2, z, dz, zi, S, pre + ".cbn2", reducer)
    ops.conv2d_transpose_bwd_weight(y1, dt, g[pre + ".convT.w"], stride=stride)
    ops.bias_grad(dt, g[pre + ".convT.b"])
    dy1 = ops.conv2d_transpose_bwd_data(dt, p[pre + ".convT.w"], stride=stride)
    dx = _cbn_bwd(c1, dy1, z, dz, zi, S, pre + ".cbn1", reducer)
    ops.conv2d_transpose_bwd_weight(x, dout, g[pre + ".short.w"], stride=stride)
    ops.conv2d_transpose_bwd_data(dout, p[pre + ".short.w"], stride=stride, out=dx, accum=True)
    return dx


# --------------------------------------------------------------------------------------------
# Bidirectional(LSTM(H, return_sequences=True, dropout=p)) (net_architecture.py:146-150)
# --------------------------------------------------------------------------------------------
def lstm_forget_bias(shape, gen=None):
    """Keras LSTM bias: zeros with unit_forget_bias=True -> ones on the forget-gate quarter (order i,f,c,o)."""
    b = torch.zeros(tuple(shape))
    h = shape[0] // 4
    b[h:2 * h] = 1.0
    return b


def lstm_recurrent_orthogonal(shape, gen):
    return orthogonal(shape, gen)


def bilstm_specs(pre: str, in_dim: int, H: int):
    s = []
    for d in ("fw", "bw"):
        s += [(pre + "." + d + ".W", (in_dim, 4 * H), glorot_uniform, True), (pre + "." + d + ".U", (H, 4 * H), lstm_recurrent_orthogonal, True),
              (pre + "." + d + ".b", (4 * H,), lstm_forget_bias, True)]
    return s


def _lstm_dir_fwd(xm, W, U, b, out, col_off, reverse: bool):
    """xm [B,T,I] (already dropout-masked) -> writes h_t into out[:, t, col_off:col_off+H].  Returns ctx."""
    B, T, I = xm.shape
    H = U.shape[0]
    # x W + b for every timestep at once on the MFMA conv kernel (a 1x1 convolution over [B,T,1,I])
    z = ops.conv2d_fwd(xm.view(B, T, 1, I), W.view(1, 1, I, 4 * H), b).view(B, T, 4 * H)
    cs = ops.empty(T, B, H, like=xm)                    # c_t, time-major
    hprev = torch.zeros(B, T, H, device=xm.device)      # h_{t-1} as seen by step t (zeros at the first step)
    order = range(T - 1, -1, -1) if reverse else range(T)
    prev_t = None
    for t in order:
        if prev_t is not None:                          # z_t += h_{t-1} U   (in place, strided rows)
            ops.gemm(hprev, U, B, 4 * H, H, T * H, 4 * H, out=z, ldc=T * 4 * H, beta=1.0, A_off=t * H, out_off=t * 4 * H)
        nxt = t - 1 if reverse else t + 1
        has_next = 0 <= nxt < T
        ops.lstm_cell_fwd(z, t * 4 * H, T * 4 * H, None if prev_t is None else cs[prev_t], cs[t], out, t * out.shape[2] + col_off,
                          T * out.shape[2], hprev if has_next else None, (nxt * H) if has_next else 0, T * H, B, H)
        prev_t = t
    return (xm, z, cs, hprev, reverse)


def _lstm_dir_bwd(ctx, dout, col_off, W, U, gW, gU, gb):
    """dout [B,T,2H]; accumulates weight grads; returns d(xm) [B,T,I]."""
    xm, gates, cs, hprev, reverse = ctx
    B, T, I = xm.shape
    H = U.shape[0]
    order = list(range(T - 1, -1, -1) if reverse else range(T))
    dh_next, dc_next = None, None
    for k in range(T - 1, -1, -1):                      # walk the recurrence backwards
        t = order[k]
        t_prev = order[k - 1] if k > 0 else None
        dc_prev = ops.empty(B, H, like=xm)
        ops.lstm_cell_bwd(gates, t * 4 * H, T * 4 * H, None if t_prev is None else cs[t_prev], cs[t], dout, t * dout.shape[2] + col_off,
                          T * dout.shape[2], dh_next, dc_next, dc_prev, B, H)
        if t_prev is not None:                          # dh_{t-1} = dz_t U^T
            dh_next = ops.gemm(gates, U, B, H, 4 * H, T * 4 * H, 4 * H, transB=True, A_off=t * 4 * H)
        dc_next = dc_prev
    dz = gates                                           # now d(pre-activations) for all timesteps [B,T,4H]
    ops.gemm(hprev.view(B * T, H), dz.view(B * T, 4 * H), H, 4 * H, B * T, H, 4 * H, transA=True, out=gU, beta=1.0)
    ops.conv2d_bwd_weight(xm.view(B, T, 1, I), dz.view(B, T, 1, 4 * H), gW.view(1, 1, I, 4 * H))
    ops.bias_grad(dz.view(B * T, 4 * H), gb)
    return ops.conv2d_bwd_data(dz.view(B, T, 1, 4 * H), W.view(1, 1, I, 4 * H), (T, 1)).view(B, T, I)


def bilstm_fwd(x, S: ParamStore, pre: str, masks=None):
    """x [B,T,I]; masks = (mask_fw, mask_bw) each [B,I] with values 0 or 1/(1-rate) (Keras input dropout: one mask per
    sample shared by all timesteps), or None when not training."""
    B, T, I = x.shape
    H = S.p[pre + ".fw.U"].shape[0]
    out = ops.empty(B, T, 2 * H, like=x)
    ctxs = []
    for di, d in enumerate(("fw", "bw")):
        xm = x if masks is None else ops.mul_mask(x, masks[di], rows_per_mask=T)
        ctxs.append(_lstm_dir_fwd(xm, S.p[pre + "." + d + ".W"], S.p[pre + "." + d + ".U"], S.p[pre + "." + d + ".b"], out, di * H,
                                  reverse=(d == "bw")))
    return out, (ctxs, masks)


def bilstm_bwd(ctx, dout, S: ParamStore, pre: str, want_dw=True):
    ctxs, masks = ctx
    H = S.p[pre + ".fw.U"].shape[0]
    T = dout.shape[1]
    dx = None
    for di, d in enumerate(("fw", "bw")):
        if want_dw:
            gW, gU, gb = S.g[pre + "." + d + ".W"], S.g[pre + "." + d + ".U"], S.g[pre + "." + d + ".b"]
        else:
            gW, gU, gb = (torch.zeros_like(S.p[pre + "." + d + k]) for k in (".W", ".U", ".b"))
        dxm = _lstm_dir_bwd(ctxs[di], dout, di * H, S.p[pre + "." + d + ".W"], S.p[pre + "." + d + ".U"], gW, gU, gb)
        if masks is not None:
            dxm = ops.mul_mask(dxm, masks[di], rows_per_mask=T)
        dx = dxm if dx is None else ops.add(dx, dxm, out=dx)
    return dx


# --------------------------------------------------------------------------------------------
# Conv2D(3x3, strides (2,2), padding 'same') of make_my_discriminator (net_architecture.py:425-443), on the transposed-conv
# kernels: a strided convolution IS the adjoint of Conv2DTranspose(padding='same') (even input,
# pad_before 0, pad_after 1), so   forward = convT data-grad,  data-grad = convT forward,  weight-grad = convT weight-grad
# with the operands swapped -- all with the SAME [kh,kw,Cin,Cout] kernel, no layout change.
# --------------------------------------------------------------------------------------------
def strided_conv_fwd(x, w, b):
    kh, kw, Cin, Cout = w.shape
    if Cin == 1:
        # one input channel: the thin stride-1 kernel on every position, then the odd positions (stride-2 SAME on an even
        # extent samples x[2i + k], stride-1 SAME x[y + k - 1]: out2[i, j] = out1[2i + 1, 2j + 1]); 4x the work of a layer
        # that has 0.03 % of the network's FLOPs
        full = ops.conv2d_fwd(x, w, b)
        return full[:, 1::2, 1::2, :].contiguous()
    y = ops.conv2d_transpose_bwd_data(x, w, stride=(2, 2))
    return ops.bias_add(y, b)


def strided_conv_bwd(x, w, dy, dw, db, want_dx: bool):
    kh, kw, Cin, Cout = w.shape
    ops.bias_grad(dy, db)
    if Cin == 1:
        B, H, W, _ = x.shape
        dfull = torch.zeros(B, H, W, Cout, device=x.device)
        dfull[:, 1::2, 1::2, :] = dy                          # scatter back to the odd positions (data movement only)
        ops.conv2d_bwd_weight(x, dfull, dw)
        return ops.conv2d_bwd_data(dfull, w, (H, W)) if want_dx else None
    ops.conv2d_transpose_bwd_weight(dy, x, dw, stride=(2, 2))
    return ops.conv2d_transpose_fwd(dy, w, stride=(2, 2)) if want_dx else None
