"""Model factories with the reference's names and signatures
(/root/reference/src/bigacgan/net_architecture.py): make_recognizer :9-79, make_my_recognizer :82-179,
make_generator :182-296, make_discriminator :299-355, make_style_promoter :358-414, make_gan :531-561,
get_in_out_channels_gen :565-573, get_in_out_channels_disc :576-586.

The returned objects duck-type the subset of tf.keras.Model the reference's callers touch
(`model(inputs, training=...)`, `.trainable`, `.trainable_variables`, `.save_weights`, `.summary`) and
add the explicit `forward` / `backward` pair the MI355X train_step drives (no tapes, no autograd):
every FLOP runs in libscrabble_hip.so.

Deliberate, documented behaviours carried over from the reference:
  * kernel_reg (spectral_norm) is accepted and, like Keras' lazily-evaluated regularizer losses that
    nobody reads, NOT applied to the weights in the forward pass (fact 2);
  * NonLocalBlock kernels are re-drawn (orthogonal) on every call and untrained in nl_mode='reference'
    (fact 3); nl_mode='persistent' keeps them as trainable weights (the evident intent);
  * the recognizer's BatchNorm runs in inference mode whenever the model is frozen (fact 4).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import nn, ops
from .nn import LOCAL, ParamStore, Reducer


def get_in_out_channels_gen(resolution=32):
    ch = 64
    if resolution == 32:
        mult = [8, 4, 2, 1]
    else:
        raise ValueError("Unsupported resolution: {}".format(resolution))
    return [ch * c for c in mult[:-1]], [ch * c for c in mult[1:]]


def get_in_out_channels_disc(colors=1, resolution=32):
    ch = 64
    if colors not in [1, 3]:
        raise ValueError("Unsupported color channels: {}".format(colors))
    if resolution == 32:
        mult = [1, 8, 16, 16]
    else:
        raise ValueError("Unsupported resolution: {}".format(resolution))
    out_channels = [ch * c for c in mult]
    return [colors] + out_channels[:-1], out_channels


_DEFAULTS = {"device": None, "seed": 0, "nl_mode": "reference", "reducer": LOCAL, "kernel_reg_mode": "reference"}


def configure(device=None, seed=None, nl_mode=None, reducer=None, conv_dtype=None, deterministic=None, sync_bn=None,
              kernel_reg_mode=None):
    """Process-wide construction defaults (device, init seed, NonLocalBlock mode, DP reducer) and the matrix-core
    operand type of the convolutions ('f32' = parity mode, 'bf16' = BASELINE config c3).

    deterministic=True: the forward / data-grad conv kernels never split a reduction across workgroups, so no activation
    and no image gradient is summed by float atomics: they are bitwise reproducible from launch to launch and independent
    of the per-rank batch size (a few percent slower: the tail tiles of a launch no longer balance over the CUs).  The
    weight-gradient, bias-gradient and attention-dK/dV kernels still add partial sums with float atomics, so weights agree
    between runs to fp32 rounding, not bitwise.  Default runs are NOT bitwise reproducible (README)."""
    if conv_dtype is not None:          # 'f32' | 'bf16' | 'fp8'
        ops.set_conv_dtype(conv_dtype)
    if deterministic is not None:
        ops.set_deterministic(bool(deterministic))
    if kernel_reg_mode is not None:
        # 'reference': kernel_reg is accepted and never applied, like the lazily evaluated Keras regularizer losses nobody
        # reads -- the parity mode.  'applied': every forward pass convolves with w / sigma (nn.AppliedSNStore).
        assert kernel_reg_mode in ("reference", "applied")
        _DEFAULTS["kernel_reg_mode"] = kernel_reg_mode
    if sync_bn is not None:          # data parallel: global (True, default) or per-rank (False) BatchNorm statistics
        (reducer if reducer is not None else _DEFAULTS["reducer"]).sync_bn = bool(sync_bn)
    if device is not None:
        _DEFAULTS["device"] = torch.device(device)
    if seed is not None:
        _DEFAULTS["seed"] = seed
    if nl_mode is not None:
        assert nl_mode in ("reference", "persistent")
        _DEFAULTS["nl_mode"] = nl_mode
    if reducer is not None:
        _DEFAULTS["reducer"] = reducer


def _device():
    if _DEFAULTS["device"] is None:
        if not torch.cuda.is_available():
            raise RuntimeError("scrabble_gan_amd needs an MI355X (no HIP device visible); there is no CPU path")
        _DEFAULTS["device"] = torch.device("cuda", torch.cuda.current_device())
    return _DEFAULTS["device"]


_model_counter = [0]


def _gen_for(name: str) -> torch.Generator:
    _model_counter[0] += 1
    return torch.Generator().manual_seed(_DEFAULTS["seed"] * 1000 + _model_counter[0])


class _Model:
    """The slice of tf.keras.Model that main.py / data_utils.py use."""

    def __init__(self, name: str, specs, gen: torch.Generator):
        ops.weights_changed()            # packed / transposed filter copies of earlier models are dropped
        self.name = name
        self.device = _device()
        self.store = ParamStore(specs, self.device, gen)
        self.trainable = True
        self.reducer: Reducer = _DEFAULTS["reducer"]
        self.kernel_reg_mode = _DEFAULTS["kernel_reg_mode"]
        self.sn_gen = torch.Generator().manual_seed(gen.initial_seed() + 13)

    def _pass_store(self):
        """The parameter view of ONE forward pass (and its backward sweeps): the store itself, or, with kernel_reg applied,
        spectrally normalised kernels with a fresh u (only when a kernel_reg callable was given to the factory)."""
        if self.kernel_reg_mode == "applied" and getattr(self, "kernel_reg", None) is not None:
            return nn.AppliedSNStore(self.store, nn.sn_names(self.store), self.sn_gen)
        return self.store

    @staticmethod
    def _fold(S):
        ops.side_join()                  # the weight gradients of the sweep (queued on the side stream) are complete from here on
        if isinstance(S, nn.AppliedSNStore):
            S.fold()

    @property
    def trainable_variables(self) -> List[torch.Tensor]:
        if not self.trainable:
            return []
        return [self.store.p[n] for n in self.store.trainable_names()]

    @property
    def variables(self) -> Dict[str, torch.Tensor]:
        return dict(self.store.p)

    def count_params(self) -> int:
        return self.store.num_params()

    def summary(self):
        print('Model: "%s"' % self.name)
        for n in self.store.names:
            print("  %-28s %-22s %s" % (n, tuple(self.store.shapes[n]), "" if self.store.trainable[n] else "(non-trainable)"))
        print("Trainable params: {:,}".format(self.count_params()))

    def save_weights(self, prefix: str):
        """generator.save_weights(prefix) (data_utils.py:346-348).  Written as <prefix>.safetensors."""
        from safetensors.torch import save_file
        d = os.path.dirname(prefix)
        if d:
            os.makedirs(d, exist_ok=True)
        save_file({k: v.contiguous() for k, v in self.store.export().items()}, prefix + ".safetensors")

    def load_weights(self, prefix: str):
        from safetensors.torch import load_file
        self.store.load(load_file(prefix + ".safetensors"))


# --------------------------------------------------------------------------------------------
# the 4-block "down" trunk shared by D, the style promoter and G's style encoder
# --------------------------------------------------------------------------------------------
class _DownTrunk:
    def __init__(self, block_fmt: str, attn_blocks: str, colors: int, resolution: int, nl_mode: str):
        self.cin, self.cout = get_in_out_channels_disc(colors, resolution)
        self.names = [block_fmt.format(i + 1) for i in range(len(self.cin))]
        self.attn = [n for n in self.names if n in attn_blocks]          # substring test (net_architecture.py:278,336)
        self.nl_mode = nl_mode

    def specs(self):
        s = []
        for n, ci, co in zip(self.names, self.cin, self.cout):
            s += nn.block_down_specs(n, ci, co)
            if n in self.attn:
                s.append(("NL_" + n + ".sigma", (), nn.zeros, True))
                if self.nl_mode == "persistent":
                    s += [("NL_" + n + ".theta", (co, co // 8), nn.orthogonal, True), ("NL_" + n + ".phi", (co, co // 8), nn.orthogonal, True),
                          ("NL_" + n + ".g", (co, co // 2), nn.orthogonal, True), ("NL_" + n + ".o", (co // 2, co), nn.orthogonal, True)]
        return s

    def _nlw(self, S: ParamStore, n: str, C: int, nl, nl_gen):
        if self.nl_mode == "persistent":
            return {k: S.p["NL_" + n + "." + k] for k in ("theta", "phi", "g", "o")}
        if nl is not None:
            return nl
        return nn.nonlocal_weights(C, nl_gen, S.device)       # fact 3: fresh orthogonal kernels per call

    def fwd(self, x, S: ParamStore, nl=None, nl_gen=None, segments=None):
        """`segments` = [(lo, hi, nl_kernels), ...]: batch ranges that are separate NonLocalBlock calls of the
        reference (each with its own freshly drawn 1x1 kernels) riding in ONE pass of the conv trunk."""
        ctxs = []
        net = x
        for i, n in enumerate(self.names):
            net, c = nn.block_down_fwd(net, S, n, is_last=(i == len(self.names) - 1))
            nlc = None
            if n in self.attn:
                sigma = S.p["NL_" + n + ".sigma"]
                if segments is None:
                    net, nlc = nn.nonlocal_fwd(net, self._nlw(S, n, self.cout[i], nl, nl_gen), sigma)
                else:
                    out, segc = torch.empty_like(net), []
                    for lo, hi, nl_seg in segments:
                        _, cseg = nn.nonlocal_fwd(net[lo:hi], self._nlw(S, n, self.cout[i], nl_seg, nl_gen), sigma, out=out[lo:hi])
                        segc.append((lo, hi, cseg))
                    net, nlc = out, ("seg", segc)
            ctxs.append((c, nlc))
        h = ops.gap_fwd(net, relu=True)                         # tf.nn.relu + GlobalAveragePooling2D
        return h, (ctxs, net)

    @staticmethod
    def slice_ctx(ctx, lo, hi):
        """The saved context of batch rows [lo, hi) of a multi-segment pass (segment-aligned)."""
        ctxs, net = ctx
        out = []
        for (x, c1, xp), nlc in ctxs:
            if nlc is not None:
                assert nlc[0] == "seg"
                nlc = ("seg", [(a - lo, b - lo, c) for a, b, c in nlc[1] if a >= lo and b <= hi])
            out.append(((x[lo:hi], c1[lo:hi], None if xp is None else xp[lo:hi]), nlc))
        return out, net[lo:hi]

    def bwd(self, ctx, dh, S: ParamStore, want_dx: bool, want_dw: bool, wscale=None, on_block_done=None):
        """`on_block_done(lo, hi)` (optional) is called after each block's backward with the range of the flat gradient buffer
        that is final from then on (that block's kernels and biases and its NonLocalBlock's parameters)."""
        ctxs, net = ctx
        d = ops.gap_bwd(dh, net, relu=True)
        for i in reversed(range(len(self.names))):
            n = self.names[i]
            c, nlc = ctxs[i]
            if nlc is not None:
                dsig = S.g["NL_" + n + ".sigma"] if want_dw else torch.zeros(1, device=S.device)
                dnlw = None
                if self.nl_mode == "persistent" and want_dw:
                    dnlw = {k: S.g["NL_" + n + "." + k] for k in ("theta", "phi", "g", "o")}
                sigma = S.p["NL_" + n + ".sigma"]
                if isinstance(nlc, tuple) and len(nlc) == 2 and nlc[0] == "seg":
                    dn = torch.empty_like(d)
                    for lo, hi, cseg in nlc[1]:
                        nn.nonlocal_bwd(cseg, d[lo:hi], sigma, dsig, dnlw, out=dn[lo:hi],
                                        wscale=None if wscale is None else wscale[lo:hi])
                    d = dn
                else:
                    d = nn.nonlocal_bwd(nlc, d, sigma, dsig, dnlw, wscale=wscale)
            d = nn.block_down_bwd(c, d, S, n, i == len(self.names) - 1, want_dx or i > 0, want_dw, wscale=wscale)
            if on_block_done is not None:
                ops.side_join()          # this block's weight gradients (side stream) before their range is handed out
                lo = S._off[n + ".conv1.w"]
                hi = S._off[self.names[i + 1] + ".conv1.w"] if i + 1 < len(self.names) else self.end_offset(S)
                on_block_done(lo, hi)
        return d

    def end_offset(self, S: ParamStore) -> int:
        """Offset in the flat trainable buffer right behind this trunk's last parameter (trunk specs come first in every model)."""
        last = self.specs()[-1]
        import math as _m
        n = int(_m.prod(last[1])) if len(last[1]) else 1
        return S._off[last[0]] + (n + 3) // 4 * 4


class DiscriminatorModel(_Model):
    """make_discriminator / make_style_promoter body: trunk -> Dense(1, no bias); make_style_extractor: Dense(128)."""

    def __init__(self, name, input_dim, kernel_reg, blocks_with_attention, units=1):
        h, w, c = input_dim
        self.trunk = _DownTrunk("B{}", blocks_with_attention, c, h, _DEFAULTS["nl_mode"])
        gen = _gen_for(name)
        super().__init__(name, self.trunk.specs() + [("dense.w", (self.trunk.cout[-1], units), nn.orthogonal, True)], gen)
        self.units = units
        self.kernel_reg = kernel_reg
        self.nl_gen = torch.Generator().manual_seed(gen.initial_seed() + 7)

    def forward(self, x, nl=None):
        x = _as_nhwc1(x, self.device)
        S = self._pass_store()
        h, tctx = self.trunk.fwd(x, S, nl, self.nl_gen)
        logits = ops.dense_fwd(h, S.p["dense.w"])                              # [B,1]
        return logits, (tctx, h, S)

    def forward_multi(self, xs, nls):
        """Several reference calls of this model (same input width) as ONE pass over the concatenated batch: the
        nets have no BatchNorm, so samples are independent and the result equals separate calls; only the
        NonLocalBlock kernels differ per call (`nls[i]`).  -> ([logits_i], ctx, [(lo_i, hi_i)])."""
        xs = [_as_nhwc1(x, self.device) for x in xs]
        bounds, lo = [], 0
        for x in xs:
            bounds.append((lo, lo + x.shape[0]))
            lo += x.shape[0]
        segs = [(a, b, nl) for (a, b), nl in zip(bounds, nls)]
        S = self._pass_store()
        h, tctx = self.trunk.fwd(torch.cat(xs, 0), S, None, self.nl_gen, segments=segs)
        logits = ops.dense_fwd(h, S.p["dense.w"])
        return [logits[a:b] for a, b in bounds], (tctx, h, S), bounds

    def slice_ctx(self, ctx, lo, hi):
        tctx, h, S = ctx
        return self.trunk.slice_ctx(tctx, lo, hi), h[lo:hi], S

    def backward(self, ctx, dlogits, want_dx: bool, want_dw: bool, wscale=None):
        """`wscale` [B]: the data-gradient chain runs with upstream `dlogits`, while sample b contributes to the weight
        gradients with `wscale[b] * dlogits[b]` -- one sweep serving two targets (backprop is linear per sample)."""
        tctx, h, S = ctx
        dlogits = dlogits.reshape(-1, self.units).contiguous()
        if want_dw:
            dl_w = dlogits if wscale is None else ops.rowscale(dlogits, wscale)
            ops.dense_bwd_weight(h, dl_w, S.g["dense.w"])
        dh = ops.dense_bwd_input(dlogits, S.p["dense.w"])
        dx = self.trunk.bwd(tctx, dh, S, want_dx, want_dw, wscale=wscale)
        if want_dw:
            self._fold(S)
        return dx

    def __call__(self, inputs, training=False):
        x = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        return self.forward(x)[0]


def _as_nhwc1(x, device):
    """Accept numpy / torch, [B,H,W] or [B,H,W,1]; cast to fp32 on the device."""
    if not torch.is_tensor(x):
        import numpy as np
        x = torch.from_numpy(np.ascontiguousarray(x))
    if x.dim() == 3:
        x = x.unsqueeze(-1)
    return x.to(device=device, dtype=torch.float32).contiguous()


def _as_labels(y, device, n_classes=None):
    """int32 labels on the device.  Host arrays (the train loop's case) are range-checked here when `n_classes` is given:
    TF's ctc_batch_cost / embedding_lookup reject indices outside [0, n_classes), and a bad char_vector or label file must
    not train on silently clamped targets.  (Device-resident labels: the CTC kernel poisons the sample's cost with +inf.)"""
    if not torch.is_tensor(y):
        import numpy as np
        y = np.ascontiguousarray(y)
        if n_classes is not None and y.size and (y.min() < 0 or y.max() >= n_classes):
            raise ValueError("label outside [0, %d): min %d, max %d" % (n_classes, y.min(), y.max()))
        y = torch.from_numpy(y)
    elif n_classes is not None and not y.is_cuda and y.numel() and (int(y.min()) < 0 or int(y.max()) >= n_classes):
        raise ValueError("label outside [0, %d)" % n_classes)
    return y.to(device=device, dtype=torch.int32).contiguous()


# --------------------------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------------------------
class GeneratorModel(_Model):
    def __init__(self, latent_dim, input_dim, embed_y, kernel_reg, blocks_with_attention, vocab_size):
        h, w, c = input_dim
        self.in_ch, self.out_ch = get_in_out_channels_gen(h)
        if tuple(embed_y) != (32, 8192):
            raise ValueError("embed_y must be (32, 8192): the seed reshape hard-codes 512x4x4 (net_architecture.py:269-270)")
        self.trunk = _DownTrunk("B_style{}", "B_style1", c, h, _DEFAULTS["nl_mode"])      # :241-246
        self.up_names = ["B{}".format(i + 1) for i in range(len(self.in_ch))]
        self.up_attn = [n for n in self.up_names if n in blocks_with_attention]           # :278
        specs = self.trunk.specs() + [("zdense.w", (self.trunk.cout[-1], 128), nn.orthogonal, True),
                                      ("filter_bank", (vocab_size, embed_y[0], embed_y[1]), nn.glorot_uniform, True)]
        for n, ci, co in zip(self.up_names, self.in_ch, self.out_ch):
            specs += nn.block_up_specs(n, ci, co)
            if n in self.up_attn:
                specs.append(("NL_" + n + ".sigma", (), nn.zeros, True))
                if _DEFAULTS["nl_mode"] == "persistent":
                    specs += [("NL_" + n + ".theta", (co, co // 8), nn.orthogonal, True), ("NL_" + n + ".phi", (co, co // 8), nn.orthogonal, True),
                              ("NL_" + n + ".g", (co, co // 2), nn.orthogonal, True), ("NL_" + n + ".o", (co // 2, co), nn.orthogonal, True)]
        cl = self.out_ch[-1]
        specs += [("bn.gamma", (cl,), nn.ones, True), ("bn.beta", (cl,), nn.zeros, True),
                  ("bn.mm", (cl,), nn.zeros, False), ("bn.mv", (cl,), nn.ones, False),
                  ("final.w", (3, 3, cl, c), nn.orthogonal, True), ("final.b", (c,), nn.zeros, True)]
        gen = _gen_for("generator")
        super().__init__("generator", specs, gen)
        self.nl_mode = _DEFAULTS["nl_mode"]
        self.nl_gen = torch.Generator().manual_seed(gen.initial_seed() + 7)
        self.kernel_reg = kernel_reg
        self.latent_dim = latent_dim          # unused, as in the reference (Appendix C-9)

    def _up_nlw(self, n, C, nl):
        if self.nl_mode == "persistent":
            return {k: self.store.p["NL_" + n + "." + k] for k in ("theta", "phi", "g", "o")}
        return nl if nl is not None else nn.nonlocal_weights(C, self.nl_gen, self.device)

    def forward(self, style, y, nl_style=None, nl_up=None, training=True):
        S = self._pass_store()
        p = S.p
        style = _as_nhwc1(style, self.device)
        y = _as_labels(y, self.device)
        h, tctx = self.trunk.fwd(style, S, nl_style, self.nl_gen)                 # :241-250
        z = ops.dense_fwd(h, p["zdense.w"])                                        # :251-257  [B,128]
        net = ops.filterbank_fwd(z, y, p["filter_bank"])                           # :229-231,259-271
        up_ctx = []
        for i, n in enumerate(self.up_names):
            if training:
                net, c = nn.block_up_fwd(net, z, i + 1, S, n, i == len(self.up_names) - 1, self.reducer)
            else:
                net, c = _block_up_infer(net, z, i + 1, S, n, i == len(self.up_names) - 1), None
            nlc = None
            if n in self.up_attn:
                net, nlc = nn.nonlocal_fwd(net, self._up_nlw(n, self.out_ch[i], nl_up), p["NL_" + n + ".sigma"])
            up_ctx.append((c, nlc))
        if training:
            yb, bctx = nn.bn_train_fwd(net, p["bn.gamma"], p["bn.beta"], False, True, self.reducer)   # :281-282
            ops.bn_update_moving(p["bn.mm"], p["bn.mv"], bctx[2], bctx[3], bctx[5])
        else:
            yb, bctx = ops.bn_apply(net, p["bn.mm"], p["bn.mv"], p["bn.gamma"], p["bn.beta"], False, True), None
        img = ops.conv2d_fwd(yb, p["final.w"], p["final.b"], tanh_out=True)        # :283-289
        return img, (tctx, h, z, y, up_ctx, bctx, yb, img, S)

    def backward(self, ctx, dimg, on_tail_ready=None, on_slice_ready=None):
        """`on_tail_ready(offset)` (optional) is called once every gradient in `store.grad[offset:]` is final;
        `on_slice_ready(lo, hi)` (optional) after each style-encoder block's backward with the range of the flat gradient
        buffer that is final from then on -- data parallelism reduces G's 214 MB in slices while the rest of its backward
        still runs, so only the last (smallest) block's bytes are exposed."""
        tctx, h, z, y, up_ctx, bctx, yb, img, S = ctx
        p, g = S.p, S.g
        applied = isinstance(S, nn.AppliedSNStore)
        if applied:
            on_tail_ready = on_slice_ready = None        # the shadow gradients are folded at the end: no early slice of the flat buffer is final
        d_pre = ops.tanh_bwd(img, dimg)
        ops.conv2d_bwd_weight(yb, d_pre, g["final.w"])
        ops.bias_grad(d_pre, g["final.b"])
        dyb = ops.conv2d_bwd_data(d_pre, p["final.w"], (yb.shape[1], yb.shape[2]))
        d, _, _, _ = nn.bn_train_bwd(bctx, dyb, False, True, self.reducer, dgamma_c=g["bn.gamma"], dbeta_c=g["bn.beta"])
        dz = torch.zeros_like(z)
        for i in reversed(range(len(self.up_names))):
            n = self.up_names[i]
            c, nlc = up_ctx[i]
            if nlc is not None:
                dnlw = {k: g["NL_" + n + "." + k] for k in ("theta", "phi", "g", "o")} if self.nl_mode == "persistent" else None
                d = nn.nonlocal_bwd(nlc, d, p["NL_" + n + ".sigma"], g["NL_" + n + ".sigma"], dnlw)
            d = nn.block_up_bwd(c, d, z, dz, i + 1, S, n, self.reducer)
        ops.filterbank_bwd(z, y, p["filter_bank"], d, g["filter_bank"], dz)
        ops.dense_bwd_weight(h, dz, g["zdense.w"])
        if on_tail_ready is not None:
            # every gradient from zdense.w to the end of the flat buffer (filter bank, up blocks, final BN / conv) is
            # complete: data parallelism starts reducing that slice while the style encoder's backward still runs
            on_tail_ready(S._off["zdense.w"])
        dh = ops.dense_bwd_input(dz, p["zdense.w"])
        self.trunk.bwd(tctx, dh, S, want_dx=False, want_dw=True, on_block_done=on_slice_ready)
        self._fold(S)

    def __call__(self, inputs, training=False):
        style, y = inputs[0], inputs[1]
        return self.forward(style, y, training=training)[0]


def _block_up_infer(x, z, zi, S: ParamStore, pre: str, is_last: bool):
    """ResNetBlockUp with BatchNorm in inference mode (moving statistics): generator(..., training=False)
    of data_utils.py:507."""
    p = S.p
    stride = (2, 1) if is_last else (2, 2)
    B = x.shape[0]

    def cbn(t, name):
        C = t.shape[-1]
        gamma = ops.gemm(z, p[name + ".gamma.w"], B, C, 32, 128, C, A_off=32 * zi)
        beta = ops.gemm(z, p[name + ".beta.w"], B, C, 32, 128, C, A_off=32 * zi)
        return ops.bn_apply(t, p[name + ".mm"], p[name + ".mv"], gamma, beta, True, True)
    t = ops.conv2d_transpose_fwd(cbn(x, pre + ".cbn1"), p[pre + ".convT.w"], p[pre + ".convT.b"], stride=stride)
    out = ops.conv2d_fwd(cbn(t, pre + ".cbn2"), p[pre + ".conv.w"], p[pre + ".conv.b"])
    ops.conv2d_transpose_fwd(x, p[pre + ".short.w"], p[pre + ".short.b"], stride=stride, out=out, accum=True)
    return out


# --------------------------------------------------------------------------------------------
# recognizer (fully convolutional CRNN + CTC): make_recognizer, net_architecture.py:9-79
# --------------------------------------------------------------------------------------------
_REC = [(3, 64, (2, 2)), (3, 128, (2, 2)), (3, 256, None), (3, 256, (2, 1)), (3, 512, None), (3, 512, (2, 1)), (2, 512, None)]


class RecognizerModel(_Model):
    def __init__(self, input_dim, sequence_length, output_classes):
        h, w, c = input_dim
        specs, cin = [], c
        for i, (k, co, _) in enumerate(_REC):
            specs += [("conv%d.w" % (i + 1), (k, k, cin, co), nn.glorot_uniform, True), ("conv%d.b" % (i + 1), (co,), nn.zeros, True)]
            cin = co
        for pre in ("bn5", "bn6"):
            specs += [(pre + ".gamma", (512,), nn.ones, True), (pre + ".beta", (512,), nn.zeros, True),
                      (pre + ".mm", (512,), nn.zeros, False), (pre + ".mv", (512,), nn.ones, False)]
        specs += [("dense.w", (512, output_classes), nn.glorot_uniform, True), ("dense.b", (output_classes,), nn.zeros, True)]
        super().__init__("recognizer", specs, _gen_for("recognizer"))
        self.classes = output_classes

    def _bn_fwd(self, x, pre, bn_training):
        p = self.store.p
        if bn_training:
            y, ctx = nn.bn_train_fwd(x, p[pre + ".gamma"], p[pre + ".beta"], False, False, self.reducer)
            ops.bn_update_moving(p[pre + ".mm"], p[pre + ".mv"], ctx[2], ctx[3], ctx[5])
            return y, ("train", ctx)
        y = ops.bn_apply(x, p[pre + ".mm"], p[pre + ".mv"], p[pre + ".gamma"], p[pre + ".beta"], False, False)
        return y, ("infer", x)

    def _bn_bwd(self, ctx, dy, pre, want_dw):
        p, g = self.store.p, self.store.g
        dg, db = (g[pre + ".gamma"], g[pre + ".beta"]) if want_dw else (None, None)
        if ctx[0] == "train":
            return nn.bn_train_bwd(ctx[1], dy, False, False, self.reducer, dgamma_c=dg, dbeta_c=db)[0]
        x = ctx[1]
        if want_dw:
            ops.bn_bwd_reduce(dy, None, x, p[pre + ".mm"], p[pre + ".mv"], p[pre + ".gamma"], False, False, dgamma_c=dg, dbeta_c=db)
        return ops.bn_bwd_apply(dy, None, x, p[pre + ".mm"], p[pre + ".mv"], p[pre + ".gamma"], False, None, 1, False, False)

    def forward(self, x, labels, input_length, label_length, training=True, need_grad=True):
        """-> per-sample CTC cost [B] (the model's OUTPUT is the loss, net_architecture.py:71-74), ctx."""
        with ops.bf16_only():                  # config c5: the recognizer stays bf16 when G / D / S run fp8
            return self._forward(x, labels, input_length, label_length, training, need_grad)

    def _forward(self, x, labels, input_length, label_length, training=True, need_grad=True):
        p = self.store.p
        x = _as_nhwc1(x, self.device)
        labels = _as_labels(labels, self.device)
        bn_training = bool(training and self.trainable)        # TF2: training AND layer.trainable (fact 4)
        acts, net = [], x
        for i, (k, co, pool) in enumerate(_REC):
            a = ops.conv2d_fwd(net, p["conv%d.w" % (i + 1)], p["conv%d.b" % (i + 1)], same=(k == 3), relu_out=True)
            rec = {"in": net, "a": a}
            net = a
            if i in (4, 5):
                net, rec["bn"] = self._bn_fwd(net, "bn%d" % (i + 1), bn_training)
            if pool is not None:
                net, rec["idx"] = ops.maxpool_fwd(net, *pool)
            acts.append(rec)
        B, one, T, C = net.shape
        assert one == 1, net.shape
        feat = net.view(B * T, C)
        logits = ops.dense_fwd(feat, p["dense.w"], p["dense.b"]).view(B, T, self.classes)     # Dense (softmax fused below)
        loss, dlogits = ops.softmax_ctc(logits, labels, int(input_length), int(label_length), need_grad)
        return loss, (acts, feat, dlogits, (B, T))

    def can_merge(self, training=True) -> bool:
        """Calls may share one pass only while BatchNorm is in inference mode (per-sample independent)."""
        return type(self) is RecognizerModel and not (training and self.trainable)

    def forward_multi(self, xs, labels_list, input_length, label_length, training=True):
        xs = [_as_nhwc1(x, self.device) for x in xs]
        labs = [_as_labels(l, self.device) for l in labels_list]
        bounds, lo = [], 0
        for x in xs:
            bounds.append((lo, lo + x.shape[0]))
            lo += x.shape[0]
        loss, ctx = self.forward(torch.cat(xs, 0), torch.cat(labs, 0), input_length, label_length, training)
        return [loss[a:b] for a, b in bounds], ctx, bounds

    @staticmethod
    def slice_ctx(ctx, lo, hi):
        acts, feat, dlogits, (B, T) = ctx
        out = []
        for rec in acts:
            r = {"in": rec["in"][lo:hi], "a": rec["a"][lo:hi]}
            if "idx" in rec:
                r["idx"] = rec["idx"][lo:hi]
            if "bn" in rec:
                assert rec["bn"][0] == "infer"
                r["bn"] = ("infer", rec["bn"][1][lo:hi])
            out.append(r)
        return out, feat.view(B, T, -1)[lo:hi].reshape((hi - lo) * T, -1), dlogits[lo:hi], (hi - lo, T)

    def backward(self, ctx, upstream, want_dx: bool, want_dw: bool):
        """upstream [B] = d(target)/d(cost_b).  Returns d(target)/d(images) if want_dx."""
        with ops.bf16_only():
            return self._backward(ctx, upstream, want_dx, want_dw)

    def _backward(self, ctx, upstream, want_dx: bool, want_dw: bool):
        p, g = self.store.p, self.store.g
        acts, feat, dlogits_unit, (B, T) = ctx
        dl = ops.rowscale(dlogits_unit, upstream.reshape(-1).contiguous()).view(B * T, self.classes)
        if want_dw:
            ops.dense_bwd_weight(feat, dl, g["dense.w"])
            ops.bias_grad(dl, g["dense.b"])
        d = ops.dense_bwd_input(dl, p["dense.w"]).view(B, 1, T, feat.shape[1])
        for i in reversed(range(len(_REC))):
            k, co, pool = _REC[i]
            rec = acts[i]
            if pool is not None:
                d = ops.maxpool_bwd(d, rec["idx"], *pool)
            if "bn" in rec:
                d = self._bn_bwd(rec["bn"], d, "bn%d" % (i + 1), want_dw)
            d = ops.relu_mask(d, rec["a"])                       # activation='relu' backward (mask by the output)
            if want_dw:
                ops.conv2d_bwd_weight(rec["in"], d, g["conv%d.w" % (i + 1)], same=(k == 3), db=g["conv%d.b" % (i + 1)])
            if i == 0 and not want_dx:
                return None
            xin = rec["in"]
            d = ops.conv2d_bwd_data(d, p["conv%d.w" % (i + 1)], (xin.shape[1], xin.shape[2]), same=(k == 3))
        return d

    def __call__(self, inputs, training=False):
        x, labels, il, ll = inputs
        return self.forward(x, labels, int(_scalar(il)), int(_scalar(ll)), training, need_grad=False)[0].view(-1, 1)


def _scalar(v):
    if torch.is_tensor(v):
        return v.reshape(-1)[0].item()
    try:
        import numpy as np
        return np.asarray(v).reshape(-1)[0]
    except Exception:
        return v


class CompositeGAN:
    """make_gan (net_architecture.py:531-561): G -> {D, R, S}; freezing D/R/S leaves G's variables
    as the composite's trainable_variables."""

    def __init__(self, g_model, d_model, r_model, w_model):
        self.generator, self.discriminator, self.recognizer, self.style_promoter = g_model, d_model, r_model, w_model

    @property
    def trainable_variables(self):
        out = []
        for m in (self.generator, self.discriminator, self.recognizer, self.style_promoter):
            out += m.trainable_variables
        return out

    def __call__(self, inputs, training=False):
        style, y, il, ll = inputs
        img = self.generator.forward(style, y, training=training)[0]
        d = self.discriminator.forward(img)[0]
        r = self.recognizer.forward(img, y, int(_scalar(il)), int(_scalar(ll)), training, need_grad=False)[0].view(-1, 1)
        s = self.style_promoter.forward(img)[0]
        return [img, d, r, s]


# --------------------------------------------------------------------------------------------
# factories (reference names, positional order and defaults)
# --------------------------------------------------------------------------------------------
def make_generator(latent_dim, input_dim, embed_y, kernel_reg, blocks_with_attention, vocab_size, vis_model=True):
    m = GeneratorModel(latent_dim, input_dim, embed_y, kernel_reg, blocks_with_attention, vocab_size)
    if vis_model:
        m.summary()
    return m


def make_discriminator(input_dim, kernel_reg, blocks_with_attention, vis_model=True):
    m = DiscriminatorModel("discriminator", input_dim, kernel_reg, blocks_with_attention)
    if vis_model:
        m.summary()
    return m


def make_style_promoter(input_dim, kernel_reg, blocks_with_attention, vis_model=True):
    m = DiscriminatorModel("style_promoter", input_dim, kernel_reg, blocks_with_attention)
    if vis_model:
        m.summary()
    return m


def make_style_extractor(input_dim, kernel_reg, blocks_with_attention, vis_model=True):
    """The discriminator trunk with a 128-wide linear head (net_architecture.py:465-498); not used by main.py."""
    m = DiscriminatorModel("style_extractor", input_dim, kernel_reg, blocks_with_attention, units=128)
    if vis_model:
        m.summary()
    return m


def make_recognizer(input_dim, sequence_length, output_classes, vis_model=True):
    m = RecognizerModel(input_dim, sequence_length, output_classes)
    if vis_model:
        m.summary()
    return m


_MYREC_FILTERS = [16, 32, 48, 64, 80, 128, 144]
_MYREC_POOLS = [(2, 2), (2, 2), (2, 1), (2, 1), (2, 1), None, None]


class MyRecognizerModel(RecognizerModel):
    """make_my_recognizer (net_architecture.py:82-179): 7 x [Conv3x3 -> BatchNorm -> LeakyReLU(0.01)] with five
    max-pools and Dropout(0.2) before convs 3-7, then 5 x Bidirectional(LSTM(256, dropout=0.5)), Dropout(0.5),
    Dense(softmax) and the same CTC head.  T = W/4 frames; CTC reads the first 4L-1 of them.  Dropout masks are
    drawn on the device when training (or passed explicitly for parity tests)."""

    LSTM_H = 256

    def __init__(self, input_dim, sequence_length, output_classes):
        h, w, c = input_dim
        specs, cin = [], c
        for i, co in enumerate(_MYREC_FILTERS):
            k = i + 1
            specs += [("conv%d.w" % k, (3, 3, cin, co), nn.glorot_uniform, True), ("conv%d.b" % k, (co,), nn.zeros, True),
                      ("bn%d.gamma" % k, (co,), nn.ones, True), ("bn%d.beta" % k, (co,), nn.zeros, True),
                      ("bn%d.mm" % k, (co,), nn.zeros, False), ("bn%d.mv" % k, (co,), nn.ones, False)]
            cin = co
        for l in range(5):
            specs += nn.bilstm_specs("lstm%d" % (l + 1), cin, self.LSTM_H)
            cin = 2 * self.LSTM_H
        specs += [("dense.w", (cin, output_classes), nn.glorot_uniform, True), ("dense.b", (output_classes,), nn.zeros, True)]
        _Model.__init__(self, "my_recognizer", specs, _gen_for("my_recognizer"))
        self.classes = output_classes
        self.mask_gen = torch.Generator(device=self.device)
        self.mask_gen.manual_seed(_DEFAULTS["seed"] * 1000 + 991)

    def _drop(self, shape, rate):
        keep = torch.rand(shape, device=self.device, generator=self.mask_gen) >= rate
        return keep.float() / (1.0 - rate)

    def draw_masks(self, B, W):
        """Dropout masks of one training call (values 0 or 1/(1-rate))."""
        m = {}
        hh, ww, cin = 32, W, 1
        for i, (co, pool) in enumerate(zip(_MYREC_FILTERS, _MYREC_POOLS)):
            if i + 1 >= 3:
                m["drop%d" % (i + 1)] = self._drop((B, hh, ww, cin), 0.2)
            if pool is not None:
                hh, ww = hh // pool[0], ww // pool[1]
            cin = co
        for l in range(5):
            m["lstm%d" % (l + 1)] = (self._drop((B, cin), 0.5), self._drop((B, cin), 0.5))
            cin = 2 * self.LSTM_H
        m["drop_out"] = self._drop((B, ww, cin), 0.5)
        return m

    def forward(self, x, labels, input_length, label_length, training=True, need_grad=True, masks=None):
        p, S = self.store.p, self.store
        x = _as_nhwc1(x, self.device)
        labels = _as_labels(labels, self.device)
        bn_training = bool(training and self.trainable)        # TF2: training AND layer.trainable
        if training and masks is None:
            masks = self.draw_masks(x.shape[0], x.shape[2])      # Keras Dropout follows `training` only
        acts, net = [], x
        for i, (co, pool) in enumerate(zip(_MYREC_FILTERS, _MYREC_POOLS)):
            k = i + 1
            rec = {}
            if masks is not None and k >= 3:
                net = ops.mul_mask(net, masks["drop%d" % k])
            rec["in"] = net
            conv = ops.conv2d_fwd(net, p["conv%d.w" % k], p["conv%d.b" % k])
            bn_out, rec["bn"] = self._bn_fwd(conv, "bn%d" % k, bn_training)
            rec["bn_out"] = bn_out
            net = ops.leaky_relu_fwd(bn_out, 0.01)
            if pool is not None:
                net, rec["idx"] = ops.maxpool_fwd(net, *pool)
            acts.append(rec)
        B, one, T, C = net.shape
        assert one == 1, net.shape
        seq = net.view(B, T, C)
        lstm_ctx = []
        for l in range(5):
            seq, c = nn.bilstm_fwd(seq, S, "lstm%d" % (l + 1), None if masks is None else masks["lstm%d" % (l + 1)])
            lstm_ctx.append(c)
        if masks is not None:
            seq = ops.mul_mask(seq, masks["drop_out"])
        feat = seq.reshape(B * T, seq.shape[2])
        logits = ops.dense_fwd(feat, p["dense.w"], p["dense.b"]).view(B, T, self.classes)
        loss, dlogits = ops.softmax_ctc(logits, labels, int(input_length), int(label_length), need_grad)
        return loss, (acts, lstm_ctx, feat, dlogits, masks, (B, T))

    def backward(self, ctx, upstream, want_dx: bool, want_dw: bool):
        p, g, S = self.store.p, self.store.g, self.store
        acts, lstm_ctx, feat, dlogits_unit, masks, (B, T) = ctx
        dl = ops.rowscale(dlogits_unit, upstream.reshape(-1).contiguous()).view(B * T, self.classes)
        if want_dw:
            ops.dense_bwd_weight(feat, dl, g["dense.w"])
            ops.bias_grad(dl, g["dense.b"])
        d = ops.dense_bwd_input(dl, p["dense.w"]).view(B, T, feat.shape[1])
        if masks is not None:
            d = ops.mul_mask(d, masks["drop_out"])
        for l in reversed(range(5)):
            d = nn.bilstm_bwd(lstm_ctx[l], d, S, "lstm%d" % (l + 1), want_dw)
        d = d.reshape(B, 1, T, d.shape[2])
        for i in reversed(range(len(_MYREC_FILTERS))):
            k = i + 1
            rec, pool = acts[i], _MYREC_POOLS[i]
            if pool is not None:
                d = ops.maxpool_bwd(d, rec["idx"], *pool)
            d = ops.leaky_relu_bwd(d, rec["bn_out"], 0.01)
            d = self._bn_bwd(rec["bn"], d, "bn%d" % k, want_dw)
            if want_dw:
                ops.conv2d_bwd_weight(rec["in"], d, g["conv%d.w" % k], db=g["conv%d.b" % k])
            if i == 0 and not want_dx:
                return None
            xin = rec["in"]
            d = ops.conv2d_bwd_data(d, p["conv%d.w" % k], (xin.shape[1], xin.shape[2]))
            if masks is not None and k >= 3:
                d = ops.mul_mask(d, masks["drop%d" % k])
        return d


class MyDiscriminatorModel(_Model):
    """make_my_discriminator (net_architecture.py:417-462): four Conv2D 3x3 stride (2,2) 'same' (16, 32, 64, 128 filters,
    orthogonal, bias) each followed by LeakyReLU(0.3), a NonLocalBlock after the second, a second LeakyReLU after the
    fourth, GlobalAveragePooling and Dense(1, no bias).  The NonLocalBlock runs on C = 32 (d_k = 4, d_v = 16): its freshly
    drawn 1x1 kernels are zero-padded to the attention kernel's d_k = 8 / d_v = 32, which changes nothing
    (zero key / query channels add 0 to every score, zero value channels meet zero rows of the output kernel)."""

    FILTERS = (16, 32, 64, 128)
    supports_fused_passes = False          # train_step runs every call of this model as its own pass, every tape its own sweep

    def __init__(self, input_dim, kernel_reg):
        h, w, c = input_dim
        get_in_out_channels_disc(colors=c, resolution=h)          # the reference's argument checks (:421)
        specs, cin = [], c
        for i, co in enumerate(self.FILTERS):
            specs += [("conv%d.w" % (i + 1), (3, 3, cin, co), nn.orthogonal, True), ("conv%d.b" % (i + 1), (co,), nn.zeros, True)]
            cin = co
        specs += [("NL_B1.sigma", (), nn.zeros, True), ("dense.w", (cin, 1), nn.orthogonal, True)]
        gen = _gen_for("my_discriminator")
        super().__init__("my_discriminator", specs, gen)
        self.kernel_reg = kernel_reg
        self.nl_gen = torch.Generator().manual_seed(gen.initial_seed() + 7)

    @staticmethod
    def pad_nl(nlw, device):
        """The C = 32 NonLocalBlock kernels (theta / phi [32,4], g [32,16], o [16,32]) zero-padded to d_k 8 / d_v 32."""
        t, ph, g, o = (nlw[k].reshape(nlw[k].shape[-2:]).to(device) for k in ("theta", "phi", "g", "o"))
        C = t.shape[0]
        tp, pp = torch.zeros(C, 8, device=device), torch.zeros(C, 8, device=device)
        gp, op = torch.zeros(C, 32, device=device), torch.zeros(32, C, device=device)
        tp[:, :t.shape[1]], pp[:, :ph.shape[1]], gp[:, :g.shape[1]], op[:o.shape[0]] = t, ph, g, o
        return {"theta": tp, "phi": pp, "g": gp, "o": op}

    def forward(self, x, nl=None):
        S = self._pass_store()          # kernel_reg 'applied': spectrally normalised conv / dense kernels for this pass (net_architecture.py:425-450)
        p = S.p
        x = _as_nhwc1(x, self.device)
        if nl is None:
            nl = nn.nonlocal_weights(32, self.nl_gen, torch.device("cpu"))
        acts, net, nlc = [], x, None
        for i in range(4):
            pre = nn.strided_conv_fwd(net, p["conv%d.w" % (i + 1)], p["conv%d.b" % (i + 1)])
            acts.append((net, pre))
            net = ops.leaky_relu_fwd(pre, 0.3)
            if i == 1:
                net, nlc = nn.nonlocal_fwd(net, self.pad_nl(nl, self.device), p["NL_B1.sigma"])
        last = ops.leaky_relu_fwd(net, 0.3)                         # the second LeakyReLU of :446
        h = ops.gap_fwd(last, relu=False)
        return ops.dense_fwd(h, p["dense.w"]), (acts, nlc, net, last, h, S)

    def backward(self, ctx, dlogits, want_dx: bool, want_dw: bool, wscale=None):
        if wscale is not None:
            raise NotImplementedError("shared backward sweeps are not wired for make_my_discriminator: pass share_backward=False")
        acts, nlc, net4, last, h, S = ctx
        p, g = S.p, S.g
        dlogits = dlogits.reshape(-1, 1).contiguous()
        if want_dw:
            ops.dense_bwd_weight(h, dlogits, g["dense.w"])
        d = ops.gap_bwd(ops.dense_bwd_input(dlogits, p["dense.w"]), last, relu=False)
        d = ops.leaky_relu_bwd(d, net4, 0.3)
        for i in reversed(range(4)):
            xin, pre = acts[i]
            if i == 1:
                dsig = g["NL_B1.sigma"] if want_dw else torch.zeros(1, device=self.device)
                d = nn.nonlocal_bwd(nlc, d, p["NL_B1.sigma"], dsig)
            d = ops.leaky_relu_bwd(d, pre, 0.3)
            gw = g["conv%d.w" % (i + 1)] if want_dw else torch.zeros_like(p["conv%d.w" % (i + 1)])
            gb = g["conv%d.b" % (i + 1)] if want_dw else torch.zeros_like(p["conv%d.b" % (i + 1)])
            d = nn.strided_conv_bwd(xin, p["conv%d.w" % (i + 1)], d, gw, gb, want_dx or i > 0)
        if want_dw:
            self._fold(S)
        return d

    def __call__(self, inputs, training=False):
        x = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        return self.forward(x)[0]


def make_my_discriminator(gen_path, input_dim, kernel_reg, vis_model=True):
    """The plain strided-conv discriminator alternative (net_architecture.py:417-462; `shared_specs.my_disc = 1`)."""
    m = MyDiscriminatorModel(input_dim, kernel_reg)
    if vis_model:
        m.summary()
    return m


def make_my_recognizer(input_dim, sequence_length, output_classes, vis_model=True):
    """CRNN with 5 BiLSTM layers (net_architecture.py:82-179); selected by `shared_specs.my_rec = 1`."""
    m = MyRecognizerModel(input_dim, sequence_length, output_classes)
    if vis_model:
        m.summary()
    return m


def make_gan(g_model, d_model, r_model, w_model, vis_model=True):
    d_model.trainable = False          # :543-545
    r_model.trainable = False
    w_model.trainable = False
    return CompositeGAN(g_model, d_model, r_model, w_model)
