"""The hot path and its host loop, with the reference's names and signatures
(/root/reference/src/bigacgan/data_utils.py): train_step :358-473, apply_gradient_balancing :476-490,
train :198-352, load_random_word_list :550-574, and the label/index bookkeeping of load_prepare_data
:14-84.

train_step keeps the 22-positional-parameter signature and the 16-scalar return.  Where the
reference holds four GradientTapes open, this build runs the eight forward passes once, computes the
per-sample upstream gradients of all four targets in one fused loss-head kernel, runs the backward
sweeps against the SAME weight snapshot, and only then applies the four optimizer
updates -- so results are those of the reference while D/S/R updates cannot leak into G's backward.
"""
from __future__ import annotations

import os
import random
import time
from typing import List, Sequence

import numpy as np
import torch

from . import ops
from .net_architecture import _as_labels, _as_nhwc1

# ------------------------------------------------------------------------------------------------
# label / index bookkeeping (integer: bit-exact with the reference)
# ------------------------------------------------------------------------------------------------
def encode_word(word: str, char_vector: str) -> List[int]:
    """'auto' -> [0, 20, 19, 14]  (data_utils.py:49,572: char_vector.index(char))."""
    return [char_vector.index(ch) for ch in word]


def bucket_width(input_dim, bucket: int) -> int:
    """image width of data bucket `bucket` (1-based): int(h/2 * bucket)  (data_utils.py:81)."""
    return int((input_dim[0] / 2) * bucket)


def normalize_images(image_batch: np.ndarray, input_dim, bucket: int) -> np.ndarray:
    """uint8 [B,h,w] -> float32 [B,h,16*bucket,c] in [-1,1]: (x-127.5)/127.5  (data_utils.py:77-82)."""
    h, _, c = input_dim
    x = np.asarray(image_batch).astype("float32").reshape(-1, h, bucket_width(input_dim, bucket), c)
    return (x - 127.5) / 127.5


def ctc_input_length(sequence_length: int) -> int:
    """-1 + sequence_length * 4  (data_utils.py:400,413)."""
    return -1 + sequence_length * 4


def bucket_words(words: Sequence[str], bucket_size: int, char_vector: str) -> List[List[List[int]]]:
    """Word-list bucketing of load_random_word_list (data_utils.py:561-574): 0-based bucket = len-1,
    words longer than bucket_size dropped, order preserved."""
    random_words: List[List[List[int]]] = [[] for _ in range(bucket_size)]
    for word in words:
        word = word.strip()
        bucket = len(word)
        if bucket <= bucket_size:
            random_words[bucket - 1].append(encode_word(word, char_vector))
    return random_words


def load_random_word_list(reading_dir, bucket_size, char_vector):
    """random_words.txt lives three directory levels above reading_dir (data_utils.py:565-566)."""
    path = os.path.dirname(os.path.dirname(os.path.dirname(reading_dir)))
    with open(os.path.join(path, "random_words.txt"), "r") as f:
        return bucket_words(list(f), bucket_size, char_vector)


def draw_fake_labels(random_words, bucket_size: int, batch_size: int):
    """The host draws of train_step (data_utils.py:386-387), same `random` call sequence."""
    random_bucket_idx = random.randint(0, bucket_size - 1)
    fake_labels = np.array([random.choice(random_words[random_bucket_idx]) for _ in range(batch_size)], np.int32)
    return random_bucket_idx, fake_labels


# ------------------------------------------------------------------------------------------------
def apply_gradient_balancing(r_fake_logits, g_loss, alpha=1):
    """data_utils.py:476-490 on device tensors: returns (g_balanced, r_loss_balanced, alpha, r_std, g_std).
    (train_step uses the fused loss head; this mirrors the public helper.)"""
    r = r_fake_logits.reshape(-1).contiguous()
    g = g_loss.reshape(-1).contiguous()
    z = torch.zeros_like(r)
    # loss-head kernels give the two population stds; the rest is the definition
    sums = ops.loss_sums(z, z, z, z, z, r, z, 0)
    r_mean, r_sq = (sums[7] / sums[11]).item(), (sums[10] / sums[11]).item()
    sums_g = ops.loss_sums(z, z, z, z, z, g, z, 0)
    g_mean, g_sq = (sums_g[7] / sums_g[11]).item(), (sums_g[10] / sums_g[11]).item()
    r_std, g_std = max(r_sq - r_mean ** 2, 0.0) ** 0.5, max(g_sq - g_mean ** 2, 0.0) ** 0.5
    ratio = torch.full((1,), alpha * g_std / r_std, device=r.device)
    r_bal = ops.scale(_pad4(r), ratio)[: r.numel()]
    g_bal = ops.add(g, r_bal.contiguous())
    return g_bal.view(-1, 1), r_bal.view(-1, 1), alpha, r_std, g_std


def _pad4(t):
    n = (t.numel() + 3) // 4 * 4
    if n == t.numel():
        return t
    out = torch.zeros(n, device=t.device)
    out[: t.numel()] = t
    return out


DEBUG_KEEP = None      # tests set a dict here to receive the generator's saved context of the next train_step


# ------------------------------------------------------------------------------------------------
def train_step(epoch_idx, batch_idx, batch_per_epoch, images, labels, discriminator, recognizer, style_promoter, composite_gan,
               generator_optimizer, discriminator_optimizer, recognizer_optimizer, stylepromoter_optimizer, my_imgs,
               batch_size, latent_dim, loss_fn, disc_iters, apply_gradient_balance, random_words, bucket_size, gen_path,
               fake_labels=None, nl=None, verbose=True, sync=True, fuse_passes=True, share_backward=True):
    """One optimisation step (data_utils.py:358-473).

    Extra keyword arguments (not in the reference): `fake_labels` overrides the host draw of :386-387
    (parity tests, data-parallel ranks that received their shard from rank 0); `nl` maps pass names
    ('G.style','G.up','D.fake','D.real','S.fake','S.style','S.real') to explicit NonLocalBlock
    kernels; `sync=False` returns the 16 scalars as a device tensor without a host sync and `sync="lazy"` a
    StepScalars sequence that waits for its device-to-host copy on first access; `fuse_passes=False`
    keeps every reference call a separate pass even when the input widths match; `share_backward=False` runs the
    weight-gradient and image-gradient sweeps through D(x_f) / S(x_f) separately, as the reference's tapes do.
    """
    G = composite_gan.generator
    D, R, S = discriminator, recognizer, style_promoter
    ops.new_step()                                                  # bf16 twins of the previous step's activations are released
    dev = G.device
    red = G.reducer
    nl = nl or {}

    # ---- host RNG + shapes (:385-395); tf.random.normal `noise` is unused by the reference graph ----
    if fake_labels is None:
        random_bucket_idx, fake_labels = draw_fake_labels(random_words, bucket_size, batch_size)
        if red.world_size > 1:
            # every rank must shard the SAME global batch (same L_f, same words): take rank 0's draw, whatever the state of
            # this rank's `random` stream (main.py seeds all ranks alike, other callers may not) -- ADVICE r1
            fake_labels = red.broadcast_object(fake_labels)
    images = _as_nhwc1(images, dev)
    n_chars = G.store.shapes["filter_bank"][0]                      # vocabulary size = CTC classes - 1 (blank is the last class)
    labels_t = _as_labels(labels, dev, n_chars)
    fake_t = _as_labels(fake_labels, dev, n_chars)
    if isinstance(my_imgs, (list, tuple)):
        my_imgs = np.stack([np.asarray(m) for m in my_imgs], axis=0) if not torch.is_tensor(my_imgs[0]) else torch.stack(list(my_imgs), 0)
    style = _as_nhwc1(my_imgs, dev)
    if red.world_size > 1 and getattr(red, "shard_inputs", True):       # data parallel: this rank's slice of every per-step tensor
        images, labels_t, fake_t, style = (red.shard(t) for t in (images, labels_t, fake_t, style))
    L_r, L_f = labels_t.shape[1], fake_t.shape[1]

    # NonLocalBlock re-draws its four 1x1 kernels on every call.  Draw all seven sets of this
    # step now -- same generators, same order as the per-call draws -- with one async upload, so no forward pass
    # stalls the launch queue on a host->device copy.
    if G.nl_mode == "reference":
        order = [("G.style", G, G.trunk.attn), ("G.up", G, G.up_attn), ("D.fake", D, D.trunk.attn), ("S.fake", S, S.trunk.attn),
                 ("D.real", D, D.trunk.attn), ("S.style", S, S.trunk.attn), ("S.real", S, S.trunk.attn)]
        def chan(m, attn, up):
            return m.out_ch[m.up_names.index(attn[0])] if up else m.trunk.cout[m.trunk.names.index(attn[0])]
        # (passes with several attention blocks keep the per-call draw inside forward)
        todo = [(name, m, chan(m, attn, name == "G.up")) for name, m, attn in order if len(attn) == 1 and name not in nl]
        if todo:
            from .nn import nonlocal_weights_batch
            drawn = nonlocal_weights_batch([(c, m.nl_gen) for _, m, c in todo], dev)
            todo = [(name, m) for name, m, _ in todo]
            nl = dict(nl)
            nl.update({name: d for (name, _), d in zip(todo, drawn)})

    # ---- forward passes, all with the pre-update weights (:398-415) ----
    # The reference calls D twice, S three times and R twice per step.  Where the inputs have the same width
    # (always for the fixed-shape configs) those calls ride in ONE pass over the concatenated batch: D and S have
    # no BatchNorm and the frozen R normalises with moving statistics, so samples are independent and the result is
    # the same; each call keeps its own NonLocalBlock kernels (passes may be re-ordered).
    x_f, ctx_g = G.forward(style, fake_t, nl.get("G.style"), nl.get("G.up"), training=True)
    if DEBUG_KEEP is not None:
        DEBUG_KEEP["ctx_g"] = ctx_g                                 # (tests: the ReLU decisions the forward pass took)
    B = x_f.shape[0]
    il_f, il_r = ctc_input_length(L_f), ctc_input_length(L_r)
    plain = all(getattr(m, "supports_fused_passes", True) for m in (D, S))     # (make_my_discriminator: separate passes and sweeps)
    fuse = fuse_passes and plain and x_f.shape == images.shape
    use_ns = False
    if fuse:
        fuse_style = style.shape == x_f.shape
        # small per-GPU batches (the data-parallel shards): S's passes on a second stream beside D's and R's (ops.net_stream)
        use_ns = fuse_style and ops.net_stream_enabled(3 * B)
        if use_ns:
            nls = [nl.get("S.fake"), nl.get("S.style"), nl.get("S.real")]
            with ops.net_stream(x_f, style, images, *[t for d in nls if d is not None for t in d.values()]) as ns_f:
                (s_f, s_my, s_r), ctx_S, _ = S.forward_multi([x_f, style, images], nls)
        (d_f, d_r), ctx_D, _ = D.forward_multi([x_f, images], [nl.get("D.fake"), nl.get("D.real")])
        if use_ns:
            pass
        elif fuse_style:
            (s_f, s_my, s_r), ctx_S, _ = S.forward_multi([x_f, style, images], [nl.get("S.fake"), nl.get("S.style"), nl.get("S.real")])
        else:
            (s_f, s_r), ctx_S, _ = S.forward_multi([x_f, images], [nl.get("S.fake"), nl.get("S.real")])
            s_my, ctx_smy = S.forward(style, nl.get("S.style"))
        fuse_r = R.can_merge(True)
        if fuse_r:
            (r_f, r_r), ctx_R, _ = R.forward_multi([x_f, images], [fake_t, labels_t], il_f, L_f, training=True)
        else:
            r_f, ctx_rf = R.forward(x_f, fake_t, il_f, L_f, training=True)
            r_r, ctx_rr = R.forward(images, labels_t, il_r, L_r, training=True)
    else:
        d_f, ctx_df = D.forward(x_f, nl.get("D.fake"))
        s_f, ctx_sf = S.forward(x_f, nl.get("S.fake"))
        r_f, ctx_rf = R.forward(x_f, fake_t, il_f, L_f, training=True)
        d_r, ctx_dr = D.forward(images, nl.get("D.real"))
        s_my, ctx_smy = S.forward(style, nl.get("S.style"))
        s_r, ctx_sr = S.forward(images, nl.get("S.real"))
        r_r, ctx_rr = R.forward(images, labels_t, il_r, L_r, training=True)

    if use_ns:
        ns_f.join(s_f, s_my, s_r)
    if DEBUG_KEEP is not None:                                      # (tests: the recognizer's ReLU / max-pool decisions)
        DEBUG_KEEP["R_f"] = R.slice_ctx(ctx_R, 0, B) if (fuse and fuse_r) else ctx_rf
        DEBUG_KEEP["R_r"] = R.slice_ctx(ctx_R, B, 2 * B) if (fuse and fuse_r) else ctx_rr
        if plain:                                                   # D / S contexts per reference call (fake, real / fake, style, real)
            DEBUG_KEEP["D_f"], DEBUG_KEEP["D_r"] = (D.slice_ctx(ctx_D, 0, B), D.slice_ctx(ctx_D, B, 2 * B)) if fuse else (ctx_df, ctx_dr)
            if fuse and fuse_style:
                DEBUG_KEEP["S_f"], DEBUG_KEEP["S_my"], DEBUG_KEEP["S_r"] = (S.slice_ctx(ctx_S, i * B, (i + 1) * B) for i in range(3))
            elif fuse:
                DEBUG_KEEP["S_f"], DEBUG_KEEP["S_my"], DEBUG_KEEP["S_r"] = S.slice_ctx(ctx_S, 0, B), ctx_smy, S.slice_ctx(ctx_S, B, 2 * B)
            else:
                DEBUG_KEEP["S_f"], DEBUG_KEEP["S_my"], DEBUG_KEEP["S_r"] = ctx_sf, ctx_smy, ctx_sr

    # ---- losses, gradient balancing, statistics and the upstream gradients of all four targets (:418-442) ----
    mode = getattr(loss_fn, "mode", None)
    if mode is None:
        raise TypeError("loss_fn must be scrabble_gan_amd.net_loss.hinge or .not_saturating")
    v = [t.reshape(-1) for t in (d_r, d_f, s_my, s_f, s_r)]
    r_f, r_r = r_f.reshape(-1), r_r.reshape(-1)
    sums = red.all_reduce_sum(ops.loss_sums(*v, r_f, r_r, mode))
    scalars, (gD_r, gD_f, gS_my, gS_f, gG_d, gG_s, gG_r, shD, shS) = ops.loss_grads(*v, r_f, mode, bool(apply_gradient_balance), 1.0, sums)

    # ---- backward sweeps against the same weight snapshot (:449-468) ----
    # Each network's flat gradient buffer starts its SUM all-reduce (data parallel; the targets are [B,1]
    # vectors) as soon as its sweeps are queued, so the D/R/S exchanges overlap G's backward.
    #
    # Shared sweeps: the reference back-propagates through D(x_f) twice (tape of d_loss for D's weights, tape of
    # g_final for the image gradient) and likewise through S(x_f).  Backprop is linear in the per-sample upstream
    # scalar, so ONE sweep with upstream u_b produces both: sample b enters the weight gradients with factor
    # gD_f[b]/u_b and the image gradient is rescaled by gG_d[b]/u_b (factors from the loss-head kernel, |.| <= 1).
    g_step = (batch_idx + 1) % disc_iters == 0
    share = share_backward and g_step and plain
    ones_b = torch.ones(B, device=dev)
    for m in (D, R, S):
        m.store.zero_grad()
    discriminator.trainable = True
    dx_d = dx_s = None
    ns_b = None
    if use_ns and share:               # S's sweep first, on the network stream; D's and R's follow on the launch stream
        style_promoter.trainable = True
        with ops.net_stream(shS, gS_my, ones_b) as ns_b:
            dxs_all = S.backward(S.slice_ctx(ctx_S, 0, 2 * B), torch.cat([shS[0], gS_my]), want_dx=True, want_dw=True,
                                 wscale=torch.cat([shS[1], ones_b]))
            dx_s = ops.rowscale(dxs_all[:B], shS[2])
    if fuse and share:
        dx_all = D.backward(ctx_D, torch.cat([shD[0], gD_r]), want_dx=True, want_dw=True, wscale=torch.cat([shD[1], ones_b]))
        dx_d = ops.rowscale(dx_all[:B], shD[2])
    elif fuse:
        D.backward(ctx_D, torch.cat([gD_f, gD_r]), want_dx=False, want_dw=True)
    else:
        D.backward(ctx_dr, gD_r, want_dx=False, want_dw=True)
        if share:
            dx_d = ops.rowscale(D.backward(ctx_df, shD[0], want_dx=True, want_dw=True, wscale=shD[1]), shD[2])
        else:
            D.backward(ctx_df, gD_f, want_dx=False, want_dw=True)
    pending = [red.all_reduce_sum_async(D.store.grad)]
    recognizer.trainable = True
    ones = torch.ones_like(r_r)                                   # target r_real_logits: CTC on real only
    if fuse and fuse_r:
        R.backward(R.slice_ctx(ctx_R, B, 2 * B), ones, want_dx=False, want_dw=True)
    else:
        R.backward(ctx_rr, ones, want_dx=False, want_dw=True)
    pending.append(red.all_reduce_sum_async(R.store.grad))
    style_promoter.trainable = True
    if ns_b is not None:
        ns_b.join(dx_s)                # (S's gradients and image gradient are complete from here on, on the launch stream)
    elif fuse and fuse_style:
        if share:
            dxs_all = S.backward(S.slice_ctx(ctx_S, 0, 2 * B), torch.cat([shS[0], gS_my]), want_dx=True, want_dw=True,
                                 wscale=torch.cat([shS[1], ones_b]))
            dx_s = ops.rowscale(dxs_all[:B], shS[2])
        else:
            S.backward(S.slice_ctx(ctx_S, 0, 2 * B), torch.cat([gS_f, gS_my]), want_dx=False, want_dw=True)
    else:
        S.backward(ctx_smy, gS_my, want_dx=False, want_dw=True)
        ctx_sf_ = S.slice_ctx(ctx_S, 0, B) if fuse else ctx_sf
        if share:
            dx_s = ops.rowscale(S.backward(ctx_sf_, shS[0], want_dx=True, want_dw=True, wscale=shS[1]), shS[2])
        else:
            S.backward(ctx_sf_, gS_f, want_dx=False, want_dw=True)
    pending.append(red.all_reduce_sum_async(S.store.grad))
    if g_step:
        recognizer.trainable = False
        discriminator.trainable = False
        style_promoter.trainable = False
        G.store.zero_grad()
        if dx_d is None:
            dx_d = D.backward(D.slice_ctx(ctx_D, 0, B) if fuse else ctx_df, gG_d, want_dx=True, want_dw=False)
        if dx_s is None:
            dx_s = S.backward(S.slice_ctx(ctx_S, 0, B) if fuse else ctx_sf, gG_s, want_dx=True, want_dw=False)
        dx = ops.add(dx_d, dx_s, out=dx_d)
        ctx_rf_ = R.slice_ctx(ctx_R, 0, B) if (fuse and fuse_r) else ctx_rf
        ops.add(dx, R.backward(ctx_rf_, gG_r, want_dx=True, want_dw=False), out=dx)
        # G's buffer is reduced in slices as its backward finalises them: [zdense.w .. end] (filter bank, up blocks, head)
        # first, under the style encoder's backward; then each style-encoder block's range right after that block's
        # backward (B_style4: 80 MB, B_style3: 59 MB, B_style2: 11 MB, B_style1: 0.2 MB) -- only the last, smallest slices
        # are not hidden behind compute.  Whatever the callbacks did not cover is reduced at the end.
        g_parts = []          # (lo, hi, handle)
        applied = getattr(G, "kernel_reg_mode", "reference") == "applied" and getattr(G, "kernel_reg", None) is not None
        if red.world_size > 1 and not applied:
            n_flat = G.store.grad.numel()
            G.backward(ctx_g, dx,
                       on_tail_ready=lambda off: g_parts.append((off, n_flat, red.all_reduce_sum_async(G.store.grad[off:]))),
                       on_slice_ready=lambda lo, hi: g_parts.append((lo, hi, red.all_reduce_sum_async(G.store.grad[lo:hi]))))
        else:
            G.backward(ctx_g, dx)

    # ---- finish the gradient exchange and update: each network as soon as ITS exchange is done, so the updates of
    #      D / R / S run under G's all-reduce ----
    g_handles = []
    if g_step:
        n_flat = G.store.grad.numel()
        covered = sorted((lo, hi) for lo, hi, _ in g_parts)
        g_handles = [h for _, _, h in g_parts]
        pos = 0
        for lo, hi in covered + [(n_flat, n_flat)]:          # the gaps the callbacks left (everything, on one rank / applied mode)
            if lo > pos:
                g_handles.append(red.all_reduce_sum_async(G.store.grad[pos:lo]))
            pos = max(pos, hi)
    for h, opt, m in zip(pending, (discriminator_optimizer, recognizer_optimizer, stylepromoter_optimizer), (D, R, S)):
        red.wait(h)
        opt.apply_flat(m.store)
    if g_step:
        for h in g_handles:
            red.wait(h)
        generator_optimizer.apply_flat(G.store)

    ops.end_step()
    if not sync:
        return scalars
    out = StepScalars(scalars, (epoch_idx, batch_idx, batch_per_epoch) if verbose else None)
    if sync == "lazy":
        return out
    return out.resolve()                                         # the one host sync of the step


class StepScalars:
    """The 16 scalars of data_utils.py:470-473 behind ONE asynchronous device-to-host copy.
    `train_step(..., sync="lazy")` returns this sequence unresolved so the host can queue the next step while this
    one still runs; the first access to a value (indexing, iteration) waits for the copy.  With the default
    `sync=True` train_step resolves it and returns the plain 16-tuple, which is the reference's behaviour."""

    def __init__(self, scalars, progress=None):
        self._host = torch.empty(scalars.numel(), dtype=scalars.dtype, pin_memory=scalars.is_cuda)
        self._host.copy_(scalars, non_blocking=True)
        self._event = None
        if scalars.is_cuda:
            self._event = torch.cuda.Event()
            self._event.record(torch.cuda.current_stream(scalars.device))
        self._values = None
        self._progress = progress

    def resolve(self):
        if self._values is None:
            if self._event is not None:
                self._event.synchronize()
            s = self._host.tolist()
            self._values = (s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], s[9], 1, s[11], s[12], s[13], s[14], s[15])
            if self._progress is not None:                      # the reference's per-step console line (:466-468)
                e, b, n = self._progress
                print('>%d, %d/%d, d=%.3f, d_real=%.3f, d_fake=%.3f, g_trad=%.3f, r_loss_fake=%.3f, g_loss=%.3f, r=%.3f, s=%.3f' % (
                    e + 1, b + 1, n, s[6], s[7], s[8], s[3], s[0], s[9], s[1], s[14]))
        return self._values

    def __len__(self):
        return 16

    def __iter__(self):
        return iter(self.resolve())

    def __getitem__(self, i):
        return self.resolve()[i]

    def __repr__(self):
        return repr(self.resolve())


# ------------------------------------------------------------------------------------------------
# Full training state: the reference saves G and R weights per epoch and cannot resume; this
# writes EVERYTHING a restart needs -- parameters and BN moving statistics of all four networks, the optimizer slots and
# step counters -- into one safetensors file (no pickle), plus the epoch / batch position.
def save_training_state(path, generator, discriminator, recognizer, style_promoter, generator_optimizer, discriminator_optimizer,
                        recognizer_optimizer, stylepromoter_optimizer, epoch_idx=0, batch_idx=0):
    from safetensors.torch import save_file
    out = {"position": torch.tensor([int(epoch_idx), int(batch_idx)], dtype=torch.int64)}
    for tag, m, opt in (("G", generator, generator_optimizer), ("D", discriminator, discriminator_optimizer),
                        ("R", recognizer, recognizer_optimizer), ("S", style_promoter, stylepromoter_optimizer)):
        for k, v in m.store.export().items():
            out["%s/%s" % (tag, k)] = v.contiguous()
        for k, v in opt.flat_state(m.store).items():
            out["%s.opt/%s" % (tag, k)] = v.detach().cpu().contiguous()
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    save_file(out, path + ".tmp")
    os.replace(path + ".tmp", path)                       # a crash while writing never clobbers the previous state


def load_training_state(path, generator, discriminator, recognizer, style_promoter, generator_optimizer, discriminator_optimizer,
                        recognizer_optimizer, stylepromoter_optimizer):
    """Restores what save_training_state wrote; -> (epoch_idx, batch_idx) of the save."""
    from safetensors.torch import load_file
    st = load_file(path)
    for tag, m, opt in (("G", generator, generator_optimizer), ("D", discriminator, discriminator_optimizer),
                        ("R", recognizer, recognizer_optimizer), ("S", style_promoter, stylepromoter_optimizer)):
        pre, opre = tag + "/", tag + ".opt/"
        m.store.load({k[len(pre):]: v for k, v in st.items() if k.startswith(pre)})
        opt.load_flat_state(m.store, {k[len(opre):]: v for k, v in st.items() if k.startswith(opre)})
    ops.weights_changed()
    pos = st["position"].tolist()
    return int(pos[0]), int(pos[1])


SUMMARY_HEADER = ("disc_loss;disc_loss_real;disc_loss_fake;r_loss_real;r_loss_fake;r_loss_balanced;g_loss;g_lossT;g_lossS;"
                  "g_loss_final;alpha;r_loss_fake_std;g_loss_std;s_loss;s_loss_real;s_loss_fake\n")


def train(dataset, generator, discriminator, recognizer, style_promoter, composite_gan, checkpoint, checkpoint_prefix,
          generator_optimizer, discriminator_optimizer, recognizer_optimizer, stylepromoter_optimizer, my_imgs,
          seed_labels, buffer_size, batch_size, epochs, model_path, latent_dim, gen_path, loss_fn, disc_iters,
          apply_gradient_balance, random_words, bucket_size, char_vector, max_batches_per_epoch=None, state_path=None):
    """Epoch/batch loop, 16-column ';' summaries and per-epoch G/R weight saves (data_utils.py:198-352).
    The summary rows carry the ';' the reference drops between g_loss_std and s_loss (Appendix C-10).
    `state_path` (not in the reference): full training state written there after every epoch and, when the file exists
    at start, loaded first -- training continues with the epoch after the saved one."""
    generator_save_dir = os.path.join(checkpoint_prefix, 'generator/')
    recognizer_save_dir = os.path.join(checkpoint_prefix, 'recognizer/')
    # data parallel: every rank runs the loop (same seeded host streams -> same batches, each rank takes its slice inside
    # train_step); only rank 0 writes summaries and checkpoints
    is_main = getattr(generator.reducer, "rank", 0) == 0
    if is_main:
        os.makedirs(generator_save_dir, exist_ok=True)
        os.makedirs(recognizer_save_dir, exist_ok=True)
        os.makedirs(gen_path, exist_ok=True)
    batch_per_epoch = int(buffer_size / batch_size) + 1
    if max_batches_per_epoch is not None:
        batch_per_epoch = min(batch_per_epoch, max_batches_per_epoch)
    print('no. training samples: ', buffer_size)
    print('batch size:           ', batch_size)
    print('no. batch_per_epoch:  ', batch_per_epoch)
    print('epoch size:           ', epochs)
    order = (6, 7, 8, 1, 0, 2, 3, 4, 5, 9, 10, 11, 12, 13, 14, 15)      # return tuple -> summary column order
    first_epoch = 0
    if state_path is not None and os.path.exists(state_path):
        saved_epoch, _ = load_training_state(state_path, generator, discriminator, recognizer, style_promoter, generator_optimizer,
                                             discriminator_optimizer, recognizer_optimizer, stylepromoter_optimizer)
        first_epoch = saved_epoch + 1
        print('resumed from %s: continuing with epoch %d' % (state_path, first_epoch + 1))
    mode = "a" if first_epoch > 0 else "w"                   # a resumed run appends to the summaries of the first one
    with open(os.path.join(gen_path, "batch_summary.txt") if is_main else os.devnull, mode) as batch_summary, \
            open(os.path.join(gen_path, "epoch_summary.txt") if is_main else os.devnull, mode) as epoch_summary:
        if first_epoch == 0:
            epoch_summary.write(SUMMARY_HEADER)
            batch_summary.write(SUMMARY_HEADER)
        for epoch_idx in range(first_epoch, epochs):
            start = time.time()
            totals = [0.0] * 16
            pending = None          # the previous step's scalars: read back after the next step is queued, so the
                                    # device-to-host copy of the 16 values never drains the launch queue
            def flush(res):
                nonlocal totals
                batch_summary.write(";".join(str(res[i]) for i in order) + "\n")
                totals = [t + float(o) for t, o in zip(totals, res)]
            for batch_idx in range(batch_per_epoch):
                image_batch, label_batch = next(dataset)
                my_img_batch = random.choices(my_imgs, k=batch_size)
                out = train_step(epoch_idx, batch_idx, batch_per_epoch, image_batch, label_batch, discriminator, recognizer,
                                 style_promoter, composite_gan, generator_optimizer, discriminator_optimizer,
                                 recognizer_optimizer, stylepromoter_optimizer, my_img_batch, batch_size, latent_dim, loss_fn,
                                 disc_iters, apply_gradient_balance, random_words, bucket_size, gen_path, sync="lazy")
                if pending is not None:
                    flush(pending)
                pending = out
            if pending is not None:
                flush(pending)
            epoch_summary.write(";".join(str(totals[i] / batch_per_epoch) for i in order) + "\n")
            print('Time for epoch {} is {} sec'.format(epoch_idx + 1, time.time() - start))
            if not is_main:
                continue
            generator.save_weights(os.path.join(generator_save_dir, str(epoch_idx + 1), 'cktp-' + str(epoch_idx + 1)))
            recognizer.save_weights(os.path.join(recognizer_save_dir, str(epoch_idx + 1), 'cktp-' + str(epoch_idx + 1)))
            if state_path is not None:
                save_training_state(state_path, generator, discriminator, recognizer, style_promoter, generator_optimizer,
                                    discriminator_optimizer, recognizer_optimizer, stylepromoter_optimizer, epoch_idx, batch_per_epoch)


# ------------------------------------------------------------------------------------------------
def synthetic_batch(batch_size: int, L_real: int, input_dim=(32, 160, 1), n_classes=52, seed=0):
    """Synthetic inputs of SURVEY section 8(d): U(-1,1) images / style images, U{0..51} labels."""
    rng = np.random.default_rng(seed)
    h, w, c = input_dim
    images = rng.uniform(-1, 1, (batch_size, h, 16 * L_real, c)).astype(np.float32)
    labels = rng.integers(0, n_classes, (batch_size, L_real)).astype(np.int32)
    my_imgs = rng.uniform(-1, 1, (batch_size, h, w, c)).astype(np.float32)
    return images, labels, my_imgs


def synthetic_random_words(bucket_size=10, words_per_bucket=1000, n_classes=52, seed=0):
    rng = np.random.default_rng(seed + 1)
    return [[list(map(int, rng.integers(0, n_classes, k + 1))) for _ in range(words_per_bucket)] for k in range(bucket_size)]
