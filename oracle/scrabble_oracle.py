"""CPU ORACLE for the ScrabbleGAN train step -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (UtkuKaradeniz/scrabble-gan) is pure Python on TensorFlow 2 /
gin / cv2, none of which exist in this image, and it ships no tests, golden vectors or fixtures
(SURVEY.md section 8c).  This file is therefore a *restatement* of the reference arithmetic on
torch-CPU primitives (fp64 or fp32), pinned only by the analytic known-answer tests and the naive
numpy loop restatements in tests/test_oracle_*.py.  Nothing in the product package may import it:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker.

Every function cites the reference file:line it restates (paths under /root/reference/src).
Tensors are NHWC like Keras; weights use the TF layouts:
    Conv2D kernel           [kh, kw, Cin, Cout]
    Conv2DTranspose kernel  [kh, kw, Cout, Cin]
    Dense kernel            [in, out]
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-3          # Keras BatchNormalization default epsilon (resnet_ops.py:14, net_architecture.py:42)
BN_MOMENTUM = 0.99     # Keras default momentum


# --------------------------------------------------------------------------------------------
# primitives (TF semantics)
# --------------------------------------------------------------------------------------------
RELU_HOOK = None       # tests only: callable(x) -> relu(x), sees every ReLU ACTIVATION site in call order (tests/step_fixture.py)


def relu(x: Tensor) -> Tensor:
    """tf.nn.relu / layers.ReLU at an activation site of the networks (resnet_ops.py:51,63,97,102; net_architecture.py:250,282)."""
    return RELU_HOOK(x) if RELU_HOOK is not None else torch.relu(x)


def _nchw(x: Tensor) -> Tensor:
    return x.permute(0, 3, 1, 2)


def _nhwc(x: Tensor) -> Tensor:
    return x.permute(0, 2, 3, 1)


def conv2d(x: Tensor, w: Tensor, b: Optional[Tensor] = None, padding: str = "same") -> Tensor:
    """layers.Conv2D stride 1 (resnet_ops.py:65,98,103,109; net_architecture.py:28-49,283).
    Cross-correlation; SAME pads (k-1)/2 on every side for odd k; VALID pads nothing."""
    kh, kw = w.shape[0], w.shape[1]
    pad = (kh // 2, kw // 2) if padding == "same" else (0, 0)
    y = F.conv2d(_nchw(x).contiguous(), w.permute(3, 2, 0, 1).contiguous(), b, stride=1, padding=pad)
    return _nhwc(y)


def conv2d_transpose(x: Tensor, w: Tensor, b: Optional[Tensor], stride: Tuple[int, int]) -> Tensor:
    """layers.Conv2DTranspose(padding='same') (resnet_ops.py:57,69) = adjoint of the SAME forward
    convolution from the output shape (Appendix A-3): k=3,s=2: out[2i+k] += x[i] w[k], cropped to
    2n; k=3,s=1: out[j+k-1]; k=1,s=2: out[2i] = x[i] w; bias on every output position."""
    kh, kw = w.shape[0], w.shape[1]
    sh, sw = stride
    B, H, W, _ = x.shape
    pads, outp = [], []
    for k, s in ((kh, sh), (kw, sw)):
        if s == 1:
            pads.append(k // 2)
            outp.append(0)
        else:
            pads.append(0)
            outp.append(s - 1 if k == 1 else 0)
    y = F.conv_transpose2d(_nchw(x).contiguous(), w.permute(3, 2, 0, 1).contiguous(), None, stride=(sh, sw),
                           padding=tuple(pads), output_padding=tuple(outp))
    y = y[:, :, : H * sh, : W * sw]
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return _nhwc(y)


def avg_pool2(x: Tensor) -> Tensor:
    """tf.nn.pool(AVG, 2x2, SAME, stride 2) on even dims = plain 2x2 mean (resnet_ops.py:106,113)."""
    return _nhwc(F.avg_pool2d(_nchw(x), 2))


MAXPOOL_HOOK = None    # tests only: callable(x, ph, pw) -> pooled tensor, sees every MaxPool2D site in call order


def max_pool(x: Tensor, ph: int, pw: int) -> Tensor:
    """layers.MaxPool2D(pool_size=(ph,pw)) VALID, stride = pool (arch_ops.py:47,58;
    net_architecture.py:29,32,38,47)."""
    if MAXPOOL_HOOK is not None:
        return MAXPOOL_HOOK(x, ph, pw)
    return _nhwc(F.max_pool2d(_nchw(x), (ph, pw)))


def batch_norm_train(x: Tensor, eps: float = BN_EPS) -> Tuple[Tensor, Tensor, Tensor]:
    """BatchNormalization in training mode: batch mean / biased variance over N,H,W
    (Appendix A-4).  Returns (x_hat, mean, biased_var)."""
    mean = x.mean(dim=(0, 1, 2))
    var = x.var(dim=(0, 1, 2), unbiased=False)
    return (x - mean) * torch.rsqrt(var + eps), mean, var


def l2_normalize(x: Tensor) -> Tensor:
    """tf.nn.l2_normalize over the whole tensor (Appendix A-8)."""
    return x * torch.rsqrt(torch.clamp((x * x).sum(), min=1e-12))


def spectral_norm(w: Tensor, u: Tensor, power_iteration: int = 1) -> Tensor:
    """arch_ops.py:98-126.  `u` is the N(0,1) [1,N] draw of :110, passed explicitly."""
    w_shape = w.shape
    w2 = w.reshape(-1, w_shape[-1])
    u_hat, v_hat = u.reshape(1, -1), None
    for _ in range(power_iteration):
        v_hat = l2_normalize(u_hat @ w2.t())            # :115-116
        u_hat = l2_normalize(v_hat @ w2)                # :118-119
    sigma = (v_hat @ w2) @ u_hat.t()                    # :121
    return (w2 / sigma).reshape(w_shape)                # :123-124


def nonlocal_block(x: Tensor, w_theta: Tensor, w_phi: Tensor, w_g: Tensor, w_o: Tensor,
                   sigma: Tensor) -> Tensor:
    """NonLocalBlock.call (arch_ops.py:32-67) as a pure function of explicit 1x1 kernels
    (SURVEY fact 3).  w_theta/w_phi [C, C/8], w_g [C, C/2], w_o [C/2, C]."""
    B, H, W, C = x.shape
    theta = (x @ w_theta).reshape(B, H * W, -1)                         # :38-41
    phi = max_pool(x @ w_phi, 2, 2).reshape(B, (H // 2) * (W // 2), -1)  # :44-48
    attn = torch.softmax(theta @ phi.transpose(1, 2), dim=-1)            # :51-52 (no 1/sqrt(d))
    g = max_pool(x @ w_g, 2, 2).reshape(B, (H // 2) * (W // 2), -1)      # :55-59
    attn_g = (attn @ g).reshape(B, H, W, -1)                             # :61-62
    return sigma * (attn_g @ w_o) + x                                    # :63-67


def filter_bank_seed(z0: Tensor, y: Tensor, table: Tensor) -> Tensor:
    """SpatialEmbedding lookup (arch_ops.py:89-90) + z0 contraction and the raw reshape/transpose
    seed layout (net_architecture.py:262-271).  z0 [B,32], y [B,L] int, table [V,32,8192]
    -> seed [B, 4, 4L, 512]."""
    B, L = y.shape
    se = table[y.long()]                                   # [B,L,32,8192]
    net = torch.einsum("bk,blkj->blj", z0, se)             # matmul(tile(z0), se) + squeeze
    net = net.reshape(B, 512, 4, 4, -1)                    # :269
    net = net.reshape(B, -1, 512, 4)                       # :270
    return net.permute(0, 3, 1, 2)                         # :271  -> [B,4,4L,512]


def conditional_batch_norm(x: Tensor, z: Tensor, w_gamma: Tensor, w_beta: Tensor,
                           stats: Optional[dict] = None, moving: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """ConditionalBatchNorm.call (resnet_ops.py:13-28): BN without affine, then per-sample
    gamma=Dense(z), beta=Dense(z) (gamma is NOT 1+gamma).  `moving` = (moving_mean, moving_variance)
    selects inference mode (`training=False`, data_utils.py:505-507): normalise with the moving statistics."""
    if moving is not None:
        x_hat = (x - moving[0]) * torch.rsqrt(moving[1] + BN_EPS)
    else:
        x_hat, mean, var = batch_norm_train(x)
    if stats is not None and moving is None:
        stats["mean"], stats["var"], stats["count"] = mean.detach(), var.detach(), x.numel() // x.shape[-1]
    gamma = (z @ w_gamma).reshape(-1, 1, 1, x.shape[-1])
    beta = (z @ w_beta).reshape(-1, 1, 1, x.shape[-1])
    return x_hat * gamma + beta


def ctc_batch_cost(y_true: Tensor, y_pred: Tensor, input_length: int, label_length: int) -> Tensor:
    """K.ctc_batch_cost (net_architecture.py:57-64; Appendix A-7): log(p+1e-7), then ctc_loss
    applies its own log-softmax; blank = last class.  Returns [B,1]."""
    B, T, C = y_pred.shape
    lp = torch.log_softmax(torch.log(y_pred + 1e-7), dim=-1)[:, :input_length]
    loss = F.ctc_loss(lp.transpose(0, 1), y_true.long()[:, :label_length],
                      torch.full((B,), input_length, dtype=torch.long),
                      torch.full((B,), label_length, dtype=torch.long),
                      blank=C - 1, reduction="none", zero_infinity=False)
    return loss.reshape(B, 1)


# --------------------------------------------------------------------------------------------
# blocks and nets (explicit weight dicts; names documented in DESIGN.md)
# --------------------------------------------------------------------------------------------
def resnet_block_down(x: Tensor, p: Dict[str, Tensor], pre: str, is_last: bool) -> Tensor:
    """ResNetBlockDown.call (resnet_ops.py:93-115)."""
    net = conv2d(relu(x), p[pre + ".conv1.w"], p[pre + ".conv1.b"])            # :97-99
    net = conv2d(relu(net), p[pre + ".conv2.w"], p[pre + ".conv2.b"])          # :102-104
    if not is_last:
        net = avg_pool2(net)                                                   # :105-106
    sc = conv2d(x, p[pre + ".short.w"], p[pre + ".short.b"])                   # :109-111
    if not is_last:
        sc = avg_pool2(sc)                                                     # :112-113
    return net + sc


def resnet_block_up(x: Tensor, z: Tensor, p: Dict[str, Tensor], pre: str, is_last: bool,
                    bn_stats: Optional[dict] = None, training: bool = True) -> Tensor:
    """ResNetBlockUp.call (resnet_ops.py:46-74).  training=False: both ConditionalBatchNorms use their moving statistics."""
    stride = (2, 1) if is_last else (2, 2)                                     # :54
    s1 = {} if bn_stats is not None else None
    s2 = {} if bn_stats is not None else None
    m1 = None if training else (p[pre + ".cbn1.mm"], p[pre + ".cbn1.mv"])
    m2 = None if training else (p[pre + ".cbn2.mm"], p[pre + ".cbn2.mv"])
    net = relu(conditional_batch_norm(x, z, p[pre + ".cbn1.gamma.w"], p[pre + ".cbn1.beta.w"], s1, m1))
    net = conv2d_transpose(net, p[pre + ".convT.w"], p[pre + ".convT.b"], stride)
    net = relu(conditional_batch_norm(net, z, p[pre + ".cbn2.gamma.w"], p[pre + ".cbn2.beta.w"], s2, m2))
    net = conv2d(net, p[pre + ".conv.w"], p[pre + ".conv.b"])
    sc = conv2d_transpose(x, p[pre + ".short.w"], p[pre + ".short.b"], stride)
    if bn_stats is not None and training:
        bn_stats[pre + ".cbn1"], bn_stats[pre + ".cbn2"] = s1, s2
    return net + sc


DISC_CHANNELS = [64, 512, 1024, 1024]       # get_in_out_channels_disc (net_architecture.py:576-586)
GEN_CHANNELS = [(512, 256), (256, 128), (128, 64)]   # get_in_out_channels_gen (:565-573)


def disc_trunk(x: Tensor, p: Dict[str, Tensor], nl: Optional[Dict[str, Tensor]], attn_blocks: str,
               block_fmt: str = "B{}") -> Tensor:
    """Shared body of make_discriminator / make_style_promoter / G's style encoder
    (net_architecture.py:241-250, 332-341, 391-400): 4 ResNetBlockDown, NonLocalBlock after
    the blocks whose name is a substring of `attn_blocks`, then ReLU + GlobalAveragePooling."""
    net = x
    for i in range(4):
        name = block_fmt.format(i + 1)
        net = resnet_block_down(net, p, name, is_last=(i == 3))
        if name in attn_blocks:
            net = nonlocal_block(net, nl["theta"], nl["phi"], nl["g"], nl["o"], p["NL_" + name + ".sigma"])
    return relu(net).mean(dim=(1, 2))


def discriminator(x: Tensor, p: Dict[str, Tensor], nl: Optional[Dict[str, Tensor]] = None,
                  attn_blocks: str = "B1") -> Tensor:
    """make_discriminator / make_style_promoter forward (net_architecture.py:299-355, 358-414)."""
    return disc_trunk(x, p, nl, attn_blocks) @ p["dense.w"]


def conv2d_stride2_same(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """layers.Conv2D(3x3, strides=(2,2), padding='same') on even extents (net_architecture.py:425-443; Appendix A-1):
    pad_before 0, pad_after 1."""
    xp = F.pad(_nchw(x), (0, 1, 0, 1))
    return _nhwc(F.conv2d(xp, w.permute(3, 2, 0, 1).contiguous(), b, stride=2, padding=0))


MYDISC_FILTERS = [16, 32, 64, 128]


def my_discriminator(x: Tensor, p: Dict[str, Tensor], nl: Dict[str, Tensor]) -> Tensor:
    """make_my_discriminator forward (net_architecture.py:417-462): 4 x [Conv2D 3x3 s2 same -> LeakyReLU(0.3)], a
    NonLocalBlock after the second (C = 32), a second LeakyReLU after the fourth (:446), GAP, Dense(1, no bias)."""
    net = x
    for i in range(4):
        net = F.leaky_relu(conv2d_stride2_same(net, p["conv%d.w" % (i + 1)], p["conv%d.b" % (i + 1)]), 0.3)
        if i == 1:
            net = nonlocal_block(net, nl["theta"], nl["phi"], nl["g"], nl["o"], p["NL_B1.sigma"])
    net = F.leaky_relu(net, 0.3)
    return net.mean(dim=(1, 2)) @ p["dense.w"]


def init_my_discriminator(gen: torch.Generator, dtype=torch.float64, colors=1) -> Dict[str, Tensor]:
    p: Dict[str, Tensor] = {}
    cin = colors
    for i, co in enumerate(MYDISC_FILTERS):
        p["conv%d.w" % (i + 1)] = orthogonal((3, 3, cin, co), gen, dtype)
        p["conv%d.b" % (i + 1)] = torch.zeros(co, dtype=dtype)
        cin = co
    p["NL_B1.sigma"] = torch.zeros((), dtype=dtype)
    p["dense.w"] = orthogonal((cin, 1), gen, dtype)
    return p


def generator(style: Tensor, y: Tensor, p: Dict[str, Tensor], nl_style: Dict[str, Tensor],
              nl_up: Optional[Dict[str, Tensor]], attn_blocks: str = "B3",
              bn_stats: Optional[dict] = None, training: bool = True) -> Tensor:
    """make_generator forward (net_architecture.py:182-296).  training=True: batch statistics in the seven
    BatchNorms (and `bn_stats` collects them for the moving-average update); training=False =
    `generator([style, labels], training=False)` of generate_and_save_images (data_utils.py:505-507): every
    BatchNormalization normalises with its moving mean / variance (Appendix A-4)."""
    if style.dim() == 3:                                    # Appendix C-8
        style = style.unsqueeze(-1)
    h = disc_trunk(style, p, nl_style, "B_style1", block_fmt="B_style{}")     # :241-250
    z = h @ p["zdense.w"]                                                      # :251-257
    z0, z1, z2, z3 = torch.split(z, 32, dim=1)                                 # :260-262
    net = filter_bank_seed(z0, y, p["filter_bank"])                            # :265-271
    for i, zi in enumerate((z1, z2, z3)):
        name = "B{}".format(i + 1)
        net = resnet_block_up(net, zi, p, name, is_last=(i == 2), bn_stats=bn_stats, training=training)   # :274-277
        if name in attn_blocks:                                                # :278-279
            net = nonlocal_block(net, nl_up["theta"], nl_up["phi"], nl_up["g"], nl_up["o"],
                                 p["NL_" + name + ".sigma"])
    if not training:
        x_hat = (net - p["bn.mm"]) * torch.rsqrt(p["bn.mv"] + BN_EPS)
        net = torch.relu(x_hat * p["bn.gamma"] + p["bn.beta"])
        return torch.tanh(conv2d(net, p["final.w"], p["final.b"]))
    x_hat, mean, var = batch_norm_train(net)                                   # :281
    if bn_stats is not None:
        bn_stats["bn"] = {"mean": mean.detach(), "var": var.detach(), "count": net.numel() // net.shape[-1]}
    net = relu(x_hat * p["bn.gamma"] + p["bn.beta"])                           # :282
    net = conv2d(net, p["final.w"], p["final.b"])                              # :283-287
    return torch.tanh(net)                                                     # :289


def recognizer_probs(x: Tensor, p: Dict[str, Tensor], bn_training: bool = False) -> Tensor:
    """make_recognizer conv stack + per-frame softmax (net_architecture.py:28-55).
    bn_training=False is SURVEY fact 4 (frozen layer => inference-mode BN)."""
    def bn(t, pre):
        if bn_training:
            x_hat, _, _ = batch_norm_train(t)
        else:
            x_hat = (t - p[pre + ".mm"]) * torch.rsqrt(p[pre + ".mv"] + BN_EPS)
        return x_hat * p[pre + ".gamma"] + p[pre + ".beta"]
    net = max_pool(relu(conv2d(x, p["conv1.w"], p["conv1.b"])), 2, 2)
    net = max_pool(relu(conv2d(net, p["conv2.w"], p["conv2.b"])), 2, 2)
    net = relu(conv2d(net, p["conv3.w"], p["conv3.b"]))
    net = max_pool(relu(conv2d(net, p["conv4.w"], p["conv4.b"])), 2, 1)
    net = bn(relu(conv2d(net, p["conv5.w"], p["conv5.b"])), "bn5")
    net = max_pool(bn(relu(conv2d(net, p["conv6.w"], p["conv6.b"])), "bn6"), 2, 1)
    net = relu(conv2d(net, p["conv7.w"], p["conv7.b"], padding="valid"))
    net = net.squeeze(1)                                                        # :52
    return torch.softmax(net @ p["dense.w"] + p["dense.b"], dim=-1)             # :55


def recognizer(x: Tensor, labels: Tensor, input_length: int, label_length: int,
               p: Dict[str, Tensor], bn_training: bool = False) -> Tensor:
    """make_recognizer: the model output IS the CTC cost [B,1] (net_architecture.py:57-74)."""
    return ctc_batch_cost(labels, recognizer_probs(x, p, bn_training), input_length, label_length)


MYREC_FILTERS = [16, 32, 48, 64, 80, 128, 144]            # net_architecture.py:102-136
MYREC_POOLS = [(2, 2), (2, 2), (2, 1), (2, 1), (2, 1), None, None]


def lstm_direction(x: Tensor, W: Tensor, U: Tensor, b: Tensor, reverse: bool) -> Tensor:
    """One direction of layers.LSTM(units, return_sequences=True) (net_architecture.py:146): Keras gate order
    i, f, c~, o; sigmoid recurrent activation, tanh activation; zero initial state.  x [B,T,I] -> [B,T,H]."""
    B, T, _ = x.shape
    H = U.shape[0]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        z = x[:, t] @ W + h @ U + b
        i, f, g, o = torch.sigmoid(z[:, :H]), torch.sigmoid(z[:, H:2 * H]), torch.tanh(z[:, 2 * H:3 * H]), torch.sigmoid(z[:, 3 * H:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def bilstm(x: Tensor, p: Dict[str, Tensor], pre: str, masks=None) -> Tensor:
    """Bidirectional(LSTM(256, return_sequences=True, dropout=0.5)), merge_mode concat.  `masks` = (fw, bw) input-dropout
    masks [B,I] already scaled by 1/(1-rate), one per direction, shared by all timesteps (Keras implementation 2)."""
    outs = []
    for k, d in enumerate(("fw", "bw")):
        xm = x if masks is None else x * masks[k].unsqueeze(1)
        outs.append(lstm_direction(xm, p[pre + "." + d + ".W"], p[pre + "." + d + ".U"], p[pre + "." + d + ".b"], reverse=(d == "bw")))
    return torch.cat(outs, dim=-1)


def my_recognizer_probs(x: Tensor, p: Dict[str, Tensor], bn_training: bool = False, masks: Optional[dict] = None) -> Tensor:
    """make_my_recognizer conv + BiLSTM stack + per-frame softmax (net_architecture.py:102-154).  `masks` holds the
    dropout masks of a training call (already scaled): 'drop3'..'drop7' (activation-shaped, rate 0.2, :112-135),
    'lstm{l}' = (fw, bw) [B,I] (rate 0.5), 'drop_out' [B,T,512] (:153); None = inference (no dropout)."""
    net = x
    for i, (co, pool) in enumerate(zip(MYREC_FILTERS, MYREC_POOLS)):
        k = i + 1
        if masks is not None and k >= 3:
            net = net * masks["drop%d" % k]
        net = conv2d(net, p["conv%d.w" % k], p["conv%d.b" % k])
        if bn_training:
            x_hat, _, _ = batch_norm_train(net)
        else:
            x_hat = (net - p["bn%d.mm" % k]) * torch.rsqrt(p["bn%d.mv" % k] + BN_EPS)
        net = F.leaky_relu(x_hat * p["bn%d.gamma" % k] + p["bn%d.beta" % k], 0.01)
        if pool is not None:
            net = max_pool(net, *pool)
    net = net.squeeze(1)                                                    # :143
    for l in range(5):
        net = bilstm(net, p, "lstm%d" % (l + 1), None if masks is None else masks["lstm%d" % (l + 1)])
    if masks is not None:
        net = net * masks["drop_out"]
    return torch.softmax(net @ p["dense.w"] + p["dense.b"], dim=-1)          # :154


def my_recognizer(x, labels, input_length, label_length, p, bn_training=False, masks=None) -> Tensor:
    return ctc_batch_cost(labels, my_recognizer_probs(x, p, bn_training, masks), input_length, label_length)


def init_my_recognizer(gen: torch.Generator, dtype=torch.float64, classes=53, H=256) -> Dict[str, Tensor]:
    p: Dict[str, Tensor] = {}
    cin = 1
    for i, co in enumerate(MYREC_FILTERS):
        k = i + 1
        p["conv%d.w" % k] = glorot_uniform((3, 3, cin, co), gen, dtype)
        p["conv%d.b" % k] = torch.zeros(co, dtype=dtype)
        p["bn%d.gamma" % k], p["bn%d.beta" % k] = torch.ones(co, dtype=dtype), torch.zeros(co, dtype=dtype)
        p["bn%d.mm" % k], p["bn%d.mv" % k] = torch.zeros(co, dtype=dtype), torch.ones(co, dtype=dtype)
        cin = co
    for l in range(5):
        for d in ("fw", "bw"):
            pre = "lstm%d.%s" % (l + 1, d)
            p[pre + ".W"] = glorot_uniform((cin, 4 * H), gen, dtype)
            p[pre + ".U"] = orthogonal((H, 4 * H), gen, dtype)
            b = torch.zeros(4 * H, dtype=dtype)
            b[H:2 * H] = 1.0                                                 # unit_forget_bias
            p[pre + ".b"] = b
        cin = 2 * H
    p["dense.w"] = glorot_uniform((2 * H, classes), gen, dtype)
    p["dense.b"] = torch.zeros(classes, dtype=dtype)
    return p


# --------------------------------------------------------------------------------------------
# losses (net_loss.py) and gradient balancing (data_utils.py:476-490)
# --------------------------------------------------------------------------------------------
def hinge(d_real, d_fake, s_real, s_fake, _ignored=None):
    """net_loss.py:38-54; 5th positional argument accepted and ignored (Appendix C-1)."""
    d_loss_real = torch.relu(1.0 - d_real)
    d_loss_fake = torch.relu(1.0 + d_fake)
    s_loss_real = torch.relu(1.0 - s_real)
    s_loss_fake = torch.relu(1.0 + s_fake)
    g_loss = -(d_fake + s_fake)
    return (d_loss_real + d_loss_fake, d_loss_real, d_loss_fake, g_loss,
            s_loss_real + s_loss_fake, s_loss_real, s_loss_fake)


def _sce(logits: Tensor, label: float) -> Tensor:
    """tf.nn.sigmoid_cross_entropy_with_logits (Appendix A-12)."""
    return torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-logits.abs()))


def not_saturating(d_real, d_fake, s_styleimgs, s_trainingimgs, s_fake):
    """net_loss.py:4-35, argument meaning as declared there (call-site swap: Appendix C-2)."""
    d_loss_real = _sce(d_real, 1.0)
    d_loss_fake = _sce(d_fake, 0.0)
    s_style = _sce(s_styleimgs, 1.0)
    s_iam = _sce(s_trainingimgs, 0.0)
    g_loss = _sce(d_fake, 1.0) + _sce(s_fake, 1.0)
    return d_loss_real + d_loss_fake, d_loss_real, d_loss_fake, g_loss, s_style + s_iam, s_style, s_iam


def apply_gradient_balancing(r_fake: Tensor, g_loss: Tensor, alpha: float = 1):
    """data_utils.py:476-490 (population std, no stop-gradient)."""
    r_std = r_fake.std(unbiased=False)
    g_std = g_loss.std(unbiased=False)
    r_bal = alpha * ((g_std / r_std) * r_fake)
    return g_loss + r_bal, r_bal, alpha, r_std, g_std


# --------------------------------------------------------------------------------------------
# optimizers (main.py:25-35; Keras semantics, Appendix A-5)
# --------------------------------------------------------------------------------------------
def adam_update(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: dict, lr: float,
                beta_1: float, beta_2: float, eps: float = 1e-7) -> None:
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    lr_t = lr * math.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
    for k, g in grads.items():
        if g is None:
            continue
        m = state.setdefault("m." + k, torch.zeros_like(params[k]))
        v = state.setdefault("v." + k, torch.zeros_like(params[k]))
        m.mul_(beta_1).add_(g, alpha=1.0 - beta_1)
        v.mul_(beta_2).addcmul_(g, g, value=1.0 - beta_2)
        params[k] = params[k] - lr_t * m / (v.sqrt() + eps)


def rmsprop_update(params, grads, state, lr: float, rho: float = 0.9, eps: float = 1e-7) -> None:
    """tf.keras.optimizers.RMSprop(lr) of main.py:29-30 (rho 0.9, momentum 0, epsilon 1e-7, not centered), TF 2.1 dense
    path: rms = rho rms + (1-rho) g^2 ; var -= lr g / (sqrt(rms) + epsilon)  -- epsilon is added OUTSIDE the root."""
    for k, g in grads.items():
        if g is None:
            continue
        ms = state.setdefault("ms." + k, torch.zeros_like(params[k]))
        ms.mul_(rho).addcmul_(g, g, value=1.0 - rho)
        params[k] = params[k] - lr * g / (torch.sqrt(ms) + eps)


# --------------------------------------------------------------------------------------------
# initialisers (Appendix A-6) and parameter factories
# --------------------------------------------------------------------------------------------
def orthogonal(shape: Sequence[int], gen: torch.Generator, dtype=torch.float64) -> Tensor:
    rows = int(math.prod(shape[:-1]))
    cols = int(shape[-1])
    flat = (cols, rows) if rows < cols else (rows, cols)
    a = torch.randn(flat, generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if rows < cols:
        q = q.t()
    return q.reshape(tuple(shape)).to(dtype).contiguous()


def glorot_uniform(shape: Sequence[int], gen: torch.Generator, dtype=torch.float64) -> Tensor:
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(math.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(tuple(shape), generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


TRAINABLE_EXCLUDE = (".mm", ".mv")     # BN moving statistics are non-trainable


def _disc_block_params(p, pre_fmt, gen, dtype, colors=1):
    cin = colors
    for i, cout in enumerate(DISC_CHANNELS):
        pre = pre_fmt.format(i + 1)
        p[pre + ".conv1.w"] = orthogonal((3, 3, cin, cout), gen, dtype)
        p[pre + ".conv1.b"] = torch.zeros(cout, dtype=dtype)
        p[pre + ".conv2.w"] = orthogonal((3, 3, cout, cout), gen, dtype)
        p[pre + ".conv2.b"] = torch.zeros(cout, dtype=dtype)
        p[pre + ".short.w"] = orthogonal((1, 1, cin, cout), gen, dtype)
        p[pre + ".short.b"] = torch.zeros(cout, dtype=dtype)
        cin = cout


def init_discriminator(gen: torch.Generator, dtype=torch.float64, attn_blocks="B1") -> Dict[str, Tensor]:
    p: Dict[str, Tensor] = {}
    _disc_block_params(p, "B{}", gen, dtype)
    for i in range(4):
        if "B{}".format(i + 1) in attn_blocks:
            p["NL_B{}.sigma".format(i + 1)] = torch.zeros((), dtype=dtype)
    p["dense.w"] = orthogonal((1024, 1), gen, dtype)
    return p


def init_generator(gen: torch.Generator, dtype=torch.float64, vocab=52, attn_blocks="B3") -> Dict[str, Tensor]:
    p: Dict[str, Tensor] = {}
    _disc_block_params(p, "B_style{}", gen, dtype)
    p["NL_B_style1.sigma"] = torch.zeros((), dtype=dtype)
    p["zdense.w"] = orthogonal((1024, 128), gen, dtype)
    p["filter_bank"] = glorot_uniform((vocab, 32, 8192), gen, dtype)
    for i, (cin, cout) in enumerate(GEN_CHANNELS):
        pre = "B{}".format(i + 1)
        p[pre + ".cbn1.gamma.w"] = orthogonal((32, cin), gen, dtype)
        p[pre + ".cbn1.beta.w"] = orthogonal((32, cin), gen, dtype)
        p[pre + ".convT.w"] = orthogonal((3, 3, cout, cin), gen, dtype)
        p[pre + ".convT.b"] = torch.zeros(cout, dtype=dtype)
        p[pre + ".cbn2.gamma.w"] = orthogonal((32, cout), gen, dtype)
        p[pre + ".cbn2.beta.w"] = orthogonal((32, cout), gen, dtype)
        p[pre + ".conv.w"] = orthogonal((3, 3, cout, cout), gen, dtype)
        p[pre + ".conv.b"] = torch.zeros(cout, dtype=dtype)
        p[pre + ".short.w"] = orthogonal((1, 1, cout, cin), gen, dtype)
        p[pre + ".short.b"] = torch.zeros(cout, dtype=dtype)
        for cb, c in ((".cbn1", cin), (".cbn2", cout)):
            p[pre + cb + ".mm"] = torch.zeros(c, dtype=dtype)
            p[pre + cb + ".mv"] = torch.ones(c, dtype=dtype)
        if pre in attn_blocks:
            p["NL_" + pre + ".sigma"] = torch.zeros((), dtype=dtype)
    p["bn.gamma"] = torch.ones(64, dtype=dtype)
    p["bn.beta"] = torch.zeros(64, dtype=dtype)
    p["bn.mm"] = torch.zeros(64, dtype=dtype)
    p["bn.mv"] = torch.ones(64, dtype=dtype)
    p["final.w"] = orthogonal((3, 3, 64, 1), gen, dtype)
    p["final.b"] = torch.zeros(1, dtype=dtype)
    return p


REC_CONVS = [(3, 1, 64), (3, 64, 128), (3, 128, 256), (3, 256, 256), (3, 256, 512), (3, 512, 512), (2, 512, 512)]


def init_recognizer(gen: torch.Generator, dtype=torch.float64, classes=53) -> Dict[str, Tensor]:
    p: Dict[str, Tensor] = {}
    for i, (k, cin, cout) in enumerate(REC_CONVS):
        p["conv{}.w".format(i + 1)] = glorot_uniform((k, k, cin, cout), gen, dtype)
        p["conv{}.b".format(i + 1)] = torch.zeros(cout, dtype=dtype)
    for pre in ("bn5", "bn6"):
        p[pre + ".gamma"] = torch.ones(512, dtype=dtype)
        p[pre + ".beta"] = torch.zeros(512, dtype=dtype)
        p[pre + ".mm"] = torch.zeros(512, dtype=dtype)
        p[pre + ".mv"] = torch.ones(512, dtype=dtype)
    p["dense.w"] = glorot_uniform((512, classes), gen, dtype)
    p["dense.b"] = torch.zeros(classes, dtype=dtype)
    return p


def init_nonlocal(C: int, gen: torch.Generator, dtype=torch.float64) -> Dict[str, Tensor]:
    """The four orthogonal 1x1 kernels NonLocalBlock.call constructs (arch_ops.py:38-65)."""
    return {"theta": orthogonal((C, C // 8), gen, dtype), "phi": orthogonal((C, C // 8), gen, dtype),
            "g": orthogonal((C, C // 2), gen, dtype), "o": orthogonal((C // 2, C), gen, dtype)}


def is_trainable(name: str) -> bool:
    return not name.endswith(TRAINABLE_EXCLUDE)


# --------------------------------------------------------------------------------------------
# the hot path: one train_step (data_utils.py:358-473), data-flow contract of SURVEY Appendix E
# --------------------------------------------------------------------------------------------
def train_step(images: Tensor, labels: Tensor, style: Tensor, fake_labels: Tensor,
               G: Dict[str, Tensor], D: Dict[str, Tensor], S: Dict[str, Tensor], R: Dict[str, Tensor],
               nl: Dict[str, Dict[str, Tensor]], opt: dict, loss_fn=hinge, apply_gradient_balance=False,
               lr=(2e-4, 2e-4, 2e-4, 2e-4), beta_1=0.0, beta_2=0.999, rmsprop=False,
               update_generator=True, r_bn_training=False):
    """One optimisation step on explicit weights/inputs.

    `nl` holds the per-call NonLocalBlock kernels keyed by the pass name:
    'G.style', 'G.up', 'D.fake', 'D.real', 'S.fake', 'S.style', 'S.real'.
    `opt` holds the four optimizer states {'G':{}, 'D':{}, 'R':{}, 'S':{}} and is updated in place.
    The host draws of :385-392 (random bucket, fake labels) are inputs, not re-drawn here.
    Returns (16 scalars in the order of :470-473, dict of per-net gradients, fake images);
    G/D/S/R are updated in place (new tensors stored under the same keys)."""
    nets = {"G": G, "D": D, "S": S, "R": R}
    leaves = {}
    for n, P in nets.items():
        for k in P:
            if is_trainable(k):
                P[k] = P[k].detach().clone().requires_grad_(True)
                leaves[(n, k)] = P[k]
    B = images.shape[0]
    L_r, L_f = labels.shape[1], fake_labels.shape[1]
    bn_stats: dict = {}
    x_f = generator(style, fake_labels, G, nl["G.style"], nl.get("G.up"), bn_stats=bn_stats)   # :401
    d_f = discriminator(x_f, D, nl["D.fake"])
    s_f = discriminator(x_f, S, nl["S.fake"])
    r_f = recognizer(x_f, fake_labels, 4 * L_f - 1, L_f, R, r_bn_training)
    d_r = discriminator(images, D, nl["D.real"])                                               # :406
    style4 = style if style.dim() == 4 else style.unsqueeze(-1)
    s_my = discriminator(style4, S, nl["S.style"])                                             # :409
    s_r = discriminator(images, S, nl["S.real"])                                               # :410
    r_r = recognizer(images, labels, 4 * L_r - 1, L_r, R, r_bn_training)                       # :414
    d_loss, d_lr, d_lf, g_loss, s_loss, s_a, s_b = loss_fn(d_r, d_f, s_my, s_f, s_r)           # :418
    g_bal, r_bal, alpha, r_std, g_std = apply_gradient_balancing(r_f, g_loss, alpha=1)         # :421
    g_added = g_loss + r_f                                                                     # :423
    g_final = g_bal if apply_gradient_balance else g_added                                     # :424-427

    def grads_of(target, net):
        names = [k for k in nets[net] if is_trainable(k)]
        gs = torch.autograd.grad(target.sum(), [leaves[(net, k)] for k in names],
                                 retain_graph=True, allow_unused=True)
        return dict(zip(names, gs))

    all_grads = {"D": grads_of(d_loss, "D"), "R": grads_of(r_r, "R"), "S": grads_of(s_loss, "S")}
    if update_generator:
        all_grads["G"] = grads_of(g_final, "G")
    scalars = (r_f.mean().item(), r_r.mean().item(), r_bal.mean().item(), g_loss.mean().item(),
               g_added.mean().item(), g_bal.mean().item(), d_loss.mean().item(), d_lr.mean().item(),
               d_lf.mean().item(), g_final.mean().item(), alpha, r_std.item(), g_std.item(),
               s_loss.mean().item(), s_a.mean().item(), s_b.mean().item())
    with torch.no_grad():
        for n, P in nets.items():
            for k in list(P):
                P[k] = P[k].detach()
        for n, i in (("D", 1), ("R", 2), ("S", 3), ("G", 0)):
            if n not in all_grads:
                continue
            if n == "R" and rmsprop:
                rmsprop_update(nets[n], all_grads[n], opt[n], lr[i])
            else:
                adam_update(nets[n], all_grads[n], opt[n], lr[i], beta_1, beta_2)
        # G's BN moving statistics advance once per step (Appendix A-4, E): fused BN updates the
        # moving variance with the Bessel-corrected batch variance.
        for pre, st in bn_stats.items():
            n = st["count"]
            mm_k, mv_k = pre + ".mm", pre + ".mv"
            if mm_k in G:
                G[mm_k] = BN_MOMENTUM * G[mm_k] + (1 - BN_MOMENTUM) * st["mean"]
                G[mv_k] = BN_MOMENTUM * G[mv_k] + (1 - BN_MOMENTUM) * st["var"] * (n / max(n - 1, 1))
    return scalars, all_grads, x_f.detach()
