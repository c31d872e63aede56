"""fp8 (OCP e4m3) matrix-core convolutions -- BASELINE config c5, first slice -- against the CPU oracle.

The forward and data-grad launches of the >= 128-channel convolutions quantise both MFMA operands per tensor:
q = e4m3(clamp(v * 448 / amax, +-448)), amax = max|tensor| (device scalar), fp32 accumulation, result * amax_a amax_w / 448^2.
Two bars per op: (1) against the oracle evaluated on operands quantised THE SAME WAY (torch.float8_e4m3fn, same fp32
scale arithmetic) the kernel must be fp32-accurate, tol 5e-5 of max|ref| -- the only deviation of the path is the operand
quantisation; (2) against the exact fp64 oracle the error is e4m3 rounding noise (2^-4 relative per operand, random signs
over K >= 1152 terms): <= 6e-2 of max|ref|."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import scrabble_oracle as O  # noqa: E402  (checker only)
from tests.test_ops_gpu import close, g32, rnd  # noqa: E402


@pytest.fixture()
def gen():
    return torch.Generator().manual_seed(808)


@pytest.fixture()
def fp8_mode():
    from scrabble_gan_amd import ops
    ops.set_conv_dtype("fp8")
    yield ops
    ops.set_conv_dtype("f32")


def q8(t):
    """What sg_amax_f32 + sg_cvt_fp8 / sg_pack_filter_fp8 do, on the host: -> (dequantised fp64 values, fp32 amax)."""
    t32 = t.detach().float()
    amax = t32.abs().max()
    s = (torch.tensor(448.0) / amax).float() if amax > 0 else torch.tensor(1.0)
    q = (t32 * s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float64)
    return q, amax


CASES = [
    # B, H, W, Cin, Cout, k
    (2, 8, 12, 128, 256, 3),
    (1, 16, 24, 256, 512, 3),      # several M tiles, 2 N tiles, 2 k-chunks per tap
    (3, 8, 40, 512, 256, 1),       # 1x1 shortcut shape
    (5, 7, 5, 128, 256, 3),        # odd spatial dims: M = 175, one partial tile
    (16, 8, 40, 256, 256, 3),      # enough tiles for the reduction split (atomic partial tiles)
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", CASES)
def test_conv2d_fp8_fwd_dgrad(dev, gen, fp8_mode, B, H, W, Cin, Cout, k):
    ops = fp8_mode
    x = rnd(gen, B, H, W, Cin)
    w = rnd(gen, k, k, Cin, Cout) / math.sqrt(k * k * Cin)
    b = rnd(gen, Cout)
    dy = rnd(gen, B, H, W, Cout)
    xg, wg, bg, dyg = (g32(t, dev) for t in (x, w, b, dy))
    y = ops.conv2d_fwd(xg, wg, bg, relu_in=True)
    qx, ax = q8(torch.relu(x.float()))          # the kernel's amax is taken over x itself (>= that of relu(x)):
    t32 = x.float()
    ax = t32.abs().max()
    sx = (torch.tensor(448.0) / ax).float()
    qx = (torch.relu(t32) * sx).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float64)
    qw, aw = q8(w)
    osc = (ax * aw * torch.tensor(1.0 / (448.0 * 448.0))).double()
    ref_q = O.conv2d(qx, qw, None) * osc + b
    close(y, ref_q, 5e-5, "fwd vs oracle on the same fp8-quantised operands")
    close(y, O.conv2d(torch.relu(x), w, b), 6e-2, "fwd vs exact oracle")
    if Cin % 256 == 0:          # data-grad: reduction over Cout (% 128), output channels Cin (% 256)
        qd, ad = q8(dy)
        xr = x.clone().requires_grad_(True)
        O.conv2d(xr, qw, None).backward(qd)
        osd = (ad * aw * torch.tensor(1.0 / (448.0 * 448.0))).double()
        base = rnd(gen, B, H, W, Cin)
        dx = ops.conv2d_bwd_data(dyg, wg, (H, W), mask=xg, out=g32(base, dev), accum=True)
        close(dx, xr.grad * osd * (x > 0) + base, 5e-5, "dgrad (mask, accum) vs oracle on the same fp8-quantised operands")
        xe = x.clone().requires_grad_(True)
        O.conv2d(xe, w, None).backward(dy)
        close(ops.conv2d_bwd_data(dyg, wg, (H, W)), xe.grad, 6e-2, "dgrad vs exact oracle")


def q5(t32, amax):
    """sg_cvt_fp8_grad's e5m2 operand on the host: e5m2(clamp(v * 57344 / amax)) as fp64 (v already carries its row factor)."""
    s = (torch.tensor(57344.0) / amax.float()).float() if amax > 0 else torch.tensor(1.0)
    return (t32.float() * s).clamp(-57344.0, 57344.0).to(torch.float8_e5m2).to(torch.float64)


WGRAD_CASES = [
    # B, H, W, Cin, Cout, k, scaled
    (2, 8, 12, 256, 256, 3, False),
    (3, 8, 40, 512, 256, 1, True),      # 1x1 shortcut shape, two c-tiles, per-sample factors
    (5, 7, 5, 256, 512, 3, True),       # odd spatial dims: M = 175 pixels, ragged last 128-pixel tile, two n-tiles
    (40, 4, 20, 256, 256, 3, False),    # 3200 pixels: several pixel chunks per (tap, tile) -> atomically summed partial dW
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,scaled", WGRAD_CASES)
def test_conv2d_fp8_wgrad(dev, gen, fp8_mode, B, H, W, Cin, Cout, k, scaled):
    """fp8 weight gradient (e4m3 activations x e5m2 gradients, v_mfma_scale_f32_32x32x64_f8f6f4, transposing LDS reads) with the
    fused bias gradient and the per-sample factors of the shared sweep: 5e-5 of max|ref| against the oracle on operands
    quantised the same way (torch.float8_e4m3fn / float8_e5m2, same fp32 scale arithmetic), 0.15 against the exact oracle
    (e5m2 keeps 2 mantissa bits: up to 2^-3 relative per gradient element, e4m3 2^-4 per activation, random signs over as few as
    175 pixels; the largest of 1.2 M elements measured 9.3e-2 of max|ref| on the 175-pixel case)."""
    ops = fp8_mode
    x = rnd(gen, B, H, W, Cin)
    dy = rnd(gen, B, H, W, Cout)
    sc = (torch.rand(B, generator=gen, dtype=torch.float64) * 2 - 0.5) if scaled else None
    xg, dyg = g32(x, dev), g32(dy, dev)
    scg = None if sc is None else g32(sc, dev)
    dw0 = rnd(gen, k, k, Cin, Cout)
    dw, db = g32(dw0, dev), torch.zeros(Cout, device=dev)
    ops.conv2d_bwd_weight(xg, dyg, dw, relu_in=True, db=db, sample_scale=scg)
    x32 = x.float()
    ax = x32.abs().max()
    qx = (torch.relu(x32) * (torch.tensor(448.0) / ax).float()).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float64)
    dys32 = dy.float() if sc is None else dy.float() * sc.float().view(B, 1, 1, 1)
    aq = dys32.abs().max()
    qd = q5(dys32, aq)
    w = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
    O.conv2d(qx, w, None).backward(qd)
    scale = (ax.double() / 448.0) * (aq.double() / 57344.0)
    close(dw, w.grad * scale + dw0, 5e-5, "fp8 dW (+=) vs oracle on the same quantised operands")
    close(db, dys32.double().sum(dim=(0, 1, 2)), 5e-5, "fused bias gradient (fp32 sums of the scaled fp32 values)")
    we = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
    O.conv2d(torch.relu(x), we, None).backward(dys32.double())
    close(dw - g32(dw0, dev), we.grad, 0.15, "fp8 dW vs exact oracle")
    # the data-grad launch of the same gradient reads the e4m3 copy that sweep wrote beside the e5m2 one
    got4, a4 = ops.fp8_of(dyg)
    want4 = (dy.float() * (torch.tensor(448.0) / dy.float().abs().max()).float()).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(got4.cpu(), want4) and a4.item() == dy.float().abs().max().item()


def test_fp8_gradient_operand_matches_torch_e5m2(dev, gen, fp8_mode):
    """sg_amax2_f32 + sg_cvt_fp8_grad bit for bit against torch.float8_e5m2 / float8_e4m3fn: scaled rows, values down in
    e5m2's subnormal range and at the saturating end."""
    ops = fp8_mode
    B, C = 6, 64
    dy = torch.cat([rnd(gen, B, 4, 5, C) * 3, rnd(gen, B, 4, 5, C) * 1e-4], 1).float().contiguous()      # [B, 8, 5, C]
    sc = torch.tensor([1.0, -0.5, 2.0, 0.25, 1e-3, 0.0])
    dyg, scg = dy.to(dev), sc.to(dev)
    out5, a5, colsum = ops.grad_operand_fp8(dyg, scg, True)
    dys = dy * sc.view(B, 1, 1, 1)
    assert a5.item() == dys.abs().max().item()
    want5 = (dys * (torch.tensor(57344.0) / dys.abs().max()).float()).clamp(-57344, 57344).to(torch.float8_e5m2).view(torch.uint8)
    g5, w5 = out5.cpu(), want5
    assert torch.equal(g5 & 0x7f, w5 & 0x7f) and torch.equal((g5 ^ w5)[(w5 & 0x7f) != 0], torch.zeros_like(g5)[(w5 & 0x7f) != 0])   # (sign of a zero may differ)
    close(colsum, dys.double().sum(dim=(0, 1, 2)), 2e-5, "column sums")


def test_fp8_conversion_matches_torch_e4m3(dev, gen, fp8_mode):
    """sg_amax_f32 + sg_cvt_fp8 bit for bit against torch.float8_e4m3fn (same fp32 scale arithmetic), including values that
    land in e4m3's subnormal range and the saturating end."""
    ops = fp8_mode
    x = torch.cat([rnd(gen, 4096) * 3, rnd(gen, 2048) * 1e-3, torch.tensor([0.0, -0.0, 7.25, -7.25])]).float()
    x = x[: x.numel() // 8 * 8].contiguous()
    xg = x.to(dev)
    got, amax = ops.fp8_of(xg)
    assert abs(amax.item() - x.abs().max().item()) == 0.0
    s = (torch.tensor(448.0) / x.abs().max()).float()
    want = (x * s).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(got.cpu(), want), (got.cpu()[:16], want[:16])
    got_r, _ = ops.fp8_of(xg, relu=True)
    want_r = (torch.relu(x) * s).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(got_r.cpu() & 0x7f, want_r & 0x7f)      # (+0 / -0 of a flushed negative may differ in the sign bit)


def test_train_step_fp8_tracks_fp32(dev, fp8_mode):
    """One whole train_step in fp8 mode (fp8 forward / data-grad for the >= 128-channel convs of G / D / S, bf16 elsewhere and
    in the recognizer) against the fp32-mode step on identical weights and inputs: finite, scalars within 0.2 * max(1, |.|)
    (E4M3 keeps 4 significant bits per operand; the x70 logits of this fixture moved 10.4 % in one run),
    every network's flat gradient within 25 degrees of the fp32 one (cosine > 0.9; G > 0.8)."""
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, nn, optimizers
    ops = fp8_mode
    NA.configure(device=dev, seed=3)
    gen = torch.Generator().manual_seed(11)
    B, L = 8, 4
    images = (torch.rand(B, 32, 16 * L, 1, generator=gen) * 2 - 1).numpy()
    style = (torch.rand(B, 32, 16 * L, 1, generator=gen) * 2 - 1).numpy()
    labels = torch.randint(0, 52, (B, L), generator=gen).numpy().astype(np.int32)
    fake = torch.randint(0, 52, (B, L), generator=gen).numpy().astype(np.int32)
    results = {}
    for mode in ("f32", "fp8"):
        ops.set_conv_dtype(mode)
        NA._model_counter[0] = 0
        G = NA.make_generator(128, (32, 160, 1), (32, 8192), None, "B3", 52, vis_model=False)
        D = NA.make_discriminator((32, 160, 1), None, "B1", vis_model=False)
        R = NA.make_recognizer((32, 160, 1), None, 53, vis_model=False)
        S = NA.make_style_promoter((32, 160, 1), None, "B1", vis_model=False)
        gan = NA.make_gan(G, D, R, S, vis_model=False)
        for m in (D, S):                      # logits O(1): std(g_loss) is then not a difference of nearly equal numbers
            m.store.p["dense.w"].mul_(70.0)
        g2 = torch.Generator().manual_seed(5)
        nl = {n: {k: v.to(dev) for k, v in nn.nonlocal_weights(64, g2, torch.device("cpu")).items()}
              for n in ("G.style", "G.up", "D.fake", "D.real", "S.fake", "S.style", "S.real")}
        opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
        out = DU.train_step(0, 0, 1, images, labels, D, R, S, gan, opts[0], opts[1], opts[2], opts[3], style, B, 128,
                            net_loss.hinge, 1, 1, None, 10, "", fake_labels=fake, nl=nl, verbose=False)
        results[mode] = (np.array(out, np.float64), {n: m.store.grad.clone() for n, m in (("G", G), ("D", D), ("R", R), ("S", S))})
    s32, g32_ = results["f32"]
    s8, g8 = results["fp8"]
    assert np.all(np.isfinite(s8)), s8
    assert np.all(np.abs(s8 - s32) <= 0.2 * np.maximum(1.0, np.abs(s32))), (s8, s32)
    for n in ("D", "R", "S", "G"):
        a, b = g32_[n].double(), g8[n].double()
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert cos > (0.8 if n == "G" else 0.9), "%s: cosine %.4f" % (n, cos)


@pytest.mark.parametrize("B", [2, 40])       # 2: every tile is a reduction-split tail tile (amax from the tail sweep); 40: full rounds + tail
def test_fp8_producer_amax_is_the_amax_of_the_result(dev, gen, fp8_mode, B):
    """Config c5: the conv epilogue records max |result| (and max |scale_b result| for a gradient) while it writes the result, so
    the fp8 conversion of that result needs no amax sweep.  The recorded pair must be BITWISE the amax of the tensor the kernel
    wrote (forward with bias, data-grad with ReLU mask / accumulate / per-sample factors), and ops.fp8_of / grad_operand_fp8 must
    produce the same bytes with it as with the sweep."""
    ops = fp8_mode
    H, W, C = 8, 40, 512
    x = g32(rnd(gen, B, H, W, C), dev)
    w = g32(rnd(gen, 3, 3, C, C) / math.sqrt(9 * C), dev)
    b = g32(rnd(gen, C), dev)
    sc = g32(torch.rand(B, generator=gen, dtype=torch.float64) * 2 - 0.5, dev)
    y = ops.conv2d_fwd(x, w, b, relu_in=True)
    rec = ops._amax_get(y)
    assert rec is not None and rec[0].item() == y.abs().max().item() and rec[1].item() == rec[0].item()
    got, a = ops.fp8_of(y, relu=True)
    ops.new_step()                                   # forget the recorded amax: the sweep path
    ref, a_ref = ops.fp8_of(y, relu=True)
    assert a.item() == a_ref.item() and torch.equal(got, ref)
    dy = g32(rnd(gen, B, H, W, C), dev)
    base = g32(rnd(gen, B, H, W, C), dev)
    dx = ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=base.clone(), accum=True, amax_scale=sc)
    rec = ops._amax_get(dx, sc, need_scaled=True)
    assert rec is not None and rec[0].item() == dx.abs().max().item()
    assert rec[1].item() == (dx.abs() * sc.abs().view(B, 1, 1, 1)).max().item()
    assert ops._amax_get(dx, None, need_scaled=True) is None            # recorded for other factors: not usable unscaled
    o5, a5, cs = ops.grad_operand_fp8(dx, sc, True)
    o4, a4 = ops.fp8_of(dx)
    ops.new_step()
    o5r, a5r, csr = ops.grad_operand_fp8(dx, sc, True)
    o4r, a4r = ops.fp8_of(dx)
    assert a5.item() == a5r.item() and a4.item() == a4r.item() and torch.equal(o5, o5r) and torch.equal(o4, o4r)


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_operand_only_conv_result(dev, gen, mode):
    """conv1 -> conv2 of a ResNetBlockDown (resnet_ops.py:97-104) in configs c3 / c5: ops.conv2d_fwd(want16="only") writes the bf16
    operand copy of c1 ALONE; the fp32 tensor it returns is a never-written handle (poisoned with NaN in the tests).  The copy must be
    bit-identical to the twin of the ordinary launch; conv2, the ReLU mask of its data-grad and its weight-grad must give the SAME
    results from the handle as from the real tensor (bf16 mode: bitwise, they read the same bytes); in fp8 mode c1's e4m3 operand is
    quantised from the bf16 values with the amax the epilogue recorded OF THOSE VALUES (bit-exact against torch.float8_e4m3fn)."""
    from scrabble_gan_amd import ops
    try:
        ops.set_conv_dtype(mode)
        B, H, W, Cin, C = 64, 16, 80, 64, 512            # B2.conv1 at the shard batch of c5: 640 tiles of 256 x 256
        x = g32(rnd(gen, B, H, W, Cin), dev)
        w1 = g32(rnd(gen, 3, 3, Cin, C) / math.sqrt(9 * Cin), dev)
        b1 = g32(rnd(gen, C), dev)
        w2 = g32(rnd(gen, 3, 3, C, C) / math.sqrt(9 * C), dev)
        dy = g32(rnd(gen, B, H, W, C), dev)
        assert ops.operand_only_ok(B, H, W, C)
        # reference = the ordinary launches WITHOUT reduction-split tails (an operand-only launch has none: partial tiles meet in
        # the fp32 result), so that every output element sums the same products in the same order
        ops.set_deterministic(True)
        c_ref = ops.conv2d_fwd(x, w1, b1, relu_in=True, want16=True)
        t_ref = ops.bf16_of(c_ref).clone()
        y_ref = ops.conv2d_fwd(c_ref, w2, relu_in=True)
        d_ref = ops.conv2d_bwd_data(dy, w2, (H, W), mask=c_ref)
        dw_ref = torch.zeros_like(w2)
        ops.conv2d_bwd_weight(c_ref, dy, dw_ref, relu_in=True)
        ops.new_step()
        ops.set_deterministic(False)
        c_g = ops.conv2d_fwd(x, w1, b1, relu_in=True, want16="only")
        assert torch.isnan(c_g).all(), "the fp32 handle of an operand-only result must not be written"
        t_g = ops.bf16_of(c_g)
        assert torch.equal(t_g, t_ref)
        y_g = ops.conv2d_fwd(c_g, w2, relu_in=True)
        d_g = ops.conv2d_bwd_data(dy, w2, (H, W), mask=c_g)
        dw_g = torch.zeros_like(w2)
        ops.conv2d_bwd_weight(c_g[: B // 2], dy[: B // 2].contiguous(), dw_g, relu_in=True)          # (a batch slice of the handle, as the fused passes take)
        ops.conv2d_bwd_weight(c_g[B // 2:], dy[B // 2:].contiguous(), dw_g, relu_in=True)
        assert torch.isfinite(y_g).all() and torch.isfinite(d_g).all() and torch.isfinite(dw_g).all()
        if mode == "bf16":
            close(y_g, y_ref, 2e-6, "conv2 (same operand bytes; the default launch splits its tail tiles)")
            close(d_g, d_ref, 2e-6, "data-grad with the handle as ReLU mask")
            close(dw_g, dw_ref, 1e-5, "dW from the two halves of the handle")
        else:
            q, am = ops.fp8_of(c_g, relu=True)
            tf = t_g.float()
            assert am.item() == tf.abs().max().item()
            want = (torch.relu(tf) * (torch.tensor(448.0, device=dev) / tf.abs().max()).float()).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
            assert torch.equal(q & 0x7f, want & 0x7f)
            # (e4m3 of the bf16 value against e4m3 of the fp32 value: where the bf16 rounding crosses an e4m3 rounding boundary the
            #  operand moves by one e4m3 step, 2^-3 of the element; over 4 608 terms: measured 2.1e-2 of max|y|)
            close(y_g, y_ref, 5e-2, "conv2 on the e4m3 operand taken from bf16 values vs from fp32 values")
            close(d_g, d_ref, 2e-6, "data-grad with the handle as ReLU mask")       # (the mask is the bf16 twin either way)
            close(dw_g, dw_ref, 5e-2, "fp8 dW, operand from bf16 values, whole-tensor amax for the batch slices")
    finally:
        ops.set_deterministic(False)
        ops.set_conv_dtype("f32")
