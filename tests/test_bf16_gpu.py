"""bf16 matrix-core convolutions (BASELINE config c3) against the CPU oracle.

Two bars per op: (1) against the oracle evaluated on bf16-ROUNDED operands the kernel must be fp32-accurate
(tol 5e-5 of max|ref|): the only deviation of the bf16 path is the round-to-nearest-even of the two MFMA operands;
(2) against the exact fp64 oracle the error is bf16 rounding noise: <= 1e-2 of max|ref| (2^-9 relative per operand,
random signs over K >= 72 terms)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import scrabble_oracle as O  # noqa: E402  (checker only)
from tests.test_ops_gpu import close, g32, rnd  # noqa: E402


@pytest.fixture()
def gen():
    return torch.Generator().manual_seed(4321)


@pytest.fixture()
def bf16_mode():
    from scrabble_gan_amd import ops
    ops.set_conv_dtype("bf16")
    yield ops
    ops.set_conv_dtype("f32")


def r16(t):
    return t.detach().to(torch.bfloat16).to(torch.float64)


CASES = [
    # B, H, W, Cin, Cout, k
    (2, 8, 12, 64, 128, 3),
    (1, 16, 24, 128, 256, 3),      # several M tiles, 2 N tiles
    (2, 4, 20, 48, 96, 3),         # K = 48: a partial 32-wide k-tile per tap; N not a multiple of the tile
    (3, 8, 40, 512, 64, 1),        # 1x1 shortcut shape, N = 64 tile
    (5, 7, 5, 40, 72, 3),          # odd spatial dims, K % 8 == 0 only
    (16, 8, 40, 256, 256, 3),      # enough tiles for the tail split (atomic partial tiles)
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", CASES)
def test_conv2d_bf16_fwd_dgrad(dev, gen, bf16_mode, B, H, W, Cin, Cout, k):
    ops = bf16_mode
    x = rnd(gen, B, H, W, Cin)
    w = rnd(gen, k, k, Cin, Cout) / math.sqrt(k * k * Cin)
    b = rnd(gen, Cout)
    dy = rnd(gen, B, H, W, Cout)
    xg, wg, bg, dyg = (g32(t, dev) for t in (x, w, b, dy))

    y = ops.conv2d_fwd(xg, wg, bg, relu_in=True)
    ref_r = O.conv2d(r16(torch.relu(x)), r16(w), b)
    ref_x = O.conv2d(torch.relu(x), w, b)
    close(y, ref_r, 5e-5, "fwd vs bf16-rounded-operand oracle")
    close(y, ref_x, 1e-2, "fwd vs exact oracle")

    # data-grad with the ReLU mask and accumulate epilogues
    xr = x.clone().requires_grad_(True)
    O.conv2d(xr, r16(w), None).backward(r16(dy))
    base = rnd(gen, B, H, W, Cin)
    dx = ops.conv2d_bwd_data(dyg, wg, (H, W), mask=xg, out=g32(base, dev), accum=True)
    ref = xr.grad * (x > 0) + base
    close(dx, ref, 5e-5, "dgrad (mask, accum) vs bf16-rounded-operand oracle")
    xe = x.clone().requires_grad_(True)
    O.conv2d(xe, w, None).backward(dy)
    close(ops.conv2d_bwd_data(dyg, wg, (H, W)), xe.grad, 1e-2, "dgrad vs exact oracle")


def test_pack_filter_layouts(dev, gen, bf16_mode):
    ops = bf16_mode
    w = rnd(gen, 3, 3, 40, 72)
    wg = g32(w, dev)
    pf = ops.packed_filter(wg, "fwd").view(9, 72, 40).float().cpu()
    pb = ops.packed_filter(wg, "bwd").view(9, 40, 72).float().cpu()
    ref = w.to(torch.bfloat16).float().view(9, 40, 72)
    assert torch.equal(pf, ref.transpose(1, 2).contiguous())
    assert torch.equal(pb, ref)
    assert ops.packed_filter(wg, "fwd") is ops.packed_filter(wg, "fwd")      # cached until the weights change
    ops.weights_changed()
    assert ops.packed_filter(wg, "fwd").data_ptr() != 0


WGRAD_CASES = [
    # B, H, W, Cin, Cout, k, scaled
    (2, 8, 12, 64, 128, 3, False),
    (3, 8, 16, 128, 256, 3, True),      # H*W = 128: one per-sample factor per 32-pixel k-tile (QSCALE 2)
    (4, 4, 20, 256, 128, 3, True),      # H*W = 80: 32-pixel k-tiles straddle samples (QSCALE 1)
    (2, 16, 80, 64, 512, 1, False),     # 1x1 shortcut
    (5, 7, 5, 72, 96, 3, True),         # odd spatial dims, channel counts off the 128 tile
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,scaled", WGRAD_CASES)
def test_conv2d_bf16_wgrad(dev, gen, bf16_mode, B, H, W, Cin, Cout, k, scaled):
    ops = bf16_mode
    x = rnd(gen, B, H, W, Cin)
    dy = rnd(gen, B, H, W, Cout)
    sc = torch.rand(B, generator=gen, dtype=torch.float64) * 2 - 0.5 if scaled else None
    dys = dy if sc is None else dy * sc.view(B, 1, 1, 1)
    base_w, base_b = rnd(gen, k, k, Cin, Cout), rnd(gen, Cout)

    def oracle(xx, dd):
        w = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
        O.conv2d(xx, w, None).backward(dd)
        return w.grad

    dw, db = g32(base_w, dev), g32(base_b, dev)
    ops.conv2d_bwd_weight(g32(x, dev), g32(dy, dev), dw, relu_in=True, db=db, sample_scale=None if sc is None else g32(sc, dev))
    # the kernel rounds relu(x) and (factor * dy) to bf16; the bias gradient sums the fp32 values
    dys32 = dy.float() if sc is None else dy.float() * sc.float().view(B, 1, 1, 1)      # the fp32 product the kernel rounds
    ref_r = oracle(r16(torch.relu(x).float()), r16(dys32))
    if W % 4:      # image rows that 4-pixel blocks would straddle: the launcher keeps the fp32 kernel (exact oracle, fp32 bar)
        close(dw - g32(base_w, dev), oracle(torch.relu(x), dys), 1e-4, "dW (fp32 fallback) vs exact oracle")
    else:
        close(dw - g32(base_w, dev), ref_r, 1e-4, "dW vs bf16-rounded-operand oracle")
        close(dw - g32(base_w, dev), oracle(torch.relu(x), dys), 1e-2, "dW vs exact oracle")
    close(db - g32(base_b, dev), dys.sum(dim=(0, 1, 2)), 5e-5, "bias gradient (fp32 sums)")


def test_train_step_bf16_tracks_fp32(dev, bf16_mode):
    """One whole train_step with bf16 matrix-core convolutions against the same step in fp32 mode (same weights, inputs,
    NonLocalBlock kernels): the 16 scalars agree to bf16 accuracy and every network's flat gradient points the same way."""
    import numpy as np
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, optimizers
    ops = bf16_mode
    NA.configure(device=dev, seed=3)
    gen = torch.Generator().manual_seed(11)
    B, L = 4, 2
    images = (torch.rand(B, 32, 16 * L, 1, generator=gen) * 2 - 1).numpy()
    style = (torch.rand(B, 32, 32, 1, generator=gen) * 2 - 1).numpy()
    labels = torch.randint(0, 52, (B, L), generator=gen).numpy().astype(np.int32)
    fake = torch.randint(0, 52, (B, L), generator=gen).numpy().astype(np.int32)
    results = {}
    for mode in ("f32", "bf16"):
        ops.set_conv_dtype(mode)
        NA._model_counter[0] = 0                       # identical initial weights in both runs
        G = NA.make_generator(128, (32, 160, 1), (32, 8192), None, "B3", 52, vis_model=False)
        D = NA.make_discriminator((32, 160, 1), None, "B1", vis_model=False)
        R = NA.make_recognizer((32, 160, 1), None, 53, vis_model=False)
        S = NA.make_style_promoter((32, 160, 1), None, "B1", vis_model=False)
        gan = NA.make_gan(G, D, R, S, vis_model=False)
        from scrabble_gan_amd import nn
        g2 = torch.Generator().manual_seed(5)
        nl = {n: {k: v.to(dev) for k, v in nn.nonlocal_weights(64, g2, torch.device("cpu")).items()}
              for n in ("G.style", "G.up", "D.fake", "D.real", "S.fake", "S.style", "S.real")}
        opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
        out = DU.train_step(0, 0, 1, images, labels, D, R, S, gan, opts[0], opts[1], opts[2], opts[3], style, B, 128,
                            net_loss.hinge, 1, 0, None, 10, "", fake_labels=fake, nl=nl, verbose=False)
        results[mode] = (np.array(out, np.float64), {n: m.store.grad.clone() for n, m in (("G", G), ("D", D), ("R", R), ("S", S))})
    s32, g32_ = results["f32"]
    s16, g16 = results["bf16"]
    assert np.all(np.isfinite(s16))
    assert np.all(np.abs(s16 - s32) <= 3e-2 * np.maximum(1.0, np.abs(s32))), (s16, s32)
    for n in ("D", "R", "S", "G"):
        a, b = g32_[n].double(), g16[n].double()
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        rel = float((a - b).norm() / (a.norm() + 1e-30))
        assert cos > (0.97 if n == "G" else 0.995), "%s: cosine %.5f, relative error %.3e" % (n, cos, rel)


@pytest.mark.parametrize("H,W,Cin,Cout,k", [(16, 80, 512, 512, 3), (8, 40, 1024, 1024, 3), (4, 20, 1024, 1024, 3), (16, 80, 64, 512, 3),
                                           (8, 40, 512, 1024, 1)])
def test_bf16_kernels_full_size_on_bf16_representable_inputs(dev, bf16_mode, H, W, Cin, Cout, k):
    """BASELINE sizes (bs 128).  With operands that ARE bf16 values the rounding inside the bf16 kernels is the identity,
    so forward / data-grad / weight-grad must (1) agree with the fp32 kernels up to fp32 summation order and (2) satisfy
    the adjoint identity <conv(x,w),dy> == <x,dgrad> == <w,wgrad> -- size-independent checks of the bf16 path."""
    from tests.test_fullsize_gpu import dot, rel
    ops = bf16_mode
    B = 128
    g = torch.Generator(device=dev).manual_seed(H * W + Cin + 1)
    q = lambda t: t.to(torch.bfloat16).float()
    x = q(torch.randn(B, H, W, Cin, device=dev, generator=g))
    w = q(torch.randn(k, k, Cin, Cout, device=dev, generator=g) / (k * Cin ** 0.5))
    dy = q(torch.randn(B, H, W, Cout, device=dev, generator=g))
    res = {}
    for mode in ("bf16", "f32"):
        ops.set_conv_dtype(mode)
        y = ops.conv2d_fwd(x, w)
        dx = ops.conv2d_bwd_data(dy, w, (H, W))
        dw = torch.zeros_like(w)
        ops.conv2d_bwd_weight(x, dy, dw)
        res[mode] = (y, dx, dw)
    ops.set_conv_dtype("bf16")
    for name, a, b in zip(("y", "dx", "dw"), res["bf16"], res["f32"]):
        err = (a - b).abs().max().item()
        assert err <= 2e-4 * b.abs().max().item(), "%s: bf16 kernel vs fp32 kernel %.3e (scale %.3e)" % (name, err, b.abs().max().item())
    y, dx, dw = res["bf16"]
    a, b, c = dot(y, dy), dot(x, dx), dot(w, dw)
    assert rel(a, b) < 2e-3 and rel(a, c) < 2e-3, (a, b, c)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride", [(2, 4, 8, 64, 64, 3, (2, 2)), (1, 8, 20, 256, 128, 3, (2, 2)), (2, 8, 16, 128, 64, 3, (2, 1)),
                                                     (4, 4, 40, 512, 256, 3, (2, 2))])
def test_conv2d_transpose_bf16(dev, gen, bf16_mode, B, H, W, Cin, Cout, k, stride):
    """Conv2DTranspose forward / data-grad with bf16 operands (the generator's up path in config c3)."""
    ops = bf16_mode
    x = rnd(gen, B, H, W, Cin)
    w = rnd(gen, k, k, Cout, Cin) / math.sqrt(k * k * Cin)
    b = rnd(gen, Cout)
    xg, wg, bg = g32(x, dev), g32(w, dev), g32(b, dev)
    y = ops.conv2d_transpose_fwd(xg, wg, bg, stride=stride)
    close(y, O.conv2d_transpose(r16(x), r16(w), b, stride), 5e-5, "convT fwd vs bf16-rounded-operand oracle")
    close(y, O.conv2d_transpose(x, w, b, stride), 1e-2, "convT fwd vs exact oracle")
    dy = rnd(gen, *y.shape)
    xr = x.clone().requires_grad_(True)
    O.conv2d_transpose(xr, r16(w), None, stride).backward(r16(dy))
    dx = ops.conv2d_transpose_bwd_data(g32(dy, dev), wg, stride=stride, mask=xg)
    close(dx, xr.grad * (x > 0), 5e-5, "convT dgrad (mask) vs bf16-rounded-operand oracle")


@pytest.mark.parametrize("scaled", [False, True])
def test_cvt_bf16_bias_fused_sweep(dev, scaled):
    """sg_cvt_bf16_bias: bf16 twin of (per-sample factor x) dy and the fp32 bias gradient in one sweep; the bf16 weight-grad
    path takes it when no twin of dy exists yet.  Twin bit-exact against torch's rounding of the fp32 product, column sums
    against fp64."""
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    B, H, W, C = 7, 5, 9, 192                        # rows not a multiple of the row lanes, C / 8 = 24 column lanes (not a power of two)
    dy = torch.randn(B, H, W, C, device=dev, generator=g)
    sc = (torch.rand(B, device=dev, generator=g) * 2 - 0.5) if scaled else None
    db = torch.full((C,), 0.25, device=dev)
    t16, p16 = ops.cvt_bf16_bias(dy, sc, db, want_plain=True)
    prod = dy if sc is None else dy * sc.view(B, 1, 1, 1)
    assert torch.equal(t16.view(torch.int16), prod.to(torch.bfloat16).view(torch.int16))
    assert torch.equal(p16.view(torch.int16), dy.to(torch.bfloat16).view(torch.int16))       # the unscaled twin of the same sweep
    ref = prod.double().sum(dim=(0, 1, 2)) + 0.25
    err = (db.double() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item(), err


def test_grad_operand_bundle_shared_by_conv2_and_shortcut(dev, bf16_mode):
    """ops.grad_operand: the two weight-grads that share a gradient (conv2 and the 1x1 shortcut of a ResNetBlockDown) get the
    same scaled bf16 copy and the same column sums from ONE sweep; both bias gradients equal the fp64 sums; the plain twin for
    the data-grad launch comes out of that sweep as well."""
    ops = bf16_mode
    g = torch.Generator(device=dev).manual_seed(9)
    B, H, W, Cin, Cout = 4, 8, 10, 64, 256
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    c1 = torch.randn(B, H, W, Cout, device=dev, generator=g)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    sc = torch.rand(B, device=dev, generator=g) + 0.5
    ops.new_step()
    dw2, db2 = torch.zeros(3, 3, Cout, Cout, device=dev), torch.zeros(Cout, device=dev)
    dws, dbs = torch.zeros(1, 1, Cin, Cout, device=dev), torch.full((Cout,), 2.0, device=dev)
    ops.conv2d_bwd_weight(c1, dy, dw2, relu_in=True, db=db2, sample_scale=sc)
    n_entries = len(ops._TWINS)
    ops.conv2d_bwd_weight(x, dy, dws, db=dbs, sample_scale=sc)
    assert len(ops._TWINS) == n_entries + 1                     # only x's twin is new: dy's bundle was reused
    ref = (dy.double() * sc.double().view(B, 1, 1, 1)).sum(dim=(0, 1, 2))
    assert (db2.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert (dbs.double() - 2.0 - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert torch.equal(ops._twin_get(dy).view(torch.int16), dy.to(torch.bfloat16).view(torch.int16))


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride", [
    (2, 8, 10, 512, 256, 3, (2, 2)),       # G block 1 shape class: 256 x 256 tiles, two Cin tiles
    (3, 8, 12, 256, 128, 3, (2, 2)),       # 64 x 256 tiles (two Cout tiles of 64); M = 288 pixels: ragged last tile of 64
    (2, 8, 16, 128, 64, 3, (2, 2)),        # 64 x 64 tiles with two Cin tiles
    (2, 8, 10, 256, 256, 1, (2, 2)),       # the 1x1 stride-2 shortcut
    (2, 16, 5, 256, 256, 3, (2, 1)),       # anisotropic stride (the generator's last up block keeps the height)
])
def test_conv2d_transpose_weight_grad_bf16(dev, gen, bf16_mode, B, H, W, Cin, Cout, k, stride):
    """bf16 weight gradient of Conv2DTranspose (sg_conv2d_transpose_bwd_weight_bf16v2: the DMA-fed weight-grad kernel reading
    the gradient on its own, finer grid with a stride) against the oracle on bf16-rounded operands, accumulating into dw."""
    ops = bf16_mode
    x = rnd(gen, B, H, W, Cin)
    dy = rnd(gen, B, stride[0] * H, stride[1] * W, Cout)
    w = (rnd(gen, k, k, Cout, Cin) * 0).requires_grad_(True)
    O.conv2d_transpose(r16(x), w, None, stride).backward(r16(dy))
    dw = torch.full((k, k, Cout, Cin), 0.5, device=dev)
    ops.conv2d_transpose_bwd_weight(g32(x, dev), g32(dy, dev), dw, stride=stride)
    close(dw - 0.5, w.grad, 1e-4, "convT weight-grad vs bf16-rounded-operand oracle")


def test_tile128_configuration_when_forced(dev):
    """The 128 x 128 / two-workgroups-per-CU configuration of the DMA-fed kernel (off by default: measured slower,
    profiles/r03_probe_tile128.txt) stays parity-green: forced through sg_debug_set_tile128(2), forward (ReLU, bias, split tails
    and full rounds) and data-grad (mask, accumulate) against the oracle on bf16-rounded operands, 5e-5."""
    import math
    from scrabble_gan_amd import ops
    from scrabble_gan_amd._lib import lib
    g = torch.Generator().manual_seed(128)
    try:
        ops.set_conv_dtype("bf16")
        lib().sg_debug_set_tile128(2)
        for (B, H, W, Cin, Cout, k) in ((3, 8, 40, 256, 256, 3), (40, 4, 20, 128, 384, 1), (5, 7, 5, 512, 128, 3)):
            x = torch.randn(B, H, W, Cin, generator=g, dtype=torch.float64)
            w = torch.randn(k, k, Cin, Cout, generator=g, dtype=torch.float64) / math.sqrt(k * k * Cin)
            b = torch.randn(Cout, generator=g, dtype=torch.float64)
            dy = torch.randn(B, H, W, Cout, generator=g, dtype=torch.float64)
            r16 = lambda t: t.float().to(torch.bfloat16).to(torch.float64)
            xg, wg, bg, dyg = (t.float().to(dev).contiguous() for t in (x, w, b, dy))
            y = ops.conv2d_fwd(xg, wg, bg, relu_in=True)
            ref = O.conv2d(r16(torch.relu(x)), r16(w), b)
            assert (y.double().cpu() - ref).abs().max().item() <= 5e-5 * ref.abs().max().item(), (B, Cin, Cout, "fwd")
            base = torch.randn(B, H, W, Cin, generator=g, dtype=torch.float64)
            dx = ops.conv2d_bwd_data(dyg, wg, (H, W), mask=xg, out=base.float().to(dev), accum=True)
            xr = x.clone().requires_grad_(True)
            O.conv2d(xr, r16(w), None).backward(r16(dy))
            refd = xr.grad * (x > 0) + base.float().double()
            assert (dx.double().cpu() - refd).abs().max().item() <= 5e-5 * refd.abs().max().item(), (B, Cin, Cout, "dgrad")
    finally:
        lib().sg_debug_set_tile128(0)
        ops.set_conv_dtype("f32")
