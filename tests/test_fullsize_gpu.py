"""Full-size (BASELINE config c2: bs 128, 32x160) checks through size-independent properties -- the
oracle cannot run these sizes in seconds, so the MFMA kernels are checked against each other:

  adjoint identity   <conv(x,w), dy> == <x, conv_data_grad(dy,w)> == <w, conv_weight_grad(x,dy)>
  linearity          conv(x, a*w) == a*conv(x, w)   (through the accumulate epilogue)

on the layer shapes that carry >90 % of the step's FLOPs, plus the transposed convolutions of the
generator.  fp32 reductions over up to 1e8 terms: relative tolerance 2e-3."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import margins  # noqa: E402

SHAPES = [  # H, W, Cin, Cout, k
    (32, 160, 64, 64, 3), (16, 80, 64, 512, 3), (16, 80, 512, 512, 3), (8, 40, 512, 1024, 3), (8, 40, 1024, 1024, 3),
    (4, 20, 1024, 1024, 3), (8, 40, 512, 1024, 1), (4, 40, 512, 512, 3),
]


def dot(a, b):
    from scrabble_gan_amd import ops
    n = a.numel() // 4 * 4
    out = torch.zeros(1, device=a.device)
    ops.dot_accum(a.reshape(-1)[:n].contiguous(), b.reshape(-1)[:n].contiguous(), out)
    return out.item()


def rel(a, b):
    return abs(a - b) / (abs(a) + abs(b) + 1e-30)


@pytest.mark.parametrize("H,W,Cin,Cout,k", SHAPES)
def test_conv_adjoint_identity_full_size(dev, H, W, Cin, Cout, k):
    from scrabble_gan_amd import ops
    B = 128
    g = torch.Generator(device=dev).manual_seed(H * W + Cin)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(k, k, Cin, Cout, device=dev, generator=g) / (k * Cin ** 0.5)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    y = ops.conv2d_fwd(x, w)
    dx = ops.conv2d_bwd_data(dy, w, (H, W))
    dw = torch.zeros_like(w)
    ops.conv2d_bwd_weight(x, dy, dw)
    a, b, c = dot(y, dy), dot(x, dx), dot(w, dw)
    assert rel(a, b) < 2e-3 and rel(a, c) < 2e-3, (a, b, c)
    # linearity through the accumulate epilogue: conv(x, w) + conv(x, -w) == 0 up to fp32 rounding
    out = y.clone()
    ops.conv2d_fwd(x, -w, out=out, accum=True)
    assert out.abs().max().item() <= 1e-4 * y.abs().max().item()


@pytest.mark.parametrize("H,W,Cin,Cout,k,stride", [(4, 40, 512, 256, 3, (2, 2)), (8, 80, 256, 128, 3, (2, 2)),
                                                 (16, 160, 128, 64, 3, (2, 1)), (4, 40, 512, 256, 1, (2, 2))])
def test_conv_transpose_adjoint_identity_full_size(dev, H, W, Cin, Cout, k, stride):
    from scrabble_gan_amd import ops
    B = 128
    g = torch.Generator(device=dev).manual_seed(H + Cin)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(k, k, Cout, Cin, device=dev, generator=g) / (k * Cin ** 0.5)
    dy = torch.randn(B, stride[0] * H, stride[1] * W, Cout, device=dev, generator=g)
    y = ops.conv2d_transpose_fwd(x, w, stride=stride)
    dx = ops.conv2d_transpose_bwd_data(dy, w, stride=stride)
    dw = torch.zeros_like(w)
    ops.conv2d_transpose_bwd_weight(x, dy, dw, stride=stride)
    a, b, c = dot(y, dy), dot(x, dx), dot(w, dw)
    assert rel(a, b) < 2e-3 and rel(a, c) < 2e-3, (a, b, c)


def test_split_k_matches_single_pass(dev):
    """Small per-GPU batches (data parallel) take the split-K path (float-atomic partial tiles): it must agree with
    the oracle-verified single-pass result of the same kernel at a batch where the grid fills the chip."""
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(7)
    H, W, C = 4, 20, 1024
    x = torch.randn(128, H, W, C, device=dev, generator=g)
    w = torch.randn(3, 3, C, C, device=dev, generator=g) / (3 * 32)
    b = torch.randn(C, device=dev, generator=g)
    full = ops.conv2d_fwd(x, w, b, relu_in=True)                 # 1280 tiles: single pass
    part = ops.conv2d_fwd(x[:8].contiguous(), w, b, relu_in=True)   # 80 tiles: split-K
    assert (part - full[:8]).abs().max().item() <= 2e-5 * full.abs().max().item()
    dy = torch.randn(128, H, W, C, device=dev, generator=g)
    dfull = ops.conv2d_bwd_data(dy, w, (H, W), mask=x)
    dpart = ops.conv2d_bwd_data(dy[:8].contiguous(), w, (H, W), mask=x[:8].contiguous())
    assert (dpart - dfull[:8]).abs().max().item() <= 2e-5 * dfull.abs().max().item()


# ------------------------------------------------------------------------------------------------------------------
# Oracle slices at the REAL launch geometry: the checks above compare the kernels with each
# other; these put the headline launch path -- full_tiles >= 768, the XCD remap, the CU-quantum tail split, the
# weight-grad's pixel chunking at B = 128..384 -- against the CPU oracle.  Samples of a convolution are independent,
# so for y / dx the oracle evaluates only the first and last two samples of the batch (the tail tiles live at the
# end); dW sums over the whole batch, so there the oracle runs the full batch on the two cheapest full-geometry layers.
# Tolerances (max|got-ref| <= tol * max|ref|): fp32 kernels vs the fp64 oracle 2e-5 for y / dx, 1e-4 for dW (sums of up
# to 82k products with float-atomic partial sums); bf16 mode vs the oracle on bf16-rounded operands 5e-5 / 1e-4.
# ------------------------------------------------------------------------------------------------------------------
import math  # noqa: E402

from oracle import scrabble_oracle as O  # noqa: E402  (checker only)

GEOM = [  # B, H, W, Cin, Cout, k   (the fused-pass batch sizes of profiles/r01e_shapes_bs128.txt)
    (128, 16, 80, 512, 512, 3), (384, 16, 80, 512, 512, 3), (256, 8, 40, 512, 1024, 3), (128, 8, 40, 1024, 1024, 3),
    (384, 8, 40, 1024, 1024, 3), (256, 4, 20, 1024, 1024, 3), (384, 32, 160, 64, 64, 3), (256, 16, 80, 64, 512, 3),
    (384, 8, 40, 512, 1024, 1), (128, 32, 160, 1, 64, 3), (128, 32, 160, 64, 1, 3),      # (+ the thin first / last layers)
]


def _edge(t, n=2):
    return torch.cat([t[:n], t[-n:]], 0).double().cpu()


def _close(got, ref, tol, name):
    got, ref = got.double().cpu(), ref.double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item() + 1e-30
    margins.record(name, err / scale, tol)
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (rel %.3e > %.1e)" % (name, err, scale, err / scale, tol)


def _r16(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.fixture(params=["f32", "bf16", "fp8"])
def conv_mode(request):
    from scrabble_gan_amd import ops
    ops.set_conv_dtype(request.param)
    yield request.param
    ops.set_conv_dtype("f32")


def _q8(t, amax):
    """e4m3 quantisation exactly as sg_cvt_fp8 / sg_pack_filter_fp8 do it (tests/test_fp8_gpu.py:q8), with the amax of the
    WHOLE device tensor (the edge samples are quantised with the scale the kernel used for the full batch)."""
    s = (torch.tensor(448.0) / amax.float().cpu()).float()
    return (t.float() * s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float64)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", GEOM)
def test_conv_fwd_dgrad_vs_oracle_at_launch_geometry(dev, conv_mode, B, H, W, Cin, Cout, k):
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(B + H * W + Cin + k)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(k, k, Cin, Cout, device=dev, generator=g) / math.sqrt(k * k * Cin)
    b = torch.randn(Cout, device=dev, generator=g)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    if conv_mode == "fp8":
        # config c5: the >= 128-channel launches quantise both operands to e4m3 per tensor; the oracle gets the SAME quantised
        # operands (5e-5: only the fp32 accumulation order differs), VERDICT r2 weak #3.  64-channel / thin rows of GEOM run
        # exactly as in bf16 mode and are covered there.
        fwd8, dg8 = ops._fp8_ok(Cin, Cout, k, k, True), ops._fp8_ok(Cout, Cin, k, k, True)
        if not (fwd8 or dg8):
            pytest.skip("not an fp8 launch (bf16 path, covered by the bf16 mode)")
        y = ops.conv2d_fwd(x, w, b, relu_in=True)
        dx = ops.conv2d_bwd_data(dy, w, (H, W), mask=x)
        ax, aw, ad = x.abs().max(), w.abs().max(), dy.abs().max()
        xe, we, be, dye = _edge(x), w.cpu(), b.double().cpu(), _edge(dy)
        qw = _q8(we, aw)
        if fwd8:
            osc = (ax.cpu() * aw.cpu() * torch.tensor(1.0 / (448.0 * 448.0))).double()
            _close(_edge(y), O.conv2d(_q8(torch.relu(xe), ax), qw, None) * osc + be, 5e-5, "fp8 y (first/last 2 samples)")
        else:
            _close(_edge(y), O.conv2d(_r16(torch.relu(xe)), _r16(we), be), 5e-5, "bf16 y (first/last 2 samples)")
        xr = xe.clone().requires_grad_(True)
        if dg8:
            osd = (ad.cpu() * aw.cpu() * torch.tensor(1.0 / (448.0 * 448.0))).double()
            O.conv2d(xr, qw, None).backward(_q8(dye, ad))
            _close(_edge(dx), xr.grad * osd * (xe > 0), 5e-5, "fp8 dx (first/last 2 samples)")
        else:
            O.conv2d(xr, _r16(we), None).backward(_r16(dye))
            _close(_edge(dx), xr.grad * (xe > 0), 5e-5, "bf16 dx (first/last 2 samples)")
        return
    rq = _r16 if (conv_mode == "bf16" and Cin > 1 and Cout > 1) else (lambda t: t)     # (thin layers stay fp32 in every mode)
    y = ops.conv2d_fwd(x, w, b, relu_in=True)
    dx = ops.conv2d_bwd_data(dy, w, (H, W), mask=x)
    xe, we, be, dye = _edge(x), w.double().cpu(), b.double().cpu(), _edge(dy)
    _close(_edge(y), O.conv2d(rq(torch.relu(xe)), rq(we), be), 5e-5 if conv_mode == "bf16" else 2e-5, "y (first/last 2 samples)")
    xr = xe.clone().requires_grad_(True)
    O.conv2d(xr, rq(we), None).backward(rq(dye))
    _close(_edge(dx), xr.grad * (xe > 0), 5e-5 if conv_mode == "bf16" else 2e-5, "dx (first/last 2 samples)")


def _oracle_dw(x, dy, k, chunk=32):
    """fp64 weight gradient of the SAME conv over the whole batch, accumulated chunk by chunk."""
    Cin, Cout = x.shape[-1], dy.shape[-1]
    tot = torch.zeros(k, k, Cin, Cout, dtype=torch.float64)
    for lo in range(0, x.shape[0], chunk):
        w = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
        O.conv2d(x[lo:lo + chunk], w, None).backward(dy[lo:lo + chunk])
        tot += w.grad
    return tot


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,scaled", [(128, 4, 20, 1024, 1024, 3, False), (256, 4, 20, 1024, 1024, 3, True),
                                                      (256, 8, 40, 512, 1024, 1, True), (128, 8, 40, 512, 1024, 1, False),
                                                      (128, 32, 160, 1, 64, 3, True), (128, 32, 160, 64, 1, 3, False)])
def test_conv_wgrad_vs_oracle_at_launch_geometry(dev, conv_mode, B, H, W, Cin, Cout, k, scaled):
    """Whole-tensor dW (and the fused bias gradient) of the full batch: pixel chunking, the per-sample factors of the
    shared backward sweep (`scaled`) and the float-atomic partial sums all take part."""
    from scrabble_gan_amd import ops
    if conv_mode == "fp8" and not ops._fp8_wgrad_ok(Cin, Cout, k, k, True):
        pytest.skip("not an fp8 launch (covered by the bf16 mode)")
    g = torch.Generator(device=dev).manual_seed(B + H + Cin + k)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    sc = (torch.rand(B, device=dev, generator=g) * 2 - 0.5) if scaled else None
    dw = torch.zeros(k, k, Cin, Cout, device=dev)
    db = torch.zeros(Cout, device=dev) if Cout > 1 else None
    ops.conv2d_bwd_weight(x, dy, dw, relu_in=True, db=db, sample_scale=sc)
    dys32 = dy if sc is None else dy * sc.view(B, 1, 1, 1)            # the fp32 product the bf16 kernel rounds
    xr = torch.relu(x).cpu()
    if conv_mode == "fp8":
        # config c5: e4m3 activations x e5m2 gradients with per-tensor scales; the oracle gets the SAME quantised operands
        ax, aq = x.abs().max().cpu(), dys32.abs().max().cpu()
        q5 = (dys32.cpu() * (torch.tensor(57344.0) / aq).float()).clamp(-57344.0, 57344.0).to(torch.float8_e5m2).to(torch.float64)
        ref = _oracle_dw(_q8(xr, ax), q5, k) * ((ax.double() / 448.0) * (aq.double() / 57344.0))
    elif conv_mode == "bf16" and Cin > 1 and Cout > 1:                # (the thin first / last layers stay fp32 in every mode)
        ref = _oracle_dw(_r16(xr), _r16(dys32.cpu()), k)
    else:
        dys = dy.double().cpu() if sc is None else dy.double().cpu() * sc.double().cpu().view(B, 1, 1, 1)
        ref = _oracle_dw(xr.double(), dys, k)
    _close(dw, ref, 1e-4, "dW (whole tensor, whole batch)")
    if db is not None:
        _close(db, dys32.double().cpu().sum(dim=(0, 1, 2)), 5e-5, "fused bias gradient")


@pytest.mark.parametrize("H,W,Cin,Cout,k,stride", [(4, 40, 512, 256, 3, (2, 2)), (8, 80, 256, 128, 3, (2, 2)),
                                                 (16, 160, 128, 64, 3, (2, 1)), (4, 40, 512, 256, 1, (2, 2))])
def test_conv_transpose_vs_oracle_at_launch_geometry(dev, conv_mode, H, W, Cin, Cout, k, stride):
    """The generator's three Conv2DTranspose layers (+ a 1x1 stride-2 shortcut) at bs 128: y / dx on the first and
    last two samples; dW of the whole batch on the cheapest layer."""
    from scrabble_gan_amd import ops
    if conv_mode == "fp8":
        pytest.skip("transposed convs run the bf16 kernels in fp8 mode (covered by the bf16 mode)")
    B = 128
    g = torch.Generator(device=dev).manual_seed(H + Cin + k)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(k, k, Cout, Cin, device=dev, generator=g) / math.sqrt(k * k * Cin)
    b = torch.randn(Cout, device=dev, generator=g)
    dy = torch.randn(B, stride[0] * H, stride[1] * W, Cout, device=dev, generator=g)
    # bf16 mode: the 1x1 stride-2 shortcut's FORWARD keeps the fp32 kernel (a parity class without a tap only writes its
    # bias); its data-grad and every 3x3 launch round both MFMA operands to bf16
    ident = lambda t: t
    rq = _r16 if (conv_mode == "bf16" and k == 3) else ident
    rq_d = _r16 if conv_mode == "bf16" else ident
    tol = 5e-5 if conv_mode == "bf16" else 2e-5
    y = ops.conv2d_transpose_fwd(x, w, b, stride=stride)
    dx = ops.conv2d_transpose_bwd_data(dy, w, stride=stride)
    xe, we, be, dye = _edge(x), w.double().cpu(), b.double().cpu(), _edge(dy)
    _close(_edge(y), O.conv2d_transpose(rq(xe), rq(we), be, stride), tol, "convT y")
    xr = xe.clone().requires_grad_(True)
    O.conv2d_transpose(xr, rq_d(we), None, stride).backward(rq_d(dye))
    _close(_edge(dx), xr.grad, tol, "convT dx")
    if (H, k) in ((4, 3), (4, 1)):
        dw = torch.zeros_like(w)
        ops.conv2d_transpose_bwd_weight(x, dy, dw, stride=stride)
        tot = torch.zeros(k, k, Cout, Cin, dtype=torch.float64)
        xc, dyc = rq_d(x.cpu()).double(), rq_d(dy.cpu()).double()      # (bf16 mode: the weight-grad rounds both operands too, round 2)
        for lo in range(0, B, 32):
            wz = torch.zeros(k, k, Cout, Cin, dtype=torch.float64, requires_grad=True)
            O.conv2d_transpose(xc[lo:lo + 32], wz, None, stride).backward(dyc[lo:lo + 32])
            tot += wz.grad
        _close(dw, tot, 1e-4, "convT dW (whole batch)")


@pytest.mark.parametrize("mode", ["f32-v2", "bf16", "fp8"])
def test_deterministic_mode_covers_both_kernel_generations(dev, mode):
    """configure(deterministic=True) -> sg_set_deterministic(1) must reach the second-generation (DMA-fed) kernels too (ADVICE
    r2): with it, a sample's forward / data-grad result is BITWISE independent of the batch it is launched in, at shapes whose
    tile count is not a multiple of the 256 CUs (the default launch would cut the tail tiles along the reduction and add
    partial tiles with float atomics)."""
    from scrabble_gan_amd import ops
    from scrabble_gan_amd._lib import lib
    g = torch.Generator(device=dev).manual_seed(3)
    H, W, C = 8, 40, 512
    x = torch.randn(13, H, W, C, device=dev, generator=g)
    w = torch.randn(3, 3, C, C, device=dev, generator=g) / 64
    dy = torch.randn(13, H, W, C, device=dev, generator=g)
    old_min = ops.F32_V2_MIN_TILES
    try:
        if mode == "f32-v2":
            ops.F32_V2_MIN_TILES = 0
        else:
            ops.set_conv_dtype(mode)
        ops.set_deterministic(True)
        assert lib().sg_set_deterministic(1) == 1                  # (returns the previous setting)
        y5, y13 = ops.conv2d_fwd(x[:5].contiguous(), w), ops.conv2d_fwd(x, w)
        d5, d13 = ops.conv2d_bwd_data(dy[:5].contiguous(), w, (H, W)), ops.conv2d_bwd_data(dy, w, (H, W))
        if mode == "fp8":      # per-tensor amax depends on the batch: compare two launches of the same batch instead
            assert torch.equal(ops.conv2d_fwd(x, w), y13) and torch.equal(ops.conv2d_bwd_data(dy, w, (H, W)), d13)
        else:
            assert torch.equal(y5, y13[:5]) and torch.equal(d5, d13[:5])
    finally:
        ops.set_deterministic(False)
        ops.F32_V2_MIN_TILES = old_min
        ops.set_conv_dtype("f32")


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp8"])
def test_deterministic_weight_gradient(dev, mode):
    """sg_set_deterministic(1) also fixes the summation order of dW and the bias gradient: one pixel chunk per
    (tap, tile) and one workgroup for the column sums, i.e. ONE adder per address -- two launches on the same operands, adding
    into a NON-zero gradient buffer (the accumulate contract), give bitwise identical results, on the MFMA kernels of every mode
    and on the thin first-layer kernel.  The default launch of the same shape cuts the pixels into chunks that meet through
    float atomics; it must agree with the deterministic result to fp32 rounding (1e-5 of max|dW|; bf16 / fp8: same operands)."""
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(17)
    try:
        ops.set_conv_dtype(mode)
        for (B, H, W, Cin, Cout, k) in ((40, 4, 20, 256, 256, 3), (24, 8, 40, 64, 512, 1), (16, 32, 160, 1, 64, 3)):
            x = torch.randn(B, H, W, Cin, device=dev, generator=g)
            dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
            sc = torch.rand(B, device=dev, generator=g) + 0.5
            base_w = torch.randn(k, k, Cin, Cout, device=dev, generator=g)
            base_b = torch.randn(Cout, device=dev, generator=g)
            res = []
            for det in (True, True, False):
                ops.new_step()
                ops.set_deterministic(det)
                dw, db = base_w.clone(), base_b.clone()
                ops.conv2d_bwd_weight(x, dy, dw, relu_in=True, db=db, sample_scale=sc)
                res.append((dw, db))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (mode, Cin, Cout)
            inc = (res[0][0] - base_w).abs().max().item()
            assert (res[0][0] - res[2][0]).abs().max().item() <= 1e-5 * inc + 1e-6, (mode, Cin, Cout)
            assert (res[0][1] - res[2][1]).abs().max().item() <= 1e-5 * (res[0][1] - base_b).abs().max().item() + 1e-5
    finally:
        ops.set_deterministic(False)
        ops.set_conv_dtype("f32")


@pytest.mark.parametrize("mode,C", [("bf16", 512), ("bf16", 64), ("fp8", 512), ("fp8", 1024)])
def test_pool_backward_operand_copies_are_bit_identical(dev, mode, C):
    """ops.avgpool2_bwd_operands (sg_avgpool2_bwd_bf16 / _fp8: conv2's gradient operands written straight from the pooled
    gradient, no fp32 tensor in HBM) against the unfused path (sg_avgpool2_bwd, then the conversion sweeps): the plain and the
    per-sample-scaled copies and the fp8 amax scalars must be BIT-identical (0.25 x is exact); the bias gradient (column sums
    taken over dout instead of over its 4x replication) agrees to 2e-5."""
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(C)
    B, Ho, Wo = 24, 8, 40
    dout = torch.randn(B, Ho, Wo, C, device=dev, generator=g) * torch.rand(B, 1, 1, 1, device=dev, generator=g)
    sc = torch.rand(B, device=dev, generator=g) * 2 - 0.5
    try:
        ops.set_conv_dtype(mode)
        d = ops.avgpool2_bwd(dout)
        gh = ops.avgpool2_bwd_operands(dout, sc, True)
        assert torch.isnan(gh).all(), "the handle must not be written (tests poison it: ops.GHOST_NAN)"
        if mode == "bf16":
            t_u, cs_u = ops.grad_operand(d, sc, True)
            t_f, cs_f = ops.grad_operand(gh, sc, True)
            assert torch.equal(t_u, t_f) and torch.equal(ops.bf16_of(d), ops.bf16_of(gh))
        else:
            o5_u, a5_u, cs_u = ops.grad_operand_fp8(d, sc, True)
            o5_f, a5_f, cs_f = ops.grad_operand_fp8(gh, sc, True)
            (o4_u, a4_u), (o4_f, a4_f) = ops.fp8_of(d), ops.fp8_of(gh)
            assert torch.equal(o5_u, o5_f) and torch.equal(o4_u, o4_f) and a5_u.item() == a5_f.item() and a4_u.item() == a4_f.item()
        _close(cs_f, cs_u, 2e-5, "bias gradient from the pooled gradient")
        # without per-sample factors the weight-grad operand IS the plain copy
        gh2 = ops.avgpool2_bwd_operands(dout, None, True)
        if mode == "bf16":
            assert torch.equal(ops.grad_operand(gh2, None, True)[0], ops.bf16_of(d))
    finally:
        ops.set_conv_dtype("f32")
