"""Winograd-domain fp32 3x3 convolutions (scrabble_gan_amd/csrc/conv_winograd.hip; round 3) against the fp64 oracle, both forms:
F(4x4, 3x3) (the default wherever H and W are multiples of 4) and F(2x2, 3x3).

The path replaces the direct implicit-GEMM launch of the stride-1 SAME 3x3 convolutions over >= 64 channels in fp32 mode
(ops._wino_ok; resnet_ops.py:65,98,103 of the reference: the ResNet blocks' convolutions; net_architecture.py:28-49: the
recognizer's) -- same contract, 9 / 36 or 16 / 36 of the multiplies.  Tolerances (max |got - ref| <= tol * max |ref|): 2e-5 for
y / dx and 1e-4 for dW / db against the fp64 oracle, the bounds the direct fp32 kernels are held to (tests/test_fullsize_gpu.py);
the measured errors are printed (F(2x2) <= 1e-6, F(4x4) <= 5e-6 here and <= 1.3e-5 against the direct kernels at launch
geometry).  The launch-geometry rows of tests/test_fullsize_gpu.py (f32 mode) and the whole-network / train_step tests run
through the default path too."""
import math

import pytest
import torch

from oracle import scrabble_oracle as O  # checker only

pytestmark = pytest.mark.gpu

from tests import margins  # noqa: E402


@pytest.fixture(params=[2, 4], ids=["F2x2", "F4x4"])
def wino_everywhere(request):
    """Every eligible shape (K % 32 == 0, N % 64 == 0, even H and W) through the Winograd path; F4x4: F(4x4, 3x3) wherever H and W
    are multiples of 4 (the other shapes of the lists then run F(2x2, 3x3) again)."""
    from scrabble_gan_amd import ops
    old = (ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.USE_WINOGRAD, ops.WINO_TILE, ops.WINO4_WGRAD_MIN_TILES, ops.WINO_ROW_GAIN)
    ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.USE_WINOGRAD, ops.WINO_TILE, ops.WINO4_WGRAD_MIN_TILES, ops.WINO_ROW_GAIN = {2: 32, 4: 32}, {2: 0, 4: 0}, True, request.param, 0, 0.0
    yield ops
    ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.USE_WINOGRAD, ops.WINO_TILE, ops.WINO4_WGRAD_MIN_TILES, ops.WINO_ROW_GAIN = old


@pytest.fixture(params=[2, 4], ids=["F2x2", "F4x4"])
def wino_tile(request):
    from scrabble_gan_amd import ops
    old = ops.WINO_TILE
    ops.WINO_TILE = request.param
    yield request.param
    ops.WINO_TILE = old


def _close(got, ref, tol, name):
    got, ref = got.double().cpu(), ref.double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item() + 1e-30
    print("%s: rel err %.3e" % (name, err / scale))
    margins.record(name, err / scale, tol)
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (rel %.3e > %.1e)" % (name, err, scale, err / scale, tol)


# B, H, W, Cin, Cout: one tile per sample; tile counts below / not a multiple of the 128-row plane padding; odd batches; the
# recognizer-like wide rows; Cin != Cout both ways (the data-grad runs when Cin % 128 == 0 too)
SMALL = [(1, 2, 2, 32, 128), (1, 4, 4, 32, 128), (3, 4, 8, 128, 128), (3, 2, 4, 128, 128), (5, 4, 10, 96, 128), (2, 8, 40, 256, 128), (7, 6, 6, 32, 384), (2, 16, 80, 64, 128),
         (33, 4, 20, 256, 256), (16, 4, 20, 128, 256),
         (2, 8, 8, 128, 64), (3, 4, 12, 64, 64), (2, 16, 80, 512, 64), (5, 6, 10, 64, 192)]      # 64-wide product tiles (round 4): Cout % 128 == 64


@pytest.mark.parametrize("B,H,W,Cin,Cout", SMALL)
def test_forward_and_data_grad_vs_oracle(dev, wino_everywhere, B, H, W, Cin, Cout):
    ops = wino_everywhere
    assert ops._wino_ok(Cin, Cout, 3, 3, True, H, W)
    dgrad_too = ops._wino_ok(Cout, Cin, 3, 3, True, H, W)      # (the data-grad's output channels are Cin: % 128)
    g = torch.Generator(device=dev).manual_seed(B * 1000 + H * W + Cin)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g) / math.sqrt(9 * Cin)
    b1 = torch.randn(Cout, device=dev, generator=g)
    b2 = torch.randn(Cout, device=dev, generator=g)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    prev = torch.randn(B, H, W, Cout, device=dev, generator=g)
    xd, wd, dyd = x.double().cpu(), w.double().cpu(), dy.double().cpu()
    # forward: plain; operand ReLU + two biases; accumulate; output ReLU
    _close(ops.conv2d_fwd(x, w), O.conv2d(xd, wd, None), 2e-5, "y")
    ref = O.conv2d(torch.relu(xd), wd, b1.double().cpu()) + b2.double().cpu()
    _close(ops.conv2d_fwd(x, w, b1, b2, relu_in=True), ref, 2e-5, "y (relu_in, bias, bias2)")
    out = prev.clone()
    ops.conv2d_fwd(x, w, b1, out=out, accum=True)
    _close(out, O.conv2d(xd, wd, b1.double().cpu()) + prev.double().cpu(), 2e-5, "y (accumulate)")
    _close(ops.conv2d_fwd(x, w, b1, relu_out=True), torch.relu(O.conv2d(xd, wd, b1.double().cpu())), 2e-5, "y (relu_out)")
    if not dgrad_too:
        return
    # data-grad: plain; ReLU mask; mask + accumulate
    xr = xd.clone().requires_grad_(True)
    O.conv2d(xr, wd, None).backward(dyd)
    _close(ops.conv2d_bwd_data(dy, w, (H, W)), xr.grad, 2e-5, "dx")
    _close(ops.conv2d_bwd_data(dy, w, (H, W), mask=x), xr.grad * (xd > 0), 2e-5, "dx (mask)")
    acc = torch.randn(B, H, W, Cin, device=dev, generator=g)
    ref = xr.grad * (xd > 0) + acc.double().cpu()
    ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=acc, accum=True)
    _close(acc, ref, 2e-5, "dx (mask, accumulate)")


@pytest.mark.parametrize("B,H,W,Cin,Cout,scaled", [(3, 2, 4, 64, 64, False), (3, 4, 8, 64, 64, True), (5, 4, 10, 96, 128, True), (2, 8, 40, 128, 64, True),
                                                   (7, 6, 6, 64, 192, False), (33, 4, 20, 256, 256, True), (16, 8, 40, 512, 1024, True)])
def test_weight_grad_vs_oracle(dev, wino_everywhere, B, H, W, Cin, Cout, scaled):
    """dW += and the bias gradient of the same sweep, with the per-sample factors of the shared backward sweep (`scaled`) and the
    operand ReLU: 1e-4 of max |dW| against the fp64 oracle (the bound of the direct weight-grad kernels)."""
    ops = wino_everywhere
    assert ops._wino_wgrad_ok(Cin, Cout, 3, 3, True, H, W)
    g = torch.Generator(device=dev).manual_seed(B * 100 + H + Cin)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    sc = (torch.rand(B, device=dev, generator=g) * 2 - 0.5) if scaled else None
    dw0 = torch.randn(3, 3, Cin, Cout, device=dev, generator=g)
    db0 = torch.randn(Cout, device=dev, generator=g)
    dw, db = dw0.clone(), db0.clone()
    ops.conv2d_bwd_weight(x, dy, dw, relu_in=True, db=db, sample_scale=sc)
    dys = (dy if sc is None else dy * sc.view(B, 1, 1, 1)).double().cpu()
    wz = torch.zeros(3, 3, Cin, Cout, dtype=torch.float64, requires_grad=True)
    O.conv2d(torch.relu(x.double().cpu()), wz, None).backward(dys)
    _close(dw - dw0, wz.grad, 1e-4, "dW")
    _close(db - db0, dys.sum((0, 1, 2)), 1e-4, "db")


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(128, 16, 80, 512, 512), (48, 8, 40, 1024, 1024), (16, 4, 20, 1024, 1024), (24, 8, 40, 512, 1024)])
def test_agrees_with_the_direct_kernels_at_launch_geometry(dev, wino_tile, B, H, W, Cin, Cout):
    """The same launch through both forms (SG_WINOGRAD on / off) at the headline batch and the 8-way shard batch: both are held to
    2e-5 (dW, db: 1e-4) of the fp64 oracle elsewhere, so they may differ by twice that."""
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(B + Cin)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, device=dev, generator=g)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    assert ops._wino_ok(Cin, Cout, 3, 3, True, H, W)
    sc = torch.rand(B, device=dev, generator=g) + 0.5
    yw = ops.conv2d_fwd(x, w, b, relu_in=True)
    dxw = ops.conv2d_bwd_data(dy, w, (H, W), mask=x)
    dww, dbw = torch.zeros_like(w), torch.zeros_like(b)
    ops.conv2d_bwd_weight(x, dy, dww, relu_in=True, db=dbw, sample_scale=sc)
    old = ops.USE_WINOGRAD
    ops.USE_WINOGRAD = False
    try:
        yd = ops.conv2d_fwd(x, w, b, relu_in=True)
        dxd = ops.conv2d_bwd_data(dy, w, (H, W), mask=x)
        dwd, dbd = torch.zeros_like(w), torch.zeros_like(b)
        ops.conv2d_bwd_weight(x, dy, dwd, relu_in=True, db=dbd, sample_scale=sc)
    finally:
        ops.USE_WINOGRAD = old
    _close(yw, yd, 4e-5, "y: Winograd vs direct")
    _close(dxw, dxd, 4e-5, "dx: Winograd vs direct")
    _close(dww, dwd, 2e-4, "dW: Winograd vs direct")
    _close(dbw, dbd, 2e-4, "db: Winograd vs direct")


def test_c_abi_entry_points_and_workspace_contract(dev, wino_everywhere):
    """sg_conv2d_fwd_wino / sg_conv2d_bwd_data_wino with a caller-provided workspace: bitwise the result of the three exported
    steps the host path calls; a short workspace and an odd height are refused before anything is launched."""
    ops = wino_everywhere
    from scrabble_gan_amd._lib import lib
    L = lib()
    B, H, W, Cin, Cout = 4, 4, 20, 128, 256
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, device=dev, generator=g)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    tile = ops._wino_tile(H, W)
    P = (tile + 2) ** 2
    nbytes = L.sg_wino_workspace_bytes(B, H, W, Cin, Cout, tile)
    T = B * (H // tile) * (W // tile)
    assert L.sg_wino_plane_rows(B, H, W, tile) == -(-T // 128) * 128 and nbytes == 4 * P * (-(-T // 128) * 128) * (Cin + Cout)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    s = ops._stream()
    y = torch.empty(B, H, W, Cout, device=dev)
    u = ops.packed_filter(w, "wino_fwd%d" % tile)
    assert L.sg_conv2d_fwd_wino(x.data_ptr(), u.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout, 1, tile, ws.data_ptr(), nbytes, s) == 0
    assert torch.equal(y, ops.conv2d_fwd(x, w, b, relu_in=True))
    dx = torch.empty(B, H, W, Cin, device=dev)
    ub = ops.packed_filter(w, "wino_bwd%d" % tile)
    assert L.sg_conv2d_bwd_data_wino(dy.data_ptr(), ub.data_ptr(), x.data_ptr(), dx.data_ptr(), B, H, W, Cin, Cout, 0, tile, ws.data_ptr(), nbytes, s) == 0
    assert torch.equal(dx, ops.conv2d_bwd_data(dy, w, (H, W), mask=x))
    nb2 = L.sg_wino_wgrad_workspace_bytes(B, H, W, Cin, Cout, tile)
    assert nb2 == nbytes + 4 * P * Cin * Cout + 4 * 64 * Cout
    ws2 = torch.empty(nb2, device=dev, dtype=torch.uint8)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    assert L.sg_conv2d_bwd_weight_wino(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), None, B, H, W, Cin, Cout, 1, tile, ws2.data_ptr(), nb2, s) == 0
    dw_h, db_h = torch.zeros_like(w), torch.zeros_like(b)
    ops.conv2d_bwd_weight(x, dy, dw_h, relu_in=True, db=db_h)
    _close(dw, dw_h, 1e-5, "dW: C entry vs host path (float-atomic order only)")
    _close(db, db_h, 1e-5, "db: C entry vs host path")
    assert L.sg_conv2d_bwd_weight_wino(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, None, B, H, W, Cin, Cout, 1, tile, ws2.data_ptr(), nb2 - 1, s) == -1
    assert L.sg_conv2d_fwd_wino(x.data_ptr(), u.data_ptr(), None, None, y.data_ptr(), B, H, W, Cin, Cout, 0, tile, ws.data_ptr(), nbytes - 1, s) == -1
    assert L.sg_conv2d_fwd_wino(x.data_ptr(), u.data_ptr(), None, None, y.data_ptr(), B, 3, W, Cin, Cout, 0, tile, ws.data_ptr(), nbytes, s) == -3
    assert L.sg_wino_workspace_bytes(B, 3, W, Cin, Cout, tile) == 0 and L.sg_wino_workspace_bytes(B, H, W, Cin, Cout, 3) == 0


def test_filter_transform_vs_definition(dev):
    """U = G g G^T per channel pair, forward layout [16][Cout][Cin] and mirrored-tap data-grad layout [16][Cin][Cout] (F(2x2, 3x3);
    the F(4x4, 3x3) matrices are generated and checked in fp64 by tools/gen_winograd_f43.py and exercised by the oracle tests above)."""
    from scrabble_gan_amd import ops
    Cin, Cout = 32, 64
    g = torch.Generator(device=dev).manual_seed(3)
    w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    wd = w.double().cpu()
    uf = torch.einsum("ia,abck,jb->ijkc", G, wd, G).reshape(16, Cout * Cin)
    ub = torch.einsum("ia,abck,jb->ijck", G, wd.flip(0, 1), G).reshape(16, Cin * Cout)
    assert (ops.packed_filter(w, "wino_fwd2").double().cpu() - uf).abs().max().item() < 1e-6
    assert (ops.packed_filter(w, "wino_bwd2").double().cpu() - ub).abs().max().item() < 1e-6


def test_sample_results_do_not_depend_on_the_batch_in_deterministic_mode(dev, wino_tile):
    """Deterministic mode (no reduction split): a sample's result is bitwise the same alone and inside a batch, run to run."""
    from scrabble_gan_amd import ops
    H, W, C = 4, 20, 1024
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(40, H, W, C, device=dev, generator=g)
    w = torch.randn(3, 3, C, C, device=dev, generator=g) / 96.0
    ops.set_deterministic(True)
    try:
        a = ops.conv2d_fwd(x, w, relu_in=True)
        b = ops.conv2d_fwd(x, w, relu_in=True)
        c = ops.conv2d_fwd(x[:3].contiguous(), w, relu_in=True)
    finally:
        ops.set_deterministic(False)
    assert torch.equal(a, b) and torch.equal(a[:3], c)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(6, 8, 40, 128, 256), (5, 4, 20, 256, 256)])
def test_weight_grad_reuses_the_forward_transform(dev, wino_everywhere, B, H, W, Cin, Cout):
    """Inside a step (ops.new_step() ... ops.end_step()) the forward launch keeps its transformed input and the weight gradient of the
    same tensor -- or of a batch slice of it, as the per-target sweeps of a fused pass use -- reads it instead of transforming
    relu(x) again: same dW / db as without the registry (float-atomic order only), and the registry is empty after the step."""
    ops = wino_everywhere
    g = torch.Generator(device=dev).manual_seed(B + Cin)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g) / math.sqrt(9 * Cin)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    sc = torch.rand(B, device=dev, generator=g) + 0.5

    def grads(lo, hi):
        dw, db = torch.zeros_like(w), torch.zeros(Cout, device=dev)
        ops.conv2d_bwd_weight(x[lo:hi], dy[lo:hi].contiguous(), dw, relu_in=True, db=db, sample_scale=sc[lo:hi].contiguous())
        return dw, db

    ref = [grads(0, B), grads(2, B), grads(0, 2)]
    ops.new_step()
    try:
        y = ops.conv2d_fwd(x, w, relu_in=True)
        assert ops._WINO_V["map"], "the forward launch kept nothing"
        kept = ops._wino_v_get(x[2:], True, ops._wino_tile(H, W))
        assert kept is not None and kept[1] == -(-(B * (H // ops._wino_tile(H, W)) * (W // ops._wino_tile(H, W))) // 128) * 128
        got = [grads(0, B), grads(2, B), grads(0, 2)]
        assert ops._wino_v_get(x, False, ops._wino_tile(H, W)) is None            # (another operand ReLU: not the same transform)
    finally:
        ops.end_step()
    assert not ops._WINO_V["map"] and not ops._WINO_V["on"]
    for (dw, db), (rw, rb), name in zip(got, ref, ("whole batch", "samples 2..", "samples ..2")):
        _close(dw, rw, 1e-5, "dW with the kept transform, " + name)
        _close(db, rb, 1e-5, "db, " + name)
    _close(y, ops.conv2d_fwd(x, w, relu_in=True), 1e-7, "forward result with V in its own tensor")


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 4, 8, 128, 128), (2, 16, 80, 64, 128), (5, 8, 40, 96, 256), (2, 8, 12, 128, 64), (33, 4, 20, 256, 256)])
def test_pooled_output_epilogue_vs_oracle(dev, B, H, W, Cin, Cout):
    """SG_POOL2_OUT (round 4): the F(4x4) output transform writes avg_pool2x2(conv + bias) as [B, H/2, W/2, N] -- the conv2 -> AVG pool
    pair of a ResNetBlockDown (resnet_ops.py:102-106) -- plain and accumulating; against the fp64 oracle at the forward bound 2e-5.
    Also through ops.conv2d_avgpool_fwd, whose fallback (convolution, then the pooling kernel) must give the same tensor."""
    from scrabble_gan_amd import ops
    from scrabble_gan_amd._lib import call
    g = torch.Generator(device=dev).manual_seed(B * 7 + H + Cout)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g) / math.sqrt(9 * Cin)
    b1 = torch.randn(Cout, device=dev, generator=g)
    ref_full = O.conv2d(torch.relu(x.double().cpu()), w.double().cpu(), b1.double().cpu())
    ref = ref_full.reshape(B, H // 2, 2, W // 2, 2, Cout).mean(dim=(2, 4))
    old = (ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.WINO_ROW_GAIN, ops.FUSE_POOL)
    ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.WINO_ROW_GAIN = {2: 32, 4: 32}, {2: 0, 4: 0}, 0.0
    try:
        assert ops._wino_ok(Cin, Cout, 3, 3, True, H, W, B) and ops._wino_tile(H, W) == 4
        ops.FUSE_POOL = True
        fused = ops.conv2d_avgpool_fwd(x, w, b1, relu_in=True)
        ops.FUSE_POOL = False
        unfused = ops.conv2d_avgpool_fwd(x, w, b1, relu_in=True)
        # the accumulating form, through the C-ABI steps: y += avg_pool(conv + bias)
        tile, P = 4, 36
        T = B * (H // 4) * (W // 4)
        Tp = -(-T // 128) * 128
        V = torch.empty(P * Tp * Cin, device=dev)
        Mt = torch.empty(P * Tp * Cout, device=dev)
        u = ops.packed_filter(w, "wino_fwd4")
        prev = torch.randn(B, H // 2, W // 2, Cout, device=dev, generator=g)
        acc = prev.clone()
        s = ops._stream()
        call("sg_wino_input", x.data_ptr(), V.data_ptr(), B, H, W, Cin, 1, tile, s)
        call("sg_wino_gemm", V.data_ptr(), u.data_ptr(), Mt.data_ptr(), B, H, W, Cin, Cout, tile, s)
        call("sg_wino_output", Mt.data_ptr(), acc.data_ptr(), b1.data_ptr(), None, None, B, H, W, Cout, ops.POOL2_OUT | ops.ACCUM, tile, s)
        # refused combinations: F(2x2), an output ReLU, a mask
        from scrabble_gan_amd._lib import lib
        assert lib().sg_wino_output(Mt.data_ptr(), acc.data_ptr(), None, None, None, B, H, W, Cout, ops.POOL2_OUT | ops.RELU_OUT, tile, s) == -3
        assert lib().sg_wino_output(Mt.data_ptr(), acc.data_ptr(), None, None, None, B, H, W, Cout, ops.POOL2_OUT, 2, s) == -3
    finally:
        ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.WINO_ROW_GAIN, ops.FUSE_POOL = old
    assert fused.shape == (B, H // 2, W // 2, Cout)
    _close(fused, ref, 2e-5, "avg_pool(conv + b), fused epilogue")
    _close(unfused, ref, 2e-5, "avg_pool(conv + b), convolution then pooling kernel")
    _close(acc, ref + prev.double().cpu(), 2e-5, "y += avg_pool(conv + b)")


@pytest.mark.parametrize("B,H,W,C,scaled", [(3, 8, 16, 128, True), (2, 16, 80, 128, False), (5, 8, 40, 256, True), (33, 4, 20, 256, True)])
def test_pooled_gradient_folded_into_the_transforms_vs_oracle(dev, B, H, W, C, scaled):
    """SG_UPS2_IN (round 4): the backward of `conv3x3 -> avg_pool2x2` with the pooled gradient read directly by both gradient transforms --
    d_c = 0.25 * upsample2x2(dout) is never written.  dx (ReLU-masked), dW and db (with per-sample factors and the operand ReLU) against
    the fp64 oracle's autograd through avg_pool(conv(relu(x))), at the bounds of the unfused launches (2e-5 / 1e-4)."""
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(B * 11 + H + C)
    x = torch.randn(B, H, W, C, device=dev, generator=g)
    w = torch.randn(3, 3, C, C, device=dev, generator=g) / math.sqrt(9 * C)
    dout = torch.randn(B, H // 2, W // 2, C, device=dev, generator=g)
    sc = (torch.rand(B, device=dev, generator=g) * 2 - 0.5) if scaled else None
    old = (ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.WINO_ROW_GAIN, ops.WINO4_WGRAD_MIN_TILES)
    ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.WINO_ROW_GAIN, ops.WINO4_WGRAD_MIN_TILES = {2: 32, 4: 32}, {2: 0, 4: 0}, 0.0, 0
    try:
        assert ops.pooled_grad_foldable(B, H, W, C, C, True)
        dw0, db0 = torch.randn(3, 3, C, C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
        dw, db = dw0.clone(), db0.clone()
        dx = ops.conv2d_avgpool_bwd(x, dout, w, x, dw=dw, db=db, sample_scale=sc, relu_in=True)
    finally:
        ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.WINO_ROW_GAIN, ops.WINO4_WGRAD_MIN_TILES = old
    xd = x.double().cpu().requires_grad_(True)
    wd = w.double().cpu().requires_grad_(True)
    bd = torch.zeros(C, dtype=torch.float64, requires_grad=True)
    y = O.conv2d(torch.relu(xd), wd, bd).reshape(B, H // 2, 2, W // 2, 2, C).mean(dim=(2, 4))
    dd = dout.double().cpu()
    # the image gradient carries the plain upstream, the weight / bias gradients the per-sample factors
    gx, = torch.autograd.grad((y * dd).sum(), xd, retain_graph=True)
    dds = dd if sc is None else dd * sc.double().cpu().view(B, 1, 1, 1)
    gw, gb = torch.autograd.grad((y * dds).sum(), (wd, bd))
    _close(dx, gx, 2e-5, "dx through avg_pool and conv (masked by relu(x))")
    _close(dw - dw0, gw, 1e-4, "dW")
    _close(db - db0, gb, 1e-4, "db")
