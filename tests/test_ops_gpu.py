"""GPU parity of every C-ABI kernel family against the CPU oracle (fp64) on seeded inputs.

Tolerances are fp32: the kernels accumulate in fp32 (MFMA f32 = fmaf chain), the oracle in fp64;
`close()` bounds max|gpu - ref| by tol * max|ref| (tol stated per call)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import margins  # noqa: E402

from oracle import scrabble_oracle as O  # noqa: E402  (checker only)


def close(got, ref, tol=2e-5, name=""):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), name
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    margins.record(name, err / scale, tol)
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (rel %.3e > %.1e)" % (name, err, scale, err / scale, tol)


def rnd(gen, *shape):
    return torch.randn(*shape, generator=gen, dtype=torch.float64)


def g32(t, dev):
    return t.float().to(dev).contiguous()


@pytest.fixture()
def gen():
    return torch.Generator().manual_seed(1234)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, same
    (2, 8, 12, 64, 128, 3, True),
    (3, 6, 10, 32, 8, 1, True),       # attention theta/phi-sized 1x1
    (2, 4, 20, 48, 96, 3, True),      # channel counts that are not multiples of the 32/128 tiles
    (1, 16, 24, 128, 256, 3, True),   # several M tiles, 2 N tiles
    (2, 2, 9, 64, 64, 2, False),      # recognizer conv7: 2x2 VALID
    (2, 8, 8, 1, 64, 3, True),        # thin expand (first layer)
    (2, 8, 8, 1, 64, 1, True),        # thin 1x1 shortcut
    (2, 8, 8, 64, 1, 3, True),        # thin contract (generator head)
    (5, 7, 5, 16, 32, 3, True),       # odd spatial dims, M not a multiple of anything
    (3, 9, 21, 1, 64, 3, True),       # thin kernels (round 2: LDS-staged rows / strips): odd dims, a ragged last strip of rows
    (3, 9, 21, 64, 1, 3, True),
    (2, 6, 9, 1, 64, 2, False),       # ... VALID 2x2 (tap window 0..1)
    (2, 6, 9, 64, 1, 2, False),
    (1, 32, 368, 64, 1, 3, True),     # ... the widest bucket (L = 23): the contraction's strip shrinks to fit its LDS planes
    (1, 32, 368, 1, 64, 3, True),
    (3, 4, 9, 256, 256, 3, True),     # 256-channel tiles: ragged last pixel tile, sample boundary inside a tile
    (2, 5, 7, 256, 512, 1, True),     # ... 1x1, two Cout tiles
]


@pytest.fixture(params=["default-route", "f32-v2"])
def f32_route(request):
    """fp32 convs through the default routing (small shapes: first-generation kernels) and with every eligible shape forced
    through the DMA-fed second-generation kernel (ops.F32_V2_MIN_TILES = 0)."""
    from scrabble_gan_amd import ops
    old = ops.F32_V2_MIN_TILES
    ops.F32_V2_MIN_TILES = 0 if request.param == "f32-v2" else old
    yield request.param
    ops.F32_V2_MIN_TILES = old


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,same", CONV_CASES)
def test_conv2d_fwd_bwd(dev, gen, f32_route, B, H, W, Cin, Cout, k, same):
    from scrabble_gan_amd import ops
    x = rnd(gen, B, H, W, Cin).requires_grad_(True)
    w = (rnd(gen, k, k, Cin, Cout) / math.sqrt(k * k * Cin)).requires_grad_(True)
    b = rnd(gen, Cout).requires_grad_(True)
    y = O.conv2d(torch.relu(x), w, b, padding="same" if same else "valid")
    dy = rnd(gen, *y.shape)
    y.backward(dy)
    xg, wg, bg, dyg = g32(x, dev), g32(w, dev), g32(b, dev), g32(dy, dev)
    yg = ops.conv2d_fwd(xg, wg, bg, same=same, relu_in=True)
    close(yg, y, name="fwd")
    # data grad with the ReLU mask of the input fused into the epilogue
    dxg = ops.conv2d_bwd_data(dyg, wg, (H, W), mask=xg, same=same)
    close(dxg, x.grad, name="bwd_data")
    dwg = torch.zeros_like(wg)
    dbf = torch.zeros_like(bg) if Cout > 1 else None          # bias-grad fused into the weight-grad sweep
    ops.conv2d_bwd_weight(xg, dyg, dwg, same=same, relu_in=True, db=dbf)
    close(dwg, w.grad, tol=5e-5, name="bwd_weight")
    if dbf is not None:
        close(dbf, b.grad, tol=5e-5, name="fused bias_grad")
    dbg = torch.zeros_like(bg)
    ops.bias_grad(dyg, dbg)
    close(dbg, b.grad, tol=5e-5, name="bias_grad")
    if Cout > 1:      # per-sample weighting of the weight / bias gradients (shared backward sweeps)
        s = rnd(gen, B)
        w2, b2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
        O.conv2d(torch.relu(x.detach()), w2, b2, padding="same" if same else "valid").backward(dy * s.view(-1, 1, 1, 1))
        dws, dbs = torch.zeros_like(wg), torch.zeros_like(bg)
        ops.conv2d_bwd_weight(xg, dyg, dws, same=same, relu_in=True, db=dbs, sample_scale=g32(s, dev))
        close(dws, w2.grad, tol=5e-5, name="bwd_weight sample_scale")
        close(dbs, b2.grad, tol=5e-5, name="bias_grad sample_scale")


def test_conv2d_bwd_data_transposed_filter_copy(dev, gen):
    """The optional data-grad path through a transposed fp32 filter copy (ops.TRANSPOSED_DGRAD_FILTERS) gives the same dx."""
    from scrabble_gan_amd import ops
    B, H, W, Cin, Cout, k = 2, 8, 12, 128, 256, 3
    dy, w, x = rnd(gen, B, H, W, Cout), rnd(gen, k, k, Cin, Cout) / math.sqrt(k * k * Cin), rnd(gen, B, H, W, Cin)
    xr = x.clone().requires_grad_(True)
    O.conv2d(xr, w, None).backward(dy)
    ref = xr.grad * (x > 0)
    try:
        ops.TRANSPOSED_DGRAD_FILTERS = True
        wg = g32(w, dev)
        wt = ops.packed_filter(wg, "bwd_f32")
        assert torch.equal(wt.cpu(), wg.cpu().permute(0, 1, 3, 2).contiguous())
        close(ops.conv2d_bwd_data(g32(dy, dev), wg, (H, W), mask=g32(x, dev)), ref, 5e-5, "dx via transposed copy")
    finally:
        ops.TRANSPOSED_DGRAD_FILTERS = False
        ops.weights_changed()
    close(ops.conv2d_bwd_data(g32(dy, dev), g32(w, dev), (H, W), mask=g32(x, dev)), ref, 5e-5, "dx")


def test_conv2d_epilogues(dev, gen, f32_route):
    from scrabble_gan_amd import ops
    B, H, W, Cin, Cout = 2, 8, 12, 64, 128
    x, w = rnd(gen, B, H, W, Cin), rnd(gen, 3, 3, Cin, Cout) / 24
    b1, b2, prev = rnd(gen, Cout), rnd(gen, Cout), rnd(gen, B, H, W, Cout)
    xg, wg = g32(x, dev), g32(w, dev)
    y = torch.relu(O.conv2d(x, w, b1 + b2))
    close(ops.conv2d_fwd(xg, wg, g32(b1, dev), g32(b2, dev), relu_out=True), y, name="bias2+relu_out")
    out = g32(prev, dev)
    ops.conv2d_fwd(xg, wg, g32(b1, dev), out=out, accum=True)
    close(out, O.conv2d(x, w, b1) + prev, name="accum")
    # bwd_data: mask then accumulate
    dy, mask, prevx = rnd(gen, B, H, W, Cout), rnd(gen, B, H, W, Cin), rnd(gen, B, H, W, Cin)
    xr = x.clone().requires_grad_(True)
    O.conv2d(xr, w).backward(dy)
    outx = g32(prevx, dev)
    ops.conv2d_bwd_data(g32(dy, dev), wg, (H, W), mask=g32(mask, dev), out=outx, accum=True)
    close(outx, xr.grad * (mask > 0) + prevx, name="bwd_data mask+accum")
    # thin contract with tanh
    w1, bb = rnd(gen, 3, 3, Cin, 1) / 24, rnd(gen, 1)
    close(ops.conv2d_fwd(xg, g32(w1, dev), g32(bb, dev), tanh_out=True), torch.tanh(O.conv2d(x, w1, bb)), name="tanh_out")


CONVT_CASES = [
    (2, 4, 8, 64, 32, 3, (2, 2)),
    (2, 4, 8, 64, 32, 3, (2, 1)),
    (2, 4, 8, 64, 32, 1, (2, 2)),
    (2, 4, 8, 64, 32, 1, (2, 1)),
    (1, 8, 20, 256, 128, 3, (2, 2)),
    (4, 4, 40, 512, 256, 3, (2, 2)),    # the 8-way-shard shape of G.B1: few tiles per parity class -> pre-zeroed output, split reduction
    (4, 8, 80, 256, 128, 3, (2, 1)),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride", CONVT_CASES)
def test_conv2d_transpose(dev, gen, B, H, W, Cin, Cout, k, stride):
    from scrabble_gan_amd import ops
    x = rnd(gen, B, H, W, Cin).requires_grad_(True)
    w = (rnd(gen, k, k, Cout, Cin) / math.sqrt(k * k * Cin)).requires_grad_(True)
    b = rnd(gen, Cout)
    y = O.conv2d_transpose(x, w, b, stride)
    dy = rnd(gen, *y.shape)
    y.backward(dy)
    xg, wg, dyg = g32(x, dev), g32(w, dev), g32(dy, dev)
    close(ops.conv2d_transpose_fwd(xg, wg, g32(b, dev), stride=stride), y, name="fwd")
    close(ops.conv2d_transpose_bwd_data(dyg, wg, stride=stride), x.grad, name="bwd_data")
    dwg = torch.zeros_like(wg)
    ops.conv2d_transpose_bwd_weight(xg, dyg, dwg, stride=stride)
    close(dwg, w.grad, tol=5e-5, name="bwd_weight")


def test_pool_and_elementwise(dev, gen):
    from scrabble_gan_amd import ops
    a, b = rnd(gen, 2, 8, 12, 64), rnd(gen, 2, 8, 12, 64)
    ag, bg = g32(a, dev), g32(b, dev)
    close(ops.avgpool2_add_fwd(ag, bg), O.avg_pool2(a) + O.avg_pool2(b), name="avgpool_add")
    close(ops.avgpool2_add_fwd(ag), O.avg_pool2(a), name="avgpool")
    d = rnd(gen, 2, 4, 6, 64)
    ar = a.clone().requires_grad_(True)
    O.avg_pool2(ar).backward(d)
    close(ops.avgpool2_bwd(g32(d, dev)), ar.grad, name="avgpool_bwd")
    d1 = rnd(gen, 2, 4, 6, 1)
    a1 = rnd(gen, 2, 8, 12, 1).requires_grad_(True)
    O.avg_pool2(a1).backward(d1)
    close(ops.avgpool2_bwd(g32(d1, dev)), a1.grad, name="avgpool_bwd_c1")
    close(ops.add(ag, bg), a + b, name="add")
    close(ops.relu_mask(ag, bg), a * (b > 0), name="relu_mask")
    t = torch.tanh(a)
    close(ops.tanh_bwd(g32(t, dev), bg), b * (1 - t * t), name="tanh_bwd")
    for ph, pw in ((2, 2), (2, 1)):
        ar = a.clone().requires_grad_(True)
        y = O.max_pool(ar, ph, pw)
        dy = rnd(gen, *y.shape)
        y.backward(dy)
        yg, idx = ops.maxpool_fwd(ag, ph, pw)
        close(yg, y, name="maxpool")
        close(ops.maxpool_bwd(g32(dy, dev), idx, ph, pw), ar.grad, name="maxpool_bwd")
    x = rnd(gen, 3, 4, 10, 128).requires_grad_(True)
    y = torch.relu(x).mean(dim=(1, 2))
    dy = rnd(gen, 3, 128)
    y.backward(dy)
    xg = g32(x, dev)
    close(ops.gap_fwd(xg), y, name="gap")
    close(ops.gap_bwd(g32(dy, dev), xg), x.grad, name="gap_bwd")
    sig = torch.tensor([0.37], dtype=torch.float64)
    close(ops.scale_add(ag, bg, g32(sig, dev)), sig * a + b, name="scale_add")
    close(ops.scale(ag, g32(sig, dev)), sig * a, name="scale")
    acc = torch.zeros(1, device=dev)
    ops.dot_accum(ag, bg, acc)
    close(acc, (a * b).sum().reshape(1), tol=1e-4, name="dot")
    s = rnd(gen, 2)
    close(ops.rowscale(ag, g32(s, dev)), a * s.view(2, 1, 1, 1), name="rowscale")
    for N in (1, 53, 64, 1024):
        dyb = rnd(gen, 300, N)
        db = torch.zeros(N, device=dev)
        ops.bias_grad(g32(dyb, dev), db)
        close(db, dyb.sum(0), tol=5e-5, name="bias_grad%d" % N)


@pytest.mark.parametrize("tA,tB", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm(dev, gen, tA, tB):
    from scrabble_gan_amd import ops
    M, N, K = 37, 53, 70
    A = rnd(gen, K, M) if tA else rnd(gen, M, K)
    B = rnd(gen, N, K) if tB else rnd(gen, K, N)
    bias, C0 = rnd(gen, N), rnd(gen, M, N)
    ref = 0.5 * ((A.t() if tA else A) @ (B.t() if tB else B)) + bias + 2.0 * C0
    out = g32(C0, dev)
    ops.gemm(g32(A, dev), g32(B, dev), M, N, K, A.shape[1], B.shape[1], transA=tA, transB=tB, bias=g32(bias, dev), out=out,
             alpha=0.5, beta=2.0)
    close(out, ref, name="gemm")


@pytest.mark.parametrize("per_sample,relu,C", [(True, True, 64), (True, True, 512), (False, True, 64), (False, False, 128)])
def test_batchnorm(dev, gen, per_sample, relu, C):
    from scrabble_gan_amd import ops
    B, H, W = 3, 4, 10
    x = (rnd(gen, B, H, W, C) * 1.7 + 0.3).requires_grad_(True)
    gamma = rnd(gen, B if per_sample else 1, C).requires_grad_(True)
    beta = rnd(gen, B if per_sample else 1, C).requires_grad_(True)
    x_hat, mean, var = O.batch_norm_train(x)
    y = x_hat * gamma.view(-1, 1, 1, C) + beta.view(-1, 1, 1, C)
    if relu:
        y = torch.relu(y)
    dy = rnd(gen, B, H, W, C)
    y.backward(dy)
    xg, gg, bg, dyg = g32(x, dev), g32(gamma, dev), g32(beta, dev), g32(dy, dev)
    sums = ops.bn_stats_sums(xg)
    n = B * H * W
    mg, vg = ops.bn_stats_finalize(sums, n, xg)
    close(mg, mean, name="mean")
    close(vg, var, name="var")
    yg = ops.bn_apply(xg, mg, vg, gg, bg, per_sample, relu)
    close(yg, y, name="apply")
    dgam, dbet, chan = ops.bn_bwd_reduce(dyg, yg, xg, mg, vg, gg, per_sample, relu)
    dxg = ops.bn_bwd_apply(dyg, yg, xg, mg, vg, gg, per_sample, chan, n, relu, True)
    close(dxg, x.grad, tol=5e-5, name="dx")
    if per_sample:
        close(dgam, gamma.grad, tol=5e-5, name="dgamma")
        close(dbet, beta.grad, tol=5e-5, name="dbeta")
    else:
        close(chan[2 * C:3 * C].float(), gamma.grad[0], tol=5e-5, name="dgamma_c")
        close(chan[3 * C:].float(), beta.grad[0], tol=5e-5, name="dbeta_c")
    # inference-mode BN backward (frozen recognizer): dx = dz * gamma * rstd
    mm, mv = rnd(gen, C) * 0.1, torch.rand(C, generator=gen, dtype=torch.float64) + 0.5
    xr = x.detach().clone().requires_grad_(True)
    g1, b1 = gamma.detach()[0], beta.detach()[0]
    yi = (xr - mm) * torch.rsqrt(mv + O.BN_EPS) * g1 + b1
    yi.backward(dy)
    mmg, mvg, g1g, b1g = g32(mm, dev), g32(mv, dev), g32(g1, dev), g32(b1, dev)
    close(ops.bn_apply(xg, mmg, mvg, g1g, b1g, False, False), yi, name="inference apply")
    close(ops.bn_bwd_apply(dyg, None, xg, mmg, mvg, g1g, False, None, 1, False, False), xr.grad, name="inference dx")
    # moving statistics
    m0, v0 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ops.bn_update_moving(m0, v0, mg, vg, n)
    close(m0, 0.01 * mean, name="mm")
    close(v0, 0.99 + 0.01 * var * n / (n - 1), name="mv")


def test_filterbank(dev, gen):
    from scrabble_gan_amd import ops
    B, L, V = 3, 4, 7
    table = (rnd(gen, V, 32, 8192) * 0.05).requires_grad_(True)
    z = rnd(gen, B, 128).requires_grad_(True)
    y = torch.randint(0, V, (B, L), generator=gen)
    seed = O.filter_bank_seed(z[:, :32], y, table)
    assert seed.shape == (B, 4, 4 * L, 512)
    d = rnd(gen, *seed.shape)
    seed.backward(d)
    zg, tg, yg = g32(z, dev), g32(table, dev), y.int().to(dev)
    close(ops.filterbank_fwd(zg, yg, tg), seed, name="seed")
    dt, dz = torch.zeros_like(tg), torch.zeros_like(zg)
    ops.filterbank_bwd(zg, yg, tg, g32(d, dev), dt, dz)
    close(dt, table.grad, tol=5e-5, name="dtable")
    close(dz, z.grad, tol=5e-5, name="dz")


# B = 2: 32 queries / 32 keys per workgroup and the z-split of the key sweep; B = 260 with Nq = 1024: 256 queries per workgroup
# (the full-batch step); (200, 300): several key tiles and ragged tails; (20 | 48, 1280, 320): the discriminator site at the
# 8-way shard sizes -> 64 / 128 queries per workgroup
@pytest.mark.parametrize("B,Nq,Nk", [(2, 128, 32), (2, 384, 96), (2, 640, 160), (2, 300, 77), (3, 200, 300), (260, 1024, 72), (20, 1280, 320), (48, 1280, 320)])
def test_attention(dev, gen, B, Nq, Nk):
    from scrabble_gan_amd import ops
    th = (rnd(gen, B, Nq, 8) * 1.5).requires_grad_(True)
    ph = (rnd(gen, B, Nk, 8) * 1.5).requires_grad_(True)
    g = rnd(gen, B, Nk, 32).requires_grad_(True)
    out = torch.softmax(th @ ph.transpose(1, 2), dim=-1) @ g
    d = rnd(gen, *out.shape)
    out.backward(d)
    thg, phg, gg = g32(th, dev), g32(ph, dev), g32(g, dev)
    og, lse = ops.attention_fwd(thg, phg, gg)
    close(og, out, name="attn out")
    dth, dph, dg = ops.attention_bwd(thg, phg, gg, og, lse, g32(d, dev))
    close(dth, th.grad, tol=5e-5, name="dtheta")
    close(dph, ph.grad, tol=5e-5, name="dphi")
    close(dg, g.grad, tol=5e-5, name="dg")


@pytest.mark.parametrize("L", [1, 3, 10, 23])      # 23: the widest bucket of config c4 (T = 91, S = 47, > 64 KB of LDS)
def test_softmax_ctc(dev, gen, L):
    from scrabble_gan_amd import ops
    B, C = 4, 53
    T = 4 * L - 1
    logits = (rnd(gen, B, T, C) * 2).requires_grad_(True)
    labels = torch.randint(0, C - 1, (B, L), generator=gen)
    labels[0, :] = labels[0, 0]          # repeated characters force the blank transitions
    cost = O.ctc_batch_cost(labels, torch.softmax(logits, -1), T, L)
    cost.sum().backward()
    loss, dl = ops.softmax_ctc(g32(logits, dev), labels.int().to(dev), T, L)
    close(loss, cost[:, 0], name="ctc loss")
    close(dl, logits.grad, tol=1e-4 if L <= 10 else 3e-4, name="ctc dlogits")     # fp32 log-space alpha/beta over T = 4L-1 frames


@pytest.mark.parametrize("mode,balance", [(0, False), (0, True), (1, False), (1, True)])
def test_loss_head(dev, gen, mode, balance):
    from scrabble_gan_amd import ops
    B = 37
    v = [(rnd(gen, B, 1) * 1.5).requires_grad_(True) for _ in range(5)]
    r_f = (torch.rand(B, 1, generator=gen, dtype=torch.float64) * 30 + 5).requires_grad_(True)
    r_r = torch.rand(B, 1, generator=gen, dtype=torch.float64) * 30
    fn = O.hinge if mode == 0 else O.not_saturating
    d_loss, d_lr, d_lf, g_loss, s_loss, s_a, s_b = fn(*v)
    g_bal, r_bal, alpha, r_std, g_std = O.apply_gradient_balancing(r_f, g_loss)
    g_added = g_loss + r_f
    g_final = g_bal if balance else g_added
    ref = [r_f.mean(), r_r.mean(), r_bal.mean(), g_loss.mean(), g_added.mean(), g_bal.mean(), d_loss.mean(), d_lr.mean(),
           d_lf.mean(), g_final.mean(), torch.tensor(1.0), r_std, g_std, s_loss.mean(), s_a.mean(), s_b.mean()]
    gD = torch.autograd.grad(d_loss.sum(), [v[0], v[1]], retain_graph=True)
    gS = torch.autograd.grad(s_loss.sum(), [v[2], v[3]], retain_graph=True)
    gG = torch.autograd.grad(g_final.sum(), [v[1], v[3], r_f], retain_graph=True, allow_unused=True)
    dv = [g32(t.detach().reshape(-1), dev) for t in v]
    rfg, rrg = g32(r_f.detach().reshape(-1), dev), g32(r_r.reshape(-1), dev)
    sums = ops.loss_sums(*dv, rfg, rrg, mode)
    scalars, outs = ops.loss_grads(*dv, rfg, mode, balance, 1.0, sums)
    close(scalars, torch.stack([t.detach().reshape(()) for t in ref]), tol=1e-5, name="scalars")
    refs = [gD[0], gD[1], gS[0], gS[1], gG[0], gG[1] if gG[1] is not None else torch.zeros(B, 1), gG[2]]
    for i, (o, r) in enumerate(zip(outs, refs)):
        close(o, r.reshape(-1), tol=2e-4 if balance else 1e-5, name="upstream%d" % i)
    # shared-sweep factors: u * w = d sum(d_loss)/d d_f (resp. s_loss/s_f), u * x = d sum(g_final)/d d_f (resp. s_f); |w|,|x| <= 1
    for sh, a, c, nm in ((outs[7], refs[1], refs[4], "D"), (outs[8], refs[3], refs[5], "S")):
        assert sh.shape == (3, B) and sh[1:].abs().max().item() <= 1.0 + 1e-6
        close(sh[0] * sh[1], a.reshape(-1), tol=2e-4 if balance else 1e-5, name="shared w " + nm)
        close(sh[0] * sh[2], c.reshape(-1), tol=2e-4 if balance else 1e-5, name="shared x " + nm)


def test_adam_rmsprop_spectral(dev, gen):
    from scrabble_gan_amd import ops
    n = 1003
    p, g = rnd(gen, n), rnd(gen, n)
    P, st = {"w": p.clone()}, {}
    pg, gg = g32(p, dev), g32(g, dev)
    m, v = torch.zeros_like(pg), torch.zeros_like(pg)
    for t in (1, 2, 3):
        O.adam_update(P, {"w": g * t}, st, 2e-4, 0.0, 0.999)
        lr_t = 2e-4 * math.sqrt(1 - 0.999 ** t) / (1 - 0.0 ** t)
        ops.adam_update(pg, gg * t, m, v, lr_t, 0.0, 0.999)
    close(pg, P["w"], tol=1e-6, name="adam")
    P2, st2 = {"w": p.clone()}, {}
    pg2, ms = g32(p, dev), torch.zeros(n, device=dev)
    for t in (1, 2):
        O.rmsprop_update(P2, {"w": g}, st2, 2e-4)
        ops.rmsprop_update(pg2, gg, ms, 2e-4)
    close(pg2, P2["w"], tol=1e-6, name="rmsprop")
    # gradients far below sqrt(eps): epsilon sits OUTSIDE the root (TF 2.1 dense path), hand-computed first step
    gs = torch.tensor([1.0, -1e-3, 1e-6, 1e-9] * 2, dtype=torch.float64)
    pz, msz = torch.zeros(8, device=dev), torch.zeros(8, device=dev)
    ops.rmsprop_update(pz, g32(gs, dev), msz, 2e-4)
    want = -2e-4 * gs / (math.sqrt(0.1) * gs.abs() + 1e-7)
    assert ((pz.double().cpu() - want).abs() <= 2e-6 * want.abs()).all(), (pz, want)
    w, u = rnd(gen, 3, 3, 16, 40), rnd(gen, 1, 40)
    close(ops.spectral_norm(g32(w, dev), g32(u.reshape(-1), dev)), O.spectral_norm(w, u), tol=2e-5, name="spectral_norm")
    close(ops.spectral_norm(g32(w, dev), g32(u.reshape(-1), dev), 3), O.spectral_norm(w, u, 3), tol=2e-5, name="spectral_norm3")


def test_launch_status_keeps_an_earlier_failure(dev):
    """The C-ABI convention 'returns 0 or a negative code' must hold for entry points that queue several launches: a launch
    refused in the MIDDLE of the sequence (2048 threads per workgroup) is reported although the launch after it succeeds
    (sg_common.h: SG_KERNEL folds every launch's status into the flag sg_launch_status() returns), and the flag is reset."""
    from scrabble_gan_amd import _lib, ops
    scratch = torch.zeros(256, device=dev)
    rc = _lib.lib().sg_selftest_launch_status(scratch.data_ptr(), ops._stream())
    assert rc == -2, rc
    torch.cuda.synchronize()
    assert scratch[0].item() == 3.0                     # the launches around the refused one ran
    out = ops.add(torch.ones(8, device=dev), torch.ones(8, device=dev))     # the next entry point starts from a clean slate
    assert out.sum().item() == 16.0
    with pytest.raises(TypeError):                      # arity is checked on the host before anything reaches the GPU
        _lib.call("sg_add", scratch.data_ptr(), scratch.data_ptr(), scratch.data_ptr(), 8)


def test_one_adder_variants_of_deterministic_mode_vs_oracle(dev):
    """`sg_set_deterministic(1)` switches the float-atomic reductions outside the convolutions to one-adder forms (round 4): BatchNorm's
    backward reduce with one workgroup per sample, the filter bank's dz kernel with one workgroup per sample (a different kernel:
    k_filterbank_bwd_dz_det), the attention key sweep without the query split, the sigma dot product on one workgroup.  The same
    oracle checks as in default mode, run with the switch on."""
    from scrabble_gan_amd import ops
    ops.set_deterministic(True)
    try:
        test_filterbank(dev, torch.Generator().manual_seed(1234))
        for per_sample, relu, C in ((True, True, 64), (False, False, 512)):
            test_batchnorm(dev, torch.Generator().manual_seed(1234), per_sample, relu, C)
        for B, Nq, Nk in ((2, 640, 160), (3, 200, 300), (2, 1280, 320), (1, 5120, 1280)):      # (the last two would split the query range)
            test_attention(dev, torch.Generator().manual_seed(1234), B, Nq, Nk)
        a, b = torch.randn(4096, device=dev), torch.randn(4096, device=dev)
        out = torch.zeros(1, device=dev)
        from scrabble_gan_amd._lib import call
        call("sg_dot_accum", a.data_ptr(), b.data_ptr(), out.data_ptr(), 4096, ops._stream())
        close(out, (a.double() * b.double()).sum().reshape(1), tol=1e-5, name="dot (one workgroup)")
    finally:
        ops.set_deterministic(False)
