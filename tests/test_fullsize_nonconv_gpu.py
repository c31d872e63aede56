"""Every NON-convolution kernel of the step against the CPU oracle AT THE REAL LAUNCH GEOMETRY.

tests/test_ops_gpu.py checks these kernels at toy sizes; the launch configuration of most of them is a function of the
problem size (attention: workgroup width / key-sweep width / z-split from B * Nq and B * Nk; BatchNorm: number of fp32
partials per channel; filter bank: ballot scan over B * L labels; Adam: grid-stride over the flat buffer), so the sizes of
BASELINE configs c2 (bs 128) and the 8-way shard (bs 16) are launched here as they are in a step and compared with the
oracle.  Where samples are independent the oracle evaluates the first and last two samples of the batch (edge technique
of tests/test_fullsize_gpu.py); reductions over the batch (BN statistics, filter-bank table gradient, bias gradient) are
compared on the whole batch in fp64.

Tolerances: max|got - ref| <= tol * max|ref| (fp32 kernels vs the fp64 oracle), tol stated per check."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import margins  # noqa: E402

from oracle import scrabble_oracle as O  # noqa: E402  (checker only)


def _close(got, ref, tol, name):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), name
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item() + 1e-30
    margins.record(name, err / scale, tol)
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (rel %.3e > %.1e)" % (name, err, scale, err / scale, tol)


def _edge(t, n=2):
    return torch.cat([t[:n], t[-n:]], 0).double().cpu()


def _close_l2(got, ref, tol, name):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    err, scale = (got - ref).norm().item(), ref.norm().item() + 1e-30
    assert err <= tol * scale, "%s: ||err|| %.3e vs ||ref|| %.3e (rel %.3e > %.1e)" % (name, err, scale, err / scale, tol)


# ---------------------------------------------------------------------------------------------------------------------
# (a) attention of NonLocalBlock (arch_ops.py:51-61).  G.NL3 at L = 10 is the largest attention launch of the step:
# Nq = 32*160 = 5120 queries, Nk = 1280 keys per sample; L = 23: 11776 x 2944.  The discriminator-side site (1280 x 320)
# at the fused-pass batch 384.  out / dtheta 2e-5 / 5e-5; dphi / dg (sums over up to 11776 queries) 5e-5.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Nq,Nk,edge", [(128, 5120, 1280, 2), (16, 5120, 1280, 2), (8, 11776, 2944, 1), (384, 1280, 320, 2)])
def test_attention_vs_oracle_at_launch_geometry(dev, B, Nq, Nk, edge):
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(B + Nq)
    # theta / phi as the 1x1 convs of an orthogonal kernel produce them from O(1) activations: logits of a few units
    th = torch.randn(B, Nq, 8, device=dev, generator=g) * 1.2
    ph = torch.randn(B, Nk, 8, device=dev, generator=g) * 1.2
    gg = torch.randn(B, Nk, 32, device=dev, generator=g)
    d = torch.randn(B, Nq, 32, device=dev, generator=g)
    out, lse = ops.attention_fwd(th, ph, gg)
    dth, dph, dg = ops.attention_bwd(th, ph, gg, out, lse, d)
    the, phe, gge = (_edge(t, edge).requires_grad_(True) for t in (th, ph, gg))
    ref = torch.softmax(the @ phe.transpose(1, 2), dim=-1) @ gge            # plain softmax(theta phi^T) g, no 1/sqrt(d)
    ref.backward(_edge(d, edge))
    _close(_edge(out, edge), ref, 2e-5, "attention out")
    _close(_edge(dth, edge), the.grad, 5e-5, "dtheta")
    _close(_edge(dph, edge), phe.grad, 5e-5, "dphi")
    _close(_edge(dg, edge), gge.grad, 5e-5, "dg")


# ---------------------------------------------------------------------------------------------------------------------
# (b) ConditionalBatchNorm (resnet_ops.py:14-27) on the WHOLE batch in fp64: statistics over 655 360 / 20 480 pixels (fp32
# partials -> fp64 combine), per-sample affine + ReLU, backward reduce (per-sample dgamma / dbeta) and backward apply.
# mean / var 1e-5 (var: 2e-5), y 2e-5, dx / dgamma / dbeta 5e-5.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,C,per_sample", [(128, 32, 160, 64, True), (128, 4, 40, 512, True), (128, 32, 160, 64, False),
                                                  (16, 16, 160, 128, True)])
def test_batchnorm_vs_oracle_at_launch_geometry(dev, B, H, W, C, per_sample):
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(B + H + C)
    chan_mu = torch.randn(C, device=dev, generator=g) * 0.8                 # channel means of the order of the spread: the
    x = torch.randn(B, H, W, C, device=dev, generator=g) * 1.3 + chan_mu    # E[x^2] - E[x]^2 cancellation takes part
    gamma = torch.randn(B if per_sample else 1, C, device=dev, generator=g)
    beta = torch.randn(B if per_sample else 1, C, device=dev, generator=g) * 0.5
    dy = torch.randn(B, H, W, C, device=dev, generator=g)
    n = B * H * W
    sums = ops.bn_stats_sums(x)
    mean, var = ops.bn_stats_finalize(sums, n, x)
    gk, bk = (gamma, beta) if per_sample else (gamma[0].contiguous(), beta[0].contiguous())
    y = ops.bn_apply(x, mean, var, gk, bk, per_sample, True)
    dgam, dbet, chan = ops.bn_bwd_reduce(dy, y, x, mean, var, gk, per_sample, True)
    dx = ops.bn_bwd_apply(dy, y, x, mean, var, gk, per_sample, chan, n, True, True)
    xr = x.double().cpu().requires_grad_(True)
    gr, br = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    x_hat, m_ref, v_ref = O.batch_norm_train(xr)
    yr = torch.relu(x_hat * gr.view(-1, 1, 1, C) + br.view(-1, 1, 1, C))
    yr.backward(dy.double().cpu())
    _close(mean, m_ref, 1e-5, "batch mean")
    _close(var, v_ref, 2e-5, "batch variance")
    _close(y, yr, 2e-5, "CBN + ReLU")
    _close(dx, xr.grad, 5e-5, "dx")
    if per_sample:
        _close(dgam, gr.grad, 5e-5, "dgamma [B,C]")
        _close(dbet, br.grad, 5e-5, "dbeta [B,C]")
    else:
        _close(chan[2 * C:3 * C], gr.grad[0], 5e-5, "dgamma [C]")
        _close(chan[3 * C:], br.grad[0], 5e-5, "dbeta [C]")


# ---------------------------------------------------------------------------------------------------------------------
# (c) SpatialEmbedding + seed (net_architecture.py:265-271, arch_ops.py:89-90) at the real vocabulary (V = 52, the 54.5 MB
# table) and B * L = 1280 / 2944 labels: forward seed, dz and the atomic-free table gradient (ballot scan over all labels;
# classes that do not occur must stay untouched).  seed 2e-5; dz / dtable (sums of up to ~60 label hits x 512 terms) 5e-5.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,L", [(128, 10), (128, 23), (16, 10)])
def test_filterbank_vs_oracle_at_launch_geometry(dev, B, L):
    from scrabble_gan_amd import ops
    V = 52
    g = torch.Generator(device=dev).manual_seed(B * L)
    table = torch.randn(V, 32, 8192, device=dev, generator=g) * 0.02
    z = torch.randn(B, 128, device=dev, generator=g)
    y = torch.randint(0, V - 2, (B, L), device=dev, generator=g, dtype=torch.int32)     # classes V-2, V-1 never occur
    y[0, :] = y[0, 0]                                                                    # one word of a repeated character
    d = torch.randn(B, 4, 4 * L, 512, device=dev, generator=g)
    seed = ops.filterbank_fwd(z, y, table)
    sentinel = 7.0
    dt = torch.zeros_like(table)
    dt[V - 2:] = sentinel
    dz = torch.zeros_like(z)
    ops.filterbank_bwd(z, y, table, d, dt, dz)
    assert torch.equal(dt[V - 2:], torch.full_like(dt[V - 2:], sentinel)), "table rows of absent classes were written"
    tc, zc, yc, dc = table.double().cpu(), z.double().cpu(), y.cpu(), d.double().cpu()
    dt_ref = torch.zeros_like(tc)
    chunk = 16
    for lo in range(0, B, chunk):                        # the oracle's [B,L,32,8192] gather, 16 samples at a time
        tr = tc.clone().requires_grad_(True)
        zr = zc[lo:lo + chunk].clone().requires_grad_(True)
        s_ref = O.filter_bank_seed(zr[:, :32], yc[lo:lo + chunk], tr)
        s_ref.backward(dc[lo:lo + chunk])
        _close(seed[lo:lo + chunk], s_ref, 2e-5, "seed [%d:%d]" % (lo, lo + chunk))
        _close(dz[lo:lo + chunk], zr.grad, 5e-5, "dz [%d:%d]" % (lo, lo + chunk))
        dt_ref += tr.grad
    _close(dt[:V - 2], dt_ref[:V - 2], 5e-5, "dtable")


# ---------------------------------------------------------------------------------------------------------------------
# (d) pools, bias gradient, Adam, CTC at step sizes.
# ---------------------------------------------------------------------------------------------------------------------
def test_pools_vs_oracle_at_launch_geometry(dev):
    """avg-pool + residual add (resnet_ops.py:105-114) at D.B1's fused-pass size, max-pool fwd/bwd of the NonLocalBlock
    (arch_ops.py:47,58) and of the recognizer ((2,1) pools, net_architecture.py:38,47), GAP + ReLU; edge samples, 1e-6
    (pure selections / 4-term sums) and 2e-5 for GAP (sums over 80 pixels)."""
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(91)
    B = 384
    a = torch.randn(B, 32, 160, 64, device=dev, generator=g)
    b = torch.randn(B, 32, 160, 64, device=dev, generator=g)
    out = ops.avgpool2_add_fwd(a, b)
    _close(_edge(out), O.avg_pool2(_edge(a)) + O.avg_pool2(_edge(b)), 1e-6, "avgpool2(a) + avgpool2(b)")
    dout = torch.randn(B, 16, 80, 64, device=dev, generator=g)
    ar = _edge(a).requires_grad_(True)
    O.avg_pool2(ar).backward(_edge(dout))
    _close(_edge(ops.avgpool2_bwd(dout)), ar.grad, 1e-6, "avgpool2 backward")
    del a, b, out
    for (Bp, H, W, C, ph, pw) in ((128, 32, 160, 32, 2, 2), (128, 32, 160, 8, 2, 2), (256, 8, 40, 256, 2, 1)):
        x = torch.randn(Bp, H, W, C, device=dev, generator=g)
        yk, idx = ops.maxpool_fwd(x, ph, pw)
        xr = _edge(x).requires_grad_(True)
        yr = O.max_pool(xr, ph, pw)
        dy = torch.randn(Bp, H // ph, W // pw, C, device=dev, generator=g)
        yr.backward(_edge(dy))
        _close(_edge(yk), yr, 1e-6, "maxpool %dx%d C=%d" % (ph, pw, C))
        _close(_edge(ops.maxpool_bwd(dy, idx, ph, pw)), xr.grad, 1e-6, "maxpool backward %dx%d C=%d" % (ph, pw, C))
    x = torch.randn(384, 4, 20, 1024, device=dev, generator=g)
    xr = _edge(x).requires_grad_(True)
    yr = torch.relu(xr).mean(dim=(1, 2))
    dy = torch.randn(384, 1024, device=dev, generator=g)
    yr.backward(_edge(dy))
    _close(_edge(ops.gap_fwd(x)), yr, 2e-5, "ReLU + GAP")
    _close(_edge(ops.gap_bwd(dy, x)), xr.grad, 2e-5, "ReLU + GAP backward")


@pytest.mark.parametrize("M,N", [(128 * 32 * 160, 64), (128 * 8 * 80, 256), (384 * 4 * 20, 1024), (128 * 39, 53)])
def test_bias_grad_vs_oracle_at_launch_geometry(dev, M, N):
    """Column sums over up to 655 360 rows (the ConvT / shortcut biases of block_up_bwd, the recognizer's dense bias):
    fp32 partial sums with float atomics vs fp64, 5e-5 of max|ref| (|ref| ~ sqrt(M))."""
    from scrabble_gan_amd import ops
    g = torch.Generator(device=dev).manual_seed(M % 9973 + N)
    dy = torch.randn(M, N, device=dev, generator=g) + 0.1
    db = torch.randn(N, device=dev, generator=g)
    ref = db.double().cpu() + dy.double().sum(0).cpu()            # (fp64 sum on the device tensor: torch, checker side only)
    ops.bias_grad(dy, db)
    _close(db, ref, 5e-5, "bias gradient %dx%d" % (M, N))


def test_adam_on_the_generator_sized_flat_buffer(dev):
    """One fused Adam launch over a 53.7 M-element flat buffer (G's parameter count), three steps, against the
    oracle's Keras Adam on the whole buffer in fp64: 1e-6 of max|p| (the update is elementwise: only fp32 rounding).  The
    second moment is held to 3e-5: the kernel forms 1 - beta_2 in fp32 as TF's fp32 ResourceApplyAdam does
    (1 - fp32(0.999) = 0.00099998713: 1.3e-5 relative to the oracle's fp64 0.001; measured 1.28e-5)."""
    from scrabble_gan_amd import ops
    n = 53_680_004
    g = torch.Generator(device=dev).manual_seed(5)
    p = torch.randn(n, device=dev, generator=g) * 0.05
    grads = [torch.randn(n, device=dev, generator=g) * (10.0 ** -k) for k in (1, 3, 5)]
    P, st = {"w": p.double().cpu()}, {}
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t, gr in enumerate(grads, 1):
        O.adam_update(P, {"w": gr.double().cpu()}, st, 2e-4, 0.0, 0.999)
        ops.adam_update(p, gr, m, v, 2e-4 * math.sqrt(1 - 0.999 ** t), 0.0, 0.999)
    _close(p, P["w"], 1e-6, "Adam, 53.7 M parameters, 3 steps")
    _close(m, st["m.w"], 1e-6, "first moment")
    _close(v, st["v.w"], 3e-5, "second moment")


@pytest.mark.parametrize("B,L", [(128, 10), (256, 10), (128, 23)])
def test_ctc_vs_oracle_at_launch_geometry(dev, B, L):
    """softmax + K.ctc_batch_cost (net_architecture.py:55-64) for a whole batch (one wave per sample, B waves): loss 2e-5,
    d/dlogits 1e-4 (L = 10) / 3e-4 (L = 23: fp32 log-space alpha/beta over 91 frames)."""
    from scrabble_gan_amd import ops
    gen = torch.Generator().manual_seed(B + L)
    C, T = 53, 4 * L - 1
    logits = (torch.randn(B, T, C, generator=gen, dtype=torch.float64) * 2).requires_grad_(True)
    labels = torch.randint(0, C - 1, (B, L), generator=gen)
    labels[B // 2, :] = labels[B // 2, 0]
    cost = O.ctc_batch_cost(labels, torch.softmax(logits, -1), T, L)
    cost.sum().backward()
    loss, dl = ops.softmax_ctc(logits.detach().float().to(dev), labels.int().to(dev), T, L)
    _close(loss, cost[:, 0], 2e-5, "CTC cost")
    _close(dl, logits.grad, 1e-4 if L <= 10 else 3e-4, "d cost / d logits")


# ---------------------------------------------------------------------------------------------------------------------
# (e) whole NETWORKS at the launch geometry of the step.  D / S have no BatchNorm and the frozen R normalises with moving
# statistics, so samples are independent: the fused passes of the bs-128 step (384 / 256 samples of 32 x 160) run as they do in
# train_step -- every kernel at its real grid, the fused multi-call NonLocalBlock segments, the shared backward sweep with
# per-sample weight factors -- and the oracle evaluates the first / last samples of each call: logits / CTC costs 1e-4; the image
# gradient 1e-2 in the L2 norm and 3e-2 of max|ref| per pixel (ReLU / max-pool near-ties re-route single pixels in fp32).
# ---------------------------------------------------------------------------------------------------------------------
def _perturbed(model, gen):
    w = model.store.export()
    for k, v in w.items():
        if k.endswith(".sigma"):
            w[k] = torch.tensor(0.3)
        elif k.endswith(".b") or k.endswith(".beta"):
            w[k] = torch.randn(v.shape, generator=gen) * 0.1
        elif k.endswith(".gamma"):
            w[k] = 1 + torch.randn(v.shape, generator=gen) * 0.1
        elif k.endswith(".mm"):
            w[k] = torch.randn(v.shape, generator=gen) * 0.1
        elif k.endswith(".mv"):
            w[k] = 1 + torch.rand(v.shape, generator=gen) * 0.2
    model.store.load(w)
    return {k: v.double() for k, v in w.items()}


def test_discriminator_fused_pass_vs_oracle_at_launch_geometry(dev):
    """make_discriminator (net_architecture.py:299-355) on the fused fake | style | real pass of the bs-128 step (3 x 128 samples,
    one NonLocalBlock kernel set per call), forward and the image gradient of a backward sweep with want_dw (per-sample weight
    factors as the shared sweep passes them)."""
    from scrabble_gan_amd import net_architecture as NA, nn
    NA.configure(device=dev, seed=3)
    gen = torch.Generator().manual_seed(77)
    D = NA.make_discriminator((32, 160, 1), None, "B1", vis_model=False)
    P = _perturbed(D, gen)
    B = 128
    g = torch.Generator(device=dev).manual_seed(78)
    xs = [torch.rand(B, 32, 160, 1, device=dev, generator=g) * 2 - 1 for _ in range(3)]
    nls_o = [O.init_nonlocal(64, gen) for _ in range(3)]
    nls_g = [{k: v.float().to(dev).contiguous() for k, v in d.items()} for d in nls_o]
    logits, ctx, bounds = D.forward_multi(xs, nls_g)
    up = torch.randn(3 * B, device=dev, generator=g)
    wsc = torch.rand(3 * B, device=dev, generator=g) + 0.5
    D.store.zero_grad()
    dx = D.backward(ctx, up, want_dx=True, want_dw=True, wscale=wsc)
    for c, (x, nlo, (lo, hi)) in enumerate(zip(xs, nls_o, bounds)):
        xe = _edge(x).requires_grad_(True)
        ref = O.discriminator(xe, P, nlo)
        ue = _edge(up[lo:hi])
        (ref[:, 0] * ue).sum().backward()
        _close(_edge(logits[c]), ref, 1e-4, "logits of call %d (first / last 2 samples)" % c)
        # (the image gradient crosses 9 ReLU masks and 2 max-pools over 2.6 M activations per sample: near-ties resolved
        #  differently in fp32 re-route single pixels; measured 1.2e-3 ... 6.3e-3 of max|ref| at single pixels)
        _close(_edge(dx[lo:hi]), xe.grad, 3e-2, "image gradient of call %d" % c)
        _close_l2(_edge(dx[lo:hi]), xe.grad, 1e-2, "image gradient of call %d (L2)" % c)
    assert torch.isfinite(D.store.grad).all()


def test_recognizer_fused_pass_vs_oracle_at_launch_geometry(dev):
    """make_recognizer + K.ctc_batch_cost (net_architecture.py:9-79) frozen (inference-mode BatchNorm) on the fused
    fake | real pass of the bs-128 step (256 words of 10 characters): CTC cost per sample and the image gradient of the fake half."""
    from scrabble_gan_amd import net_architecture as NA
    NA.configure(device=dev, seed=3)
    gen = torch.Generator().manual_seed(79)
    R = NA.make_recognizer((32, 160, 1), None, 53, vis_model=False)
    P = _perturbed(R, gen)
    R.trainable = False
    B, L = 128, 10
    g = torch.Generator(device=dev).manual_seed(80)
    xs = [torch.rand(B, 32, 160, 1, device=dev, generator=g) * 2 - 1 for _ in range(2)]
    labs = [torch.randint(0, 52, (B, L), device=dev, generator=g, dtype=torch.int32) for _ in range(2)]
    T = 4 * L - 1
    (loss_f, loss_r), ctx, bounds = R.forward_multi(xs, labs, T, L, training=True)
    up = torch.rand(B, device=dev, generator=g) + 0.5
    dx = R.backward(R.slice_ctx(ctx, 0, B), up, want_dx=True, want_dw=False)
    from tests import step_fixture as F
    from tests.test_nets_gpu import _with_hip_decisions
    for c, (x, lab, loss) in enumerate(zip(xs, labs, (loss_f, loss_r))):
        xe = _edge(x).requires_grad_(True)
        lo = bounds[c][0]
        # frozen BatchNorm: samples are independent, so the four edge samples can be checked on their own -- under the ReLU /
        # max-pool decisions the HIP pass took for THEM (1.3 M activations per sample: a handful of near-ties fall the other way
        # in fp32 and re-route single pixels of the image gradient by up to 1.2e-2 of max|ref|; with the decisions imposed the
        # fixed 1e-3 gradient bound of tests/test_nets_gpu.py::test_recognizer holds)
        relu, pool = F.recognizer_decisions(R.slice_ctx(ctx, lo, lo + B))
        ref = _with_hip_decisions(lambda: O.recognizer(xe, _edge_i(lab), T, L, P, bn_training=False), ([_edge(t) for t in relu], [_edge(t) for t in pool]),
                                  "frozen recognizer, call %d" % c)
        _close(_edge(loss.reshape(-1, 1)), ref, 1e-4, "CTC cost of call %d (first / last 2 samples)" % c)
        if c == 0:
            (ref[:, 0] * _edge(up)).sum().backward()
            _close(_edge(dx), xe.grad, 1e-3, "image gradient through the frozen recognizer")


def _edge_i(t, n=2):
    return torch.cat([t[:n], t[-n:]], 0).long().cpu()


def test_generator_vs_oracle_at_the_shard_geometry(dev):
    """make_generator (net_architecture.py:182-296) forward AND backward on a WHOLE batch at the launch geometry of the 8-way
    data-parallel shard (B = 16 words of 10 characters, 160-wide style images): style encoder, z, filter bank, the three
    ResNetBlockUp with ConditionalBatchNorm over the batch's own statistics (10 240 ... 81 920 pixels), the NonLocalBlock at
    5120 x 1280, final BN + conv + tanh -- image 1e-4; every gradient tensor with the criterion of tests/test_nets_gpu.py::
    test_generator (fixed max-norm bound against the fp64 oracle under the HIP pass's own ReLU / max-pool decisions, which must
    equal the oracle's except at near-ties); moving statistics."""
    from scrabble_gan_amd import net_architecture as NA
    from tests import step_fixture as F
    from tests.test_nets_gpu import GEN_GRAD_TOL, _with_hip_decisions, close, leaves, net_atol
    NA.configure(device=dev, seed=3)
    gen = torch.Generator().manual_seed(7)
    G = NA.make_generator(128, (32, 160, 1), (32, 8192), None, "B3", 52, vis_model=False)
    P = _perturbed(G, gen)
    B, L = 16, 10
    style = torch.rand(B, 32, 160, 1, generator=gen, dtype=torch.float64) * 2 - 1
    y = torch.randint(0, 52, (B, L), generator=gen)
    nls_o, nlu_o = O.init_nonlocal(64, gen), O.init_nonlocal(64, gen)
    nls_g, nlu_g = ({k: v.float().to(dev).contiguous() for k, v in d.items()} for d in (nls_o, nlu_o))
    dimg = torch.randn(B, 32, 16 * L, 1, generator=gen, dtype=torch.float64)
    img, ctx = G.forward(style.float().to(dev), y.int().to(dev), nls_g, nlu_g, training=True)
    G.store.zero_grad()
    G.backward(ctx, dimg.float().to(dev))
    lv = leaves(P)
    stats = {}
    ref = _with_hip_decisions(lambda: O.generator(style, y, P, nls_o, nlu_o, bn_stats=stats), F.generator_decisions(ctx), "generator at the shard geometry")
    (ref * dimg).sum().backward()
    close(img, ref, 1e-4, "image")
    at = net_atol([v.grad for v in lv.values()])
    for k, v in lv.items():
        close(G.store.g[k], v.grad, GEN_GRAD_TOL, "grad " + k, at)
    for pre in ("B1.cbn1", "B3.cbn2", "bn"):
        st = stats[pre]
        n = st["count"]
        close(G.store.p[pre + ".mm"], 0.99 * P[pre + ".mm"] + 0.01 * st["mean"], 1e-4, "moving mean " + pre)
        close(G.store.p[pre + ".mv"], 0.99 * P[pre + ".mv"] + 0.01 * st["var"] * n / (n - 1), 1e-4, "moving variance " + pre)
