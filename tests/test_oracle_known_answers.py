"""Pins for the CPU oracle (CPU-only).  The reference ships no tests or golden vectors and TensorFlow is
absent (parity unpinned by the reference), so the oracle is pinned by
  (i)  analytic known answers, and
  (ii) independent naive numpy restatements of the TF semantics of SURVEY Appendix A
       (loop convolutions, brute-force CTC path enumeration, the raw-reshape seed layout)."""
import itertools
import math

import numpy as np
import torch

from oracle import scrabble_oracle as O

D = torch.float64


def g(seed=0):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------- naive restatements
def naive_conv_same(x, w, b):
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    ph, pw = kh // 2, kw // 2
    y = np.zeros((B, H, W, Co))
    for i in range(H):
        for j in range(W):
            for ky in range(kh):
                for kx in range(kw):
                    ii, jj = i + ky - ph, j + kx - pw
                    if 0 <= ii < H and 0 <= jj < W:
                        y[:, i, j, :] += x[:, ii, jj, :] @ w[ky, kx]
    return y + b


def naive_conv_transpose_same(x, w, b, sh, sw):
    """Appendix A-3: adjoint of the SAME forward conv; k=3,s=2: out[2i+k] += x[i] w[k] cropped to 2n;
    s=1: out[j+k-1]; k=1,s=2: out[2i] = x[i] w; bias on all positions.  w is [kh,kw,Cout,Cin]."""
    B, H, W, Ci = x.shape
    kh, kw, Co, _ = w.shape
    pbh = kh // 2 if sh == 1 else 0
    pbw = kw // 2 if sw == 1 else 0
    y = np.zeros((B, sh * H, sw * W, Co))
    for i in range(H):
        for j in range(W):
            for ky in range(kh):
                for kx in range(kw):
                    Y, X = sh * i + ky - pbh, sw * j + kx - pbw
                    if 0 <= Y < sh * H and 0 <= X < sw * W:
                        y[:, Y, X, :] += x[:, i, j, :] @ w[ky, kx].T
    return y + b


def test_conv2d_matches_naive_loops():
    x = torch.randn(2, 5, 6, 3, generator=g(1), dtype=D)
    for k in (1, 3):
        w = torch.randn(k, k, 3, 4, generator=g(2), dtype=D)
        b = torch.randn(4, generator=g(3), dtype=D)
        np.testing.assert_allclose(O.conv2d(x, w, b).numpy(), naive_conv_same(x.numpy(), w.numpy(), b.numpy()), atol=1e-12)
    w2 = torch.randn(2, 2, 3, 4, generator=g(4), dtype=D)       # 2x2 VALID (recognizer conv7)
    y = O.conv2d(x, w2, None, padding="valid").numpy()
    assert y.shape == (2, 4, 5, 4)
    ref = sum(np.einsum("bijc,co->bijo", x.numpy()[:, ky:ky + 4, kx:kx + 5], w2.numpy()[ky, kx]) for ky in range(2) for kx in range(2))
    np.testing.assert_allclose(y, ref, atol=1e-12)


def test_conv2d_transpose_matches_naive_loops():
    x = torch.randn(2, 3, 4, 5, generator=g(5), dtype=D)
    for k, stride in ((3, (2, 2)), (3, (2, 1)), (1, (2, 2)), (1, (2, 1))):
        w = torch.randn(k, k, 6, 5, generator=g(6), dtype=D)
        b = torch.randn(6, generator=g(7), dtype=D)
        got = O.conv2d_transpose(x, w, b, stride).numpy()
        np.testing.assert_allclose(got, naive_conv_transpose_same(x.numpy(), w.numpy(), b.numpy(), *stride), atol=1e-12)


def test_conv2d_transpose_1x1_stride2_is_zero_insertion_plus_bias():
    x = torch.randn(1, 2, 3, 4, generator=g(8), dtype=D)
    w = torch.randn(1, 1, 5, 4, generator=g(9), dtype=D)
    b = torch.randn(5, generator=g(10), dtype=D)
    y = O.conv2d_transpose(x, w, b, (2, 2))
    assert torch.allclose(y[:, 1::2, :, :], b.expand_as(y[:, 1::2, :, :]))
    assert torch.allclose(y[:, :, 1::2, :], b.expand_as(y[:, :, 1::2, :]))
    assert torch.allclose(y[:, ::2, ::2, :], x @ w[0, 0].t() + b)


# ---------------------------------------------------------------- analytic known answers
def test_nonlocal_sigma_zero_is_identity():
    x = torch.randn(2, 4, 8, 16, generator=g(11), dtype=D)
    nl = O.init_nonlocal(16, g(12))
    out = O.nonlocal_block(x, nl["theta"], nl["phi"], nl["g"], nl["o"], torch.tensor(0.0, dtype=D))
    assert torch.equal(out, x)


def test_spectral_norm_rank_one():
    a, b = torch.randn(12, generator=g(13), dtype=D), torch.randn(7, generator=g(14), dtype=D)
    w = torch.outer(a, b).reshape(3, 4, 7)
    for seed in (0, 1):
        u = torch.randn(1, 7, generator=g(seed), dtype=D)
        assert torch.allclose(O.spectral_norm(w, u), w / (a.norm() * b.norm()), atol=1e-12)


def test_batchnorm_and_cbn():
    x = torch.full((2, 3, 4, 5), 3.7, dtype=D)
    x_hat, mean, var = O.batch_norm_train(x)
    assert torch.allclose(x_hat, torch.zeros_like(x)) and torch.allclose(mean, torch.full((5,), 3.7, dtype=D)) and var.abs().max() == 0
    x = torch.randn(2, 3, 4, 5, generator=g(15), dtype=D)
    z = torch.randn(2, 32, generator=g(16), dtype=D)
    wb = torch.randn(32, 5, generator=g(17), dtype=D)
    y = O.conditional_batch_norm(x, z, torch.zeros(32, 5, dtype=D), wb)       # gamma Dense = 0 -> beta broadcast
    assert torch.allclose(y, (z @ wb).view(2, 1, 1, 5).expand_as(y))


def test_losses_on_hand_values():
    v = lambda *a: torch.tensor(a, dtype=D).view(-1, 1)
    d, dr, df, gl, s, sa, sb = O.hinge(v(2.0, 0.5), v(-3.0, 0.25), v(0.0, 1.5), v(-0.5, 2.0), v(9.0, 9.0))
    assert torch.equal(dr, v(0.0, 0.5)) and torch.equal(df, v(0.0, 1.25)) and torch.equal(d, v(0.0, 1.75))
    assert torch.equal(sa, v(1.0, 0.0)) and torch.equal(sb, v(0.5, 3.0)) and torch.equal(gl, v(3.5, -2.25))
    out = O.not_saturating(v(0.0), v(0.0), v(0.0), v(0.0), v(0.0))
    ln2 = math.log(2.0)
    for t, e in zip(out, (2 * ln2, ln2, ln2, 2 * ln2, 2 * ln2, ln2, ln2)):
        assert abs(t.item() - e) < 1e-12
    r, gg = v(1.0, 3.0), v(2.0, 6.0)
    g_bal, r_bal, alpha, r_std, g_std = O.apply_gradient_balancing(r, gg)
    assert r_std.item() == 1.0 and g_std.item() == 2.0 and torch.equal(r_bal, 2 * r) and torch.equal(g_bal, gg + 2 * r)


def brute_force_ctc(logp, label, blank):
    T, C = logp.shape
    total = 0.0
    for path in itertools.product(range(C), repeat=T):
        collapsed = [k for k, _ in itertools.groupby(path) if k != blank]
        if collapsed == list(label):
            total += math.exp(sum(logp[t, c] for t, c in enumerate(path)))
    return -math.log(total)


def test_ctc_matches_path_enumeration():
    gen = g(18)
    for T, C, label in ((4, 3, [0, 1]), (5, 4, [1, 1]), (3, 3, [0])):
        p = torch.softmax(torch.randn(1, T, C, generator=gen, dtype=D), -1)
        cost = O.ctc_batch_cost(torch.tensor([label]), p, T, len(label)).item()
        lp = torch.log_softmax(torch.log(p + 1e-7), -1)[0].numpy()
        assert abs(cost - brute_force_ctc(lp, label, C - 1)) < 1e-9


def test_seed_layout_formula():
    """seed[b, r, 4l+pw, q] = (z0[b] . E[y[b,l]])[pw*2048 + q*4 + r]."""
    gen = g(19)
    B, L, V = 2, 3, 5
    table = torch.randn(V, 32, 8192, generator=gen, dtype=D)
    z0 = torch.randn(B, 32, generator=gen, dtype=D)
    y = torch.randint(0, V, (B, L), generator=gen)
    seed = O.filter_bank_seed(z0, y, table)
    assert seed.shape == (B, 4, 4 * L, 512)
    for b, l, pw, q, r in ((0, 0, 0, 0, 0), (1, 2, 3, 511, 3), (0, 1, 2, 100, 1), (1, 0, 1, 7, 2)):
        v = z0[b] @ table[y[b, l]]
        assert abs(seed[b, r, 4 * l + pw, q].item() - v[pw * 2048 + q * 4 + r].item()) < 1e-12


def test_adam_first_step_closed_form():
    p0 = torch.randn(50, generator=g(20), dtype=D)
    gr = torch.randn(50, generator=g(21), dtype=D)
    P, st = {"w": p0.clone()}, {}
    O.adam_update(P, {"w": gr}, st, 2e-4, 0.0, 0.999)
    expect = p0 - 2e-4 * gr / (gr.abs() + 1e-7 / math.sqrt(1 - 0.999))      # beta_1 = 0: first step = -lr g/(|g| + eps/sqrt(1-b2))
    assert torch.allclose(P["w"], expect, atol=1e-15)


def test_orthogonal_and_glorot_initialisers():
    for shape in ((3, 3, 8, 16), (32, 128), (64, 8)):
        w = O.orthogonal(shape, g(22)).reshape(-1, shape[-1])
        eye = w.t() @ w if w.shape[0] >= w.shape[1] else w @ w.t()
        assert torch.allclose(eye, torch.eye(eye.shape[0], dtype=D), atol=1e-10)
    fb = O.glorot_uniform((52, 32, 8192), g(23))
    assert abs(fb.abs().max().item() - math.sqrt(6.0 / (32 * 52 + 8192 * 52))) < 1e-4     # 3.746e-3 (Appendix A-6)


def test_lstm_matches_torch_nn_lstm():
    """Independent pin of the Keras-order LSTM restatement: torch.nn.LSTM uses the same gate order (i,f,g,o) with
    weights stored transposed and two bias vectors."""
    gen = g(24)
    B, T, I, H = 3, 5, 7, 4
    x = torch.randn(B, T, I, generator=gen, dtype=D)
    W, U, b = (torch.randn(I, 4 * H, generator=gen, dtype=D), torch.randn(H, 4 * H, generator=gen, dtype=D),
               torch.randn(4 * H, generator=gen, dtype=D))
    ref = torch.nn.LSTM(I, H, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for sfx in ("", "_reverse"):
            getattr(ref, "weight_ih_l0" + sfx).copy_(W.t())
            getattr(ref, "weight_hh_l0" + sfx).copy_(U.t())
            getattr(ref, "bias_ih_l0" + sfx).copy_(b)
            getattr(ref, "bias_hh_l0" + sfx).zero_()
        out, _ = ref(x)
    p = {"l.fw.W": W, "l.fw.U": U, "l.fw.b": b, "l.bw.W": W, "l.bw.U": U, "l.bw.b": b}
    assert torch.allclose(O.bilstm(x, p, "l"), out, atol=1e-12)
    # input dropout: one mask per sample shared by all timesteps == masking the input sequence
    m = (torch.rand(B, I, generator=gen, dtype=D) > 0.5).double() * 2
    assert torch.allclose(O.bilstm(x, p, "l", (m, m)), ref(x * m.unsqueeze(1))[0], atol=1e-12)


def test_my_recognizer_shapes_and_frame_count():
    P = O.init_my_recognizer(g(25))
    x = torch.rand(2, 32, 64, 1, generator=g(26), dtype=D)
    probs = O.my_recognizer_probs(x, P)
    assert probs.shape == (2, 16, 53) and torch.allclose(probs.sum(-1), torch.ones(2, 16, dtype=D))     # T = W/4
    b = P["lstm1.fw.b"]
    assert b[:256].abs().sum() == 0 and (b[256:512] == 1).all() and b[512:].abs().sum() == 0           # unit_forget_bias


def test_parameter_counts():
    """G 53.7 M, D 37.3 M, R 5.6 M trainable parameters."""
    n = lambda P: sum(v.numel() for k, v in P.items() if O.is_trainable(k))
    assert n(O.init_discriminator(g(0), torch.float32)) == 37_336_385
    assert n(O.init_recognizer(g(0), torch.float32)) == 5_578_037
    assert abs(n(O.init_generator(g(0), torch.float32)) / 1e6 - 53.68) < 0.01


def test_generator_inference_mode_uses_moving_statistics():
    """training=False (data_utils.py:505-507): with the moving statistics of every BatchNorm set to the batch statistics of
    a training-mode pass (biased variance, what the normalisation itself uses) the two modes give the same image; with
    the Keras initial values (mean 0, variance 1) they do not."""
    gen = g(5)
    P = O.init_generator(gen)
    for k in P:
        if k.endswith(".beta.w") or k.endswith(".gamma.w"):
            P[k] = P[k] * 1.0
    style = torch.rand(2, 32, 32, 1, generator=gen, dtype=torch.float64) * 2 - 1
    y = torch.randint(0, 52, (2, 2), generator=gen)
    nl_s, nl_u = O.init_nonlocal(64, gen), O.init_nonlocal(64, gen)
    stats = {}
    with torch.no_grad():
        train = O.generator(style, y, P, nl_s, nl_u, bn_stats=stats)
        cold = O.generator(style, y, P, nl_s, nl_u, training=False)
        for pre, st in stats.items():
            P[pre + ".mm"], P[pre + ".mv"] = st["mean"], st["var"]
        warm = O.generator(style, y, P, nl_s, nl_u, training=False)
    assert (train - cold).abs().max().item() > 1e-3
    assert (train - warm).abs().max().item() < 1e-12
    assert set(stats) == {"B1.cbn1", "B1.cbn2", "B2.cbn1", "B2.cbn2", "B3.cbn1", "B3.cbn2", "bn"}


def test_rmsprop_first_step_closed_form_with_small_gradients():
    """Keras RMSprop of TF 2.1 (momentum 0, not centered): rms = 0.1 g^2 after the first step, so
    delta = -lr g / (sqrt(0.1) |g| + 1e-7) -- epsilon OUTSIDE the root.  Hand values with gradients down to
    1e-9, where the two epsilon placements differ by orders of magnitude: for g = 1e-9 the step is
    -lr * 1e-9 / (3.162e-10 + 1e-7) = -lr * 9.968e-3, while eps inside the root would give -lr * 3.16e-6."""
    g_ = torch.tensor([1.0, -1e-3, 1e-6, 1e-9], dtype=torch.float64)
    P, st = {"w": torch.zeros(4, dtype=torch.float64)}, {}
    O.rmsprop_update(P, {"w": g_}, st, lr=2e-4)
    want = -2e-4 * g_ / (math.sqrt(0.1) * g_.abs() + 1e-7)
    assert torch.allclose(P["w"], want, rtol=1e-12, atol=0)
    assert abs(P["w"][3].item() / (-2e-4) - 9.96847e-3) < 1e-7
    assert abs(P["w"][0].item() / (-2e-4) - 1 / (math.sqrt(0.1) + 1e-7)) < 1e-9
