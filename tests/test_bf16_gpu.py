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
