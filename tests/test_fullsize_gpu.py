"""Full-size (BASELINE config c2: bs 128, 32x160) checks through size-independent properties -- the
oracle cannot run these sizes in seconds, so the MFMA kernels are checked against each other:

  adjoint identity   <conv(x,w), dy> == <x, conv_data_grad(dy,w)> == <w, conv_weight_grad(x,dy)>
  linearity          conv(x, a*w) == a*conv(x, w)   (through the accumulate epilogue)

on the layer shapes that carry >90 % of the step's FLOPs, plus the transposed convolutions of the
generator.  fp32 reductions over up to 1e8 terms: relative tolerance 2e-3."""
import pytest
import torch

pytestmark = pytest.mark.gpu

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
