"""Probe of the second-generation bf16 conv kernels (conv_bf16v2.hip) on the headline layer shapes: correctness of the
first / last samples against the CPU oracle on bf16-rounded operands, and HIP-event timing against the first-generation
bf16 kernel and the fp32 kernel.   python tools/probe_bf16v2.py [--batch 256] [--no-check]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops  # noqa: E402
from scrabble_gan_amd._lib import call  # noqa: E402

SHAPES = [(16, 80, 512, 512, 3), (8, 40, 512, 1024, 3), (8, 40, 1024, 1024, 3), (4, 20, 1024, 1024, 3), (16, 80, 64, 512, 3),
          (8, 40, 512, 1024, 1), (32, 160, 64, 64, 3), (8, 80, 256, 256, 3), (16, 160, 128, 128, 3)]


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    st = ops._stream
    for (H, W, Cin, Cout, k) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(H + Cin)
        x = torch.randn(B, H, W, Cin, device=dev, generator=g)
        w = torch.randn(k, k, Cin, Cout, device=dev, generator=g) / math.sqrt(k * k * Cin)
        bias = torch.randn(Cout, device=dev, generator=g)
        dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
        flops = 2.0 * B * H * W * k * k * Cin * Cout
        x16 = torch.empty(x.shape, device=dev, dtype=torch.bfloat16)
        call("sg_cvt_bf16", x.data_ptr(), x16.data_ptr(), x.numel(), 0, None, 8, st())
        dy16 = torch.empty(dy.shape, device=dev, dtype=torch.bfloat16)
        call("sg_cvt_bf16", dy.data_ptr(), dy16.data_ptr(), dy.numel(), 0, None, 8, st())
        assert torch.equal(x16, x.to(torch.bfloat16)), "cvt"
        ops.set_conv_dtype("bf16")
        wp_f, wp_b = ops.packed_filter(w, "fwd"), ops.packed_filter(w, "bwd")
        y = torch.empty(B, H, W, Cout, device=dev)
        y16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
        dx = torch.empty(B, H, W, Cin, device=dev)

        def f_v2():
            call("sg_conv2d_fwd_bf16v2", x16.data_ptr(), wp_f.data_ptr(), bias.data_ptr(), None, y.data_ptr(), y16.data_ptr(), B, H, W,
                 Cin, Cout, k, k, 1, ops.RELU_IN, None, st())

        def d_v2():
            call("sg_conv2d_bwd_data_bf16v2", dy16.data_ptr(), wp_b.data_ptr(), x.data_ptr(), None, dx.data_ptr(), None, B, H, W, Cin, Cout, k, k,
                 1, 0, None, None, st())
        line = "%3dx%3d %4d->%4d k%d B%d:" % (H, W, Cin, Cout, k, B)
        if Cout % 64 == 0:
            t = timeit(f_v2)
            line += "  fwd v2 %7.3f ms %7.1f TF/s" % (t, flops / t / 1e9)
        if Cin % 64 == 0 and Cout % 64 == 0:
            t = timeit(d_v2)
            line += "  dgrad v2 %7.3f ms %7.1f TF/s" % (t, flops / t / 1e9)
        t = timeit(lambda: ops.conv2d_fwd(x, w, bias, relu_in=True, out=y))
        line += "  | fwd v1 %7.3f ms %7.1f TF/s" % (t, flops / t / 1e9)
        t = timeit(lambda: ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=dx))
        line += "  dgrad v1 %7.3f ms %7.1f TF/s" % (t, flops / t / 1e9)
        dw = torch.zeros_like(w)

        def w_v2():
            call("sg_conv2d_bwd_weight_bf16v2", x16.data_ptr(), dy16.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout, k, k, 1, ops.RELU_IN, None, st())
        wg2 = (Cin % 64 == 0 and Cout % 256 == 0) or (Cin == 64 and Cout == 64)
        if wg2:
            t = timeit(w_v2)
            line += "  | wgrad v2 %7.3f ms %7.1f TF/s" % (t, flops / t / 1e9)
        t = timeit(lambda: ops.conv2d_bwd_weight(x, dy, dw, relu_in=True))
        line += "  wgrad v1 %7.3f ms %7.1f TF/s" % (t, flops / t / 1e9)
        print(line, flush=True)
        if args.no_check:
            continue
        if wg2:
            # whole-batch dW against the fp32 kernel on the SAME bf16-representable operands (exact products, fp32 sums)
            xq, dyq = x16.float(), dy16.float()
            ops.set_conv_dtype("f32")
            ref = torch.zeros_like(w)
            ops.conv2d_bwd_weight(xq, dyq, ref, relu_in=True)
            ops.set_conv_dtype("bf16")
            dw.zero_()
            w_v2()
            err = (dw - ref).abs().max().item() / ref.abs().max().item()
            print("    wgrad v2 vs fp32 kernel on bf16-representable operands: rel %.2e" % err, flush=True)
            assert err < 2e-4
        from oracle import scrabble_oracle as O
        r16 = lambda t: t.to(torch.bfloat16).to(torch.float64)
        edge = lambda t: torch.cat([t[:2], t[-2:]], 0).double().cpu()
        if Cout % 64 == 0:
            f_v2()
            ref = O.conv2d(r16(torch.relu(edge(x))), r16(w.double().cpu()), bias.double().cpu())
            err = (edge(y) - ref).abs().max().item() / ref.abs().max().item()
            err16 = (edge(y16.float()) - edge(y).to(torch.bfloat16).double()).abs().max().item()
            print("    fwd v2 vs oracle (bf16-rounded operands): rel %.2e   bf16 copy max diff %.2e" % (err, err16), flush=True)
            assert err < 5e-5 and err16 == 0.0
        if Cin % 64 == 0 and Cout % 64 == 0:
            d_v2()
            xe = edge(x)
            xr = xe.clone().requires_grad_(True)
            O.conv2d(xr, r16(w.double().cpu()), None).backward(r16(edge(dy)))
            ref = xr.grad * (xe > 0)
            err = (edge(dx) - ref).abs().max().item() / ref.abs().max().item()
            print("    dgrad v2 vs oracle: rel %.2e" % err, flush=True)
            assert err < 5e-5
    ops.set_conv_dtype("f32")


if __name__ == "__main__":
    main()
