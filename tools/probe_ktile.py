"""Per-workgroup constant cost vs per-k-tile cost of the DMA-fed conv kernels: forward conv of [128,16,80,Cin] -> 512 channels,
3x3, for Cin in {64 .. 1024} (same M, N and grid, only the reduction length changes); wall = c + K * slope.
    python tools/probe_ktile.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
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
    dev = torch.device("cuda:0")
    B, H, W, Cout, k = 128, 16, 80, 512, 3
    g = torch.Generator(device=dev).manual_seed(3)
    for mode in ("f32", "bf16", "fp8"):
        ops.set_conv_dtype(mode)
        for what in ("fwd", "dgrad", "wgrad"):
            pts = []
            for Cin in (64, 128, 256, 512, 1024):
                if what == "wgrad" and Cin < 256:
                    continue
                x = torch.randn(B, H, W, Cin, device=dev, generator=g)
                w = torch.randn(k, k, Cin, Cout, device=dev, generator=g) / math.sqrt(k * k * Cin)
                y = torch.empty(B, H, W, Cout, device=dev)
                if what == "fwd":
                    def fn():
                        ops.conv2d_fwd(x, w, None, relu_in=True, out=y)
                elif what == "dgrad":     # reduction over Cin as well: dy has Cin channels, dx has 512
                    wd = torch.randn(k, k, Cout, Cin, device=dev, generator=g) / math.sqrt(k * k * Cin)

                    def fn():
                        ops.conv2d_bwd_data(x, wd, (H, W), mask=y, out=y)
                else:
                    dw = torch.zeros(k, k, Cin, Cout, device=dev)
                    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)

                    def fn():
                        ops.conv2d_bwd_weight(x, dy, dw, relu_in=True)
                ops.new_step()
                fn()                       # (twins / packed filters are made here, outside the timing)
                t = timeit(fn)
                pts.append((Cin, t))
            (c0, t0), (c1, t1) = pts[0], pts[-1]
            slope = (t1 - t0) / (c1 - c0)
            const = t0 - slope * c0
            print("%-5s %-5s " % (mode, what) + "  ".join("Cin %4d: %7.3f ms" % p for p in pts) +
                  "   | constant %.3f ms, %.4f ms per 64 channels" % (const, slope * 64), flush=True)
    ops.set_conv_dtype("f32")


if __name__ == "__main__":
    main()
