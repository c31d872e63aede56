"""Timing of the thin (one-channel) convolution kernels at the first / last layer shapes: algorithmic bytes of the wide tensor
over HIP-event time.   python tools/probe_thin.py [--batch 256]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, H, W, C = a.batch, 32, 160, 64
    g = torch.Generator(device=dev).manual_seed(1)
    x1 = torch.randn(B, H, W, 1, device=dev, generator=g)
    xc = torch.randn(B, H, W, C, device=dev, generator=g)
    wide = B * H * W * C * 4
    for k in (3, 1):
        w_e = torch.randn(k, k, 1, C, device=dev, generator=g)
        w_c = torch.randn(k, k, C, 1, device=dev, generator=g)
        b_e, b_c = torch.randn(C, device=dev, generator=g), torch.randn(1, device=dev, generator=g)
        y_e, y_c = torch.empty(B, H, W, C, device=dev), torch.empty(B, H, W, 1, device=dev)
        dwe, dwc, dbe = torch.zeros_like(w_e), torch.zeros_like(w_c), torch.zeros(C, device=dev)
        rows = [
            ("expand   fwd   1->%d k%d" % (C, k), lambda: ops.conv2d_fwd(x1, w_e, b_e, out=y_e)),
            ("contract fwd   %d->1 k%d" % (C, k), lambda: ops.conv2d_fwd(xc, w_c, b_c, relu_in=True, out=y_c)),
            ("contract dgrad 1<-%d k%d" % (C, k), lambda: ops.conv2d_bwd_data(xc, w_e, (H, W), out=y_c)),
            ("expand   dgrad %d<-1 k%d" % (C, k), lambda: ops.conv2d_bwd_data(x1, w_c, (H, W), mask=xc, out=y_e)),
            ("wgrad    1->%d k%d (+bias)" % (C, k), lambda: ops.conv2d_bwd_weight(x1, xc, dwe, db=dbe)),
            ("wgrad    %d->1 k%d" % (C, k), lambda: ops.conv2d_bwd_weight(xc, x1, dwc, relu_in=True)),
        ]
        for name, fn in rows:
            t = timeit(fn)
            print("%-28s %7.3f ms  %6.2f TB/s of the wide tensor" % (name, t, wide / t / 1e9), flush=True)


if __name__ == "__main__":
    main()
