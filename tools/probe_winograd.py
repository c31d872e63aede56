"""Direct vs Winograd-domain launches of the fp32 3x3 convolutions, per layer shape and direction (ms per launch):
where the 16 / 36 product count pays for the two transform sweeps (-> ops.WINO_MIN_C / WINO_MIN_KN).
    python tools/probe_winograd.py [--iters 5]            (SG_WINO_TILE=2: F(2x2, 3x3) everywhere; default F(4x4, 3x3) where H and W allow)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, Cout
    ("D.B2.conv2 b384", 384, 16, 80, 512, 512), ("D.B3.conv2 b384", 384, 8, 40, 1024, 1024), ("D.B4.conv b256", 256, 4, 20, 1024, 1024),
    ("D.B3.conv1 b128", 128, 8, 40, 512, 1024), ("D.B2.conv2 b32", 32, 16, 80, 512, 512), ("D.B3.conv2 b32", 32, 8, 40, 1024, 1024),
    ("D.B4.conv b32", 32, 4, 20, 1024, 1024), ("D.B4.conv b16", 16, 4, 20, 1024, 1024),
    ("G.B1.conv b128", 128, 8, 80, 256, 256), ("G.B2.conv b128", 128, 16, 160, 128, 128), ("R.conv3 b256", 256, 8, 40, 128, 256),
    ("R.conv4 b256", 256, 8, 40, 256, 256), ("R.conv5 b256", 256, 4, 40, 256, 512), ("R.conv6 b256", 256, 4, 40, 512, 512),
    ("D.B2.conv1 b384", 384, 16, 80, 64, 512), ("G.B1.conv b16", 16, 8, 80, 256, 256), ("R.conv3 b32", 32, 8, 40, 128, 256),
    ("R.conv4 b32", 32, 8, 40, 256, 256), ("R.conv5 b32", 32, 4, 40, 256, 512), ("R.conv6 b32", 32, 4, 40, 512, 512),
    ("D.B1.conv2 b384", 384, 32, 160, 64, 64), ("G.B3.conv b128", 128, 32, 160, 64, 64), ("R.conv2 b256", 256, 16, 80, 64, 128),
    ("D.B2.conv1 b32", 32, 16, 80, 64, 512), ("D.B1.conv2 b32", 32, 32, 160, 64, 64),
    ("D.B2.conv1 b256", 256, 16, 80, 64, 512), ("D.B1.conv2 b256", 256, 32, 160, 64, 64), ("G.B3.conv b16", 16, 32, 160, 64, 64), ("R.conv2 b32", 32, 16, 80, 64, 128),
]


def timeit(fn, iters):
    fn()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.WINO_MIN_C, ops.WINO_MIN_KN, ops.WINO4_WGRAD_MIN_TILES, ops.WINO_ROW_GAIN = {2: 32, 4: 32}, {2: 0, 4: 0}, 0, 0.0
    print("%-18s | %21s | %21s | %21s   (ms direct / Winograd, ratio)" % ("layer", "fwd", "dgrad", "wgrad"))
    for name, B, H, W, Ci, Co in SHAPES:
        x = torch.randn(B, H, W, Ci, device=dev)
        w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05
        dy = torch.randn(B, H, W, Co, device=dev)
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.zeros_like(w)
        fns = {"fwd": lambda: ops.conv2d_fwd(x, w, relu_in=True, out=y), "dgrad": lambda: ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=dx),
               "wgrad": lambda: ops.conv2d_bwd_weight(x, dy, dw, relu_in=True)}
        cells = []
        for d in ("fwd", "dgrad", "wgrad"):
            if (d == "dgrad" and Ci % ops.WINO_N_MULT) or (d == "fwd" and Co % ops.WINO_N_MULT):
                cells.append("%21s" % "-")
                continue
            ops.USE_WINOGRAD = False
            t0 = timeit(fns[d], args.iters)
            ops.USE_WINOGRAD = True
            t1 = timeit(fns[d], args.iters)
            cells.append("%6.3f / %6.3f %5.2fx" % (t0, t1, t0 / t1))
        print("%-18s | %s" % (name, " | ".join(cells)))
        del x, w, dy, y, dx, dw


if __name__ == "__main__":
    main()
