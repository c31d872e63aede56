"""Per-shape microbenchmark of the MFMA conv kernels on the shapes of one train_step (bs 128, L=10).
    python tools/bench_conv.py [--batch 128] [--iters 5] [--only fwd|dgrad|wgrad]
Prints TFLOP/s per (layer, direction): the table the kernel tuning works from."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops  # noqa: E402

SHAPES = [  # name, H, W, Cin, Cout, k
    ("D.B1.conv2", 32, 160, 64, 64, 3), ("D.B2.conv1", 16, 80, 64, 512, 3), ("D.B2.conv2", 16, 80, 512, 512, 3),
    ("D.B2.short", 16, 80, 64, 512, 1), ("D.B3.conv1", 8, 40, 512, 1024, 3), ("D.B3.conv2", 8, 40, 1024, 1024, 3),
    ("D.B3.short", 8, 40, 512, 1024, 1), ("D.B4.conv", 4, 20, 1024, 1024, 3), ("D.B4.short", 4, 20, 1024, 1024, 1),
    ("NL.theta", 16, 80, 64, 8, 1), ("NL.g", 16, 80, 64, 32, 1), ("NL.o", 16, 80, 32, 64, 1),
    ("G.B1.conv", 8, 80, 256, 256, 3), ("G.B2.conv", 16, 160, 128, 128, 3), ("G.B3.conv", 32, 160, 64, 64, 3),
    ("R.conv2", 16, 80, 64, 128, 3), ("R.conv3", 8, 40, 128, 256, 3), ("R.conv4", 8, 40, 256, 256, 3),
    ("R.conv5", 4, 40, 256, 512, 3), ("R.conv6", 4, 40, 512, 512, 3),
]


def timeit(fn, iters):
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
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default=None)
    ap.add_argument("--filter", default=None)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (DVFS probe: same cycles, higher clock)")
    ap.add_argument("--dtype", default="f32", help="f32 | bf16 (matrix-core operand type)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.set_conv_dtype(args.dtype)
    B = args.batch
    print("%-12s %5s %5s %5s %5s | %9s %9s %9s   (TFLOP/s; ms)" % ("layer", "H", "W", "Cin", "Cout", "fwd", "dgrad", "wgrad"))
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    for name, H, W, Ci, Co, k in SHAPES:
        if args.filter and args.filter not in name:
            continue
        x = torch.randn(B, H, W, Ci, device=dev)
        w = torch.randn(k, k, Ci, Co, device=dev) * 0.05
        dy = torch.randn(B, H, W, Co, device=dev)
        if args.zeros:
            x.zero_(); w.zero_(); dy.zero_()
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.zeros_like(w)
        flops = 2.0 * B * H * W * k * k * Ci * Co
        res = {}
        if args.only in (None, "fwd"):
            res["fwd"] = timeit(lambda: ops.conv2d_fwd(x, w, relu_in=True, out=y), args.iters)
        if args.only in (None, "dgrad"):
            res["dgrad"] = timeit(lambda: ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=dx), args.iters)
        if args.only in (None, "wgrad"):
            res["wgrad"] = timeit(lambda: ops.conv2d_bwd_weight(x, dy, dw, relu_in=True), args.iters)
        cells = []
        for d in ("fwd", "dgrad", "wgrad"):
            if d in res:
                cells.append("%5.1f %5.2f" % (flops / res[d] / 1e9, res[d]))
                tot[d][0] += flops
                tot[d][1] += res[d]
            else:
                cells.append("    -     -")
        print("%-12s %5d %5d %5d %5d | %s" % (name, H, W, Ci, Co, "  ".join(cells)))
    print("aggregate: " + "  ".join("%s %.1f TF/s" % (d, v[0] / v[1] / 1e9) for d, v in tot.items() if v[1] > 0))


if __name__ == "__main__":
    main()
