"""Time the grouped Winograd weight-grad products (sg_wino_wgrad_gemm) of the step's 3x3 layers at the headline and shard batches,
to compare kernel variants chosen by environment variables (round 4: SG_WGRAD_W8).
    SG_WGRAD_W8=0 python tools/probe_wino_wgrad.py; SG_WGRAD_W8=1 python tools/probe_wino_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops  # noqa: E402
from scrabble_gan_amd._lib import call  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    print("SG_WGRAD_W8 =", os.environ.get("SG_WGRAD_W8", "default"))
    for (B, H, W, Cin, Cout) in ((256, 16, 80, 512, 512), (256, 8, 40, 1024, 1024), (256, 8, 40, 512, 1024), (256, 4, 20, 1024, 1024), (128, 16, 80, 512, 512),
                                 (128, 8, 80, 256, 256), (256, 16, 80, 64, 512), (32, 16, 80, 512, 512), (32, 8, 40, 1024, 1024), (32, 4, 20, 1024, 1024),
                                 (16, 8, 40, 1024, 1024)):
        tile = 4
        T = B * (H // tile) * (W // tile)
        if T < ops.WINO4_WGRAD_MIN_TILES:
            continue
        Tp = -(-T // 128) * 128
        V = torch.randn(36 * Tp * Cin, device=dev)
        Q = torch.randn(36 * Tp * Cout, device=dev)
        dU = torch.empty(36 * Cin * Cout, device=dev)
        s = ops._stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        for _ in range(2):
            call("sg_wino_wgrad_gemm", V.data_ptr(), Q.data_ptr(), dU.data_ptr(), B, H, W, Cin, Cout, tile, 0, s)
        e0.record()
        for _ in range(n):
            call("sg_wino_wgrad_gemm", V.data_ptr(), Q.data_ptr(), dU.data_ptr(), B, H, W, Cin, Cout, tile, 0, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("%4d x %2dx%3d %4d->%4d: %7.3f ms  %6.1f TF/s" % (B, H, W, Cin, Cout, ms, 2.0 * 36 * T * Cin * Cout / ms / 1e9))
        del V, Q, dU


if __name__ == "__main__":
    main()
