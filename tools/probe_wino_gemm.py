"""Time the grouped Winograd products (sg_wino_gemm) of the step's 3x3 layers at the headline and shard batches, to compare kernel
variants chosen by environment variables (round 4: SG2_W8 = eight waves per 128 x 128 tile; a phase-stagger experiment measured level).
    SG2_W8=0 python tools/probe_stagger.py; SG2_W8=1 python tools/probe_stagger.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    print("SG2_W8 =", os.environ.get("SG2_W8", "default"))
    for (B, H, W, Cin, Cout) in ((384, 16, 80, 512, 512), (256, 16, 80, 512, 512), (384, 8, 40, 1024, 1024), (256, 8, 40, 1024, 1024), (384, 8, 40, 512, 1024),
                                 (256, 4, 20, 1024, 1024), (128, 16, 80, 512, 512), (128, 8, 80, 256, 256), (128, 16, 160, 128, 128),
                                 (48, 16, 80, 512, 512), (32, 16, 80, 512, 512), (16, 16, 80, 512, 512), (48, 8, 40, 1024, 1024), (32, 8, 40, 1024, 1024),
                                 (16, 8, 40, 1024, 1024), (48, 4, 20, 1024, 1024), (32, 4, 20, 1024, 1024), (16, 4, 20, 1024, 1024), (16, 8, 80, 256, 256),
                                 (16, 16, 160, 128, 128), (48, 16, 80, 64, 512)):
        x = torch.randn(B, H, W, Cin, device=dev, generator=g)
        w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g) / math.sqrt(9 * Cin)
        y = torch.empty(B, H, W, Cout, device=dev)
        ops.new_step()
        for _ in range(3):
            ops.conv2d_fwd(x, w, None, relu_in=True, out=y)
        tile = 4
        T = B * (H // tile) * (W // tile)
        Tp = -(-T // 128) * 128
        V = torch.empty(36 * Tp * Cin, device=dev)
        Mt = torch.empty(36 * Tp * Cout, device=dev)
        u = ops.packed_filter(w, "wino_fwd4")
        from scrabble_gan_amd._lib import call
        s = ops._stream()
        call("sg_wino_input", x.data_ptr(), V.data_ptr(), B, H, W, Cin, 1, tile, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        for _ in range(2):
            call("sg_wino_gemm", V.data_ptr(), u.data_ptr(), Mt.data_ptr(), B, H, W, Cin, Cout, tile, s)
        e0.record()
        for _ in range(n):
            call("sg_wino_gemm", V.data_ptr(), u.data_ptr(), Mt.data_ptr(), B, H, W, Cin, Cout, tile, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("%4d x %2dx%3d %4d->%4d: %7.3f ms  %6.1f TF/s  (%d tiles = %.1f rounds of 512)" % (
            B, H, W, Cin, Cout, ms, 2.0 * 36 * Tp * Cin * Cout / ms / 1e9, 36 * (Tp // 128) * (Cout // 128), 36 * (Tp // 128) * (Cout // 128) / 512.0))
        del x, w, y, V, Mt


if __name__ == "__main__":
    main()
