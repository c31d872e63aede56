"""bf16 mode: first-generation kernels (fp32 activations converted in registers, 128 x 128 tiles, 3 workgroups per CU) against the
DMA-fed second generation (bf16 operand copies, 256-row tiles, one workgroup per CU) on the 64-channel layers, with and without
the operand conversion sweep the second generation needs when no producer wrote the copy.   python tools/probe_v1v2.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops  # noqa: E402


def timeit(fn, n=8):
    for _ in range(2):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    dev = torch.device("cuda:0")
    ops.set_conv_dtype("bf16")
    print("%-28s %10s %10s %10s   (ms: v1 | v2 with cached copies | v2 + conversion sweeps)" % ("layer", "v1", "v2", "v2+cvt"))
    for name, H, W, Ci, Co, k in (("B1.conv2 64->64 32x160", 32, 160, 64, 64, 3), ("B2.conv1 64->512 16x80", 16, 80, 64, 512, 3),
                                  ("B2.short 64->512 8x40 1x1", 8, 40, 64, 512, 1), ("G.B2 128->128 16x160", 16, 160, 128, 128, 3),
                                  ("G.B1 256->256 8x80", 8, 80, 256, 256, 3)):
        x = torch.randn(B, H, W, Ci, device=dev)
        w = torch.randn(k, k, Ci, Co, device=dev) * 0.05
        dy = torch.randn(B, H, W, Co, device=dev)
        dw = torch.zeros_like(w)
        for what, fn in (("fwd", lambda: ops.conv2d_fwd(x, w, relu_in=True)), ("dgrad", lambda: ops.conv2d_bwd_data(dy, w, (H, W), mask=x)),
                         ("wgrad", lambda: ops.conv2d_bwd_weight(x, dy, dw, relu_in=True))):
            ops.USE_V2 = False
            t1 = timeit(fn)
            ops.USE_V2 = True
            ops.new_step()
            t2 = timeit(fn)

            def cold():
                ops.new_step()
                fn()
            t3 = timeit(cold)
            print("%-28s %10.3f %10.3f %10.3f   %s" % (name, t1, t2, t3, what))


if __name__ == "__main__":
    main()
