"""Cost of the fused ReLU-mask / accumulate epilogue of the data-grad launch (1024->1024 3x3 layer, bs 128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops
dev = torch.device("cuda:0")
for (B, H, W, C, N) in ((128, 8, 40, 1024, 1024), (256, 16, 80, 512, 512)):
    dy = torch.randn(B, H, W, N, device=dev); w = torch.randn(3, 3, C, N, device=dev) * 0.03
    x = torch.randn(B, H, W, C, device=dev); dx = torch.zeros(B, H, W, C, device=dev)
    fl = 2.0 * B * H * W * 9 * C * N
    def t(fn, it=5):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it
    for name, fn in (("plain", lambda: ops.conv2d_bwd_data(dy, w, (H, W), out=dx)),
                     ("mask", lambda: ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=dx)),
                     ("mask+accum", lambda: ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=dx, accum=True)),
                     ("fwd", lambda: ops.conv2d_fwd(x, w.permute(0, 1, 3, 2).contiguous() if C != N else w, out=dy, relu_in=True))):
        ms = t(fn)
        print("%s B=%d %dx%d %d->%d: %.3f ms %.1f TF/s" % (name, B, H, W, C, N, ms, fl / ms / 1e9))
