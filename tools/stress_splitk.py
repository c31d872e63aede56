"""Stress the split-K path: repeated launches vs the single-pass result, with the consumer reading immediately."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops, _lib
lib = _lib.lib()
setk = lib.sg_debug_set_splitk
setk.argtypes = [ctypes.c_int]; setk.restype = None
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (B, H, W, Ci, Co, k) in [(2, 4, 4, 1024, 1024, 3), (2, 8, 8, 512, 1024, 3), (2, 16, 16, 64, 512, 1), (8, 4, 20, 1024, 1024, 3)]:
    x = torch.randn(B, H, W, Ci, device=dev, generator=g)
    w = torch.randn(k, k, Ci, Co, device=dev, generator=g) / (k * Ci ** 0.5)
    dy = torch.randn(B, H, W, Co, device=dev, generator=g)
    setk(1)
    ref_f = ops.conv2d_fwd(x, w, relu_in=True)
    ref_d = ops.conv2d_bwd_data(dy, w, (H, W), mask=x)
    torch.cuda.synchronize()
    for mode in ("memset", "prezero+accum"):
        bad_f = bad_d = 0
        worst = 0.0
        for it in range(200):
            setk(0 if it % 2 else 8)
            # churn the caches / allocator a bit
            junk = torch.empty(1 << 20, device=dev).normal_()
            if mode == "memset":
                f = ops.conv2d_fwd(x, w, relu_in=True)
                d = ops.conv2d_bwd_data(dy, w, (H, W), mask=x)
            else:
                f = torch.zeros_like(ref_f); d = torch.zeros_like(ref_d)
                ops.conv2d_fwd(x, w, relu_in=True, out=f, accum=True)
                ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=d, accum=True)
            # consumer kernels read immediately with plain loads
            ef = ops.add(f, f)
            ed = ops.add(d, d)
            e1 = (ef - 2 * ref_f).abs().max().item() / ref_f.abs().max().item()
            e2 = (ed - 2 * ref_d).abs().max().item() / ref_d.abs().max().item()
            bad_f += e1 > 1e-4
            bad_d += e2 > 1e-4
            worst = max(worst, e1, e2)
        print((B, H, W, Ci, Co, k), mode, "bad fwd %d/200 bad dgrad %d/200 worst rel err %.3e" % (bad_f, bad_d, worst))
