"""Time (and optionally count) the NonLocalBlock attention kernels at the generator's 32x160 site (Nq = 5120, Nk = 1280)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "128")); Nq, Nk = 5120, 1280
th = torch.randn(B, Nq, 8, device=dev); ph = torch.randn(B, Nk, 8, device=dev); g = torch.randn(B, Nk, 32, device=dev)
d = torch.randn(B, Nq, 32, device=dev)
def t(fn, it=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
o, lse = ops.attention_fwd(th, ph, g)
pairs = B * Nq * Nk
print("B=%d fwd %.3f ms (%.1f TFLOP/s at 80 flop/pair)" % (B, t(lambda: ops.attention_fwd(th, ph, g)), pairs * 80 / t(lambda: ops.attention_fwd(th, ph, g)) / 1e9))
print("B=%d bwd %.3f ms" % (B, t(lambda: ops.attention_bwd(th, ph, g, o, lse, d))))
