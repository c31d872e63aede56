"""Is there a systematic gain error in the fp8 convolution launches (config c5)?  For forward, data-grad and weight-grad of the
>= 256-channel layers: the projection coefficient <fp8 result, fp32 result> / <fp32, fp32> (1 = unbiased: quantisation noise is
uncorrelated with the signal; < 1 = the launch shrinks its result: truncating conversion, clipping at a stale amax, underflow of
small gradient values in e4m3 / e5m2) and the relative noise |fp8 - fp32| / |fp32|, on heavy-tailed operands (gradients: a
log-normal magnitude spread of `decades` decades, the shape back-propagated gradients have).  Test infrastructure.
    python tools/diag_fp8_bias.py > gpurun_out/r04_diag_fp8_bias.txt"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scrabble_gan_amd import ops  # noqa: E402


def proj(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a * b).sum() / (b * b).sum()), float((a - b).norm() / b.norm())


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    for (B, H, W, Cin, Cout) in ((64, 16, 80, 512, 512), (64, 8, 40, 1024, 1024), (64, 4, 20, 1024, 1024), (64, 8, 80, 256, 256)):
        for decades in (0.0, 1.0, 2.0, 3.0):
            x = torch.randn(B, H, W, Cin, device=dev, generator=g)
            w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g) / math.sqrt(9 * Cin)
            dy = torch.randn(B, H, W, Cout, device=dev, generator=g) * torch.exp(torch.randn(B, H, W, Cout, device=dev, generator=g) * decades * math.log(10) / 2)
            res = {}
            for md in ("f32", "fp8"):
                ops.set_conv_dtype(md)
                ops.new_step()
                y = ops.conv2d_fwd(x, w, None, relu_in=True)
                dx = ops.conv2d_bwd_data(dy, w, (H, W), mask=x)
                dw = torch.zeros_like(w)
                ops.conv2d_bwd_weight(x, dy, dw, relu_in=True)
                torch.cuda.synchronize()
                res[md] = (y.clone(), dx.clone(), dw.clone())
            ops.set_conv_dtype("f32")
            line = "%dx%d %d->%d B%d, gradient magnitude spread %.0f decades:" % (H, W, Cin, Cout, B, decades)
            for name, a, b in zip(("fwd", "dgrad", "wgrad"), res["fp8"], res["f32"]):
                p, n = proj(a, b)
                line += "  %s gain %.4f noise %.3f" % (name, p, n)
            print(line)


if __name__ == "__main__":
    main()
