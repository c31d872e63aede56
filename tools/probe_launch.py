"""Which batch sizes does one conv launch accept?  (debugging aid: python tools/probe_launch.py bf16 64 64)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops  # noqa: E402

mode, Cin, Cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ops.set_conv_dtype(mode)
dev = torch.device("cuda:0")
for B in (16, 32, 48, 64, 80, 96, 128, 192, 256):
    x = torch.randn(B, 32, 160, Cin, device=dev)
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    try:
        y = ops.conv2d_fwd(x, w, b, relu_in=True)
        torch.cuda.synchronize()
        print(B, "ok", float(y.abs().max()))
    except Exception as e:  # noqa: BLE001
        print(B, "FAILED", e)
