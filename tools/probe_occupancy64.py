import os, sys, torch
sys.path.insert(0, "/root/repo")
from scrabble_gan_amd import ops
ops.set_conv_dtype("bf16")
dev = torch.device("cuda:0")
B=768
x = torch.randn(B,32,160,64,device=dev); w = torch.randn(3,3,64,64,device=dev)*0.05; dy=torch.randn(B,32,160,64,device=dev)
def t(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
print("pad", os.environ.get("SG2_LDS_PAD","0"), "fwd %.3f ms  dgrad %.3f ms" % (t(lambda: ops.conv2d_fwd(x,w,relu_in=True)), t(lambda: ops.conv2d_bwd_data(dy,w,(32,160),mask=x))))
