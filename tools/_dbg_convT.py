import sys, math, torch
sys.path.insert(0, '/root/repo')
from scrabble_gan_amd import ops
from oracle import scrabble_oracle as O
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(3)
def rnd(*s): return torch.randn(*s, generator=gen, dtype=torch.float64)
def g32(t): return t.float().to(dev).contiguous()
def rel(a, b): return ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
for (B, H, W, Cin, Cout, k, stride) in [(8, 4, 8, 512, 256, 3, (2, 2)), (8, 4, 12, 512, 256, 3, (2, 2)), (8, 4, 8, 512, 256, 1, (2, 2)), (8, 8, 16, 256, 128, 3, (2, 2)), (8, 16, 32, 128, 64, 3, (2, 1))]:
    x = rnd(B, H, W, Cin); w = rnd(k, k, Cout, Cin) / math.sqrt(k * k * Cin); b = rnd(Cout)
    y = O.conv2d_transpose(x, w, b, stride)
    dy = rnd(*y.shape)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    O.conv2d_transpose(xr, wr, None, stride).backward(dy)
    yg = ops.conv2d_transpose_fwd(g32(x), g32(w), g32(b), stride=stride)
    base = rnd(B, H, W, Cin)
    dxg = ops.conv2d_transpose_bwd_data(g32(dy), g32(w), stride=stride)
    dxa = ops.conv2d_transpose_bwd_data(g32(dy), g32(w), stride=stride, out=g32(base), accum=True)
    dwg = torch.zeros(k, k, Cout, Cin, device=dev)
    ops.conv2d_transpose_bwd_weight(g32(x), g32(dy), dwg, stride=stride)
    print((B, H, W, Cin, Cout, k, stride), "fwd %.2e dgrad %.2e dgrad+accum %.2e wgrad %.2e" % (rel(yg, y), rel(dxg, xr.grad), rel(dxa, xr.grad + base), rel(dwg, wr.grad)), flush=True)
# CBN at the seed shape
from scrabble_gan_amd import nn
for (B, H, W, C) in [(8, 4, 8, 512), (8, 4, 12, 512)]:
    x = rnd(B, H, W, C) * 0.06; gamma = rnd(B, C) * 3; beta = rnd(B, C) * 3; dy = rnd(B, H, W, C)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    xh, mean, var = O.batch_norm_train(xr)
    yy = torch.relu(xh * gr.view(B, 1, 1, C) + br.view(B, 1, 1, C)); yy.backward(dy)
    yg, ctx = nn.bn_train_fwd(g32(x), g32(gamma), g32(beta), True, True)
    dx, dga, dbe, _ = nn.bn_train_bwd(ctx, g32(dy), True, True)
    print((B, H, W, C), "cbn y %.2e dx %.2e dgamma %.2e dbeta %.2e" % (rel(yg, yy.detach()), rel(dx, xr.grad), rel(dga, gr.grad), rel(dbe, br.grad)), flush=True)
