"""Generates tests/golden/ops_v1.npz: small seeded input/output vectors of the hot-path ops, computed by the CPU
oracle in fp64.  The reference itself cannot run offline (no TensorFlow: SURVEY 8c), so these fixtures pin the
ORACLE (a later edit that changes its arithmetic fails tests/test_golden.py) and give the HIP kernels a fixed,
reviewable target; they are data only.

    python tests/golden/make_golden.py        # rewrites ops_v1.npz (deterministic)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import scrabble_oracle as O  # noqa: E402

D = torch.float64


def build():
    g = torch.Generator().manual_seed(20261004)
    r = lambda *s: torch.randn(*s, generator=g, dtype=D)
    out = {}
    # Conv2D 3x3 SAME on relu(x) + bias, and its three gradients
    x, w, b = r(2, 4, 6, 8).requires_grad_(True), (r(3, 3, 8, 12) / 8).requires_grad_(True), r(12).requires_grad_(True)
    y = O.conv2d(torch.relu(x), w, b)
    dy = r(*y.shape)
    y.backward(dy)
    out.update(conv_x=x, conv_w=w, conv_b=b, conv_y=y, conv_dy=dy, conv_dx=x.grad, conv_dw=w.grad, conv_db=b.grad)
    # Conv2DTranspose 3x3 stride (2,1) and 1x1 stride (2,2)
    xt, wt, bt = r(2, 3, 4, 8).requires_grad_(True), (r(3, 3, 4, 8) / 8).requires_grad_(True), r(4)
    yt = O.conv2d_transpose(xt, wt, bt, (2, 1))
    dyt = r(*yt.shape)
    yt.backward(dyt)
    out.update(convT_x=xt, convT_w=wt, convT_b=bt, convT_y=yt, convT_dy=dyt, convT_dx=xt.grad, convT_dw=wt.grad)
    w1 = r(1, 1, 4, 8) / 3
    out.update(convT1_w=w1, convT1_y=O.conv2d_transpose(xt.detach(), w1, bt, (2, 2)))
    # ConditionalBatchNorm + ReLU
    xb, z = r(3, 2, 4, 8).requires_grad_(True), r(3, 32)
    wg, wb = r(32, 8) / 5, r(32, 8) / 5
    yb = torch.relu(O.conditional_batch_norm(xb, z, wg, wb))
    dyb = r(*yb.shape)
    yb.backward(dyb)
    out.update(cbn_x=xb, cbn_z=z, cbn_wg=wg, cbn_wb=wb, cbn_y=yb, cbn_dy=dyb, cbn_dx=xb.grad)
    # NonLocalBlock
    xn = r(2, 4, 8, 64).requires_grad_(True)
    nl = O.init_nonlocal(64, g)
    sig = torch.tensor(0.4, dtype=D)
    yn = O.nonlocal_block(xn, nl["theta"], nl["phi"], nl["g"], nl["o"], sig)
    dyn = r(*yn.shape)
    yn.backward(dyn)
    out.update(nl_x=xn, nl_theta=nl["theta"], nl_phi=nl["phi"], nl_g=nl["g"], nl_o=nl["o"], nl_sigma=sig, nl_y=yn, nl_dy=dyn, nl_dx=xn.grad)
    # softmax + CTC (repeated characters included)
    lg = (r(3, 11, 53) * 2).requires_grad_(True)
    lab = torch.tensor([[5, 5, 9], [0, 51, 3], [7, 8, 7]])
    cost = O.ctc_batch_cost(lab, torch.softmax(lg, -1), 11, 3)
    cost.sum().backward()
    out.update(ctc_logits=lg, ctc_labels=lab, ctc_cost=cost, ctc_dlogits=lg.grad)
    # loss head: hinge and not_saturating with gradient balancing
    v = [r(6, 1) * 1.5 for _ in range(5)]
    rf = torch.rand(6, 1, generator=g, dtype=D) * 20 + 5
    for name, fn in (("hinge", O.hinge), ("ns", O.not_saturating)):
        outs = fn(*v)
        gb = O.apply_gradient_balancing(rf, outs[3])
        out.update({"loss_%s_%d" % (name, i): t for i, t in enumerate(outs)})
        out.update({"loss_%s_gbal" % name: gb[0], "loss_%s_rbal" % name: gb[1], "loss_%s_rstd" % name: gb[3], "loss_%s_gstd" % name: gb[4]})
    out.update({"loss_in_%d" % i: t for i, t in enumerate(v)}, loss_rf=rf)
    # Adam, two steps (beta_1 = 0 as in the gin config)
    p0, gr = r(40), r(40)
    P, st = {"w": p0.clone()}, {}
    O.adam_update(P, {"w": gr}, st, 2e-4, 0.0, 0.999)
    O.adam_update(P, {"w": gr * 0.5}, st, 2e-4, 0.0, 0.999)
    out.update(adam_p0=p0, adam_g=gr, adam_p2=P["w"])
    # spectral norm
    ws, u = r(3, 3, 4, 6), r(1, 6)
    out.update(sn_w=ws, sn_u=u, sn_out=O.spectral_norm(ws, u))
    return {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ops_v1.npz")
    np.savez_compressed(path, **build())
    print("wrote", path, os.path.getsize(path), "bytes")
