"""Attribute the recognizer's gradient deviation (tests/test_nets_gpu.py::test_recognizer, bn_training=True) to its cause.

For each convolution routing (F(4x4,3x3) Winograd / F(2x2,3x3) / direct implicit GEMM) the same problem is evaluated on the
HIP path and compared per tensor with (a) the plain fp64 oracle and (b) the COUNTERFACTUAL fp64 oracle that is forced to take
the ReLU / max-pool decisions the HIP forward pass took (oracle.RELU_HOOK / MAXPOOL_HOOK).  (a) large and (b) small = a
near-tie decided the other way in fp32 (not an arithmetic error of the kernels); (a) and (b) both large = kernel rounding.

Test infrastructure: imports the oracle.  Usage (GPU box):  python tools/diag_recognizer.py > gpurun_out/diag_recognizer.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import scrabble_oracle as O  # noqa: E402
from scrabble_gan_amd import net_architecture as NA  # noqa: E402
from scrabble_gan_amd import ops  # noqa: E402
from tests.step_fixture import _windows  # noqa: E402
from tests.test_nets_gpu import leaves, perturb  # noqa: E402


def rel(got, ref):
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    return (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


def oracle_run(x, labels, L, P, bn_training, forced=None, forced_pool=None, sites=None):
    calls, pcalls = [0], [0]

    def hook(t):
        i = calls[0]
        calls[0] += 1
        if sites is not None:
            sites.append(t.detach().clone())
        if forced is not None:
            return t * forced[i].to(t.dtype)
        return torch.relu(t)

    def pool_hook(t, ph, pw):
        i = pcalls[0]
        pcalls[0] += 1
        if forced_pool is not None:
            return torch.gather(_windows(t, ph, pw), -1, forced_pool[i].long().unsqueeze(-1)).squeeze(-1)
        return _windows(t, ph, pw).max(dim=-1).values

    Pc = {k: v.clone() for k, v in P.items()}
    lv = leaves(Pc)
    xr = x.clone().requires_grad_(True)
    O.RELU_HOOK, O.MAXPOOL_HOOK = hook, pool_hook
    try:
        ref = O.recognizer(xr, labels, 4 * L - 1, L, Pc, bn_training=bn_training)
    finally:
        O.RELU_HOOK = O.MAXPOOL_HOOK = None
    return ref, xr, lv


def main():
    dev = torch.device("cuda:0")
    NA.configure(device=dev, seed=3)
    for seed, B, L in ((6, 3, 3), (6, 3, 10), (11, 8, 5)):
        gen = torch.Generator().manual_seed(seed)
        R = NA.make_recognizer((32, 160, 1), None, 53, vis_model=False)
        P = perturb(R, gen)
        x = torch.rand(B, 32, 16 * L, 1, generator=gen, dtype=torch.float64) * 2 - 1
        labels = torch.randint(0, 52, (B, L), generator=gen)
        up = torch.rand(B, generator=gen, dtype=torch.float64) + 0.5
        for bn_training in (True, False):
            sites = []
            ref, xr, lv = oracle_run(x, labels, L, P, bn_training, sites=sites)
            (ref[:, 0] * up).sum().backward()
            for mode, (wino, tile) in (("F(4x4)", (True, 4)), ("F(2x2)", (True, 2)), ("direct", (False, 4))):
                ops.USE_WINOGRAD, ops.WINO_TILE = wino, tile
                R.trainable = bn_training
                R.store.load({k: v for k, v in P.items() if k.endswith((".mm", ".mv"))})
                loss, ctx = R.forward(x.float().to(dev), labels.int().to(dev), 4 * L - 1, L, training=True)
                R.store.zero_grad()
                dx = R.backward(ctx, up.float().to(dev), want_dx=True, want_dw=True)
                acts = ctx[0]
                forced = [(rec["a"] > 0).cpu() for rec in acts]
                fpool = [rec["idx"].cpu() for rec in acts if "idx" in rec]
                flips = [int((f != (s > 0)).sum()) for f, s in zip(forced, sites)]
                # the margin of the flipped decisions: |pre-activation| of the fp64 oracle where the HIP path decided differently
                margins = [float(s[f != (s > 0)].abs().max()) if n else 0.0 for f, s, n in zip(forced, sites, flips)]
                cf, cxr, clv = oracle_run(x, labels, L, P, bn_training, forced=forced, forced_pool=fpool)
                (cf[:, 0] * up).sum().backward()
                print("seed %d B %d L %d bn_training=%s %s: loss rel %.2e (cf %.2e)  dx rel %.2e (cf %.2e)  relu flips %s max|pre| %s"
                      % (seed, B, L, bn_training, mode, rel(loss, ref[:, 0]), rel(loss, cf[:, 0]), rel(dx, xr.grad), rel(dx, cxr.grad),
                         flips, ["%.1e" % m for m in margins]))
                for k, v in lv.items():
                    a, b = rel(R.store.g[k], v.grad), rel(R.store.g[k], clv[k].grad)
                    flag = "  <-- over 2e-3" if a > 2e-3 else ""
                    print("    %-10s vs oracle %.3e   vs counterfactual %.3e%s" % (k, a, b, flag))
    ops.USE_WINOGRAD, ops.WINO_TILE = True, 4


if __name__ == "__main__":
    main()
