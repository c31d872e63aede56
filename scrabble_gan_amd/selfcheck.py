"""smoke(): one tiny train_step through the HIP path on cuda:0, checked against the CPU oracle.
(The oracle is imported here only as the checker -- see oracle/scrabble_oracle.py's header.)"""
from __future__ import annotations

import numpy as np
import torch


def tiny_problem(dev, seed=8, B=2, L_r=2, L_f=3, style_w=32, loss_name="hinge", balance=False):
    """Build the four nets with perturbed weights plus inputs; returns everything both sides need."""
    from oracle import scrabble_oracle as O
    from . import net_architecture as NA
    NA.configure(device=dev, seed=3)
    gen = torch.Generator().manual_seed(seed)
    G = NA.make_generator(128, (32, 160, 1), (32, 8192), None, "B3", 52, vis_model=False)
    D = NA.make_discriminator((32, 160, 1), None, "B1", vis_model=False)
    R = NA.make_recognizer((32, 160, 1), None, 53, vis_model=False)
    S = NA.make_style_promoter((32, 160, 1), None, "B1", vis_model=False)
    gan = NA.make_gan(G, D, R, S, vis_model=False)
    P = {}
    for name, m in (("G", G), ("D", D), ("R", R), ("S", S)):
        w = m.store.export()
        for k, v in w.items():
            if k.endswith(".sigma"):
                w[k] = torch.tensor(0.3)
            elif k.endswith(".b") or k.endswith(".beta"):
                w[k] = torch.randn(v.shape, generator=gen) * 0.1
        m.store.load(w)
        P[name] = {k: v.double() for k, v in w.items()}
    images = torch.rand(B, 32, 16 * L_r, 1, generator=gen, dtype=torch.float64) * 2 - 1
    style = torch.rand(B, 32, style_w, 1, generator=gen, dtype=torch.float64) * 2 - 1
    labels = torch.randint(0, 52, (B, L_r), generator=gen)
    fake = torch.randint(0, 52, (B, L_f), generator=gen)
    nlo, nlg = {}, {}
    for n in ("G.style", "G.up", "D.fake", "D.real", "S.fake", "S.style", "S.real"):
        nlo[n] = O.init_nonlocal(64, gen)
        nlg[n] = {k: v.float().to(dev).contiguous() for k, v in nlo[n].items()}
    return dict(G=G, D=D, R=R, S=S, gan=gan, P=P, images=images, style=style, labels=labels, fake=fake, nlo=nlo, nlg=nlg,
                loss_name=loss_name, balance=balance, B=B)


def run_both(pb):
    from oracle import scrabble_oracle as O
    from . import data_utils as DU, net_loss, optimizers
    opt = {"G": {}, "D": {}, "R": {}, "S": {}}
    loss_o = O.hinge if pb["loss_name"] == "hinge" else O.not_saturating
    P = pb["P"]
    ref = O.train_step(pb["images"], pb["labels"], pb["style"], pb["fake"], P["G"], P["D"], P["S"], P["R"], pb["nlo"], opt,
                       loss_fn=loss_o, apply_gradient_balance=pb["balance"])
    opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
    out = DU.train_step(0, 0, 1, pb["images"].float().numpy(), pb["labels"].numpy().astype(np.int32), pb["D"], pb["R"], pb["S"],
                        pb["gan"], opts[0], opts[1], opts[2], opts[3], pb["style"].float().numpy(), pb["B"], 128,
                        getattr(net_loss, pb["loss_name"]), 1, int(pb["balance"]), None, 10, "",
                        fake_labels=pb["fake"].numpy().astype(np.int32), nl=pb["nlg"], verbose=False)
    return out, ref


def smoke() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs an MI355X")
    dev = torch.device("cuda:0")
    out, (ref_scalars, ref_grads, _) = run_both(tiny_problem(dev))
    for i, (a, b) in enumerate(zip(out, ref_scalars)):
        if abs(a - b) > 2e-4 * max(1.0, abs(b)):
            raise AssertionError("train_step scalar %d: HIP %r vs oracle %r" % (i, a, b))
    print("smoke ok: 16 train_step scalars match the CPU oracle:", ["%.4f" % float(v) for v in out])
