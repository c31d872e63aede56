"""A well-conditioned train_step problem for the composed-step parity tests (test infrastructure; imports the oracle).

Why a special fixture: with gradient balancing (data_utils.py:476-490 of the reference) G's upstream contains
d std(g_loss)/d g_b = (g_b - mean)/(B std) -- the NORMALISED deviations of the per-sample generator losses.  An
untrained D/S maps all fake images to almost the same logit (std/|mean| ~ 5e-3 with 0.1-sized random biases, B = 2),
so fp32 rounding of the logits is amplified by |logit|/std in G's gradient -- for the fp32 evaluation of the ORACLE just
as for the kernels.  Here the fake images are made to differ (style images with different mean levels, an amplified
z = Dense(GAP(.))), D/S get small biases (the reference initialises them to zero) and a scaled final Dense so that the
logits are O(1) with std ~ |mean|: the hinge kinks are exercised and std(g_loss) is O(1).

`calibrate()` evaluates the oracle in fp64 and in fp32 on the same problem: |fp32 - fp64| per tensor is the yardstick the
kernels are held to (tests/test_nets_gpu.py::test_train_step)."""
import torch

from oracle import scrabble_oracle as O

NL_NAMES = ("G.style", "G.up", "D.fake", "D.real", "S.fake", "S.style", "S.real")


_WEIGHTS = {}       # (seed, logit_scale, z_scale) -> (the four networks' fp64 weights, generator state behind them): the orthogonal
                    # initialisers (QR of up to 9216 x 1024 matrices in fp64) cost ~20 s per problem on the box's host; six tests share seed 8


def make_problem(B=8, L_r=2, L_f=2, style_w=32, seed=8, logit_scale=300.0, z_scale=30.0):
    gen = torch.Generator().manual_seed(seed)
    dt = torch.float64
    key = (seed, float(logit_scale), float(z_scale))
    if key in _WEIGHTS:
        P0, state = _WEIGHTS[key]
        P = {n: {k: v.clone() for k, v in W.items()} for n, W in P0.items()}
        gen.set_state(state)
    else:
        P = {"G": O.init_generator(gen, dt), "D": O.init_discriminator(gen, dt), "S": O.init_discriminator(gen, dt),
             "R": O.init_recognizer(gen, dt)}
        for n, W in P.items():
            for k, v in W.items():
                if k.endswith(".sigma"):
                    W[k] = torch.tensor(0.3, dtype=dt)
                elif k.endswith(".b") or k.endswith(".beta"):
                    W[k] = torch.randn(v.shape, generator=gen, dtype=dt) * (0.01 if n in ("D", "S") else 0.1)
                elif k.endswith(".gamma"):
                    W[k] = 1 + torch.randn(v.shape, generator=gen, dtype=dt) * 0.1
                elif k.endswith(".mm"):
                    W[k] = torch.randn(v.shape, generator=gen, dtype=dt) * 0.1
                elif k.endswith(".mv"):
                    W[k] = 1 + torch.rand(v.shape, generator=gen, dtype=dt) * 0.2
        P["D"]["dense.w"] = P["D"]["dense.w"] * logit_scale
        P["S"]["dense.w"] = P["S"]["dense.w"] * logit_scale
        P["G"]["zdense.w"] = P["G"]["zdense.w"] * z_scale
        if len(_WEIGHTS) >= 2:
            _WEIGHTS.clear()
        _WEIGHTS[key] = ({n: {k: v.clone() for k, v in W.items()} for n, W in P.items()}, gen.get_state())
    images = torch.rand(B, 32, 16 * L_r, 1, generator=gen, dtype=dt) * 2 - 1
    noise = torch.rand(B, 32, style_w, 1, generator=gen, dtype=dt) * 2 - 1
    style = (0.3 * noise + torch.linspace(-1, 1, B, dtype=dt).view(B, 1, 1, 1)).clamp(-1, 1)
    labels = torch.randint(0, 52, (B, L_r), generator=gen)
    fake = torch.randint(0, 52, (B, L_f), generator=gen)
    nl = {n: O.init_nonlocal(64, gen) for n in NL_NAMES}
    return dict(P=P, images=images, style=style, labels=labels, fake=fake, nl=nl, B=B, L_r=L_r, L_f=L_f)


# ReLU / MaxPool2D sites of one oracle train_step in call order (O.train_step: G, D(x_f), S(x_f), R(x_f), D(real), S(style),
# S(real), R(real)).  ReLU: G 16 (style encoder B_style1..4 x (relu(x), relu(conv1)), relu before GAP, B1..B3 x (cbn1, cbn2),
# final BN), D / S 9 each (4 blocks x 2 + the one before GAP), R 7 (one per conv).  MaxPool: 2 per NonLocalBlock (phi, g), 4 per R.
RELU_RANGE = {"G": (0, 16), "D_f": (16, 25), "S_f": (25, 34), "R_f": (34, 41), "D_r": (41, 50), "S_my": (50, 59), "S_r": (59, 68), "R_r": (68, 75)}
POOL_RANGE = {"G": (0, 4), "D_f": (4, 6), "S_f": (6, 8), "R_f": (8, 12), "D_r": (12, 14), "S_my": (14, 16), "S_r": (16, 18), "R_r": (18, 22)}
G_RELU_SITES = 16


def _windows(x, ph, pw):
    B, H, W, C = x.shape
    return x.reshape(B, H // ph, ph, W // pw, pw, C).permute(0, 1, 3, 5, 2, 4).reshape(B, H // ph, W // pw, C, ph * pw)


def run_oracle(pb, dtype, loss_name="hinge", balance=False, relu_sites=None, forced=None, pool_sites=None, forced_pool=None):
    """-> (16 scalars, {net: {name: grad}}, {net: {name: post-update weight}}, fake images); pb is not modified.
    relu_sites / pool_sites (lists, optional): receive the PRE-activation tensor of every ReLU site / the input of every
    MaxPool2D site of the step, in call order (RELU_RANGE / POOL_RANGE).
    forced (dict site -> bool tensor) / forced_pool (dict site -> window-position tensor, the encoding of sg_maxpool_fwd):
    the counterfactual oracle -- at those sites the DECISION is imposed (y = x * mask, resp. y = the chosen window element, so
    the backward routing follows) instead of taken from the fp64 values."""
    calls, pcalls = [0], [0]

    def hook(x):
        i = calls[0]
        calls[0] += 1
        if relu_sites is not None:
            relu_sites.append(x.detach().clone())
        if forced is not None and forced.get(i) is not None:
            return x * forced[i].to(x.dtype)
        return torch.relu(x)

    def pool_hook(x, ph, pw):
        i = pcalls[0]
        pcalls[0] += 1
        if pool_sites is not None:
            pool_sites.append((x.detach().clone(), ph, pw))
        if forced_pool is not None and forced_pool.get(i) is not None:
            return torch.gather(_windows(x, ph, pw), -1, forced_pool[i].long().unsqueeze(-1)).squeeze(-1)
        return _windows(x, ph, pw).max(dim=-1).values

    hooked = relu_sites is not None or forced is not None or pool_sites is not None or forced_pool is not None
    O.RELU_HOOK = hook if hooked else None
    O.MAXPOOL_HOOK = pool_hook if hooked else None
    try:
        return _run_oracle(pb, dtype, loss_name, balance)
    finally:
        O.RELU_HOOK = None
        O.MAXPOOL_HOOK = None


def _run_oracle(pb, dtype, loss_name, balance):
    cast = lambda t: t.to(dtype) if t.is_floating_point() else t
    P = {n: {k: cast(v.clone()) for k, v in W.items()} for n, W in pb["P"].items()}
    nl = {n: {k: cast(v) for k, v in d.items()} for n, d in pb["nl"].items()}
    opt = {"G": {}, "D": {}, "R": {}, "S": {}}
    loss = O.hinge if loss_name == "hinge" else O.not_saturating
    scalars, grads, x_f = O.train_step(cast(pb["images"]), pb["labels"], cast(pb["style"]), pb["fake"], P["G"], P["D"], P["S"], P["R"],
                                       nl, opt, loss_fn=loss, apply_gradient_balance=balance)
    return scalars, grads, P, x_f


GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def calibrate(pb, loss_name="hinge", balance=False, tag=None):
    """fp64 reference + the per-tensor deviation of the oracle's own fp32 evaluation from it.
    `tag`: the fp32 deviations are read from the committed table tests/golden/calibration_<tag>.json when it exists (the problems
    are seeded, the oracle is deterministic on the CPU: the table saves one oracle evaluation per case -- 12-20 s of the GPU
    suite's host time each; `python -m tests.step_fixture` regenerates the tables, and a run without a table computes the
    deviations as before)."""
    import json
    import os
    sites64, pools64 = [], []
    s64, g64, w64, x64 = run_oracle(pb, torch.float64, loss_name, balance, relu_sites=sites64, pool_sites=pools64)
    path = os.path.join(GOLDEN, "calibration_%s.json" % tag) if tag else None
    if path and os.path.exists(path):
        t = json.load(open(path))
        assert set(t["err32"]) == set(g64) and all(set(t["err32"][n]) == set(g64[n]) for n in g64), "stale calibration table " + path
        # the table belongs to THIS problem: its fp64 scalars are stored beside the deviations
        assert all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(s64, t["scalars64"])), "calibration table %s is for another problem" % path
        err32, l2err32, serr32, xerr32 = t["err32"], t["l2err32"], t["scalar_err32"], t["x_err32"]
    else:
        s32, g32, w32, x32 = run_oracle(pb, torch.float32, loss_name, balance)
        err32 = {n: {k: (g32[n][k].double() - v).abs().max().item() for k, v in g64[n].items()} for n in g64}
        l2err32 = {n: {k: (g32[n][k].double() - v).norm().item() for k, v in g64[n].items()} for n in g64}
        serr32 = [abs(a - b) for a, b in zip(s32, s64)]
        xerr32 = (x32.double() - x64).abs().max().item()
        if path and os.environ.get("SG_WRITE_CALIBRATION") == "1":
            json.dump({"scalars64": [float(v) for v in s64], "err32": err32, "l2err32": l2err32, "scalar_err32": [float(v) for v in serr32],
                       "x_err32": xerr32}, open(path, "w"), indent=0)
    return dict(scalars=s64, grads=g64, weights=w64, x_f=x64, err32=err32, l2err32=l2err32, scalar_err32=serr32, x_err32=xerr32,
                relu_sites64=sites64, pool_sites64=pools64)


# the three calibrated problems of the GPU suite (tests/test_nets_gpu.py::test_train_step, tests/test_configs_gpu.py::test_c4_...)
CALIBRATED = {"not_saturating_1_L3": (dict(B=4, L_r=2, L_f=3, style_w=32, seed=8, logit_scale=70.0), "not_saturating", True),
              "hinge_1_L2": (dict(B=4, L_r=2, L_f=2, style_w=32, seed=8, logit_scale=70.0), "hinge", True),
              "c4_Lr3_Lf2": (dict(B=4, L_r=3, L_f=2, style_w=160, seed=23, logit_scale=70.0), "hinge", False)}


def load_models(NA, pb, dev):
    """The four networks of the HIP path carrying problem `pb`'s weights, + their NonLocalBlock kernels on the device."""
    from scrabble_gan_amd import nn as _nn
    _nn.FAST_INIT = True                # every weight is loaded from the fixture below: skip the QR initialisers
    try:
        G = NA.make_generator(128, (32, 160, 1), (32, 8192), None, "B3", 52, vis_model=False)
        D = NA.make_discriminator((32, 160, 1), None, "B1", vis_model=False)
        R = NA.make_recognizer((32, 160, 1), None, 53, vis_model=False)
        S = NA.make_style_promoter((32, 160, 1), None, "B1", vis_model=False)
    finally:
        _nn.FAST_INIT = False
    gan = NA.make_gan(G, D, R, S, vis_model=False)
    models = {"G": G, "D": D, "R": R, "S": S}
    for n, m in models.items():
        assert set(m.store.names) == set(pb["P"][n]), (n, set(m.store.names) ^ set(pb["P"][n]))
        m.store.load({k: v.float() for k, v in pb["P"][n].items()})
    nlg = {n: {k: v.float().to(dev).contiguous() for k, v in d.items()} for n, d in pb["nl"].items()}
    return models, gan, nlg


def _nl_pool_indices(nlc):
    """The two max-pool selections (phi, g) of one NonLocalBlock call from its saved context (plain or one-segment form)."""
    if isinstance(nlc, tuple) and len(nlc) == 2 and nlc[0] == "seg":
        assert len(nlc[1]) == 1, "one NonLocalBlock call per sliced context"
        nlc = nlc[1][0][2]
    return [nlc[3].cpu(), nlc[5].cpu()]


def _trunk_decisions(tctx):
    """ReLU decisions (9) and NonLocalBlock max-pool selections (2) of one pass through a D-shaped trunk."""
    ctxs, net = tctx
    relu, pool = [], []
    for (x, c1, _xp), nlc in ctxs:
        relu += [(x > 0).cpu(), (c1 > 0).cpu()]
        if nlc is not None:
            pool += _nl_pool_indices(nlc)
    relu.append((net > 0).cpu())
    return relu, pool


def generator_decisions(ctx_g):
    """ReLU decisions (16) and max-pool selections (4) of one generator forward from its saved context, in the oracle's call order."""
    tctx, h, z, y, up_ctx, bctx, yb, img, S = ctx_g
    dec, pl = _trunk_decisions(tctx)
    for (c, nlc) in up_ctx:
        x_in, c1, c2, stride = c
        dec += [(c1[1] > 0).cpu(), (c2[1] > 0).cpu()]
        if nlc is not None:
            pl += _nl_pool_indices(nlc)
    dec.append((yb > 0).cpu())
    assert len(dec) == G_RELU_SITES and len(pl) == 4
    return dec, pl


def recognizer_decisions(ctx_r):
    """ReLU decisions (7) and max-pool selections (4) of one recognizer forward from its saved context."""
    acts = ctx_r[0]
    dec = [(rec["a"] > 0).cpu() for rec in acts]
    pl = [rec["idx"].cpu() for rec in acts if "idx" in rec]
    assert len(pl) == 4 and len(dec) == 7
    return dec, pl


def hip_decisions(keep):
    """The ReLU / max-pool decisions the HIP forward passes took, from the contexts train_step saved (DEBUG_KEEP):
    -> ({relu site: bool tensor}, {pool site: uint8 window positions}) for ALL sites of the step: the generator (16 ReLU, 4 pools),
    every discriminator / style-promoter call (9 + 2 each) and both recognizer calls (7 + 4 each), indexed as RELU_RANGE /
    POOL_RANGE.  The D-shaped trunks apply their ReLUs in the consumers' operand loaders (decision = sign of the saved fp32
    pre-activation); the up blocks, the final BatchNorm and the recognizer's convolutions materialise relu(.) (decision =
    output > 0)."""
    relu, pool = {}, {}
    dec, pl = generator_decisions(keep["ctx_g"])
    relu.update(enumerate(dec))
    pool.update(enumerate(pl))
    for tag in ("D_f", "S_f", "D_r", "S_my", "S_r"):
        if tag not in keep:
            continue
        dec, pl = _trunk_decisions(keep[tag][0])
        assert len(dec) == 9 and len(pl) == 2
        relu.update((RELU_RANGE[tag][0] + i, d) for i, d in enumerate(dec))
        pool.update((POOL_RANGE[tag][0] + i, d) for i, d in enumerate(pl))
    for tag in ("R_f", "R_r"):
        dec, pl = recognizer_decisions(keep[tag])
        relu.update((RELU_RANGE[tag][0] + i, d) for i, d in enumerate(dec))
        pool.update((POOL_RANGE[tag][0] + i, d) for i, d in enumerate(pl))
    return relu, pool


# ---- the counterfactual oracle for single networks ----------------------------------------------------------------------------
# Gradients are discontinuous in the forward pass's DECISIONS (ReLU on / off, which element of a max-pool window is taken): where
# the fp64 pre-activation is a near-tie (|x| at the fp32 rounding level of the layer), the fp32 forward may decide the other way
# and the gradient then differs by O(1) in single elements -- for ANY fp32 evaluation, the kernels' and the oracle's own alike.
# The checker therefore separates the two questions:
#   (1) every decision of the HIP path that differs from the fp64 oracle's IS a near-tie (margin <= TIE_MARGIN of the site's
#       largest magnitude; everything else must agree exactly), and
#   (2) given the same decisions, the gradients agree to the fixed per-tensor bounds (no widened bars, no fp32-oracle yardstick).
TIE_MARGIN = 2e-5       # = the convolutions' own forward tolerance against the fp64 oracle (tests/test_ops_gpu.py)


def forced(fn, relu, pool):
    """-> (fn(), report): fn evaluates an oracle network; at ReLU site i (call order) the decision relu[i] (bool tensor) is imposed
    (y = x * mask), at MaxPool2D site j the window positions pool[j] (the encoding of sg_maxpool_fwd) -- the backward routing
    follows.  report = {"relu": [(site, differing decisions, largest |pre-activation| among them / max |pre-activation|)],
    "pool": [(site, differing selections with a value gap, largest gap / max |input|)]}."""
    calls, pcalls = [0], [0]
    rep = {"relu": [], "pool": []}

    def hook(x):
        i = calls[0]
        calls[0] += 1
        f = relu[i].reshape(x.shape)
        xd = x.detach()
        diff = f != (xd > 0)
        n = int(diff.sum())
        rep["relu"].append((i, n, float(xd[diff].abs().max() / (xd.abs().max() + 1e-300)) if n else 0.0))
        return x * f.to(x.dtype)

    def pool_hook(x, ph, pw):
        j = pcalls[0]
        pcalls[0] += 1
        w = _windows(x, ph, pw)
        chosen = torch.gather(w, -1, pool[j].long().reshape(w.shape[:-1]).unsqueeze(-1)).squeeze(-1)
        gap = (w.max(dim=-1).values - chosen).detach()
        n = int((gap > 0).sum())
        rep["pool"].append((j, n, float(gap.max() / (x.detach().abs().max() + 1e-300)) if n else 0.0))
        return chosen

    O.RELU_HOOK, O.MAXPOOL_HOOK = hook, pool_hook
    try:
        out = fn()
    finally:
        O.RELU_HOOK = O.MAXPOOL_HOOK = None
    assert calls[0] == len(relu) and pcalls[0] == len(pool), (calls[0], len(relu), pcalls[0], len(pool))
    return out, rep


def assert_near_ties(rep, name=""):
    """Question (1): the decisions that differ between the fp32 forward pass and the fp64 oracle are all near-ties."""
    for kind in ("relu", "pool"):
        for site, n, margin in rep[kind]:
            assert margin <= TIE_MARGIN, "%s %s site %d: %d decisions differ from the fp64 oracle, largest margin %.2e of the site's scale > %.0e" % (
                name, kind, site, n, margin, TIE_MARGIN)


def flips(rep):
    return sum(n for _s, n, _m in rep["relu"]), sum(n for _s, n, _m in rep["pool"])


if __name__ == "__main__":          # regenerate tests/golden/calibration_*.json (CPU only: two oracle evaluations per problem)
    import os
    os.environ["SG_WRITE_CALIBRATION"] = "1"
    for tag, (kw, loss_name, balance) in CALIBRATED.items():
        path = os.path.join(GOLDEN, "calibration_%s.json" % tag)
        if os.path.exists(path):
            os.remove(path)
        calibrate(make_problem(**kw), loss_name, balance, tag=tag)
        print("wrote", path)
