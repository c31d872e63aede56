"""The host loop end to end on the GPU: gin config -> factories -> train() on synthetic variable-width buckets
(one word length per batch, L in [1,10]) -> 16-column summaries and per-epoch G/R weight files; and the widest
bucket of config c4 (L = 23) through every network."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_main_synthetic_writes_summaries_and_checkpoints(dev, tmp_path, monkeypatch):
    from scrabble_gan_amd import gin_config as gin, main as M
    cfg = tmp_path / "cfg.gin"
    base = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "scrabble_gan_mi355x.gin")).read()
    cfg.write_text(base + "\nio.base_path = '%s/'\nshared_specs.batch_size = 4\nshared_specs.num_gen = 4\n" % tmp_path)
    gin.clear_config()
    M.main(["--gin", str(cfg), "--synthetic", "--steps", "3", "--epochs", "1", "--resumable"])
    out = tmp_path / "run" / "output"
    rows = (out / "batch_summary.txt").read_text().strip().split("\n")
    assert rows[0].startswith("disc_loss;disc_loss_real") and len(rows) == 4
    for r in rows[1:]:
        cols = r.split(";")
        assert len(cols) == 16 and all(np.isfinite(float(c)) for c in cols)
    assert len((out / "epoch_summary.txt").read_text().strip().split("\n")) == 2
    for net in ("generator", "recognizer"):
        assert (tmp_path / "run" / "checkpoints" / net / "1" / "cktp-1.safetensors").exists()
    gin.clear_config()
    # the inference script on the checkpoint just written (run_inference.py of the reference): two word lengths, 3 styles
    from PIL import Image
    from scrabble_gan_amd import run_inference as RI
    png = tmp_path / "gen" / "words.png"
    RI.main(["--weights", str(tmp_path / "run" / "checkpoints" / "generator" / "1" / "cktp-1"), "--words", "machine", "ab", "learn",
             "--synthetic-style", "--count", "3", "--out", str(png)])
    im = Image.open(png)
    assert im.mode == "L" and im.size == (16 * 7, 9 * 32 + 8 * 2)      # 3 words x 3 styles, 2-pixel separators, widest word 7 chars
    assert RI.encode("aZ") == [0, 51]
    # --resumable: the full state of epoch 1 is on disk; asking for 2 epochs now runs ONLY the second one
    assert (tmp_path / "run" / "checkpoints" / "state" / "latest.safetensors").exists()
    M.main(["--gin", str(cfg), "--synthetic", "--steps", "3", "--epochs", "2", "--resumable"])
    assert len((out / "epoch_summary.txt").read_text().strip().split("\n")) == 3          # header + epoch 1 + epoch 2
    assert len((out / "batch_summary.txt").read_text().strip().split("\n")) == 7          # header + 3 + 3 steps
    assert (tmp_path / "run" / "checkpoints" / "generator" / "2" / "cktp-2.safetensors").exists()
    gin.clear_config()


def test_widest_bucket_and_bilstm_recognizer_step(dev):
    """L_r = 23 real words, L_f = 4 fakes, the BiLSTM recognizer with device-drawn dropout: one finite step."""
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, optimizers
    NA.configure(device=dev, seed=2)
    G = NA.make_generator(128, (32, 160, 1), (32, 8192), None, "B3", 52, vis_model=False)
    D = NA.make_discriminator((32, 160, 1), None, "B1", vis_model=False)
    R = NA.make_my_recognizer((32, 160, 1), None, 53, vis_model=False)
    S = NA.make_style_promoter((32, 160, 1), None, "B1", vis_model=False)
    gan = NA.make_gan(G, D, R, S, vis_model=False)
    opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(3)] + [optimizers.RMSprop(2e-4)]
    images, labels, my_imgs = DU.synthetic_batch(2, 23, seed=5)
    words = DU.synthetic_random_words(23, 20, seed=5)
    fake = np.array(words[3][:2], np.int32)
    before = G.store.flat.clone()
    out = DU.train_step(0, 0, 1, images, labels, D, R, S, gan, opts[0], opts[1], opts[3], opts[2], my_imgs, 2, 128, net_loss.hinge, 1, 1,
                        words, 23, "", fake_labels=fake, verbose=False)
    assert len(out) == 16 and all(np.isfinite(float(v)) for v in out)
    assert torch.isfinite(G.store.flat).all() and not torch.equal(before, G.store.flat)
    img = G([my_imgs, np.array(words[22][:2], np.int32)], training=False)            # inference path, 23-char words
    assert tuple(img.shape) == (2, 32, 368, 1) and img.abs().max().item() <= 1.0


def test_shared_sweeps_and_fused_passes_equal_the_reference_schedule(dev):
    """train_step's default schedule (fused passes over concatenated batches, ONE backward sweep through D(x_f) / S(x_f)
    serving both the weight and the image gradient) against the reference's own schedule (every call its own pass, every
    tape its own sweep: fuse_passes=False, share_backward=False) on identical weights and inputs, WITH gradient balancing:
    same 16 scalars (1e-5), same gradients of all four networks -- G included -- within 1e-3 of the network's largest
    gradient, same post-Adam weights.  The problem is the well-conditioned B = 8 fixture of tests/step_fixture.py
    (std(g_loss), std(r_fake) = O(1); all widths equal, so every pass fuses) -- VERDICT r2 weak #4(b): the former B = 4
    random-weight problem divided by a near-zero std and needed 5e-2 for G."""
    import numpy as np
    import torch
    from tests import step_fixture as F
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, optimizers
    from scrabble_gan_amd import ops
    pb = F.make_problem(B=8, L_r=2, L_f=2, style_w=32, seed=8, logit_scale=70.0)
    B = pb["B"]
    res = {}
    ops.set_deterministic(True)        # batch-size dependent reduction splits would move forward bits (see test_dp_gpu.py)
    try:
        for mode, kw in (("default", {}), ("reference", {"fuse_passes": False, "share_backward": False})):
            NA._model_counter[0] = 0
            NA.configure(device=dev, seed=9)
            models, gan, nlg = F.load_models(NA, pb, dev)
            G, D, R, S = (models[n] for n in ("G", "D", "R", "S"))
            opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
            out = DU.train_step(0, 0, 1, pb["images"].float().numpy(), pb["labels"].numpy().astype(np.int32), D, R, S, gan, opts[0], opts[1],
                                opts[2], opts[3], pb["style"].float().numpy(), B, 128, net_loss.hinge, 1, 1, None, 10, "",
                                fake_labels=pb["fake"].numpy().astype(np.int32), nl=nlg, verbose=False, **kw)
            res[mode] = (np.array(out, np.float64), {n: (m.store.grad.clone(), m.store.flat.clone()) for n, m in models.items()})
    finally:
        ops.set_deterministic(False)
    sa, ga = res["default"]
    sb, gb = res["reference"]
    assert sb[12] > 0.05 and sb[11] > 0.05, "fixture lost its conditioning: %r" % (sb[11:13],)
    assert np.all(np.abs(sa - sb) <= 1e-5 * np.maximum(1.0, np.abs(sb))), (sa, sb)
    for n in ("D", "R", "S", "G"):
        (da, wa), (db, wb) = ga[n], gb[n]
        scale = db.abs().max().item()
        # fp32 summation order is all that differs between the schedules
        assert (da - db).abs().max().item() <= 1e-3 * scale, "%s gradients: %.3e vs scale %.3e" % (n, (da - db).abs().max().item(), scale)
        assert (wa - wb).abs().max().item() <= 5e-4, n


def test_schedules_agree_at_headline_tile_geometry(dev):
    """The same equivalence at a bs-128-like TILE geometry: B = 32 words of L = 10 (32x160 crops, the fused
    passes carry 64 / 96 samples -> 2 560+ output tiles per large conv launch: full rounds, the XCD remap AND the
    reduction-split tail are all active, nothing is forced through sg_debug_set_splitk).  Default schedule (fused passes,
    shared sweeps) vs fuse_passes=False / share_backward=False, hinge, no balancing, well-conditioned logits (D/S Dense
    scaled as in tests/step_fixture.py).  fp32 summation order is all that differs: scalars 1e-5, gradients 3e-4 of the
    network's largest gradient (sums over 3.3 M pixels in different orders; measured on MI355X: <= 1.3e-4 for D / S, 3e-5 for
    G and R, profiles/r02_schedule_equivalence_bs32.txt)."""
    import numpy as np
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, nn, optimizers
    B, L = 32, 10
    images, labels, style = DU.synthetic_batch(B, L, seed=31)
    words = DU.synthetic_random_words(10, 100, seed=31)
    fake = np.array(words[L - 1][:B], np.int32)
    res = {}
    for mode, kw in (("default", {}), ("unfused", {"fuse_passes": False}), ("unshared", {"share_backward": False})):
        NA._model_counter[0] = 0
        NA.configure(device=dev, seed=9)
        G = NA.make_generator(128, (32, 160, 1), (32, 8192), None, "B3", 52, vis_model=False)
        D = NA.make_discriminator((32, 160, 1), None, "B1", vis_model=False)
        R = NA.make_recognizer((32, 160, 1), None, 53, vis_model=False)
        S = NA.make_style_promoter((32, 160, 1), None, "B1", vis_model=False)
        gan = NA.make_gan(G, D, R, S, vis_model=False)
        for m in (G, D, S):
            for k in m.store.names:
                if k.endswith(".sigma"):
                    m.store.p[k].fill_(0.25)
        for m in (D, S):
            m.store.p["dense.w"].mul_(70.0)
        g2 = torch.Generator().manual_seed(5)
        nl = {n: {k: v.to(dev) for k, v in nn.nonlocal_weights(64, g2, torch.device("cpu")).items()}
              for n in ("G.style", "G.up", "D.fake", "D.real", "S.fake", "S.style", "S.real")}
        opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
        out = DU.train_step(0, 0, 1, images, labels, D, R, S, gan, opts[0], opts[1], opts[2], opts[3], style, B, 128,
                            net_loss.hinge, 1, 0, None, 10, "", fake_labels=fake, nl=nl, verbose=False, **kw)
        res[mode] = (np.array(out, np.float64), {n: m.store.grad.clone() for n, m in (("G", G), ("D", D), ("R", R), ("S", S))})
        del G, D, R, S, gan
    sa, ga = res["default"]
    lines, bad = [], []
    for other in ("unfused", "unshared"):
        sb, gb = res[other]
        lines.append("%s scalars rel %.3e" % (other, float(np.max(np.abs(sa - sb) / np.maximum(1.0, np.abs(sb))))))
        if not np.all(np.abs(sa - sb) <= 1e-5 * np.maximum(1.0, np.abs(sb))):
            bad.append("%s scalars %r vs %r" % (other, sa, sb))
        for n in ("D", "R", "S", "G"):
            scale = gb[n].abs().max().item()
            err = (ga[n] - gb[n]).abs().max().item()
            lines.append("%s %s rel %.3e" % (other, n, err / scale))
            if not err <= 3e-4 * scale:
                bad.append("%s: %s gradients differ by %.3e of %.3e" % (other, n, err, scale))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        open("gpurun_out/schedule_equivalence_bs32.txt", "w").write("\n".join(lines) + "\n")
    except OSError:
        pass
    assert not bad, "\n".join(bad)


def test_full_state_save_and_resume(dev, tmp_path):
    """save_training_state / load_training_state (the reference cannot resume): three steps in one go against
    two steps, save, rebuild every object from scratch, load, one more step -- same weights, BN statistics and Adam state."""
    import random
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, optimizers
    B, L = 4, 3
    images, labels, my_imgs = DU.synthetic_batch(B, L, seed=3)
    words = DU.synthetic_random_words(10, 50, seed=3)

    def build():
        NA._model_counter[0] = 0
        NA.configure(device=dev, seed=4)
        G = NA.make_generator(128, (32, 160, 1), (32, 8192), None, "B3", 52, vis_model=False)
        D = NA.make_discriminator((32, 160, 1), None, "B1", vis_model=False)
        R = NA.make_recognizer((32, 160, 1), None, 53, vis_model=False)
        S = NA.make_style_promoter((32, 160, 1), None, "B1", vis_model=False)
        return G, D, R, S, NA.make_gan(G, D, R, S, vis_model=False), [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]

    def steps(objs, first, n):
        G, D, R, S, gan, opts = objs
        for i in range(first, first + n):
            random.seed(100 + i)                              # the fake-label draw of step i
            torch.manual_seed(100 + i)
            G.nl_gen.manual_seed(1000 + i); D.nl_gen.manual_seed(2000 + i); S.nl_gen.manual_seed(3000 + i)   # per-call kernel re-draws
            DU.train_step(0, i, 9, images, labels, D, R, S, gan, opts[0], opts[1], opts[2], opts[3], my_imgs, B, 128, net_loss.hinge, 1, 0,
                          words, 10, "", verbose=False)

    # Both runs in the deterministic mode (configure(deterministic=True)): every conv output and every dW / db element has
    # one adder, so what still differs between two runs of the same step is the float-atomic remainder listed in
    # ops.set_deterministic (BatchNorm's per-sample partials, the filter bank's dz, the attention key sweep at this tiny batch).
    from scrabble_gan_amd import ops
    ops.set_deterministic(True)
    try:
        a = build()
        steps(a, 0, 3)
        b = build()
        steps(b, 0, 2)
        path = str(tmp_path / "state" / "latest.safetensors")
        DU.save_training_state(path, b[0], b[1], b[2], b[3], *b[5], epoch_idx=0, batch_idx=2)
        c = build()                                               # fresh weights, fresh optimizers
        assert DU.load_training_state(path, c[0], c[1], c[2], c[3], *c[5]) == (0, 2)
        assert all(o.iterations == 2 for o in c[5])
        steps(c, 2, 1)
    finally:
        ops.set_deterministic(False)
    lines, checks = [], []
    for ma, mc, oa, oc in zip(a[:4], c[:4], a[5], c[5]):
        assert oa.iterations == oc.iterations == 3
        # Float-atomic summation order moves gradients in their last bits; Adam (beta_1 = 0) normalises every component to
        # +-lr, so a component whose gradient is ~0 may step the other way: a handful of entries differ by up to 2 lr,
        # the mean difference stays orders of magnitude below lr.  A state that was NOT restored moves every entry by O(lr).
        diff = (ma.store.flat - mc.store.flat).abs()
        # (worst case: a component that steps the other way in each of the 3 steps = 3 x 2 lr = 1.2e-3)
        assert diff.max().item() <= 3 * 2 * 2e-4 * 1.1 and diff.mean().item() <= 2e-5, (ma.name, diff.max().item(), diff.mean().item())
        assert (ma.store.state - mc.store.state).abs().max().item() <= 1e-5, ma.name
        for k in ("m", "v"):
            sa, sc = oa.flat_state(ma.store)[k], oc.flat_state(mc.store)[k]
            # (the last gradient and its running square; deterministic mode leaves the float-atomic remainder, amplified
            #  where a ReLU / max-pool decision sits on the edge: 1e-3 of the slot's largest entry, 5e-3 before dW had a fixed order)
            err, scale = (sa - sc).abs().max().item(), sa.abs().max().item()
            lines.append("%s %s: max |resumed - continuous| %.3e of %.3e (rel %.2e)" % (ma.name, k, err, scale, err / (scale + 1e-30)))
            # The float-atomic remainder of the deterministic mode (G's batch-of-4 BatchNorm partials, the attention key sweep)
            # moves the fake images in their last bits; that reaches D / S through ReLU / max-pool decisions on the edge.
            # Measured on MI355X (profiles/r03_resume_vs_continuous.txt): R 2e-7, S 5e-4, G 1.9e-3, D 1.2e-2 of the slot's
            # largest entry.  The bar only has to separate "restored" from "not restored" (O(1) relative): 3e-2.
            checks.append((err <= 3e-2 * scale + 1e-12, lines[-1]))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        open("gpurun_out/resume_vs_continuous.txt", "w").write("\n".join(lines) + "\n")
    except OSError:
        pass
    assert all(ok for ok, _ in checks), "\n".join(ln for ok, ln in checks if not ok)


def test_device_prefetcher_matches_the_host_loader(dev, tmp_path):
    """bucket folders -> background thread -> pinned staging -> async H2D of the uint8 pixels -> GPU-side
    (x - 127.5) / 127.5.  Same seeds, same draw order: every batch is BIT-identical to what the reference-style host loader
    (load_prepare_data, data_utils.py:14-84) produces, and train_step accepts the device tensors."""
    import random
    from PIL import Image
    from scrabble_gan_amd import data_io
    cv = 'abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'
    rng = np.random.default_rng(0)
    for L, words in ((1, ["a", "Z", "q"]), (2, ["to", "Hi", "ab"]), (3, ["the", "Cat"])):
        d = tmp_path / str(L)
        d.mkdir()
        for i, wd in enumerate(words):
            Image.fromarray(rng.integers(0, 256, (32, 16 * L), dtype=np.uint8), mode="L").save(d / ("w%d.png" % i))
            (d / ("w%d.txt" % i)).write_text(wd)
    random.seed(3)
    np.random.seed(3)
    host = data_io.load_prepare_data((32, 160, 1), 5, str(tmp_path) + "/", cv, 3)
    want = [next(host) for _ in range(6)]
    random.seed(3)
    np.random.seed(3)
    pre = data_io.DevicePrefetcher(data_io.load_prepare_data((32, 160, 1), 5, str(tmp_path) + "/", cv, 3, raw=True), dev, depth=2)
    for (im_h, lab_h), _ in zip(want, range(6)):
        im_d, lab_d = next(pre)
        assert im_d.is_cuda and im_d.dtype == torch.float32 and tuple(im_d.shape) == im_h.shape
        assert np.array_equal(lab_d, lab_h)
        assert np.array_equal(im_d.cpu().numpy(), im_h), "GPU-side normalisation must be bit-identical to the numpy expression"
    pre.close()
    # The loader shares the global `random` / `np.random` streams with train_step's fake-label draw and main.py's style
    # draw (data_utils.py:386-392).  Two seeded runs that interleave those draws with next(prefetcher) must see the same
    # batches AND the same fake labels (a generator pulled from a background thread interleaves by timing).
    from scrabble_gan_amd import data_utils as DU
    words = DU.synthetic_random_words(3, 20, seed=1)

    def run():
        random.seed(11)
        np.random.seed(11)
        pf = data_io.DevicePrefetcher(data_io.load_prepare_data((32, 160, 1), 5, str(tmp_path) + "/", cv, 3, raw=True), dev, depth=3)
        seen = []
        for i in range(8):
            picks = random.choices(range(100), k=3)                  # main.py's style draw
            im, lab = next(pf)
            idx, fl = DU.draw_fake_labels(words, 3, 4)               # train_step's draw
            if i % 3 == 0:
                import time
                time.sleep(0.02)                                     # give a racing thread every chance
            seen.append((picks, im.cpu().numpy().tobytes(), lab.tobytes(), idx, fl.tobytes()))
        pf.close()
        return seen

    assert run() == run()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_graph_captured_step_equals_eager_steps(dev, mode):
    """scrabble_gan_amd.graph_step.GraphedStep: a train_step captured once into a HIP graph and replayed -- the per-call
    NonLocalBlock kernels and Adam's bias-corrected step size reach the replay through device memory.  After the capture (two
    eager warm-up steps inside) the whole training state is snapshotted; ONE graph replay from that state is compared with ONE
    eager train_step from the same state, same inputs, same NonLocalBlock kernels: the 16 scalars at 1e-4 (fp32) / 1e-3 (bf16; 2e-2 for the
    ill-conditioned std / balancing entries:
    an atomic-order difference that crosses a bf16 rounding boundary), the post-Adam weights within the Adam bar (a component
    whose gradient is ~0 may step the other way: <= 3.5 lr, mean << lr), optimizer counters advanced by the replay."""
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, ops, optimizers
    from scrabble_gan_amd.graph_step import GraphedStep
    B, L = 8, 3
    images, labels, my_imgs = DU.synthetic_batch(B, L, seed=3)
    words = DU.synthetic_random_words(10, 50, seed=3)
    fake = np.array(words[L - 1][:B], np.int32)
    dv = [torch.from_numpy(a).to(dev) for a in (images, labels, my_imgs, fake)]
    try:
        ops.set_conv_dtype(mode)
        NA._model_counter[0] = 0
        NA.configure(device=dev, seed=4)
        G = NA.make_generator(128, (32, 160, 1), (32, 8192), None, "B3", 52, vis_model=False)
        D = NA.make_discriminator((32, 160, 1), None, "B1", vis_model=False)
        R = NA.make_recognizer((32, 160, 1), None, 53, vis_model=False)
        S = NA.make_style_promoter((32, 160, 1), None, "B1", vis_model=False)
        for m in (G, D, S):
            for k in m.store.names:
                if k.endswith(".sigma"):
                    m.store.p[k].fill_(0.25)
        gan = NA.make_gan(G, D, R, S, vis_model=False)
        models = (G, D, R, S)
        opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
        gs = GraphedStep(D, R, S, gan, opts, B, net_loss.hinge, 0, warmup=2).capture(*dv)
        assert all(o.iterations == 2 for o in opts)
        torch.cuda.synchronize()
        snap = [(m.store.flat.clone(), m.store.state.clone(), tuple(t.clone() for t in o._slot(m.store.flat))) for m, o in zip(models, opts)]
        out_g = np.array(tuple(gs.step()), np.float64)
        torch.cuda.synchronize()
        assert all(o.iterations == 3 for o in opts)
        w_graph = [m.store.flat.clone() for m in models]
        for m, o, (flat, state, slots) in zip(models, opts, snap):          # back to the snapshot, then the same step eagerly
            m.store.flat.copy_(flat)
            m.store.state.copy_(state)
            for t, sv in zip(o._slot(m.store.flat), slots):
                t.copy_(sv)
            o.iterations = 2
        ops.weights_changed()
        out_e = np.array(DU.train_step(0, 0, 1, dv[0], dv[1], D, R, S, gan, opts[0], opts[1], opts[2], opts[3], dv[2], B, 128, net_loss.hinge,
                                       1, 0, words, 10, "", fake_labels=dv[3], nl=gs.nl, verbose=False), np.float64)
        w_eager = [m.store.flat.clone() for m in models]
    finally:
        ops.set_conv_dtype("f32")
    assert np.all(np.isfinite(out_g))
    tol = np.full(16, 1e-4 if mode == "f32" else 1e-3)
    tol[[2, 5, 11, 12]] = 2e-2       # std(r_fake), std(g_loss) ~ 0.1 of means of 2 ... 30 and the balancing ratio built on them (untrained nets)
    assert np.all(np.abs(out_g - out_e) <= tol * np.maximum(1.0, np.abs(out_e))), (out_g, out_e)
    for m, wg, we in zip(models, w_graph, w_eager):
        diff = (wg - we).abs()
        # (third Adam step, beta_1 = 0: |update| <= lr sqrt(1 - 0.999^3) / sqrt(0.001) = 1.73 lr for a component whose history is ~0;
        #  stepping the other way in the two runs = 3.5 lr = 6.9e-4, measured exactly that on one component in bf16 mode)
        assert diff.max().item() <= 4 * 2e-4 and diff.mean().item() <= 5e-6, (m.name, diff.max().item(), diff.mean().item())


@pytest.mark.parametrize("conv_dtype", ["f32", "bf16", "fp8"])
def test_network_and_side_streams_do_not_change_the_step(dev, conv_dtype):
    """(bf16 / fp8, round 4: the network stream alone -- the weight-grad side streams are an fp32-mode option -- with the operand
    copies and the per-stream amax pools in play; same criterion.)
    Small per-GPU batches queue S's passes on a second stream beside D's / R's (ops.net_stream) and every weight-grad launch on
    a side stream of its sweep (ops.side_stream).  The same step (the B = 8 fixture, all passes fused, no gradient balancing) with
    both switched off must give the same 16 scalars (1e-5) and the same gradients of all four networks -- a missing event / join
    would show as stale or partial gradients.  Yardstick = the run-to-run noise of the SINGLE-stream step itself (two runs: float
    atomics, amplified where a ReLU decision of the generator sits on the edge): |streams - single| <= max(3e-4 of the network's
    largest gradient, 4 x |single - single'|); per-tensor report in gpurun_out/streams_vs_single.txt.  The convolution kernels run
    with one adder per output address (sg_set_deterministic on the library alone -- the host-side switch would turn the streams
    off): what is left of the run-to-run noise comes from the float-atomic BatchNorm / filter-bank / attention partial sums, so
    the yardstick is not itself at the mercy of a ReLU flip (seen in round 3: 5e-1 vs a yardstick of 5e-2 on G.filter_bank)."""
    from tests import step_fixture as F
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, ops, optimizers
    from scrabble_gan_amd._lib import lib
    pb = F.make_problem(B=8, L_r=2, L_f=2, style_w=32, seed=8, logit_scale=70.0)
    B = pb["B"]
    res = {}
    old = (ops.NET_STREAM, ops.SIDE_WGRAD)
    if conv_dtype == "f32":       # (bf16 / fp8: the 64 -> 64 weight-grad has no one-adder form on the second-generation kernels; the yardstick covers the atomics)
        lib().sg_set_deterministic(1)
    ops.set_conv_dtype(conv_dtype)
    try:
        for mode, on in (("streams", True), ("single", False), ("single2", False)):
            ops.NET_STREAM = on
            ops.SIDE_WGRAD = on and conv_dtype == "f32"
            assert ops.net_stream_enabled(3 * B) == on
            NA._model_counter[0] = 0
            NA.configure(device=dev, seed=9)
            models, gan, nlg = F.load_models(NA, pb, dev)
            G, D, R, S = (models[n] for n in ("G", "D", "R", "S"))
            opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
            for rep in range(2 if on else 1):     # streams: twice -- the second step re-uses memory the first one freed on either stream
                if rep:                   # same weights again (the first step's Adam update is undone)
                    for n, m in models.items():
                        m.store.load({k: v.float() for k, v in pb["P"][n].items()})
                    ops.weights_changed()
                out = DU.train_step(0, 0, 1, pb["images"].float().numpy(), pb["labels"].numpy().astype(np.int32), D, R, S, gan, opts[0],
                                    opts[1], opts[2], opts[3], pb["style"].float().numpy(), B, 128, net_loss.hinge, 1, 0, None, 10, "",
                                    fake_labels=pb["fake"].numpy().astype(np.int32), nl=nlg, verbose=False)
            res[mode] = (np.array(out, np.float64), {n: {k: m.store.g[k].clone() for k in m.store.trainable_names()} for n, m in models.items()})
    finally:
        ops.NET_STREAM, ops.SIDE_WGRAD = old
        lib().sg_set_deterministic(0)
        ops.set_conv_dtype("f32")
    sa, ga = res["streams"]
    sb, gb = res["single"]
    sc, gc = res["single2"]
    # Floors of the criterion |streams - single| <= max(floor, 4 x |single - single'|): fp32 -- the conv kernels run one adder per
    # address here, what is left is float-atomic BatchNorm / filter-bank / attention partial sums: 1e-5 of a scalar, 3e-4 of a
    # network's largest gradient.  bf16 / fp8 -- the reduction-split tiles of the convolutions meet through float atomics in a
    # run-dependent order and every result is re-quantised for the next launch, so two SINGLE-stream runs already differ at the
    # operand precision (measured: scalars 2e-3 / 3e-2 relative, profiles/r04_streams_vs_single_{bf16,fp8}.txt): 1e-2 / 1e-1 of a
    # scalar, 8e-2 / 4e-1 of the largest gradient.  A missing event or join reads operand copies that are not written yet --
    # errors of the order of the values themselves -- and stays far outside either.
    # (bf16 / fp8 floors sit above the LARGEST single-stream run-to-run differences seen -- bf16: scalars 2e-3, G.filter_bank 3.5 % of the
    #  network's largest gradient; fp8: 3e-2, D.B4.conv2.b 19 % -- so that the verdict never hangs on one noisy noise estimate)
    s_floor, g_floor = {"f32": (1e-5, 3e-4), "bf16": (1e-2, 8e-2), "fp8": (1e-1, 4e-1)}[conv_dtype]
    lines, bad = [], []
    for i in range(16):
        lines.append("scalar %2d: streams %.6e  single %.6e  single' %.6e" % (i, sa[i], sb[i], sc[i]))
        if abs(sa[i] - sb[i]) > max(s_floor * max(1.0, abs(sb[i])), 4.0 * abs(sb[i] - sc[i])):
            bad.append(lines[-1])
    glines = []
    for n in ("D", "R", "S", "G"):
        scale = max(t.abs().max().item() for t in gb[n].values())
        for k in gb[n]:
            err = (ga[n][k] - gb[n][k]).abs().max().item()
            noise = (gb[n][k] - gc[n][k]).abs().max().item()
            glines.append("%s.%s: |streams - single| %.3e  |single - single'| %.3e  (largest gradient of the network %.3e)" % (n, k, err, noise, scale))
            if err > max(g_floor * scale, 4.0 * noise):
                bad.append(glines[-1])
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        open("gpurun_out/streams_vs_single%s.txt" % ("" if conv_dtype == "f32" else "_" + conv_dtype), "w").write(
            "\n".join(lines + sorted(glines, key=lambda l: -float(l.split("|streams - single| ")[1].split()[0]))[:40]) + "\n")
    except OSError:
        pass
    assert not bad, "\n".join(bad[:10])


def test_deterministic_mode_makes_the_whole_step_bitwise_reproducible(dev):
    """`configure(deterministic=True)` (-> sg_set_deterministic): every float-atomic accumulation of the step has ONE adder per address --
    no reduction splits and one pixel chunk in the convolutions (rounds 2-3), and since round 4 also BatchNorm's per-sample dgamma /
    dbeta partials (one workgroup per sample), the filter bank's dz (one workgroup per sample, fixed loop order), the attention key
    sweep (no query split) and the sigma dot product (one workgroup); the Winograd weight-grad gives way to the direct kernel's
    single-chunk form and the second stream is off.  Two runs of the same step from the same state must then agree BITWISE in all 16
    scalars and in every gradient tensor of all four networks (the mode used to cover the convolutions only).
    (Spectral norm's power iteration -- kernel_reg = 'applied' only -- still adds through float atomics and is not part of this step.)"""
    from tests import step_fixture as F
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, ops, optimizers
    pb = F.make_problem(B=8, L_r=2, L_f=2, style_w=32, seed=8, logit_scale=70.0)
    B = pb["B"]
    runs = []
    NA.configure(device=dev, seed=9, deterministic=True)
    try:
        assert ops.DETERMINISTIC and not ops.net_stream_enabled(3 * B)
        for _ in range(2):
            NA._model_counter[0] = 0
            models, gan, nlg = F.load_models(NA, pb, dev)
            G, D, R, S = (models[n] for n in ("G", "D", "R", "S"))
            opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
            out = DU.train_step(0, 0, 1, pb["images"].float().numpy(), pb["labels"].numpy().astype(np.int32), D, R, S, gan, opts[0],
                                opts[1], opts[2], opts[3], pb["style"].float().numpy(), B, 128, net_loss.hinge, 1, 1, None, 10, "",
                                fake_labels=pb["fake"].numpy().astype(np.int32), nl=nlg, verbose=False)
            runs.append((np.array(out, np.float64), {n: m.store.grad.clone() for n, m in models.items()},
                         {n: m.store.flat.clone() if hasattr(m.store, "flat") else None for n, m in models.items()}))
    finally:
        NA.configure(deterministic=False)
    (sa, ga, wa), (sb, gb, wb) = runs
    assert np.array_equal(sa, sb), (sa, sb)
    for n in ("D", "R", "S", "G"):
        diff = (ga[n] - gb[n]).abs().max().item()
        assert torch.equal(ga[n], gb[n]), "%s: gradients of two deterministic runs differ by up to %.3e" % (n, diff)
        if wa[n] is not None:
            assert torch.equal(wa[n], wb[n]), "%s: post-Adam weights differ" % n
