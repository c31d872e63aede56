"""CPU-only checks of the host logic: the C-ABI library loads and exports every symbol the header
declares (no compute calls without a GPU), the gin reader, the label/index bookkeeping (bit-exact),
and the API surface of the reference (names, arity)."""
import inspect
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from scrabble_gan_amd import _lib
    decls = _lib.parse_header()
    assert len(decls) >= 40
    lib = _lib.lib()
    for name in decls:
        assert hasattr(lib, name), name
    for must in ("sg_conv2d_fwd", "sg_conv2d_bwd_data", "sg_conv2d_bwd_weight", "sg_conv2d_transpose_fwd", "sg_bn_apply",
                 "sg_attention_fwd", "sg_filterbank_fwd", "sg_softmax_ctc", "sg_adam_update", "sg_spectral_norm", "sg_loss_grads"):
        assert must in decls
    # pure host helpers may be called without a GPU
    assert lib.sg_spectral_norm_workspace_floats(9216, 1024) == 9216 + 1024 + 4
    assert lib.sg_bn_stats_workspace_floats(4096, 64) == 128 * 2 * 64          # 32 rows per workgroup: 128 partial blocks
    assert lib.sg_bn_stats_workspace_floats(1 << 21, 64) == 1024 * 2 * 64      # capped at 2048 rows per workgroup


def test_ops_fail_loudly_without_library(monkeypatch, tmp_path):
    from scrabble_gan_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(_lib.ScrabbleHipError):
        _lib.lib()


def test_gin_reader_parses_the_config_surface():
    from scrabble_gan_amd import gin_config as gin, main as M
    gin.clear_config()
    gin.parse_config_file(os.path.join(ROOT, "configs", "scrabble_gan_mi355x.gin"))
    epochs, batch_size, latent_dim, embed_y, num_gen, kernel_reg, g_att, d_att, my_rec, my_disc = M.get_shared_specs()
    assert (epochs, batch_size, latent_dim, embed_y, num_gen, g_att, d_att, my_rec, my_disc) == (10, 16, 128, (32, 8192), 16, 'B3', 'B1', 0, 0)
    assert kernel_reg.__name__ == "spectral_norm"
    in_dim, buf_size, n_classes, seq_len, bucket_size, ckpt, gen_path, m_path, raw_dir, read_dir, char_vec = M.setup_io()
    assert in_dim == (32, 160, 1) and buf_size == 80377 and n_classes == 52 and seq_len is None and bucket_size == 10
    assert char_vec == 'abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ' and ckpt.endswith('checkpoints/')
    g_opt, d_opt, r_opt, w_opt, loss_fn, disc_iters, bal = M.setup_optimizer()
    assert g_opt.learning_rate == 2e-4 and g_opt.beta_1 == 0.0 and g_opt.beta_2 == 0.999 and g_opt.epsilon == 1e-7
    assert loss_fn.__name__ == "hinge" and loss_fn.mode == 0 and disc_iters == 1 and bal == 0
    gin.parse_config("setup_optimizer.rmsprop = 1   # trailing comment\nsetup_optimizer.loss_fn = @not_saturating\nio.base_path = 'a#b/'")
    out = M.setup_optimizer()
    assert type(out[2]).__name__ == "RMSprop" and out[4].mode == 1
    assert gin.query_parameter("io.base_path") == "a#b/"
    with pytest.raises(gin.GinError):
        gin.parse_config("setup_optimizer.loss_fn = @nope")
    with pytest.raises(gin.GinError):
        gin.parse_config("this is not gin")
    gin.clear_config()
    with pytest.raises(gin.GinError):
        M.get_shared_specs()


def test_bookkeeping_is_bit_exact():
    from scrabble_gan_amd import data_utils as DU
    cv = 'abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'
    assert DU.encode_word("auto", cv) == [0, 20, 19, 14]                     # the reference's own example (data_utils.py:48)
    assert DU.encode_word("Zz", cv) == [51, 25]
    with pytest.raises(ValueError):
        DU.encode_word("a-b", cv)
    assert [DU.bucket_width((32, 160, 1), b) for b in (1, 4, 10, 23)] == [16, 64, 160, 368]
    assert [DU.ctc_input_length(L) for L in (1, 10, 23)] == [3, 39, 91]
    words = DU.bucket_words(["a\n", "to", "  cat ", "elephantine", "Dog"], 10, cv)
    assert len(words) == 10 and words[0] == [[0]] and words[1] == [[19, 14]] and words[2] == [[2, 0, 19], [29, 14, 6]]
    assert all(len(w) == k + 1 for k, b in enumerate(words) for w in b) and words[9] == []
    img = np.array([[[0, 255], [127, 128]]], dtype=np.uint8).reshape(1, 2, 2)
    out = DU.normalize_images(np.zeros((3, 32, 48), np.uint8), (32, 160, 1), 3)
    assert out.shape == (3, 32, 48, 1) and out.dtype == np.float32 and (out == -1.0).all()
    assert np.allclose((img.astype('float32') - 127.5) / 127.5, [[[-1, 1], [-0.5 / 127.5, 0.5 / 127.5]]])
    # same `random` call sequence as data_utils.py:386-387
    rw = DU.synthetic_random_words(10, 20)
    random.seed(3)
    idx, fl = DU.draw_fake_labels(rw, 10, 5)
    random.seed(3)
    idx2 = random.randint(0, 9)
    fl2 = np.array([random.choice(rw[idx2]) for _ in range(5)], np.int32)
    assert idx == idx2 and fl.dtype == np.int32 and (fl == fl2).all() and fl.shape == (5, idx + 1)


def test_bucket_folder_reader(tmp_path):
    """load_prepare_data on the reference's on-disk layout <read_dir>/<L>/<name>.png + .txt (pixels and labels bit-exact)."""
    from PIL import Image
    from scrabble_gan_amd import data_io
    cv = 'abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'
    rng = np.random.default_rng(0)
    truth = {}
    for L, words in ((1, ["a", "Z"]), (2, ["to", "Hi", "ab"])):
        d = tmp_path / str(L)
        d.mkdir()
        for i, wd in enumerate(words):
            px = rng.integers(0, 256, (32, 16 * L), dtype=np.uint8)
            Image.fromarray(px, mode="L").save(d / ("w%d.png" % i))
            (d / ("w%d.txt" % i)).write_text(wd)
            truth[tuple(data_io.encode_word(wd, cv))] = px
    gen = data_io.load_prepare_data((32, 160, 1), 4, str(tmp_path) + "/", cv, 2)
    random.seed(1)
    np.random.seed(1)
    for _ in range(5):
        imgs, labels = next(gen)
        L = labels.shape[1]
        assert imgs.shape == (4, 32, 16 * L, 1) and imgs.dtype == np.float32 and labels.dtype == np.int32
        for im, lab in zip(imgs, labels):
            assert np.array_equal(im[:, :, 0], (truth[tuple(lab)].astype('float32') - 127.5) / 127.5)
    # style images: height-normalised to 32, cropped / white-padded to 160, in [-1,1]
    sdir = tmp_path / "style"
    sdir.mkdir()
    for i, (hh, ww) in enumerate([(64, 200), (32, 400), (40, 90)] * 7):
        Image.fromarray(rng.integers(0, 256, (hh, ww), dtype=np.uint8), mode="L").save(sdir / ("s%d.png" % i))
    tr, va = data_io.load_style_input((32, 160, 1), 4, 10, str(sdir))
    assert len(tr) == 19 and len(va) == 2 and all(t.shape == (32, 160) for t in tr)
    assert all(-1.0 - 1e-6 <= t.min() and t.max() <= 1.0 + 1e-6 for t in tr)


def test_reference_api_surface():
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss
    ref_params = ["epoch_idx", "batch_idx", "batch_per_epoch", "images", "labels", "discriminator", "recognizer", "style_promoter",
                  "composite_gan", "generator_optimizer", "discriminator_optimizer", "recognizer_optimizer",
                  "stylepromoter_optimizer", "my_imgs", "batch_size", "latent_dim", "loss_fn", "disc_iters",
                  "apply_gradient_balance", "random_words", "bucket_size", "gen_path"]
    assert list(inspect.signature(DU.train_step).parameters)[:22] == ref_params
    assert list(inspect.signature(NA.make_generator).parameters) == ["latent_dim", "input_dim", "embed_y", "kernel_reg",
                                                                    "blocks_with_attention", "vocab_size", "vis_model"]
    assert list(inspect.signature(NA.make_discriminator).parameters) == ["input_dim", "kernel_reg", "blocks_with_attention", "vis_model"]
    assert list(inspect.signature(NA.make_style_promoter).parameters) == ["input_dim", "kernel_reg", "blocks_with_attention", "vis_model"]
    assert list(inspect.signature(NA.make_style_extractor).parameters) == ["input_dim", "kernel_reg", "blocks_with_attention", "vis_model"]
    assert list(inspect.signature(NA.make_recognizer).parameters) == ["input_dim", "sequence_length", "output_classes", "vis_model"]
    assert list(inspect.signature(NA.make_my_recognizer).parameters) == ["input_dim", "sequence_length", "output_classes", "vis_model"]
    assert list(inspect.signature(NA.make_gan).parameters) == ["g_model", "d_model", "r_model", "w_model", "vis_model"]
    assert NA.get_in_out_channels_gen(32) == ([512, 256, 128], [256, 128, 64])
    assert NA.get_in_out_channels_disc(1, 32) == ([1, 64, 512, 1024], [64, 512, 1024, 1024])
    with pytest.raises(ValueError, match="Unsupported resolution"):
        NA.get_in_out_channels_gen(64)
    with pytest.raises(ValueError, match="Unsupported color channels"):
        NA.get_in_out_channels_disc(2, 32)
    assert len(inspect.signature(net_loss.not_saturating).parameters) == 5
    assert len(inspect.signature(net_loss.hinge).parameters) == 5        # 4 + the ignored 5th (Appendix C-1)


def test_param_store_layout_on_cpu():
    import torch
    from scrabble_gan_amd import nn
    specs = nn.block_down_specs("B1", 1, 64) + [("NL_B1.sigma", (), nn.zeros, True), ("bn.mm", (6,), nn.zeros, False)]
    S = nn.ParamStore(specs, torch.device("cpu"), torch.Generator().manual_seed(0))
    assert S.num_params() == 9 * 64 + 64 + 9 * 64 * 64 + 64 + 64 + 64 + 1
    for k in S.trainable_names():
        assert S.p[k].data_ptr() % 16 == 0 and S.g[k].shape == S.p[k].shape      # float4-aligned slices
    w = S.p["B1.conv2.w"].reshape(-1, 64)
    assert torch.allclose(w.t() @ w, torch.eye(64), atol=1e-5)                    # orthogonal init
    S.g["B1.conv1.b"].fill_(2.0)
    assert S.grad.sum().item() == 128.0
    S.zero_grad()
    assert S.grad.abs().sum().item() == 0.0


def test_step_scalars_sequence_protocol():
    """train_step(..., sync='lazy') returns a StepScalars: the 16 values of data_utils.py:470-473 behind one deferred
    host copy; it must behave as the 16-tuple the reference returns (len, index, iteration, str) and report alpha = 1."""
    import torch
    from scrabble_gan_amd.data_utils import StepScalars
    vals = torch.arange(16, dtype=torch.float32) * 0.5
    s = StepScalars(vals)
    assert len(s) == 16
    got = tuple(s)
    assert got[10] == 1                                   # the reference returns the constant alpha, not a tensor value
    assert [got[i] for i in range(16) if i != 10] == [0.5 * i for i in range(16) if i != 10]
    assert s[3] == 1.5 and s[-1] == 7.5
    assert repr(s) == repr(got)
    assert ";".join(str(s[i]) for i in (6, 7, 8)) == "3.0;3.5;4.0"   # the summary writer's access pattern


def test_cli_entry_points_parse_on_cpu():
    """bench.py, the trainer and the inference script must at least import and parse their arguments without a GPU
    (the driver launches them by path / module name)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cmd in ([sys.executable, os.path.join(root, "bench.py"), "--help"],
                [sys.executable, "-m", "scrabble_gan_amd.main", "--help"],
                [sys.executable, "-m", "scrabble_gan_amd.run_inference", "--help"]):
        r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (cmd, r.stderr[-500:])
        assert "usage" in r.stdout.lower()
    for flag in ("--gpus", "--steps", "--warmup", "--conv-dtype", "--bucketed"):
        assert flag in subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], cwd=root, capture_output=True,
                                      text=True, timeout=300).stdout


def test_pmc_traffic_tool_units_and_gfx950_correction(tmp_path):
    """tools/pmc_traffic.py: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled on gfx950 (128-byte requests tallied at
    64 bytes: MI355X_MICROARCH.md, HBM section); the result is bytes per launch of the named kernel family."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = '"Correlation_Id","Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n'
    (tmp_path / "f.csv").write_text(hdr + '1,1,"void sg_igemm_kernel<128, 128>(SgIgemmArgs)","FETCH_SIZE",1000.0\n'
                                          '2,2,"void sg_igemm_kernel<128, 64>(SgIgemmArgs)","FETCH_SIZE",3000.0\n'
                                          '3,3,"k_adam(float*)","FETCH_SIZE",99999.0\n')
    (tmp_path / "w.csv").write_text(hdr + '1,1,"void sg_igemm_kernel<128, 128>(SgIgemmArgs)","WRITE_SIZE",500.0\n'
                                          '2,2,"void sg_igemm_kernel<128, 64>(SgIgemmArgs)","WRITE_SIZE",1500.0\n')
    out = tmp_path / "t.json"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), str(tmp_path / "f.csv"), str(tmp_path / "w.csv"),
                        "--kernel", "sg_igemm_kernel", "--out", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    d = json.loads(out.read_text())
    assert d["launches_fetch_pass"] == 2 and d["launches_write_pass"] == 2
    assert d["fetch_bytes_per_launch"] == 2.0 * 2000.0 * 1024.0          # mean 2000 KiB, doubled
    assert d["write_bytes_per_launch"] == 1000.0 * 1024.0
    assert d["traffic_bytes_per_launch"] == d["fetch_bytes_per_launch"] + d["write_bytes_per_launch"]


def test_loss_plots_from_the_summaries(tmp_path):
    """utilities.main (/root/reference/src/utilities.py:8-49): per-epoch means of the 16-column batch summary
    that data_utils.train writes, the reference's four figures on disk, with and without the gradient-balancing curves."""
    import numpy as np
    from scrabble_gan_amd import utilities
    from scrabble_gan_amd.data_utils import SUMMARY_HEADER
    rng = np.random.default_rng(0)
    rows = rng.random((7, 16))                      # 7 batches, 3 per epoch: two full epochs and a partial one
    with open(tmp_path / "batch_summary.txt", "w") as f:
        f.write(SUMMARY_HEADER)
        for r in rows:
            f.write(";".join(repr(float(v)) for v in r) + "\n")
    names, data = utilities.read_summary(str(tmp_path / "batch_summary.txt"))
    assert names[:3] == ["d_loss", "d_loss_real", "d_loss_fake"] and names[9] == "g_final_loss" and data.shape == (7, 16)
    assert np.array_equal(data, rows)
    means = utilities.per_epoch_means(data, 3)
    assert means.shape == (3, 16)
    assert np.allclose(means[0], rows[:3].mean(0)) and np.allclose(means[2], rows[6])
    for balance, n in ((False, 4), (True, 4)):
        files = utilities.main(str(tmp_path), 3, info_per_batch=True, gradient_balance=balance)
        assert len(files) == n
        for p in files:
            assert os.path.getsize(p) > 1000 and open(p, "rb").read(4) == b"\x89PNG"
    assert len(utilities.main(str(tmp_path), 3, info_per_batch=False)) == 3
