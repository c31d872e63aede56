"""Host-side pieces of the Winograd-domain path (no GPU): the generated F(4, 3) transforms, the workspace arithmetic of the C-ABI
and the routing rules of scrabble_gan_amd.ops.  (The kernels themselves are checked against the oracle in tests/test_winograd_gpu.py.)"""
import importlib.util
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_winograd_f43", os.path.join(ROOT, "tools", "gen_winograd_f43.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_f43_matrices_satisfy_the_convolution_identities_in_fp64():
    """Y = A^T [(G g G^T) . (B^T d B)] A is the 3x3 correlation on a 4x4 output tile, and dg = G^T [(B^T d B) . (A dY A^T)] G its
    filter gradient -- the two identities the kernels rely on, for the point set the generator uses."""
    g = _gen()
    AT, G, BT = g.cook_toom(g.PTS)
    rng = np.random.default_rng(1)
    for _ in range(5):
        d, f, dy = rng.standard_normal((6, 6)), rng.standard_normal((3, 3)), rng.standard_normal((4, 4))
        y = AT @ ((G @ f @ G.T) * (BT @ d @ BT.T)) @ AT.T
        ref = np.array([[np.sum(d[i:i + 3, j:j + 3] * f) for j in range(4)] for i in range(4)])
        assert np.abs(y - ref).max() < 1e-12
        df = G.T @ ((BT @ d @ BT.T) * (AT.T @ dy @ AT)) @ G
        refg = np.array([[np.sum(d[a:a + 4, b:b + 4] * dy) for b in range(3)] for a in range(3)])
        assert np.abs(df - refg).max() < 1e-12


def test_generated_header_is_what_the_generator_writes():
    """csrc/wino_f43.h is generated code: every coefficient in it is the fp32 rounding of the generator's matrix entry."""
    g = _gen()
    AT, G, BT = g.cook_toom(g.PTS)
    text = open(os.path.join(ROOT, "scrabble_gan_amd", "csrc", "wino_f43.h")).read()
    for name, M in (("w43_bt", BT), ("w43_g", G), ("w43_at", AT), ("w43_a", AT.T), ("w43_gt", G.T)):
        assert g.emit(name, M, "x").split("\n", 1)[1] in text, name


def test_fp32_emulation_of_the_f43_pipeline_stays_inside_the_stated_tolerance():
    """Emulated fp32 pipeline (fp32 transforms, sequential fp32 accumulation over 512 channels -- harsher than the MFMA's blocked
    sums) against fp64: the error stays below the 2e-5 of max |ref| the GPU tests hold the kernels to."""
    g = _gen()
    AT, G, BT = (m.astype(np.float32) for m in g.cook_toom(g.PTS))
    rng = np.random.default_rng(0)
    f32, C, worst = np.float32, 512, 0.0
    for _ in range(3):
        d = np.maximum(rng.standard_normal((C, 6, 6)), 0).astype(f32)
        w = (rng.standard_normal((C, 3, 3)) / np.sqrt(9 * C)).astype(f32)
        ref = np.array([[np.sum(d[:, i:i + 3, j:j + 3].astype(np.float64) * w.astype(np.float64)) for j in range(4)] for i in range(4)])
        U = np.einsum("ia,cab,jb->cij", G, w, G).astype(f32)
        V = np.einsum("ia,cab,jb->cij", BT, d, BT).astype(f32)
        M = np.zeros((6, 6), f32)
        for c in range(C):
            M = (M + U[c] * V[c]).astype(f32)
        Y = (AT @ M @ AT.T).astype(f32)
        worst = max(worst, np.abs(Y - ref).max() / np.abs(ref).max())
    assert worst < 2e-5, worst


def test_workspace_arithmetic_of_the_c_abi():
    """sg_wino_plane_rows / sg_wino_workspace_bytes / sg_wino_wgrad_workspace_bytes are pure host functions: planes of
    ceil(T / 128) * 128 rows, V | Mt (or V | Qt | dU | 64 bias-gradient rows) back to back; shapes the tile size does not divide are refused with 0."""
    from scrabble_gan_amd._lib import lib
    L = lib()
    for tile in (2, 4):
        P = (tile + 2) ** 2
        for B, H, W, Ci, Co in ((1, 4, 4, 32, 128), (16, 4, 20, 1024, 1024), (384, 16, 80, 512, 512), (3, 8, 12, 64, 256)):
            T = B * (H // tile) * (W // tile)
            Tp = -(-T // 128) * 128
            assert L.sg_wino_plane_rows(B, H, W, tile) == Tp
            assert L.sg_wino_workspace_bytes(B, H, W, Ci, Co, tile) == 4 * P * Tp * (Ci + Co)
            assert L.sg_wino_wgrad_workspace_bytes(B, H, W, Ci, Co, tile) == 4 * (P * (Tp * (Ci + Co) + Ci * Co) + 64 * Co)
    assert L.sg_wino_plane_rows(2, 6, 8, 4) == 0 and L.sg_wino_plane_rows(2, 6, 8, 2) == 128
    assert L.sg_wino_workspace_bytes(2, 3, 8, 64, 128, 2) == 0 and L.sg_wino_workspace_bytes(2, 4, 8, 64, 128, 3) == 0


def test_routing_of_the_step_s_convolutions():
    """Which launches of the fp32 step leave the direct kernels (ops._wino_ok / _wino_wgrad_ok / _wino_tile): the D-shaped trunks'
    3x3 convolutions from 64 -> 512 up (and back: 512 -> 64) and the recognizer's from 128 channels; never 1x1, VALID, odd shapes, 64 -> 64
    (forward / data-grad) or another operand type; F(2x2) where W is not a multiple of 4 (odd word lengths of the bucketed widths)."""
    from scrabble_gan_amd import ops
    assert ops.CONV_DTYPE == "f32" and ops.USE_WINOGRAD and ops.WINO_TILE == 4
    yes = [(512, 512, 16, 80), (512, 1024, 8, 40), (1024, 1024, 4, 20), (64, 512, 16, 80), (256, 256, 8, 80), (128, 128, 16, 160), (128, 256, 8, 40),
           (512, 64, 16, 80)]       # (round 4: 64 output channels -- the data-grad of the 64 -> 512 convolution -- on 128 x 64 product tiles)
    for K, N, H, W in yes:
        assert ops._wino_ok(K, N, 3, 3, True, H, W) and ops._wino_tile(H, W) == 4, (K, N, H, W)
    assert ops._wino_tile(4, 10) == 2 and ops._wino_ok(1024, 1024, 3, 3, True, 4, 10)          # L = 5: W = 2 L on the 4-row layers
    assert not ops._wino_ok(128, 128, 3, 3, True, 4, 10)                                        # (below F(2x2)'s 32 768 channel-pair floor)
    for K, N, kh, same, H, W in [(64, 64, 3, True, 32, 160), (512, 96, 3, True, 16, 80), (512, 1024, 1, True, 8, 40), (512, 512, 3, False, 16, 80),
                                 (512, 512, 3, True, 5, 80), (1, 64, 3, True, 32, 160), (512, 512, 2, True, 2, 40)]:
        assert not ops._wino_ok(K, N, kh, kh, same, H, W), (K, N, kh, same, H, W)
    # few-tile launches (every frequency plane is padded to 128 rows): the recognizer's 8 x 12 map at B = 3 has 18 tiles -- 36 x 128 rows
    # against the direct form's 9 x 384 -- and stays direct; every layer of the headline batch and of its 8-way shard goes through
    assert not ops._wino_ok(128, 256, 3, 3, True, 8, 12, 3) and not ops._wino_ok(512, 512, 3, 3, True, 4, 12, 3)
    for B in (16, 128, 384):
        for K, N, H, W in yes:
            assert ops._wino_ok(K, N, 3, 3, True, H, W, B), (B, K, N, H, W)
    assert ops._wino_ok(1024, 1024, 3, 3, True, 4, 20, 16) and not ops._wino_ok(1024, 1024, 3, 3, True, 4, 16, 4)    # T = 80 / 16 tiles
    assert ops._wino_wgrad_ok(64, 64, 3, 3, True, 32, 160) and ops._wino_wgrad_ok(64, 512, 3, 3, True, 16, 80)
    assert not ops._wino_wgrad_ok(1, 64, 3, 3, True, 32, 160) and not ops._wino_wgrad_ok(512, 512, 1, 1, True, 8, 40)
    old = ops.CONV_DTYPE
    try:
        ops.CONV_DTYPE = "bf16"
        assert not ops._wino_ok(512, 512, 3, 3, True, 16, 80) and not ops._wino_wgrad_ok(512, 512, 3, 3, True, 16, 80)
    finally:
        ops.CONV_DTYPE = old
    old = ops.DETERMINISTIC
    try:
        ops.DETERMINISTIC = True            # one adder per dW address: the direct kernel's single-chunk weight-grad
        assert ops._wino_ok(512, 512, 3, 3, True, 16, 80) and not ops._wino_wgrad_ok(512, 512, 3, 3, True, 16, 80)
    finally:
        ops.DETERMINISTIC = old
