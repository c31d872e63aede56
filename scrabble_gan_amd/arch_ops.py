"""Ops with the reference's names (/root/reference/src/bigacgan/arch_ops.py): spectral_norm :98-126,
NonLocalBlock :5-72 (as a pure function of explicit kernels), SpatialEmbedding :77-95 (fused with the
z0 contraction and the seed layout of net_architecture.py:259-271)."""
from __future__ import annotations

import torch

from . import nn, ops


def spectral_norm(w, power_iteration=1, u=None, generator=None):
    """One-step power iteration; `u` is the fresh N(0,1) draw of arch_ops.py:110 (drawn here when not
    given).  gin configurable `@spectral_norm`; as in the reference it is registered as a
    kernel_regularizer and therefore never applied inside the forward pass."""
    if u is None:
        u = torch.randn(w.shape[-1], generator=generator).to(w.device)
    return ops.spectral_norm(w.contiguous(), u.reshape(-1).float().contiguous().to(w.device), power_iteration)


def non_local_block(x, w_theta, w_phi, w_g, w_o, sigma):
    """sigma * Conv1x1(softmax(theta phi^T) g) + x ; sigma == 0 is the exact identity."""
    out, _ = nn.nonlocal_fwd(x, {"theta": w_theta, "phi": w_phi, "g": w_g, "o": w_o}, sigma)
    return out


def spatial_embedding_seed(z, y, filter_bank):
    """seed[b, r, 4l+pw, q] = (z[b,:32] . E[y[b,l]])[pw*2048 + q*4 + r]  -> [B,4,4L,512]."""
    return ops.filterbank_fwd(z, y, filter_bank)
