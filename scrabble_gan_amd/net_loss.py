"""Loss configurables with the reference's names (/root/reference/src/bigacgan/net_loss.py):
not_saturating :4-35 (5 args) and hinge :38-54 (4 args + an ignored 5th, SURVEY Appendix C-1: the
reference's train_step passes 5 positional arguments, data_utils.py:418).

Called on device tensors they return the same 7 per-sample tensors [B,1] as the reference.  The
train step itself uses the fused loss head (sg_loss_sums / sg_loss_grads) and picks the formula by
the callable's `.mode` attribute."""
from __future__ import annotations

from . import ops


def _terms(mode, d_real, d_fake, a, b, c):
    B = d_real.numel()
    out = ops.loss_terms(*[t.reshape(-1).contiguous() for t in (d_real, d_fake, a, b, c)], mode)
    return tuple(out[i].view(B, 1) for i in range(7))


def hinge(d_real_logits, d_fake_logits, s_real_logits, s_fake_logits, s_real_imgs_logits=None):
    """d = relu(1-d_real)+relu(1+d_fake); s likewise; g = -(d_fake+s_fake).  Returns
    (d_loss, d_loss_real, d_loss_fake, g_loss, s_loss, s_loss_real, s_loss_fake)."""
    c = s_real_imgs_logits if s_real_imgs_logits is not None else s_real_logits
    return _terms(0, d_real_logits, d_fake_logits, s_real_logits, s_fake_logits, c)


def not_saturating(d_real_logits, d_fake_logits, s_styleimgs_logits, s_trainingimgs_logits, s_fake_logits):
    """Sigmoid cross-entropy variants, argument meaning exactly as declared in net_loss.py:4 (the call
    site passes S(G(z)) as s_trainingimgs and S(real) as s_fake: Appendix C-2, reproduced literally)."""
    return _terms(1, d_real_logits, d_fake_logits, s_styleimgs_logits, s_trainingimgs_logits, s_fake_logits)


hinge.mode = 0
not_saturating.mode = 1
