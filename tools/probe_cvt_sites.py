"""Which call sites make bf16 twins by a separate conversion pass (ops.cvt_bf16 / cvt_bf16_bias) in one bf16-mode train_step, by
bytes: the producers worth teaching to write the twin themselves.   python tools/probe_cvt_sites.py [--batch 256]"""
import argparse
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    from scrabble_gan_amd import ops
    sites = collections.Counter()
    counts = collections.Counter()

    def wrap(fn, name):
        def inner(t, *args, **kw):
            st = traceback.extract_stack(limit=8)[:-1]
            key = name + " <- " + " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st) if "scrabble_gan_amd" in f.filename)[:200]
            sites[key] += t.numel() * 6
            counts[key] += 1
            return fn(t, *args, **kw)
        return inner
    ops.cvt_bf16 = wrap(ops.cvt_bf16, "cvt")
    ops.cvt_bf16_bias = wrap(ops.cvt_bf16_bias, "cvt+bias")
    import random
    import numpy as np
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, optimizers
    from scrabble_gan_amd.main import build_models
    dev = torch.device("cuda:0")
    NA.configure(device=dev, seed=0)
    ops.set_conv_dtype("bf16")
    in_dim = (32, 160, 1)
    G, D, R, S, gan = build_models(in_dim, 128, (32, 8192), None, "B3", "B1", 52, None)
    opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
    B, L = a.batch, 10
    images, labels, my_imgs = DU.synthetic_batch(B, L, in_dim, 52, seed=0)
    words = DU.synthetic_random_words(10, 1000, 52, seed=0)
    random.seed(0)
    fake = torch.from_numpy(np.array([random.choice(words[L - 1]) for _ in range(B)], np.int32)).to(dev)
    images_d, my_d, labels_d = torch.from_numpy(images).to(dev), torch.from_numpy(my_imgs).to(dev), torch.from_numpy(labels).to(dev)

    def step():
        DU.train_step(0, 0, 2, images_d, labels_d, D, R, S, gan, opts[0], opts[1], opts[2], opts[3], my_d, B, 128, net_loss.hinge, 1, 0,
                      words, 10, "", fake_labels=fake, verbose=False, sync=True)
    step()
    torch.cuda.synchronize()
    sites.clear(); counts.clear()
    step()
    torch.cuda.synchronize()
    tot = sum(sites.values())
    print("total %.1f MB moved by conversion passes in one step" % (tot / 1e6))
    for k, v in sites.most_common(40):
        print("%8.1f MB  n=%3d  %s" % (v / 1e6, counts[k], k))


if __name__ == "__main__":
    main()
