"""Generate word images from a trained generator: the counterpart of /root/reference/src/run_inference.py:12-48
(this fork's generator is conditioned on STYLE IMAGES instead of a noise vector: data_utils.py:493-519 calls
`generator([style_imgs, labels], training=False)`).

    python -m scrabble_gan_amd.run_inference --weights run/checkpoints/generator/15/cktp-15 --words machinelearning \
        [--style-dir DIR | --synthetic-style] [--count 10] --out words.png

The checkpoint is the `<prefix>.safetensors` file `train()` writes per epoch (`_Model.save_weights`).  BatchNorm runs in
inference mode (moving statistics).  The output PNG stacks the `count` renderings of each word vertically (the
reference shows them as a 10 x 1 matplotlib grid) after the reference's (x + 1) / 2 mapping to [0, 1]."""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

CHAR_VEC = 'abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'


def encode(word: str, char_vec: str = CHAR_VEC):
    """char -> index by `char_vec.index`, as run_inference.py:33 / data_utils.py:49."""
    return [char_vec.index(c) for c in word]


def generate(generator, words, style_imgs, char_vec: str = CHAR_VEC) -> np.ndarray:
    """-> float array [len(words) * n_style, 32, 16 * L, 1] in [0, 1]; all words must have one length L."""
    L = len(words[0])
    if any(len(w) != L for w in words):
        raise ValueError("one word length per call (the generator's output width is 16 * L)")
    n = len(style_imgs)
    labels = np.array([encode(w, char_vec) for w in words for _ in range(n)], np.int32)
    style = np.concatenate([np.asarray(style_imgs, np.float32)] * len(words), axis=0)
    img = generator([style, labels], training=False)
    return ((img.detach().float().cpu().numpy() + 1.0) / 2.0).clip(0.0, 1.0)


def save_grid(imgs: np.ndarray, path: str) -> None:
    from PIL import Image
    rows = [np.round(i[:, :, 0] * 255.0).astype(np.uint8) for i in imgs]
    sep = np.full((2, rows[0].shape[1]), 255, np.uint8)
    stack = []
    for r in rows:
        stack += [r, sep]
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    Image.fromarray(np.concatenate(stack[:-1], axis=0), mode="L").save(path)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--weights", required=True, help="checkpoint prefix (without .safetensors)")
    ap.add_argument("--words", nargs="+", default=["machinelearning"])
    ap.add_argument("--style-dir", default=None, help="folder of style PNGs (read like data_utils.load_style_input)")
    ap.add_argument("--synthetic-style", action="store_true", help="uniform-noise style images (no dataset at hand)")
    ap.add_argument("--count", type=int, default=10, help="renderings per word (= style images used)")
    ap.add_argument("--out", default="generated.png")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)

    from . import net_architecture as NA
    NA.configure(device=torch.device(args.device))
    in_dim = (32, 160, 1)
    G = NA.make_generator(128, in_dim, (32, 8192), None, "B3", len(CHAR_VEC), vis_model=False)
    G.load_weights(args.weights)
    if args.style_dir and not args.synthetic_style:
        from .data_io import load_style_input
        train_imgs, val_imgs = load_style_input(in_dim, args.count, 10, style_dir=args.style_dir)
        style = np.asarray((train_imgs + val_imgs)[:args.count], np.float32).reshape(-1, 32, 160, 1)
    else:
        style = np.random.default_rng(0).uniform(-1, 1, (args.count, 32, 160, 1)).astype(np.float32)
    by_len = {}
    for w in args.words:
        by_len.setdefault(len(w), []).append(w)
    widest = 16 * max(by_len)
    rows = []
    for L, ws in sorted(by_len.items()):
        imgs = generate(G, ws, style)
        rows += [np.pad(i, ((0, 0), (0, widest - i.shape[1]), (0, 0)), constant_values=1.0) for i in imgs]
    save_grid(np.stack(rows), args.out)
    print("wrote %s: %d rows (%d words x %d styles)" % (args.out, len(rows), len(args.words), len(style)))


if __name__ == "__main__":
    main()
