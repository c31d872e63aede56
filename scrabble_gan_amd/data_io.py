"""On-disk dataset readers of the reference's host pipeline ("next" tier):
load_prepare_data (/root/reference/src/bigacgan/data_utils.py:14-84) and load_style_input (:87-195).

Bucket folders `<read_dir>/<L>/<name>.png` + `<name>.txt` (one transcription line) as written by the reference's
one-time converter; the style folder holds free-size handwriting crops that are height-normalised to 32, then
cropped / white-padded to 160.  PNGs are read with PIL (OpenCV is not installed); the `random` / `np.random`
call sequences are the reference's, so a seeded run draws the same batches.  Resampling differs from OpenCV's
INTER_AREA / INTER_CUBIC kernels by design limits of PIL (BOX / BICUBIC are the closest filters) -- this only
touches the free-size style images, never the bucketed training words, whose pixels are read unchanged."""
from __future__ import annotations

import os
import random

import numpy as np

from .data_utils import encode_word, normalize_images


def _read_gray(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("L"))


def load_prepare_data(input_dim, batch_size, reading_dir, char_vector, bucket_size):
    """Python generator of (images float32 [B,h,16L,c] in [-1,1], labels int32 [B,L]); one bucket per batch, the
    bucket drawn with probability proportional to its population, samples drawn with replacement."""
    data_buckets, bucket_weights, number_samples = {}, {}, 0
    for i in range(1, bucket_size + 1):
        imgs, labels = [], []
        bucket_dir = reading_dir + str(i) + '/'
        for file in [f for f in os.listdir(bucket_dir) if f.endswith(".txt")]:
            with open(bucket_dir + file, 'r', encoding='utf8') as f:
                label = encode_word(f.readline(), char_vector)
            imgs.append(_read_gray(os.path.join(bucket_dir, os.path.splitext(file)[0] + '.png')))
            labels.append(label)
            number_samples += 1
        data_buckets[i] = (imgs, labels)
    for i in range(1, bucket_size + 1):
        bucket_weights[i] = len(data_buckets[i][1]) / number_samples
    while True:
        bucket = int(np.random.choice(bucket_size, 1, p=list(bucket_weights.values()))[0]) + 1
        image_batch, label_batch = [], []
        for _ in range(batch_size):
            k = random.randint(0, len(data_buckets[bucket][1]) - 1)
            image_batch.append(data_buckets[bucket][0][k])
            label_batch.append(data_buckets[bucket][1][k])
        yield normalize_images(np.array(image_batch), input_dim, bucket), np.array(label_batch).astype(np.int32)


def _fit_style_image(img: np.ndarray, h: int, w: int, validate: bool) -> np.ndarray:
    from PIL import Image
    ht, wt = img.shape
    if validate:
        rate = min(h / ht, w / wt)
        dim = (int(wt * rate), h) if rate == h / ht else (w, int(ht * rate))
        resample = Image.BICUBIC
    else:
        rate = h / float(ht)
        dim = (int(wt * rate), h)
        resample = Image.BOX
    img = np.asarray(Image.fromarray(img.astype('float32'), mode="F").resize(dim, resample))
    width = img.shape[-1]
    if width > w:
        final = img[:, :w]
    elif width < w:
        final = np.ones([img.shape[0], w]) * 255
        final[:, :width] = img
    else:
        final = img
    return (final - 127.5) / 127.5


def load_style_input(input_dim, batch_size, bucket_size, style_dir="../../scrabble-gan/data/Utku_40/"):
    """-> (train_imgs, validate_imgs): lists of [32,160] arrays in [-1,1]; 95/5 split after random.shuffle."""
    h, w, _ = input_dim
    files = os.listdir(style_dir)
    random.shuffle(files)
    split = int(len(files) * 0.95)
    train = [_fit_style_image(_read_gray(os.path.join(style_dir, f)), h, w, False) for f in files[:split]]
    validate = [_fit_style_image(_read_gray(os.path.join(style_dir, f)), h, w, True) for f in files[split:]]
    return train, validate
