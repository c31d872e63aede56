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


def load_prepare_data(input_dim, batch_size, reading_dir, char_vector, bucket_size, raw=False):
    """Python generator of (images float32 [B,h,16L,c] in [-1,1], labels int32 [B,L]); one bucket per batch, the
    bucket drawn with probability proportional to its population, samples drawn with replacement.
    raw=True (not in the reference) yields the uint8 pixels [B,h,16L,c] instead: DevicePrefetcher normalises on the GPU."""
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
        if raw:
            h, _, c = input_dim
            yield (np.array(image_batch, dtype=np.uint8).reshape(-1, h, int((h / 2) * bucket), c), np.array(label_batch).astype(np.int32))
        else:
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


class DevicePrefetcher:
    """Host pipeline of SURVEY 8(f)-3 around a raw (uint8) batch generator.  The GENERATOR is pulled on the consumer's thread
    (at construction for the first `depth` batches, then one more per next()): it draws from the global `random` /
    `np.random` streams that train_step's fake-label draw and main.py's style split use too, so pulling it from a background
    thread would interleave the two consumers of those streams by timing -- a seeded run would not reproduce and
    data-parallel ranks would not see the same global batch.  Here the interleaving is a fixed function of the
    call sequence; the generator's own draw order is the reference loader's.  A background thread does the byte work only:
    it stages the pixels in PINNED host buffers (torch's caching host allocator); next() queues an asynchronous
    host-to-device copy of the bytes (a quarter of the fp32 volume) and the GPU-side pixel normalisation (sg_normalize_u8 =
    data_utils.py:82) on the current stream and returns (images fp32 on the device, labels int32 numpy): train_step takes
    device tensors as they are.  depth = batches staged ahead."""

    def __init__(self, raw_batches, device, depth: int = 2):
        import queue
        import threading
        import torch
        self._torch = torch
        self._device = torch.device(device)
        self._it = iter(raw_batches)
        self._in = queue.Queue()
        self._q = queue.Queue()
        self._exhausted = False

        def work():
            while True:
                item = self._in.get()
                if item is None:
                    self._q.put(None)
                    return
                try:
                    u8, labels = item
                    u8 = np.ascontiguousarray(u8, dtype=np.uint8)
                    n = u8.size
                    pad = (-n) % 16                                   # the kernel converts 16 pixels per thread
                    buf = torch.empty(n + pad, dtype=torch.uint8, pin_memory=self._device.type == "cuda")
                    buf[:n].copy_(torch.from_numpy(u8.reshape(-1)))
                    if pad:
                        buf[n:].zero_()
                    self._q.put((buf, u8.shape, labels))
                except Exception as e:  # noqa: BLE001  (surface staging errors in the consumer)
                    self._q.put(e)

        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()
        for _ in range(max(1, depth)):
            self._feed()

    def _feed(self):
        """Pull ONE batch from the generator on the calling thread and hand it to the staging thread."""
        if self._exhausted:
            return
        try:
            self._in.put(next(self._it))
        except StopIteration:
            self._exhausted = True
            self._in.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        from . import ops
        item = self._q.get()
        if item is None:
            self._q.put(None)
            raise StopIteration
        if isinstance(item, Exception):
            raise item
        self._feed()
        buf, shape, labels = item
        n = int(np.prod(shape))
        dev8 = buf.to(self._device, non_blocking=True)
        return ops.normalize_u8(dev8)[:n].view(*shape), labels

    def close(self):
        if not self._exhausted:
            self._exhausted = True
            self._in.put(None)
