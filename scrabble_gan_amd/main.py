"""Entry point with the reference's configurable surface (/root/reference/src/main.py):
setup_optimizer :25-35, shared_specs (get_shared_specs) :38-40, io (setup_io) :43-51 and main() :54-116.

    python -m scrabble_gan_amd.main --gin configs/scrabble_gan_mi355x.gin [--synthetic] [--steps N]

`--synthetic` swaps the IAM bucket folders (absent offline) for the synthetic generators of
SURVEY section 8(d); everything downstream (factories, optimizers, train loop) is the same code.
The dataset conversion (dinterface), PNG grids and GIF writer of the reference are out of scope."""
from __future__ import annotations

import argparse
import os
import random

import numpy as np

from . import gin_config as gin
from .arch_ops import spectral_norm
from .data_utils import load_random_word_list, synthetic_batch, synthetic_random_words, train
from .net_architecture import (configure, make_discriminator, make_gan, make_generator, make_my_discriminator, make_my_recognizer,
                               make_recognizer, make_style_promoter)
from .net_loss import hinge, not_saturating
from .optimizers import Adam, RMSprop

gin.external_configurable(hinge)
gin.external_configurable(not_saturating)
gin.external_configurable(spectral_norm)


@gin.configurable
def setup_optimizer(g_lr, d_lr, r_lr, w_lr, beta_1, beta_2, loss_fn, disc_iters, apply_gradient_balance, rmsprop):
    generator_optimizer = Adam(learning_rate=g_lr, beta_1=beta_1, beta_2=beta_2)
    discriminator_optimizer = Adam(learning_rate=d_lr, beta_1=beta_1, beta_2=beta_2)
    if rmsprop:
        recognizer_optimizer = RMSprop(learning_rate=r_lr)
    else:
        recognizer_optimizer = Adam(learning_rate=r_lr, beta_1=beta_1, beta_2=beta_2)
    stylepromoter_optimizer = Adam(learning_rate=w_lr, beta_1=beta_1, beta_2=beta_2)
    return (generator_optimizer, discriminator_optimizer, recognizer_optimizer, stylepromoter_optimizer, loss_fn, disc_iters,
            apply_gradient_balance)


@gin.configurable('shared_specs')
def get_shared_specs(epochs, batch_size, latent_dim, embed_y, num_gen, kernel_reg, g_bw_attention, d_bw_attention, my_rec, my_disc):
    return epochs, batch_size, latent_dim, embed_y, num_gen, kernel_reg, g_bw_attention, d_bw_attention, my_rec, my_disc


@gin.configurable('io')
def setup_io(base_path, checkpoint_dir, gen_imgs_dir, model_dir, raw_dir, read_dir, input_dim, buf_size, n_classes, seq_len,
             char_vec, bucket_size):
    return (input_dim, buf_size, n_classes, seq_len, bucket_size, base_path + checkpoint_dir, base_path + gen_imgs_dir,
            base_path + model_dir, base_path + raw_dir, base_path + read_dir, char_vec)


def build_models(in_dim, latent_dim, embed_y, kernel_reg, g_bw_attention, d_bw_attention, n_classes, seq_len, my_rec=0, my_disc=0,
                 vis_model=False):
    """The factory call order of main.py:73-87."""
    generator = make_generator(latent_dim, in_dim, embed_y, kernel_reg, g_bw_attention, n_classes, vis_model=vis_model)
    if my_disc:      # (the reference's call site main.py:75 omits gen_path; the factory keeps the declared signature :417)
        discriminator = make_my_discriminator("", in_dim, kernel_reg, vis_model=vis_model)
    else:
        discriminator = make_discriminator(in_dim, kernel_reg, d_bw_attention, vis_model=vis_model)
    rec_factory = make_my_recognizer if my_rec else make_recognizer
    recognizer = rec_factory(in_dim, seq_len, n_classes + 1, vis_model=vis_model)
    style_promoter = make_style_promoter(in_dim, kernel_reg, d_bw_attention, vis_model=vis_model)
    gan = make_gan(generator, discriminator, recognizer, style_promoter, vis_model=vis_model)
    return generator, discriminator, recognizer, style_promoter, gan


def synthetic_dataset(batch_size, in_dim, bucket_size, n_classes, seed=0):
    """Python generator with the contract of load_prepare_data (data_utils.py:62-84): one word length per batch."""
    rng = np.random.default_rng(seed)
    k = 0
    while True:
        L = int(rng.integers(1, bucket_size + 1))
        images, labels, _ = synthetic_batch(batch_size, L, in_dim, n_classes, seed=seed + 17 * k)
        k += 1
        yield images, labels


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gin", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs",
                                                  "scrabble_gan_mi355x.gin"))
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--resumable", action="store_true",
                    help="write the full training state to <checkpoints>/state/latest.safetensors every epoch and resume from it when it exists")
    ap.add_argument("--conv-dtype", default="f32", choices=["f32", "bf16"], help="matrix-core operand type of the convolutions")
    ap.add_argument("--steps", type=int, default=None, help="cap batches per epoch")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0, help="seed of the host RNG streams (random, numpy) and of the weight initialisers")
    ap.add_argument("--deterministic", action="store_true",
                    help="no float-atomic reduction splits in the conv kernels: bitwise reproducible runs, a few percent slower")
    args = ap.parse_args(argv)

    # Data parallel (one process per GPU under torch.distributed.run): join the job BEFORE any GPU work, give every rank the
    # same host RNG streams (bucket choice, fake words, style images: every rank must see the same global batch and take its
    # slice) and the same initial weights.  A single process gets the identity reducer.
    from . import dist as sdist
    reducer = sdist.init_from_env(os.environ.get("SG_DIST_BACKEND", "nccl"))
    random.seed(args.seed)
    np.random.seed(args.seed)

    gin.parse_config_file(args.gin)
    epochs, batch_size, latent_dim, embed_y, num_gen, kernel_reg, g_bw_attention, d_bw_attention, my_rec, my_disc = get_shared_specs()
    in_dim, buf_size, n_classes, seq_len, bucket_size, ckpt_path, gen_path, m_path, raw_dir, read_dir, char_vec = setup_io()
    epochs = args.epochs or epochs
    batch_size = args.batch_size or batch_size
    configure(conv_dtype=args.conv_dtype, seed=args.seed, reducer=reducer, deterministic=args.deterministic)

    if args.synthetic:
        random_words = synthetic_random_words(bucket_size, 1000, n_classes)
        train_dataset = synthetic_dataset(batch_size, in_dim, bucket_size, n_classes)
        train_imgs = [synthetic_batch(1, 10, in_dim, n_classes, seed=1000 + i)[2][0] for i in range(64)]
    else:
        if not os.path.exists(read_dir):
            raise FileNotFoundError("%s not found: the IAM conversion (dinterface) is out of scope; use --synthetic" % read_dir)
        from .data_io import DevicePrefetcher, load_prepare_data, load_style_input      # "next" tier
        from .net_architecture import _device
        random_words = load_random_word_list(read_dir, bucket_size, char_vec)
        # uint8 pixels staged in pinned memory by a background thread, copied asynchronously, normalised on the GPU
        train_dataset = DevicePrefetcher(load_prepare_data(in_dim, batch_size, read_dir, char_vec, bucket_size, raw=True), _device())
        train_imgs, _ = load_style_input(in_dim, batch_size, bucket_size)

    generator, discriminator, recognizer, style_promoter, gan = build_models(
        in_dim, latent_dim, embed_y, kernel_reg, g_bw_attention, d_bw_attention, n_classes, seq_len, my_rec, my_disc,
        vis_model=getattr(reducer, "rank", 0) == 0)
    (generator_optimizer, discriminator_optimizer, recognizer_optimizer, stylepromoter_optimizer, loss_fn, disc_iters,
     apply_gradient_balance) = setup_optimizer()

    random_bucket_idx = random.randint(4, bucket_size - 1)
    labels = np.array([random.choice(random_words[random_bucket_idx]) for _ in range(num_gen)], np.int32)
    train(train_dataset, generator, discriminator, recognizer, style_promoter, gan, None, ckpt_path, generator_optimizer,
          discriminator_optimizer, recognizer_optimizer, stylepromoter_optimizer, train_imgs, [None, labels], buf_size, batch_size,
          epochs, m_path, latent_dim, gen_path, loss_fn, disc_iters, apply_gradient_balance, random_words, bucket_size, char_vec,
          max_batches_per_epoch=args.steps,
          state_path=os.path.join(ckpt_path, "state", "latest.safetensors") if args.resumable else None)


if __name__ == "__main__":
    main()
