"""Loss curves of a training run from its summaries (the counterpart of /root/reference/src/utilities.py:1-67).

The reference's script reads `<base_path>/batch_summary.csv` -- a file its train loop never writes (it writes the ';'-separated
`batch_summary.txt`, data_utils.py:254-255,296-300) -- averages it per epoch and saves three per-epoch figures and
one per-batch figure.  This module reads what `scrabble_gan_amd.data_utils.train` writes (`batch_summary.txt`, 16 ';'-separated
columns under SUMMARY_HEADER; a `batch_summary.csv` with the reference's column names is accepted too), maps the columns to the
names the reference plots, and writes the same four files:

    disc_loss_vis_per_epoch.png   d_loss, d_loss_fake, d_loss_real                                    (utilities.py:24-26)
    rec_gen_vis_per_epoch.png     r_loss_fake, g_loss, g_final_loss (+ balanced / std curves)          (:27-32)
    rec_loss_vis_per_epoch.png    r_loss_fake, r_loss_real (+ balanced / std curves)                   (:34-39)
    disc_loss_vis_per_batch.png   d_loss, d_loss_fake, d_loss_real over batches (info_per_batch)       (:42-49)

Host-side only (numpy + matplotlib's Agg canvas); nothing here touches the GPU."""
from __future__ import annotations

import os

import numpy as np

# column of data_utils.SUMMARY_HEADER -> the name the reference's plots use
_RENAME = {"disc_loss": "d_loss", "disc_loss_real": "d_loss_real", "disc_loss_fake": "d_loss_fake", "g_loss_final": "g_final_loss"}


def read_summary(path: str):
    """-> (column names, float array [rows, columns]) of a ';'- or ','-separated summary with one header line."""
    with open(path) as f:
        header = f.readline().strip()
        sep = ";" if ";" in header else ","
        names = [_RENAME.get(c.strip(), c.strip()) for c in header.split(sep)]
        rows = [[float(v) for v in line.strip().split(sep)] for line in f if line.strip()]
    data = np.asarray(rows, dtype=np.float64).reshape(len(rows), len(names))
    return names, data


def per_epoch_means(data: np.ndarray, batch_per_epoch: int) -> np.ndarray:
    """Mean of every column over consecutive groups of `batch_per_epoch` rows (a trailing partial epoch is averaged over its rows,
    as pandas' groupby(arange // batch_per_epoch).mean() does in the reference)."""
    n = data.shape[0]
    groups = [data[i:i + batch_per_epoch].mean(axis=0) for i in range(0, n, batch_per_epoch)]
    return np.stack(groups) if groups else np.zeros((0, data.shape[1]))


def _plot(x, table, names, cols, xlabel, out_path):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots()
    for c in cols:
        ax.plot(x, table[:, names.index(c)], label=c)
    ax.set_xlabel(xlabel)
    ax.legend()
    fig.savefig(out_path)
    plt.close(fig)


def main(base_path, batch_per_epoch, info_per_batch=True, gradient_balance=False):
    """Same parameters as the reference's `main` (utilities.py:8).  Returns the list of files written."""
    src = os.path.join(base_path, "batch_summary.txt")
    if not os.path.exists(src):
        src = os.path.join(base_path, "batch_summary.csv")
    names, data = read_summary(src)
    if "epoch" in names:                       # (a file in the reference's own layout carries epoch / batch columns)
        keep = [i for i, c in enumerate(names) if c not in ("epoch", "batch")]
        names, data = [names[i] for i in keep], data[:, keep]
    means = per_epoch_means(data, int(batch_per_epoch))
    epochs = np.arange(1, means.shape[0] + 1)
    written = []

    def save(fname, x, table, cols, xlabel):
        path = os.path.join(base_path, fname)
        _plot(x, table, names, cols, xlabel, path)
        written.append(path)

    save("disc_loss_vis_per_epoch.png", epochs, means, ["d_loss", "d_loss_fake", "d_loss_real"], "epoch")
    if gradient_balance:
        save("rec_gen_vis_per_epoch.png", epochs, means, ["r_loss_fake", "g_loss", "r_loss_balanced", "g_final_loss", "r_loss_fake_std", "g_loss_std"], "epoch")
        save("rec_loss_vis_per_epoch.png", epochs, means, ["r_loss_fake", "r_loss_real", "r_loss_balanced", "r_loss_fake_std", "g_loss_std"], "epoch")
    else:
        save("rec_gen_vis_per_epoch.png", epochs, means, ["r_loss_fake", "g_loss", "g_final_loss"], "epoch")
        save("rec_loss_vis_per_epoch.png", epochs, means, ["r_loss_fake", "r_loss_real"], "epoch")
    if info_per_batch:
        save("disc_loss_vis_per_batch.png", np.arange(data.shape[0]), data, ["d_loss", "d_loss_fake", "d_loss_real"], "batch")
    return written


if __name__ == "__main__":
    import sys
    if len(sys.argv) < 3:
        raise SystemExit("usage: python -m scrabble_gan_amd.utilities <base_path> <batch_per_epoch> [gradient_balance 0|1]")
    for p in main(sys.argv[1], int(sys.argv[2]), True, bool(int(sys.argv[3])) if len(sys.argv) > 3 else False):
        print(p)
