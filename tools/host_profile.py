"""Where the host time of one train_step goes (cProfile over a few queued steps at a small batch)."""
import cProfile, os, pstats, random, sys, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, optimizers
from scrabble_gan_amd.main import build_models
dev = torch.device("cuda:0")
NA.configure(device=dev, seed=0)
G, D, R, S, gan = build_models((32, 160, 1), 128, (32, 8192), None, "B3", "B1", 52, None)
opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
B, L = int(os.environ.get("B", "16")), 10
images, labels, my_imgs = DU.synthetic_batch(B, L, (32, 160, 1), 52, seed=0)
words = DU.synthetic_random_words(10, 1000, 52, seed=0)
random.seed(0)
fake = torch.from_numpy(np.array([random.choice(words[L - 1]) for _ in range(B)], np.int32)).to(dev)
im, my, lb = torch.from_numpy(images).to(dev), torch.from_numpy(my_imgs).to(dev), torch.from_numpy(labels).to(dev)
def step(i):
    return DU.train_step(0, i, 9, im, lb, D, R, S, gan, opts[0], opts[1], opts[2], opts[3], my, B, 128, net_loss.hinge, 1, 0, words, 10, "",
                         fake_labels=fake, verbose=False, sync="lazy")
for i in range(3):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
outs = [step(3 + i) for i in range(4)]
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
