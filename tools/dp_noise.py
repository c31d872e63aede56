import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_dp_gpu import _run_step
from scrabble_gan_amd.nn import Reducer
dev = torch.device("cuda:0")
for balance in (False, True):
    runs = [_run_step(Reducer(), dev, balance) for _ in range(3)]
    for n in ("D", "R", "S", "G"):
        a = runs[0][1][n]
        errs = [(r[1][n] - a).abs().max().item() for r in runs[1:]]
        print("balance", balance, n, "scale %.3e" % a.abs().max().item(), "run-to-run max err", ["%.2e" % e for e in errs])
