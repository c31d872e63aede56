"""Record of how much of each parity bound the GPU run used (test infrastructure).  Every `close`-style check of the network-level
tests reports (test, tensor, measured error relative to the tensor's scale, bound); at session end the table is written to
gpurun_out/parity_margins.txt so that a bound can be judged against what was measured instead of against a comment."""
import os

ROWS = []
CURRENT = [""]


def record(name, rel, tol):
    ROWS.append((CURRENT[0], name, rel, tol))


def dump():
    if not ROWS:
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    worst = {}
    for test, name, rel, tol in ROWS:
        k = (test, tol) if "decisions differing" not in name else (test, name)
        if k not in worst or rel / tol > worst[k][1] / tol:
            worst[k] = (name, rel)
    with open(os.path.join(out, "parity_margins.txt"), "w") as f:
        f.write("# per test and bound: the tensor that used the largest share of it (measured max|gpu - ref| / max|ref|, bound, share)\n")
        for (test, tol), (name, rel) in sorted(worst.items(), key=lambda kv: (kv[0][0], str(kv[0][1]))):
            if isinstance(tol, str):
                f.write("%-90s %s\n" % (test, name))
                continue
            f.write("%-90s %-40s %.3e  bound %.1e  %5.1f %%\n" % (test, name, rel, tol, 100.0 * rel / tol))
