"""Why does a graph-replayed shard step run SLOWER than the eager one (profiles/r03_streams_bs16.txt: 53.8 vs 40.1 ms on one box)?

Measures the GPU-side cost per kernel of (a) eager launches on one stream and (b) the same launches replayed from a HIP graph
(torch.cuda.CUDAGraph = hipGraph), for chains of N dependent tiny kernels (sg_add on 4 KB) and of N kernels that each run ~20 us
(sg_add on 16 MB), N = 600 ~ the launches of one fp32 train_step.  Wall time of the whole chain with the queue kept full (one
synchronize at the end), so host launch cost only shows where the host is the bottleneck.
    python tools/probe_graph_nodes.py > gpurun_out/r04_probe_graph_nodes.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrabble_gan_amd import ops  # noqa: E402


def chain(a, b, n):
    for _ in range(n):
        ops.add(a, b, out=a)


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    dev = torch.device("cuda:0")
    N = 600
    for numel, label in ((1024, "4 KB operands (launch-bound)"), (4 << 20, "16 MB operands (~20 us of HBM traffic per kernel)")):
        a = torch.zeros(numel, device=dev)
        b = torch.ones(numel, device=dev)
        chain(a, b, 10)
        t_eager = timed(lambda: chain(a, b, N))
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            chain(a, b, N)
        g.replay()
        t_graph = timed(g.replay)
        # host-only cost of queueing the eager chain (no sync inside): time to return from the python loop
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        chain(a, b, N)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("%-52s eager %7.2f us / kernel (host enqueue %5.2f us / kernel)   graph replay %7.2f us / kernel node" % (
            label, t_eager / N * 1e6, t_host / N * 1e6, t_graph / N * 1e6))


if __name__ == "__main__":
    main()
