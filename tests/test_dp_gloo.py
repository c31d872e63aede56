"""world_size-2 data-parallel tests on CPU (gloo): the reducer's sharding / SUM all-reduce, and the
data-parallel identity the train step relies on -- per-rank gradients of [B/P,1] targets SUM to the
single-process gradient of the [B,1] target, with the loss statistics reduced as 12 sums."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import scrabble_oracle as O
        from scrabble_gan_amd.dist import DistReducer
        red = DistReducer()
        assert red.world_size == world and red.rank == rank
        # sharding + SUM
        t = torch.arange(8.0).view(4, 2)
        sh = red.shard(t)
        assert sh.shape == (2, 2) and sh[0, 0].item() == 4.0 * rank
        s = red.all_reduce_sum(sh.sum().view(1).clone())
        assert s.item() == t.sum().item()
        with pytest.raises(ValueError):
            red.shard(torch.zeros(3, 2))
        # DP identity on a small conv net with a hinge target (no BatchNorm -> no SyncBN needed)
        g = torch.Generator().manual_seed(0)
        dt = torch.float64
        w1 = torch.randn(3, 3, 1, 8, generator=g, dtype=dt).requires_grad_(True)
        w2 = torch.randn(8, 1, generator=g, dtype=dt).requires_grad_(True)
        x = torch.randn(4, 8, 8, 1, generator=g, dtype=dt)

        def net(xx):
            return torch.relu(O.conv2d(torch.relu(xx), w1)).mean(dim=(1, 2)) @ w2

        full = torch.relu(1.0 + net(x))                              # [B,1] target: gradient of its SUM
        gf = torch.autograd.grad(full.sum(), [w1, w2])
        local = torch.relu(1.0 + net(red.shard(x)))
        gl = torch.autograd.grad(local.sum(), [w1, w2])
        flat = torch.cat([t.reshape(-1) for t in gl])
        red.all_reduce_sum(flat)
        assert torch.allclose(flat, torch.cat([t.reshape(-1) for t in gf]), atol=1e-12)
        # loss statistics: mean/std from all-reduced (sum, sum of squares, count)
        v = red.shard(full.detach().reshape(-1))
        sums = torch.stack([v.sum(), (v * v).sum(), torch.tensor(float(v.numel()), dtype=dt)])
        red.all_reduce_sum(sums)
        mean = sums[0] / sums[2]
        std = (sums[1] / sums[2] - mean * mean).clamp_min(0).sqrt()
        assert torch.allclose(mean, full.mean()) and torch.allclose(std, full.std(unbiased=False))
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


def test_dp_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): bench.py launches two fresh rank processes through
    torch.distributed.run, they rendezvous on 127.0.0.1 and rank 0 prints ONE JSON line.  --dry-run keeps
    the rehearsal GPU-free: the rendezvous and one SUM all-reduce, over gloo here (RCCL on the GPU node)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["rccl_ranks"] == 2 and out["allreduce_of_ones"] == 2 and out["n_gpus"] == 2 and out["backend"] == "gloo"
    # a rank failure must surface as a non-zero exit code of the launcher
    env["SG_DIST_BACKEND"] = "no-such-backend"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
