"""Train-step throughput of the MI355X-native ScrabbleGAN hot path (BASELINE.json metric:
train-step images/sec @ 32x160 bs128, 1/2/4/8 GPU; % of the MFMA roofline).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full train_step (8 forward passes, 4 backward sweeps, 4 Adam updates,
/root/reference/src/bigacgan/data_utils.py:358-473) on synthetic 32x160 word crops (config c2 of
SURVEY section 8d: global batch 128, L_r = L_f = 10, fp32), inputs resident in HBM.  The global batch
is fixed as N grows (strong scaling, gradients SUM-all-reduced over RCCL).  Rank 0 prints ONE JSON
line; `roofline` is the implicit-GEMM conv kernel (fwd + data-grad launches) timed with HIP events on
its own stream, `cpu_baseline` is the CPU oracle's train_step on a bounded sample (bs 8).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_IMAGE = 379.1e9            # 3 F_G + 15 F_D + 5 F_R at L_r = L_f = 10 (the reference's four tapes)
FLOP_PER_IMAGE_EXECUTED = 379.1e9 - 2 * 19.891e9   # the shared backward sweep through D(x_f) and S(x_f) runs each data-grad chain once
PEAK_FP32_MFMA_TF = 157.3           # MI355X_MICROARCH.md chip table (v_mfma_f32_32x32x2_f32)


def cpu_baseline(batch=8, L=10, warmup=2, timed=5, budget_s=100.0):
    """The oracle's train_step (torch-CPU fp32, torch's default intra-op thread pool on this host) on a bounded sample,
    with the protocol of BASELINE.md section 3: bs 8, `warmup` untimed steps, median of `timed` steps.  Runs on rank 0 at
    N = 1 only, after the GPU measurement (outside the timed region).  The whole leg is bounded by `budget_s` seconds of CPU
    work: on a slow or loaded host fewer warm-up / timed steps are taken and the sample string says how many.  One progress
    line per step goes to stderr."""
    import statistics
    from oracle import scrabble_oracle as O
    cores = torch.get_num_threads()
    dt = torch.float32
    g = torch.Generator().manual_seed(1)
    G, D, S, R = O.init_generator(g, dt), O.init_discriminator(g, dt), O.init_discriminator(g, dt), O.init_recognizer(g, dt)
    images = torch.rand(batch, 32, 16 * L, 1, generator=g, dtype=dt) * 2 - 1
    style = torch.rand(batch, 32, 160, 1, generator=g, dtype=dt) * 2 - 1
    labels = torch.randint(0, 52, (batch, L), generator=g)
    fake = torch.randint(0, 52, (batch, L), generator=g)
    nl = {k: O.init_nonlocal(64, g, dt) for k in ("G.style", "G.up", "D.fake", "D.real", "S.fake", "S.style", "S.real")}
    opt = {"G": {}, "D": {}, "R": {}, "S": {}}
    times, t_start = [], time.time()
    n_warm = warmup
    while len(times) < n_warm + timed:
        t0 = time.time()
        O.train_step(images, labels, style, fake, G, D, S, R, nl, opt)
        times.append(time.time() - t0)
        print("[cpu_baseline] step %d: %.2f s (%d threads)" % (len(times), times[-1], cores), file=sys.stderr, flush=True)
        spent, per = time.time() - t_start, times[-1]
        if len(times) == 1 and per * (warmup + 2) > budget_s:
            n_warm = 1                                  # slow host: one warm-up only
        if len(times) > n_warm and spent + per > budget_s:
            break
        if len(times) == n_warm and spent + per > budget_s and len(times) > 1:
            break
    samples = times[n_warm:] if len(times) > n_warm else times[-1:]
    med = statistics.median(samples)
    return {"value": batch / med, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "torch-CPU fp32 oracle train_step at bs %d, 32x160 (L = %d): %d warm-up + median of %d timed steps "
                      "(BASELINE.md section 3 asks for 2 + 5; bounded to %d s of CPU work); stand-in for the TF2 CPU path, "
                      "TensorFlow is not installable offline" % (batch, L, len(times) - len(samples), len(samples), int(budget_s)),
            "seconds_per_step": med, "all_steps_s": [round(t, 3) for t in times]}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` from a bare shell: start the N ranks ourselves (one fresh process per GPU through
    torch.distributed.run, before this process has touched the GPU), forward rank 0's JSON line, exit with the job's code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def dry_run(args):
    """Launcher rehearsal without a GPU (tests/test_dp_gloo.py): rendezvous, one all-reduce, rank 0 prints a JSON line."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("SG_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
        t = torch.ones(1)
        dist.all_reduce(t)
        ranks, got = dist.get_world_size(), int(t.item())
        rank = dist.get_rank()
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks, got, rank = 1, 1, 0
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": args.gpus, "rccl_ranks": ranks, "backend": backend, "allreduce_of_ones": got}))


WORKLOADS = {
    "f32": "c2: synthetic random_words 32x160, global bs %d, L_r=L_f=%d, fp32 MFMA convs (3x3 over >= 64 channels as F(4x4,3x3) Winograd-domain products: exact fp32 products, fp32 accumulation), hinge, disc_iters=1",
    "bf16": "c3: synthetic random_words 32x160, global bs %d, L_r=L_f=%d, bf16 MFMA convs (fp32 accumulation, bf16 operand copies in HBM), hinge, disc_iters=1",
    "fp8": "c5: synthetic random_words 32x160, global bs %d, L_r=L_f=%d, fp8 convs of G/D/S (e4m3 forward / data-grad operands, e4m3 x e5m2 weight-grads, >= 128 channels), bf16 recognizer + CTC, hinge",
}
DTYPE_NAME = {"f32": "f32", "bf16": "bf16", "fp8": "fp8 (e4m3 / e5m2) + bf16"}
# dense matrix-core peaks of MI355X_MICROARCH.md's chip table, per kernel family
PEAK_TF = {"f32": PEAK_FP32_MFMA_TF, "bf16": 2500.0, "fp8": 5000.0}


def _git_commit_of(path):
    """Short hash and date of the last commit that touched `path` ('uncommitted' when git has none / is absent)."""
    import subprocess
    try:
        r = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h %cs", "--", path], capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or "uncommitted"
    except Exception:  # noqa: BLE001
        return "unknown"


def committed_traffic(kernel, conv_dtype, per_gpu_batch):
    """HBM-side bytes per launch from the committed PMC passes of this same command (tools/pmc_traffic.py; counters need
    their own rocprofv3 runs, so the figure is read back, not measured in this process) -> (bytes, source string carrying
    the profile file's commit and date, so a stale figure is visible as such)."""
    import glob
    tag = "%s%s_traffic_bs%d.json" % (kernel, {"f32": "", "bf16": "_bf16", "fp8": "_fp8"}[conv_dtype], per_gpu_batch)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + tag)))
    if not files:
        return None, None
    tj = json.load(open(files[-1]))
    rel = "profiles/" + os.path.basename(files[-1])
    # (the file records the commit its PMC passes ran on; git itself is not available on a box that received a snapshot)
    stamp = tj.get("source_commit") or _git_commit_of(rel)
    return tj["traffic_bytes_per_launch"], "%s @ %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/pmc_traffic.py)" % (rel, stamp)


TIMING_IN_REGION = False


def measure(conv_dtype, B, L, balance, steps, warmup, timing_steps, reducer, dev, world, bucketed=False, sync_every_step=False,
            kernel_timing=True, shape_table=None, hbm_families=True, graph=False):
    """Build the four networks, run `warmup` + `steps` train_steps of one configuration with the inputs resident in HBM, and
    return everything measured: wall time of the timed region (MAX over ranks), HIP-event summaries of the kernel families.
    The models and every per-step buffer are released before returning."""
    from scrabble_gan_amd import data_utils as DU, net_architecture as NA, net_loss, ops, optimizers
    from scrabble_gan_amd.main import build_models
    ops.set_conv_dtype(conv_dtype)
    in_dim = (32, 160, 1)
    NA.configure(device=dev, seed=0, reducer=reducer)         # same seed on every rank -> identical replicas
    G, D, R, S, gan = build_models(in_dim, 128, (32, 8192), None, "B3", "B1", 52, None)
    opts = [optimizers.Adam(2e-4, 0.0, 0.999) for _ in range(4)]
    images, labels, my_imgs = DU.synthetic_batch(B, L, in_dim, 52, seed=0)
    words = DU.synthetic_random_words(max(10, L), 1000, 52, seed=0)
    random.seed(0)
    fake = np.array([random.choice(words[L - 1]) for _ in range(B)], np.int32)      # random_bucket_idx forced to L-1
    # inputs resident in HBM before the timed region; every rank holds the global batch and takes its slice
    images_d, my_d = torch.from_numpy(images).to(dev), torch.from_numpy(my_imgs).to(dev)
    labels_d, fake_d = torch.from_numpy(labels).to(dev), torch.from_numpy(fake).to(dev)

    # sync="lazy": the 16 scalars of a step come back by an asynchronous copy and are read after the next step has
    # been queued (as scrabble_gan_amd.data_utils.train does); every step's values are read and checked below.
    pools = None
    if bucketed:          # variable-width words: inputs of every bucket resident in HBM, the per-step pair from a shared stream
        words23 = DU.synthetic_random_words(23, 1000, 52, seed=0)
        pools = {}
        for Lb in range(4, 24):
            im, lb, _ = DU.synthetic_batch(B, Lb, in_dim, 52, seed=Lb)
            fk = np.array([random.choice(words23[Lb - 1]) for _ in range(B)], np.int32)
            pools[Lb] = (torch.from_numpy(im).to(dev), torch.from_numpy(lb).to(dev), torch.from_numpy(fk).to(dev))
        pair_rng = np.random.default_rng(7)
        pairs = [tuple(int(v) for v in pair_rng.integers(4, 24, 2)) for _ in range(warmup + steps + 2 * timing_steps)]

    gs = None
    if graph:       # the step captured once into a HIP graph, then one launch per step (scrabble_gan_amd/graph_step.py)
        from scrabble_gan_amd.graph_step import GraphedStep
        gs = GraphedStep(D, R, S, gan, opts, B, net_loss.hinge, int(balance), warmup=2).capture(images_d, labels_d, my_d, fake_d)

    def step(i):
        if gs is not None:
            return gs.step()
        if pools is not None:
            L_r, L_f = pairs[i % len(pairs)]
            return DU.train_step(0, i, steps, pools[L_r][0], pools[L_r][1], D, R, S, gan, opts[0], opts[1], opts[2], opts[3], my_d,
                                 B, 128, net_loss.hinge, 1, 0, words, 23, "", fake_labels=pools[L_f][2], verbose=False,
                                 sync=True if sync_every_step else "lazy")
        return DU.train_step(0, i, steps, images_d, labels_d, D, R, S, gan, opts[0], opts[1], opts[2], opts[3], my_d, B, 128,
                             net_loss.hinge, 1, int(balance), words, max(10, L), "", fake_labels=fake_d, verbose=False,
                             sync=True if sync_every_step else "lazy")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i - warmup)            # negative indices: the warm-up draws its own pairs in --bucketed mode
    # Kernel timing (HIP events around every MFMA conv launch, on the launch stream) runs inside the timed region at
    # N = 1.  At N > 1 the per-GPU batch is small enough for the event records to cost a few percent, so there the
    # timed region runs bare and the roofline figures come from --timing-steps extra steps right after it.
    # Round 4: the timed region runs BARE at every N -- exactly the path scrabble_gan_amd.data_utils.train runs (two HIP streams: S's
    # passes beside D's and R's; one C-ABI call per Winograd-domain convolution) -- and the per-kernel HIP-event timing happens in
    # --timing-steps extra steps right after it, on one stream (a kernel's duration only measures the kernel while it has the GPU
    # to itself).  --timing-in-region restores the round-3 behaviour (events inside the timed region, single stream).
    timer = ops.KernelTimer(only=("igemm", "wgrad")) if kernel_timing else None      # the MFMA conv kernels (+ their thin variants)
    timing_in_region = timer is not None and world == 1 and TIMING_IN_REGION
    ops.PROFILER = timer if timing_in_region else None
    fence()
    calls_before = getattr(reducer, "calls", 0)          # collectives of the warm-up steps
    t0 = time.perf_counter()
    outs = []
    host_enqueue = 0.0                                            # host time spent queuing the steps (no device waits in there)
    for i in range(steps):
        t_q = time.perf_counter()
        outs.append(step(i))
        host_enqueue += time.perf_counter() - t_q
        if i:
            tuple(outs[i - 1])                                    # read back step i-1 while step i runs
    fence()
    tuple(outs[-1])
    elapsed = time.perf_counter() - t0
    calls_timed = getattr(reducer, "calls", 0) - calls_before      # collectives inside the timed region (before any extra timing steps)
    ops.PROFILER = None
    if world > 1:                      # the measurement proper is complete here: MAX over ranks first, extras afterwards
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    assert all(np.isfinite(float(v)) for o in outs for v in o), outs
    timed_steps = steps
    if timer is not None and not timing_in_region:
        ops.PROFILER = timer
        extra = [step(steps + i) for i in range(timing_steps)]
        fence()
        ops.PROFILER = None
        timed_steps = timing_steps
        for o in extra:
            tuple(o)
    # The memory-bound kernel families (BN, pools, Adam, filter bank, elementwise, attention) are timed in extra steps
    # AFTER the timed region, so that their ~1 000 event records per step never touch the headline number.
    hbm_timer = None
    if timer is not None and timing_steps > 0 and hbm_families:
        hbm_timer = ops.KernelTimer()
        ops.PROFILER = hbm_timer
        extra = [step(steps + timing_steps + i) for i in range(timing_steps)]
        fence()
        ops.PROFILER = None
        for o in extra:
            tuple(o)
    res = {"elapsed": elapsed, "steps": steps, "B": B, "L": L, "host_enqueue": host_enqueue, "calls_timed": calls_timed,
           "timed_steps": timed_steps, "timing_in_region": timing_in_region, "ks": None, "hs": None, "hroof": None, "error": None}
    if timer is not None:
        try:
            res["ks"] = timer.summary()
            if hbm_timer is not None:
                res["hs"], res["hroof"] = hbm_timer.summary(), dict(hbm_timer.roof)
            if shape_table:
                with open(shape_table, "w") as f:
                    f.write("# per-shape MFMA conv launches inside train_step, per-GPU batch %d, %d steps (HIP events on the launch stream)\n" % (B // world, timed_steps))
                    f.write("%-9s %-12s %5s %4s %4s %5s %5s %2s | %4s %9s %8s\n" % ("family", "kind", "B", "H", "W", "Cin", "Cout", "k", "n", "ms/step", "TFLOP/s"))
                    for fam, tag, n, ms_, tf in timer.by_shape():
                        f.write("%-9s %-12s %5d %4d %4d %5d %5d %2d | %4d %9.3f %8.1f\n" % ((fam,) + tuple(tag) + (n // timed_steps, ms_ / timed_steps, tf)))
        except Exception as e:  # noqa: BLE001  (the headline numbers must still be printed)
            res["error"] = repr(e)
    # release the models, optimizer slots, saved contexts and operand copies before the next configuration is built
    del G, D, R, S, gan, opts, outs, images_d, my_d, labels_d, fake_d, pools
    ops.new_step()
    ops.weights_changed()
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res


def roofline_of(res, conv_dtype, world):
    """`roofline` (the dominant kernel family of the configuration) + `kernels` (every MFMA conv family) + `hbm_kernels`."""
    ks, timed_steps = res["ks"], res["timed_steps"]
    out = {}
    if ks is None:
        if res["error"]:
            out["roofline_error"] = res["error"]
        return out
    # fp8 mode: the e4m3 / e5m2 launches are their own families ('igemm_fp8', 'wgrad_fp8', peak 5 PF); the launches that stay
    # bf16 there (64-channel layers, recognizer, transposed convs) keep 'igemm' / 'wgrad' against the bf16 peak
    def peak_of(fam):
        if fam.endswith("_fp8"):
            return PEAK_TF["fp8"]
        return PEAK_TF["f32"] if conv_dtype == "f32" else PEAK_TF["bf16"]
    dom = "igemm_fp8" if (conv_dtype == "fp8" and "igemm_fp8" in ks) else "igemm"
    if conv_dtype == "f32" and ks.get("igemm_wino", {}).get("ms", 0.0) > ks.get("igemm", {}).get("ms", 0.0):
        dom = "igemm_wino"      # the grouped Winograd-domain products (one kernel template: sg_igemm_bf16v2_kernel<128 | 64, 4, false, 128, 0 | 8>)
    ig = ks.get(dom, {"tflops": 0.0, "launches": 0, "ms": 0.0})
    try:
        traffic, traffic_src = committed_traffic(dom if dom == "igemm_wino" else "igemm", conv_dtype, res["B"] // world)
    except Exception:  # noqa: BLE001
        traffic, traffic_src = None, None
    kernel_name = {"f32": "sg_igemm_bf16v2_kernel<128, 4, false, 128, 0 | 8> (four / eight waves per 128 x 128 tile; <64, 4, false, 128, 0> for 64 output channels): the 36 (F(4x4,3x3)) or 16 (F(2x2,3x3)) grouped Winograd-domain products of a 3x3 conv fwd / data-grad in one launch, "
                          "fp32 MFMA 32x32x2, executed FLOPs; the direct launches -- 1x1, 64-channel, strided -- are the family 'igemm')" if dom == "igemm_wino" else
                          "sg_igemm_kernel + sg_igemm_bf16v2_kernel<BN, 4, RELU> for the large-grid launches (conv fwd + data-grad, fp32 MFMA 32x32x2)",
                   "bf16": "sg_igemm_bf16v2_kernel<BN, 2, RELU> / sg_igemm_bf16_kernel (conv fwd + data-grad, bf16 MFMA 32x32x16; <= 32-filter convs stay fp32)",
                   "fp8": "sg_igemm_bf16v2_kernel<256, 1, false> (conv fwd + data-grad of the >= 128-channel layers, v_mfma_scale_f32_32x32x64_f8f6f4 on e4m3 operands)"}[conv_dtype]
    out["roofline"] = {"bound": "mfma", "achieved": ig["tflops"], "peak": peak_of(dom), "unit": "TFLOP/s",
                       "frac": ig["tflops"] / peak_of(dom), "traffic": traffic, "traffic_source": traffic_src,
                       "algorithmic_bytes_per_launch": ig.get("bytes", 0.0) / max(ig["launches"], 1),
                       "algorithmic_flop_per_launch": ig.get("flops", 0.0) / max(ig["launches"], 1),
                       "kernel": kernel_name, "family": dom,
                       "launches_per_step": ig["launches"] / timed_steps, "ms_per_step": ig["ms"] / timed_steps,
                       "timed": "inside the timed region" if res["timing_in_region"] else "%d extra steps after the timed region" % timed_steps}
    out["kernels"] = {k: {"tflops": round(v["tflops"], 2), "ms_per_step": round(v["ms"] / timed_steps, 3),
                          "launches_per_step": v["launches"] / timed_steps, "peak_tflops": peak_of(k) if not k.endswith("_thin") else None,
                          "frac_of_mfma_peak": round(v["tflops"] / peak_of(k), 4) if not k.endswith("_thin") else None,
                          # every operand read once + every result written once, per launch (beside the PMC `traffic` where measured)
                          "algorithmic_bytes_per_launch": (v.get("bytes", 0.0) / max(v["launches"], 1)) or None}
                      for k, v in ks.items()}
    for fam in ("wgrad", "wgrad_wino", "igemm"):
        if fam in out["kernels"] and fam != dom:
            try:
                wt, wsrc = committed_traffic(fam, conv_dtype, res["B"] // world)
            except Exception:  # noqa: BLE001
                wt, wsrc = None, None
            out["kernels"][fam].update({"traffic": wt, "traffic_source": wsrc})
    if res["hs"] is not None:
        # memory-bound families: algorithmic bytes (every operand once + every result once) / HIP-event time,
        # against the 8 TB/s HBM3E peak of MI355X_MICROARCH.md (6.3 TB/s is what a float4 copy reaches)
        out["hbm_kernels"] = {}
        for k, v in sorted(res["hs"].items(), key=lambda kv: -kv[1]["ms"]):
            roof = res["hroof"].get(k, "hbm")
            if roof == "mfma" or v["ms"] <= 0:
                continue
            nst = max(1, res["hbm_steps"]) if "hbm_steps" in res else 1
            gbps = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            ent = {"bound": roof, "ms_per_step": round(v["ms"] / nst, 3), "launches_per_step": v["launches"] / nst,
                   "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4)}
            if v["flops"]:
                ent["tflops"] = round(v["tflops"], 2)
            out["hbm_kernels"][k] = ent
    if res["error"]:
        out["roofline_error"] = res["error"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="GLOBAL batch (fixed as N grows); default 128")
    ap.add_argument("--L", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-in-region", action="store_true", help="HIP-event kernel timing inside the timed region (single stream, N = 1) instead of in extra steps after it")
    ap.add_argument("--timing-steps", type=int, default=2, help="extra steps after the timed region for the per-kernel HIP-event timing of the memory-bound families (and, at N > 1, of the conv kernels)")
    ap.add_argument("--conv-dtype", default=None, choices=["f32", "bf16", "fp8"],
                    help="matrix-core operand type of the convolutions: f32 = the headline config c2 (default); bf16 = config c3 (run it with --batch 256); "
                         "fp8 = config c5 (fp8 forward / data-grad / weight-grad of the >= 128-channel convs of G/D/S, bf16 recognizer; --batch 512 --balance)")
    ap.add_argument("--balance", action="store_true", help="apply_gradient_balance = 1 (config c5)")
    ap.add_argument("--bucketed", action="store_true",
                    help="config c4: one (L_r, L_f) pair per step drawn U{4..23}^2 from a stream shared by all ranks (bucket_size 23)")
    ap.add_argument("--sync-every-step", action="store_true", help="read the 16 scalars back before queuing the next step")
    ap.add_argument("--shape-table", default=None, help="write the per-shape time / TFLOP/s table of the MFMA conv kernels here")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="the default N = 1 invocation also measures configs c3 (bf16, bs 256) and c5 (fp8, bs 512, balancing) after the "
                         "headline's timed region and reports them under \"configs\"; this flag skips them")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step into a HIP graph once and replay it (single GPU, fixed shapes; no per-kernel timing: the "
                         "roofline object is omitted) -- for the small-batch bf16 / fp8 steps whose host time is within 2x of the GPU time")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous + one all-reduce only, no GPU work (launcher rehearsal on CPU)")
    args = ap.parse_args()
    global TIMING_IN_REGION
    TIMING_IN_REGION = bool(args.timing_in_region)
    plain_invocation = args.conv_dtype is None and args.batch is None and not args.bucketed and not args.balance and args.L == 10 and not args.graph
    if args.conv_dtype is None:
        args.conv_dtype = "f32"
    if args.batch is None:
        args.batch = 128

    # N > 1 from a bare shell: start the ranks (nothing in this process has touched the GPU yet)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])
    if args.dry_run:
        return dry_run(args)

    from scrabble_gan_amd import dist as sdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or unset WORLD_SIZE "
                         "and let bench.py start its own ranks" % (args.gpus, world, args.gpus))
    # SG_DIST_BACKEND=gloo SG_FORCE_DEVICE=0 rehearse the multi-rank path with several ranks on ONE GPU (RCCL needs
    # one device per rank); the driver's runs use the defaults: nccl, one GPU per rank.
    backend = os.environ.get("SG_DIST_BACKEND", "nccl")
    local_rank = int(os.environ.get("SG_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if backend != "nccl":
        torch.cuda.set_device(local_rank)
    reducer = sdist.init_from_env(backend)
    rank = getattr(reducer, "rank", 0)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    B, L = args.batch, args.L
    res = measure(args.conv_dtype, B, L, args.balance, args.steps, args.warmup, args.timing_steps, reducer, dev, world,
                  bucketed=args.bucketed, sync_every_step=args.sync_every_step, kernel_timing=not (args.no_kernel_timing or args.graph),
                  shape_table=args.shape_table if rank == 0 else None, graph=args.graph and world == 1 and not args.bucketed)
    res["hbm_steps"] = args.timing_steps

    # configs c3 and c5 of BASELINE.json in the same (driver-run) invocation: after the headline's timed region and its timing
    # steps, on freshly built models, each with its own warm-up; single GPU only (the scaling runs time c2 alone)
    extra = {}
    if plain_invocation and world == 1 and not args.no_extra_configs and not args.no_kernel_timing:
        for tag, cd, eb, bal in (("c3", "bf16", 256, False), ("c4", "f32", 256, False), ("c5", "fp8", 512, True)):
            try:
                bk = tag == "c4"            # c4: one (L_r, L_f) pair per step from a shared stream, L in [4, 23] (data_utils.py:62-84,386-392)
                r = measure(cd, eb, 10, bal, max(3, min(args.steps, 10)), 2, 1, reducer, dev, world, hbm_families=True, bucketed=bk)
                r["hbm_steps"] = 1
                v = eb * r["steps"] / r["elapsed"]
                ent = {"workload": ("c4: synthetic random_words 32x(16 L), (L_r, L_f) ~ U{4..23}^2 per step, global bs %d, fp32 MFMA convs "
                                    "(F(4x4,3x3) / F(2x2,3x3) Winograd-domain products); 1 of the 8 GPUs c4 names, whole batch" % eb) if bk else
                                   WORKLOADS[cd] % (eb, 10) + (", gradient balancing on" if bal else ""), "value": v, "unit": "images/s",
                       "ms_per_step": r["elapsed"] / r["steps"] * 1e3, "steps": r["steps"], "warmup": 2, "dtype": DTYPE_NAME[cd],
                       "global_batch": eb, "host_enqueue_ms_per_step": r["host_enqueue"] / r["steps"] * 1e3,
                       "step_algorithmic_tflops": None if bk else FLOP_PER_IMAGE * v / 1e12}
                ent.update(roofline_of(r, cd, world))
                extra[tag] = ent
            except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
                extra[tag] = {"error": repr(e)}
        from scrabble_gan_amd import ops as _ops
        _ops.set_conv_dtype("f32")

    if rank == 0:
        elapsed = res["elapsed"]
        ms = elapsed / args.steps * 1e3
        value = B * args.steps / elapsed
        try:
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:  # noqa: BLE001
            metric = "train-step images/sec @ 32x160 bs128"
        line = {
            "metric": metric, "value": value, "unit": "images/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": DTYPE_NAME[args.conv_dtype], "data": "synthetic",
            "config": {"workload": ((WORKLOADS[args.conv_dtype] % (B, L)) + (", gradient balancing on" if args.balance else "")) if not args.bucketed else
                                   "c4: synthetic random_words 32x(16 L), (L_r, L_f) ~ U{4..23}^2 per step, global bs %d, %s MFMA convs" % (B, args.conv_dtype),
                       "global_batch": B, "per_gpu_batch": B // world, "parallelism": "dp%d" % world},
            "host_enqueue_ms_per_step": res["host_enqueue"] / args.steps * 1e3,
            "step_algorithmic_tflops": FLOP_PER_IMAGE * value / 1e12 if (L == 10 and not args.bucketed) else None,      # reference-tape accounting
            "step_executed_tflops": FLOP_PER_IMAGE_EXECUTED * value / 1e12 if (L == 10 and not args.bucketed) else None,  # direct form, shared sweeps counted once
        }
        if args.conv_dtype == "f32":
            line["flop_accounting"] = ("step_algorithmic_tflops / step_executed_tflops count DIRECT-form convolution FLOPs (the reference's tapes); in fp32 "
                                       "mode the 3x3 convolutions over >= 64 channels run in the Winograd domain -- F(4x4,3x3) at 9/36 of that count where H and W are "
                                       "multiples of 4, F(2x2,3x3) at 16/36 otherwise (conv_winograd.hip) -- so these rates may exceed the fp32 MFMA peak.  roofline.achieved and kernels.* count the "
                                       "products the matrix cores EXECUTE (Winograd-domain products for those launches) over HIP-event time")
        line["config"]["hip_graph"] = bool(args.graph)
        line["config"]["streams"] = "S's passes on a second HIP stream beside D's and R's (timed region); per-kernel timing on one stream"
        line["config"]["fused_passes"] = True
        line["config"]["shared_backward"] = True
        line["rccl_ranks"] = torch.distributed.get_world_size() if world > 1 else 1
        line["backend"] = (backend + (" (RCCL)" if backend == "nccl" else "")) if world > 1 else "none (single process)"
        line["collectives_per_step"] = res["calls_timed"] / max(1, args.steps) if world > 1 else 0
        line.update(roofline_of(res, args.conv_dtype, world))
        if res["ks"] is not None and res["timed_steps"]:
            # what the matrix cores executed in one step (every MFMA conv family, Winograd-domain products counted as such) over the
            # step time: the whole-step fraction of the MFMA peak, next to the dominant kernel's own fraction in `roofline`
            fl = sum(v.get("flops", 0.0) for k, v in res["ks"].items() if not k.endswith("_thin")) / res["timed_steps"]      # (rank 0's own launches)
            pk = PEAK_TF["f32"] if args.conv_dtype == "f32" else None
            line["step_mfma_executed"] = {"tflop_per_step_per_gpu": fl / 1e12, "tflops_per_gpu": fl / (ms * 1e-3) / 1e12,
                                          "frac_of_mfma_peak": (fl / (ms * 1e-3) / 1e12 / pk) if pk else None}
        if extra:
            line["configs"] = extra
        if args.gpus == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
