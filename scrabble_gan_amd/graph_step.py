"""One train_step captured into a HIP graph and replayed (VERDICT r2 #4c).

A step queues ~600 (fp32) to ~1 000 (bf16 / fp8) kernel launches; the host needs 7-30 ms for that, which is within 2x of the
GPU time of the bf16 / fp8 steps at the 8-way shard batch (34 / 50 ms).  With fixed shapes the launch sequence of a step is
the same every step, so it is captured ONCE (torch.cuda.CUDAGraph on ROCm = hipGraph; fresh capture in this process, nothing is
re-executed) and each further step is one hipGraphLaunch.  What changes from step to step lives in device memory the graph
reads: the input batch, the freshly drawn NonLocalBlock kernels (host QR per call, uploaded before the replay)
and Adam's bias-corrected step size (sg_adam_update_dlr).  Single process, Adam optimizers, fixed word lengths; the data-parallel
exchange is not captured (world size 1)."""
from __future__ import annotations

import torch

from . import data_utils as DU
from . import nn, ops

NL_NAMES = ("G.style", "G.up", "D.fake", "S.fake", "D.real", "S.style", "S.real")      # the draw order of data_utils.train_step


class GraphedStep:
    def __init__(self, discriminator, recognizer, style_promoter, composite_gan, optimizers, batch_size, loss_fn, apply_gradient_balance,
                 warmup: int = 2):
        self.D, self.R, self.S, self.gan = discriminator, recognizer, style_promoter, composite_gan
        self.G = composite_gan.generator
        self.opts = list(optimizers)               # (generator, discriminator, recognizer, style promoter) as train_step takes them
        self.B, self.loss_fn, self.balance, self.warmup = batch_size, loss_fn, int(apply_gradient_balance), warmup
        self.graph = None
        if self.G.reducer.world_size != 1:
            raise ValueError("GraphedStep captures single-process steps only")
        if self.G.nl_mode != "reference":
            raise ValueError("GraphedStep expects the per-call NonLocalBlock kernel re-draw (nl_mode='reference')")

    # ---- per-step host work: the seven kernel sets of the step, drawn like train_step does, into one pinned buffer
    def _nl_requests(self):
        G, D, S = self.G, self.D, self.S
        up_c = G.out_ch[G.up_names.index(G.up_attn[0])]
        tr = lambda m: m.trunk.cout[m.trunk.names.index(m.trunk.attn[0])]
        models = {"G.style": (G, tr(G)), "G.up": (G, up_c), "D.fake": (D, tr(D)), "S.fake": (S, tr(S)), "D.real": (D, tr(D)),
                  "S.style": (S, tr(S)), "S.real": (S, tr(S))}
        return [(models[n][1], models[n][0].nl_gen) for n in NL_NAMES]

    def _draw_nl_host(self) -> torch.Tensor:
        host = []
        nthreads = torch.get_num_threads()
        torch.set_num_threads(1)                   # (tiny QRs: see nn.nonlocal_weights_batch)
        try:
            for C, gen in self._nl_requests():
                for shp in ((1, 1, C, C // 8), (1, 1, C, C // 8), (1, 1, C, C // 2), (1, 1, C // 2, C)):
                    host.append(nn.orthogonal(shp, gen).reshape(-1))
        finally:
            torch.set_num_threads(nthreads)
        return torch.cat(host)

    def _upload_step_state(self):
        """Before a step: fresh NonLocalBlock kernels and the four lr_t values into the static device buffers (async copies)."""
        drawn = self._draw_nl_host()
        if self._copied is not None:               # the previous step's upload must have left the pinned buffers
            self._copied.synchronize()
        self._nl_pinned.copy_(drawn)
        self._nl_dev.copy_(self._nl_pinned, non_blocking=True)
        for i, o in enumerate(self.opts):
            o.iterations += 1
            self._lr_pinned[i] = o._lr_t()
            o.iterations -= 1                      # (apply_flat advances the counter itself in eager steps; replay() does it below)
        self._lr_dev.copy_(self._lr_pinned, non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record(torch.cuda.current_stream())

    def capture(self, images, labels, my_imgs, fake_labels):
        """images / labels / my_imgs / fake_labels: device tensors whose MEMORY the graph reads on every replay (copy new batches
        into them).  Runs `warmup` eager steps (caches, kernel attributes), then captures one step."""
        dev = self.G.device
        self.inputs = (images, labels, my_imgs, fake_labels)
        n = sum(2 * C * (C // 8) + 2 * C * (C // 2) for C, _gen in self._nl_requests())
        self._copied = None
        self._nl_pinned = torch.empty(n, dtype=torch.float32, pin_memory=True)
        self._nl_dev = torch.empty(n, device=dev)
        self._lr_pinned = torch.empty(4, dtype=torch.float32, pin_memory=True)
        self._lr_dev = torch.empty(4, device=dev)
        self.nl, off = {}, 0
        for name, (C, _gen) in zip(NL_NAMES, self._nl_requests()):
            d = {}
            for k, shp in (("theta", (1, 1, C, C // 8)), ("phi", (1, 1, C, C // 8)), ("g", (1, 1, C, C // 2)), ("o", (1, 1, C // 2, C))):
                m = shp[2] * shp[3]
                d[k] = self._nl_dev[off:off + m].view(shp)
                off += m
            self.nl[name] = d
        for i, o in enumerate(self.opts):
            o.lr_dev = self._lr_dev[i:i + 1]
        for _ in range(self.warmup):               # eager steps through exactly the code path that gets captured
            self._upload_step_state()
            self._run()
        torch.cuda.synchronize()                   # (nothing is executed during capture: the static buffers need no fresh content)
        saved = [o.iterations for o in self.opts]
        self.graph = torch.cuda.CUDAGraph()
        ops.CAPTURING = True
        try:
            with torch.cuda.graph(self.graph):
                self.scalars = self._run()
        finally:
            ops.CAPTURING = False
        for o, it in zip(self.opts, saved):        # capturing queues nothing: the step has not happened
            o.iterations = it
        ops.new_step()
        ops.weights_changed()
        return self

    def _run(self):
        images, labels, my_imgs, fake_labels = self.inputs
        o = self.opts
        return DU.train_step(0, 0, 1, images, labels, self.D, self.R, self.S, self.gan, o[0], o[1], o[2], o[3], my_imgs, self.B, 128,
                             self.loss_fn, 1, self.balance, None, 10, "", fake_labels=fake_labels, nl=self.nl, verbose=False, sync=False)

    def step(self):
        """One optimisation step = upload of the step's host-drawn state + one graph launch.  -> StepScalars (lazy read-back)."""
        self._upload_step_state()
        self.graph.replay()
        for o in self.opts:
            o.iterations += 1
        ops.weights_changed()                      # (the replay rewrote the parameters behind Python's back)
        return DU.StepScalars(self.scalars, None)
