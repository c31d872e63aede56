"""tf.keras.optimizers.Adam / RMSprop stand-ins (/root/reference/src/main.py:27-33) over the fused
HIP update kernels.  `apply_flat(store)` is the fast path (one launch over a network's flat buffer);
`apply_gradients(zip(grads, vars))` keeps the Keras call shape of data_utils.py:451-468."""
from __future__ import annotations

import math

import torch

from . import ops


class Adam:
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self._slots = {}
        self.lr_dev = None          # 1-element device tensor: when set, the kernels read lr_t from it (graph-captured steps)

    def _lr_t(self):
        t = self.iterations
        return self.learning_rate * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def _slot(self, var):
        key = (var.data_ptr(), var.numel())
        if key not in self._slots:
            self._slots[key] = (torch.zeros(var.numel(), device=var.device), torch.zeros(var.numel(), device=var.device))
        return self._slots[key]

    def apply_flat(self, store):
        self.iterations += 1
        m, v = self._slot(store.flat)
        ops.adam_update(store.flat, store.grad, m, v, self._lr_t() if self.lr_dev is None else self.lr_dev, self.beta_1, self.beta_2, self.epsilon)

    def apply_gradients(self, grads_and_vars):
        self.iterations += 1
        lr_t = self._lr_t()
        for g, var in grads_and_vars:
            if g is None:                      # Keras skips variables without a gradient
                continue
            m, v = self._slot(var)
            ops.adam_update(var, g.contiguous(), m, v, lr_t, self.beta_1, self.beta_2, self.epsilon)

    def state_dict(self):
        return {"iterations": self.iterations, "slots": {str(i): (m.cpu(), v.cpu()) for i, (m, v) in enumerate(self._slots.values())}}

    def flat_state(self, store):
        """{name: tensor} of this optimizer's state for `store`'s flat buffer (full-state checkpoints)."""
        m, v = self._slot(store.flat)
        return {"m": m, "v": v, "iterations": torch.tensor([self.iterations], dtype=torch.int64)}

    def load_flat_state(self, store, state):
        m, v = self._slot(store.flat)
        m.copy_(state["m"].to(m.device))
        v.copy_(state["v"].to(v.device))
        self.iterations = int(state["iterations"].reshape(-1)[0].item())


class RMSprop:
    def __init__(self, learning_rate=0.001, rho=0.9, epsilon=1e-7):
        self.learning_rate, self.rho, self.epsilon = float(learning_rate), float(rho), float(epsilon)
        self.iterations = 0
        self._slots = {}

    def _slot(self, var):
        key = (var.data_ptr(), var.numel())
        if key not in self._slots:
            self._slots[key] = torch.zeros(var.numel(), device=var.device)
        return self._slots[key]

    def apply_flat(self, store):
        self.iterations += 1
        ops.rmsprop_update(store.flat, store.grad, self._slot(store.flat), self.learning_rate, self.rho, self.epsilon)

    def flat_state(self, store):
        return {"ms": self._slot(store.flat), "iterations": torch.tensor([self.iterations], dtype=torch.int64)}

    def load_flat_state(self, store, state):
        ms = self._slot(store.flat)
        ms.copy_(state["ms"].to(ms.device))
        self.iterations = int(state["iterations"].reshape(-1)[0].item())

    def apply_gradients(self, grads_and_vars):
        self.iterations += 1
        for g, var in grads_and_vars:
            if g is None:
                continue
            ops.rmsprop_update(var, g.contiguous(), self._slot(var), self.learning_rate, self.rho, self.epsilon)
