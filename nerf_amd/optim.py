"""Adam for parameters that lie back to back in memory.

`FlatAdam` is torch.optim.Adam (upstream: run_nerf.py:217-218, stepped at run_nerf.py:868-881) with the same
constructor arguments, `param_groups`, `state` and `state_dict()` / `load_state_dict()` format.  When every parameter of
a step lies in a run of parameters whose storage is contiguous (a NeRF module keeps its parameters as views of one flat
vector) and whose gradients are laid out the same way (the fused MLP's backward returns views of one flat gradient, and
the data-parallel exchange reduces it in place), the whole update is ONE launch of the library's Adam kernel plus one
increment of the step counters, instead of the stock implementation's multi-tensor launches of 64 Ki elements per block
(three under-filled launches, 100 us, for the 1.2 M parameters of a coarse + fine pair).  The moments of a run are views
of one buffer per run, and so are the step counters, which the optimiser creates on first use -- or rebuilds from whatever
`load_state_dict` put in their place.  Anything else -- a parameter without a gradient, amsgrad, maximize, a layout that
is not contiguous -- takes torch.optim.Adam.step() unchanged, on the same state.

The step only enqueues work (learning rate and step count are read on the device when they live there), so it can be
recorded into a HIP graph like the stock capturable optimiser -- after one eager step: the first step on a layout moves the
moments into their buffers and reads one word back from the device.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from ._lib import check, lib


class _Run:
    """Parameters [i0, i1) of the flattened parameter list: contiguous storage, moments and step counters of their own."""
    __slots__ = ("params", "numel", "key", "exp_avg", "exp_avg_sq", "steps", "uniform")

    def __init__(self, params):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.key = None
        self.exp_avg = self.exp_avg_sq = self.steps = None
        self.uniform = True      # every parameter of the run has taken the same number of steps


def _contiguous_after(a: torch.Tensor, b: torch.Tensor) -> bool:
    return b.data_ptr() == a.data_ptr() + 4 * a.numel()


class FlatAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize,
                         fused=True, capturable=True)
        self._runs: Optional[List[List[_Run]]] = None      # per parameter group
        self._layout_key = None
        self._static = False
        self.fast_steps = 0                                # steps taken through the library kernel (for tests and logs)
        self._steps_all = {}                               # device -> [buffer of every run's step counters, words handed out]

    # ---- layout ------------------------------------------------------------------------------------
    @staticmethod
    def _static_ok(group) -> bool:
        """What does not change from step to step: the update rule and the parameters' type and placement."""
        if group.get("amsgrad") or group.get("maximize") or group.get("differentiable") or not group["params"]:
            return False
        return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in group["params"])

    def _find_runs(self):
        """Cuts every group's parameter list where the storage stops being contiguous."""
        runs = []
        for group in self.param_groups:
            out, cur = [], []
            for p in group["params"]:
                if cur and not _contiguous_after(cur[-1], p):
                    out.append(_Run(cur))
                    cur = []
                cur.append(p)
            if cur:
                out.append(_Run(cur))
            runs.append(out)
        return runs

    def _adopt_state(self, run: _Run) -> None:
        """Moments and step counters of a run as views of one buffer each, keeping the values the state holds already
        (a fresh optimiser: zeros; after load_state_dict: the loaded tensors)."""
        dev = run.params[0].device
        m = torch.zeros(run.numel, dtype=torch.float32, device=dev)
        v = torch.zeros(run.numel, dtype=torch.float32, device=dev)
        steps = self._step_buffer(run, dev)
        off = 0
        for i, p in enumerate(run.params):
            st = self.state[p]
            n = p.numel()
            if "exp_avg" in st:
                m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps[i].copy_(torch.as_tensor(st["step"], dtype=torch.float32))
            st["exp_avg"] = m[off:off + n].view_as(p)
            st["exp_avg_sq"] = v[off:off + n].view_as(p)
            st["step"] = steps[i]
            off += n
        run.exp_avg, run.exp_avg_sq, run.steps = m, v, steps
        run.key = tuple(id(self.state[p]["exp_avg"]) for p in run.params)
        run.uniform = bool(steps.min() == steps.max())     # (a host read: adoption happens once, outside any recording)

    def _step_buffer(self, run: _Run, dev) -> torch.Tensor:
        """The step counters of a run: a slice of ONE buffer for every run on this device, so that a step increments
        them all with one launch."""
        n_all = sum(len(r.params) for runs in self._runs for r in runs)
        buf = self._steps_all.get(dev)
        if buf is None or buf[0].numel() != n_all:
            buf = self._steps_all[dev] = [torch.zeros(n_all, dtype=torch.float32, device=dev), 0]
        first = buf[1]
        if first + len(run.params) > n_all:      # a layout adopted anew: start over (the values are copied in by the caller)
            buf = self._steps_all[dev] = [torch.zeros(n_all, dtype=torch.float32, device=dev), 0]
            first = 0
        buf[1] = first + len(run.params)
        return buf[0][first:first + len(run.params)]

    @staticmethod
    def _increment(runs) -> None:
        """steps += 1 for every run: one launch per stretch of runs whose counters are adjacent slices of one buffer."""
        i = 0
        while i < len(runs):
            j, n = i, runs[i].steps.numel()
            while (j + 1 < len(runs) and runs[j + 1].steps.data_ptr() == runs[i].steps.data_ptr() + 4 * n
                   and runs[j + 1].steps.untyped_storage().data_ptr() == runs[i].steps.untyped_storage().data_ptr()):
                j += 1
                n += runs[j].steps.numel()
            (runs[i].steps if j == i else runs[i].steps.as_strided((n,), (1,))).add_(1.0)
            i = j + 1

    def _state_intact(self, run: _Run) -> bool:
        if run.key is None:
            return False
        state = self.state
        for p, k in zip(run.params, run.key):
            st = state.get(p)
            if st is None or id(st.get("exp_avg")) != k:
                return False
        return True

    # ---- the step ----------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        key = tuple(p.data_ptr() for g in self.param_groups for p in g["params"])
        if key != self._layout_key:
            self._runs, self._layout_key = self._find_runs(), key
            self._static = all(self._static_ok(g) for g in self.param_groups)
        max_runs = lib.nerf_adam_max_runs()
        todo = []     # (group, runs of it whose gradients are laid out like the parameters)
        for group, runs in zip(self.param_groups, self._runs):
            for run in runs:
                grads = [p.grad for p in run.params]
                g0 = grads[0]
                if (not self._static or any(g is None for g in grads) or g0.dtype != torch.float32 or g0.is_sparse
                        or g0.device != run.params[0].device or not grads[-1].is_contiguous()
                        or any(not _contiguous_after(a, b) for a, b in zip(grads, grads[1:]))):
                    super().step()      # a gradient layout the kernel cannot walk: the stock step, on the same state
                    return loss
            todo.append((group, runs))
        for group, runs in todo:
            for run in runs:
                if not self._state_intact(run):
                    self._adopt_state(run)
        if not all(run.uniform for _, runs in todo for run in runs):
            super().step()              # parameters of one run at different step counts: the stock step keeps them apart
            return loss
        for group, runs in todo:
            self._increment(runs)
            beta1, beta2 = group["betas"]
            lr = group["lr"]
            lr_dev = lr.data_ptr() if torch.is_tensor(lr) and lr.is_cuda else None
            lr_val = 0.0 if lr_dev is not None else float(lr)
            if torch.is_tensor(lr) and lr_dev is not None and lr.dtype != torch.float32:
                raise TypeError("FlatAdam: a device learning rate must be a float32 tensor")
            dev = runs[0].params[0].device
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream().cuda_stream
                for a in range(0, len(runs), max_runs):
                    part = runs[a:a + max_runs]
                    n = len(part)
                    counts = (ctypes.c_int64 * n)(*[r.numel for r in part])
                    arr = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)      # noqa: E731
                    check(lib.nerf_adam_step(n, counts, arr([r.params[0].data_ptr() for r in part]),
                                             arr([r.params[0].grad.data_ptr() for r in part]),
                                             arr([r.exp_avg.data_ptr() for r in part]), arr([r.exp_avg_sq.data_ptr() for r in part]),
                                             arr([r.steps.data_ptr() for r in part]), 0.0, lr_dev, lr_val, float(beta1), float(beta2),
                                             float(group["eps"]), float(group["weight_decay"]), stream))
        self.fast_steps += 1
        return loss

    def load_state_dict(self, state_dict):
        """As torch.optim.Adam, for checkpoints of either optimiser (and of upstream's); the groups keep this class's
        fused / capturable settings whatever the checkpoint's were, so that the stock step stays usable on the state."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            group["fused"], group["capturable"], group["foreach"] = True, True, None
            for p in group["params"]:
                st = self.state.get(p)
                if st and "step" in st and not (torch.is_tensor(st["step"]) and st["step"].is_cuda):
                    st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=p.device)
        self._layout_key = None
