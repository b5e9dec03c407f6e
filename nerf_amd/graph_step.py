"""The optimisation step (run_nerf.py:856-883) recorded into HIP graphs.

At the upstream batch size (N_rand = 1024) a step is about 41 launches, of which six (the MLP kernels) carry 94 % of
the GPU time; the rest are microsecond kernels.  Eager, the launching thread needs about half the step's 4.6 ms for them,
which is fine until the host's cores are contended and the thread falls behind the GPU.  The whole
step -- render with gradients, the two MSE terms, backward, Adam -- only enqueues work on the current stream, so it can
be captured once and replayed:

    graph 1   zero the gradients, render(rays), loss, backward
    (eager)   gradient all-reduce when several ranks train data-parallel
    graph 2   optimizer.step()

Inputs live in fixed buffers (`rays`, `target`) that `step` overwrites before each replay; the loss and PSNR of the
last step are left in `loss` / `psnr` (device scalars, read them with float() when needed, not every step).
The optimizer must be capturable (nerf_amd.optim.FlatAdam, or torch.optim.Adam(..., capturable=True)), which also lets
the learning rate be a device scalar that `set_lr` updates without re-recording.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import torch

from .run_nerf_helpers import NeRF, img2mse, img2mse_pair


class GraphedTrainStep:
    def __init__(self, H: int, W: int, K, render_kwargs: dict, params: Sequence[torch.Tensor],
                 optimizer: torch.optim.Optimizer, n_rand: int, chunk: int = 1024 * 32, eager_steps: int = 3,
                 between: Optional[Callable[[], None]] = None, render_fn: Optional[Callable] = None):
        """params: the tensors `optimizer` updates; eager_steps: steps run normally before recording (the allocator
        and every lazily built table must be warm); between: called between backward and optimizer.step (the
        data-parallel gradient all-reduce)."""
        if render_fn is None:
            from .run_nerf import render as render_fn
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise ValueError("GraphedTrainStep needs a capturable optimizer: torch.optim.Adam(..., capturable=True)")
        self.H, self.W, self.K, self.chunk = int(H), int(W), K, int(chunk)
        self.kwargs = dict(render_kwargs)
        self.params = list(params)
        self.opt = optimizer
        self.between = between
        self.render_fn = render_fn
        dev = self.params[0].device
        self.dev = dev
        self.rays = torch.zeros(2, n_rand, 3, device=dev)
        self.target = torch.zeros(n_rand, 3, device=dev)
        self.loss = torch.zeros((), device=dev)
        self.psnr = torch.zeros((), device=dev)
        self.eager_left = int(eager_steps)
        self.g_grad = None
        self.g_opt = None
        self.models = [m for m in (self.kwargs.get("network_fn"), self.kwargs.get("network_fine")) if isinstance(m, NeRF)]
        self.stream = torch.cuda.Stream(device=dev)
        for group in optimizer.param_groups:       # a device scalar, so that the recorded step reads the current value
            if not torch.is_tensor(group["lr"]):
                group["lr"] = torch.tensor(float(group["lr"]), device=dev)

    # ---- the step, as upstream writes it
    def _grad_part(self):
        self.opt.zero_grad(set_to_none=True)
        rgb, disp, acc, extras = self.render_fn(self.H, self.W, self.K, chunk=self.chunk, rays=self.rays, verbose=False,
                                                retraw=True, **self.kwargs)
        if 'rgb0' in extras:
            img_loss, _, loss = img2mse_pair(rgb, extras['rgb0'], self.target)
        else:
            loss = img_loss = img2mse(rgb, self.target)
        loss.backward()
        self.loss.copy_(loss.detach())
        # mse2psnr with the constant folded on the host (a host-built tensor cannot be uploaded while recording)
        self.psnr.copy_(torch.log(img_loss.detach()) * (-10.0 / math.log(10.0)))

    def _opt_part(self):
        self.opt.step()

    def _invalidate_packs(self):
        # the recorded step re-packs the weight streams into the same buffers on every replay; the modules' host-side
        # cache key cannot see that, so an eager render after a replay must not trust it
        for m in self.models:
            m.invalidate_packs()

    def set_lr(self, value: float) -> None:
        for group in self.opt.param_groups:
            group["lr"].fill_(float(value))

    def step(self, batch_rays: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """One optimisation step on batch_rays [2, N_rand, 3] and target [N_rand, 3]; returns the (device) loss."""
        cur = torch.cuda.current_stream(self.dev)
        self.rays.copy_(batch_rays)
        self.target.copy_(target)
        if self.g_grad is None:
            # eager steps and the recording run on a side stream, as graph capture requires
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                if self.eager_left > 0:
                    self.eager_left -= 1
                    self._grad_part()
                    if self.between is not None:
                        self.between()
                    self._opt_part()
                else:
                    self.opt.zero_grad(set_to_none=True)
                    self.g_grad = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.g_grad, stream=self.stream):
                        self._grad_part()
                    self.g_opt = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.g_opt, stream=self.stream):
                        self._opt_part()
            cur.wait_stream(self.stream)
            if self.g_grad is None:
                self._invalidate_packs()
                return self.loss
        self.g_grad.replay()
        if self.between is not None:
            self.between()
        self.g_opt.replay()
        self._invalidate_packs()
        return self.loss
