"""Ray batches for the optimisation loop, kept on the device (upstream: run_nerf.py:758-854).

Two samplers with the upstream semantics:

* RayShuffler  -- `use_batching`: every pixel of every training view is one (origin, direction, colour)
  record; the records are shuffled once, consumed N_rand at a time, and reshuffled after an epoch.
* ViewSampler  -- `no_batching`: each step takes one training view, an optional central crop during the
  first `precrop_iters` steps, and N_rand distinct pixels of it.  Rays and target colours of exactly those
  pixels come from one kernel (nerf_rays_at); the full H x W ray image is never built.

With several processes (one per GPU) every rank draws from its own generator seed, or, for the shuffler,
takes its own slice of each global batch of world_size * N_rand records of the common permutation.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .run_nerf_helpers import get_rays_np


class RayShuffler:
    def __init__(self, images, poses, i_train, H, W, K, N_rand, device=None, seed=0, rank=0, world_size=1):
        """images [N,H,W,3] (numpy or tensor, values in [0,1]), poses [N,3,4+]; i_train: indices of the training views."""
        dev = device if device is not None else ops.device()
        poses = np.asarray(poses.cpu() if torch.is_tensor(poses) else poses)
        images = np.asarray(images.cpu() if torch.is_tensor(images) else images)
        # the upstream builds the rays in float64 on the host and rounds once (run_nerf.py:765-776); start-up only
        recs = []
        for i in i_train:
            ro, rd = get_rays_np(H, W, K, poses[i, :3, :4])
            recs.append(np.stack([ro, rd, images[i][..., :3]], -2).reshape(-1, 3, 3).astype(np.float32))
        self.rays_rgb = torch.from_numpy(np.concatenate(recs, 0)).to(dev)       # [sum HW, ro+rd+rgb, 3]
        self.N_rand, self.rank, self.world = int(N_rand), int(rank), int(world_size)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))                                         # common to all ranks
        self.i_batch = 0
        self.epochs = 0
        self._shuffle()

    def _shuffle(self):
        perm = torch.randperm(self.rays_rgb.shape[0], device=self.rays_rgb.device, generator=self.gen)
        self.rays_rgb = self.rays_rgb[perm]
        self.i_batch = 0

    def next(self):
        """-> batch_rays [2, N_rand, 3], target_s [N_rand, 3]."""
        lo = self.i_batch + self.rank * self.N_rand
        batch = self.rays_rgb[lo:lo + self.N_rand]
        batch = torch.transpose(batch, 0, 1)
        batch_rays, target_s = batch[:2], batch[2]
        self.i_batch += self.N_rand * self.world
        if self.i_batch + self.N_rand * self.world > self.rays_rgb.shape[0]:
            print("Shuffle data after an epoch!")
            self.epochs += 1
            self._shuffle()
        return batch_rays, target_s


class ViewSampler:
    def __init__(self, images, poses, i_train, H, W, K, N_rand, precrop_iters=0, precrop_frac=.5, device=None, seed=0):
        dev = device if device is not None else ops.device()
        self.images = ops.to_dev(torch.as_tensor(np.asarray(images) if not torch.is_tensor(images) else images)[..., :3])
        self.poses = ops.to_dev(torch.as_tensor(np.asarray(poses) if not torch.is_tensor(poses) else poses)[:, :3, :4])
        self.i_train = np.asarray(i_train)
        self.H, self.W, self.K = int(H), int(W), K
        self.N_rand, self.precrop_iters, self.precrop_frac = int(N_rand), int(precrop_iters), float(precrop_frac)
        self.host_rng = np.random.RandomState(int(seed))
        self.dev = dev

    def window(self, i):
        """(row0, col0, rows, cols) of the pixels iteration i may draw from (run_nerf.py:835-846)."""
        H, W = self.H, self.W
        if i < self.precrop_iters:
            dH = int(H // 2 * self.precrop_frac)
            dW = int(W // 2 * self.precrop_frac)
            return H // 2 - dH, W // 2 - dW, 2 * dH, 2 * dW
        return 0, 0, H, W

    def next(self, i, img_i=None):
        """-> batch_rays [2, N_rand, 3], target_s [N_rand, 3] for iteration i."""
        if img_i is None:
            img_i = int(self.host_rng.choice(self.i_train))
        win = self.window(i)
        n_pix = win[2] * win[3]
        if self.N_rand > n_pix:
            raise ValueError(f"N_rand={self.N_rand} exceeds the {n_pix} pixels of the sampling window")
        K = self.K
        # distinct pixels from a keyed permutation evaluated inside the ray kernel (one launch; a device randperm of
        # the window is a sort of a dozen launches); the key comes from this sampler's host generator
        seed = int(self.host_rng.randint(0, 2 ** 31 - 1)) * (2 ** 31) + int(self.host_rng.randint(0, 2 ** 31 - 1))
        rays, target, _ = ops.rays_random(self.N_rand, seed, self.H, self.W, win, float(K[0][0]), float(K[1][1]),
                                          float(K[0][2]), float(K[1][2]), self.poses[img_i], self.images[img_i])
        return rays, target
