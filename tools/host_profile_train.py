"""Where the HOST time of the eager 1024-ray train step goes: python3 tools/host_profile_train.py [n_rand] [steps]
cProfile over `steps` steps of bench.py's training leg (the GPU runs asynchronously; what is listed is Python and
runtime-call time on the launching thread)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from nerf_amd.raybatch import ViewSampler  # noqa: E402
from nerf_amd.run_nerf import render  # noqa: E402
from nerf_amd.run_nerf_helpers import img2mse_pair  # noqa: E402

n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
H = W = 400
K = bench.lego_camera(H, W)
kw = dict(kwargs, perturb=1.0, raw_noise_std=0.0)
params = list(models[0].parameters()) + list(models[1].parameters())
from nerf_amd.optim import FlatAdam  # noqa: E402
opt = FlatAdam(params=params, lr=5e-4, betas=(0.9, 0.999))
poses = torch.stack([bench.orbit_pose(t) for t in (0.0, 45.0, 90.0, 135.0)], 0)
images = torch.rand(4, H, W, 3, device=dev)
sampler = ViewSampler(images, poses, list(range(4)), H, W, K, n_rand, device=dev, seed=1)


def one(i):
    rays, target = sampler.next(i, img_i=i % 4)
    rgb, disp, acc, extras = render(H, W, K, chunk=1024 * 32, rays=rays, verbose=False, retraw=True, **kw)
    opt.zero_grad()
    loss = img2mse_pair(rgb, extras['rgb0'], target)[2]
    loss.backward()
    opt.step()
    return loss


for i in range(5):
    one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    one(i)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{steps} steps: host issue {1e3 * t_issue / steps:.3f} ms per step, wall {1e3 * t_all / steps:.3f} ms per step")
prof = cProfile.Profile()
prof.enable()
for i in range(steps):
    one(i)
prof.disable()
torch.cuda.synchronize()
pstats.Stats(prof).sort_stats("tottime").print_stats(28)
