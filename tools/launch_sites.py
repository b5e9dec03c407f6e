"""Which line of Python issues which device launch in the eager train step: python3 tools/launch_sites.py [n_rand]
Three steps under torch.profiler with Python stacks; prints, for one step, every device kernel / copy / memset in launch
order with its duration and the innermost frames of this repository that led to it."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from nerf_amd.optim import FlatAdam  # noqa: E402
from nerf_amd.raybatch import ViewSampler  # noqa: E402
from nerf_amd.run_nerf import render  # noqa: E402
from nerf_amd.run_nerf_helpers import img2mse_pair  # noqa: E402

n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
H = W = 400
K = bench.lego_camera(H, W)
kw = dict(kwargs, perturb=1.0, raw_noise_std=0.0)
params = list(models[0].parameters()) + list(models[1].parameters())
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(3):
        one(5 + i)
    torch.cuda.synchronize()

events = prof.events()
by_corr = {}
for e in events:
    if e.device_type == torch.autograd.DeviceType.CPU and getattr(e, "id", None) is not None:
        pass
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for e in events:
    ks = getattr(e, "kernels", None)
    if not ks:
        continue
    stack = [s for s in (e.stack or []) if root in s and "launch_sites" not in s]
    for k in ks:
        rows.append((e.time_range.start, k.name[:70], k.duration, e.name, stack[:3]))
rows.sort()
n = len(rows) // 3
print(f"{len(rows)} launches in 3 steps; the middle step ({n} launches):")
total = 0.0
for t, name, dur, op, stack in rows[n:2 * n]:
    total += dur
    where = " <- ".join(s.replace(root + "/", "").split(", ")[0] for s in stack) if stack else ""
    print(f"{dur:7.1f} us  {name:70s} {op[:28]:28s} {where}")
print(f"sum of durations {total:.1f} us")
