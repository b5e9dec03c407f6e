"""A long optimisation on every arithmetic the training step can run on, from the same start and the same batches:
python3 tools/soak_train.py [steps] [out.json]

A teacher pair of networks renders eight 100x100 views; a student pair with other weights is trained on them with the
upstream loop (run_nerf.py:840-881: 1024 random rays of one view per step, 64 + 128 samples, jittered depths, both
losses, Adam 5e-4 with the upstream decay) for `steps` steps, once per arithmetic:
  f16x2    the default: two fp16 pieces per operand in all three MLP stages (marked tiles on three pieces)
  bf16x3   three bf16 pieces per operand in all three stages
  f32_mfma the fp32 matrix instruction
Recorded per arithmetic: the loss averaged over windows of 250 steps, PSNR of a held-out view at the end, the share of
128-point tiles the two-piece stages handed to the three-piece kernels, steps per second.  The three curves must lie on
top of each other up to the divergence any two fp32 summation orders show (trajectories are chaotic; what is compared is
the level, not the bits)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_amd import ops  # noqa: E402
from nerf_amd.datasets import pose_spherical  # noqa: E402
from nerf_amd.optim import FlatAdam  # noqa: E402
from nerf_amd.raybatch import ViewSampler  # noqa: E402
from nerf_amd.run_nerf import render, run_network  # noqa: E402
from nerf_amd.run_nerf_helpers import NeRF, get_embedder, img2mse, img2mse_pair, mse2psnr  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/soak/soak_train.json"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
e, _ = get_embedder(10, 0)
ed, _ = get_embedder(4, 0)
qf = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=e, embeddirs_fn=ed, netchunk=1 << 16)  # noqa: E731


def pair(seed, gain, colour):
    nets = []
    for k in range(2):
        torch.manual_seed(seed + k)
        m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        with torch.no_grad():
            m.alpha_linear.weight.mul_(gain)
            m.alpha_linear.bias.mul_(gain).add_(1.0 if gain > 1 else 0.0)
            m.rgb_linear.weight.mul_(colour)
        nets.append(m.to(dev))
    return nets


H = W = 100
focal = 0.5 * W / np.tan(0.5 * np.radians(50.0))
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
common = dict(network_query_fn=qf, N_samples=64, N_importance=128, use_viewdirs=True, white_bkgd=True, ndc=False, near=2.0, far=6.0)
teacher = pair(100, 6.0, 8.0)
angles = (0.0, 40.0, 85.0, 130.0, 175.0, 220.0, 265.0, 310.0, 350.0)
poses = torch.stack([pose_spherical(th, -30.0, 4.0) for th in angles], 0).to(dev)
with torch.no_grad():
    images = torch.stack([render(H, W, K, chunk=1 << 15, c2w=p[:3, :4], network_fn=teacher[0], network_fine=teacher[1],
                                 perturb=0.0, raw_noise_std=0.0, **common)[0] for p in poses], 0)
print("teacher images: mean %.3f std %.3f" % (float(images.mean()), float(images.std())), flush=True)
train_views, held_out = list(range(8)), 8
result = {"steps": steps, "n_rand": 1024, "samples": "64 + 128", "views": "8 of 100x100, one held out", "arithmetics": {}}
for kind in ("f16x2", "bf16x3", "f32_mfma"):
    student = pair(200, 1.0, 1.0)
    for m in student:
        m.mlp_dtype = "f32" if kind == "f16x2" else kind
    params = list(student[0].parameters()) + list(student[1].parameters())
    opt = FlatAdam(params=params, lr=5e-4, betas=(0.9, 0.999))
    sampler = ViewSampler(images, poses, train_views, H, W, K, 1024, device=dev, seed=7)
    torch.manual_seed(11)                 # the jitter of the depth samples: the same stream for every arithmetic
    window, curve, marked = [], [], {"forward": [0, 0], "backward_data": [0, 0]}
    ops.MARKED = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        rays, target = sampler.next(i)
        rgb, _, _, ex = render(H, W, K, chunk=1 << 15, rays=rays, retraw=True, network_fn=student[0], network_fine=student[1],
                               perturb=1.0, raw_noise_std=0.0, **common)
        opt.zero_grad()
        loss = img2mse_pair(rgb, ex["rgb0"], target)[2]
        loss.backward()
        opt.step()
        for g in opt.param_groups:
            g["lr"] = 5e-4 * (0.1 ** (i / 250000.0))
        window.append(loss.detach())
        if (i + 1) % 250 == 0:
            curve.append(round(float(torch.stack(window).mean()), 6))
            window = []
            for stage, f in ops.MARKED:
                marked[stage][0] += int(f.sum())
                marked[stage][1] += f.numel()
            ops.MARKED = []
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.MARKED = None
    with torch.no_grad():
        got = render(H, W, K, chunk=1 << 15, c2w=poses[held_out][:3, :4], network_fn=student[0], network_fine=student[1],
                     perturb=0.0, raw_noise_std=0.0, **common)[0]
        psnr_held = float(mse2psnr(img2mse(got, images[held_out])))
        got = render(H, W, K, chunk=1 << 15, c2w=poses[0][:3, :4], network_fn=student[0], network_fine=student[1],
                     perturb=0.0, raw_noise_std=0.0, **common)[0]
        psnr_train = float(mse2psnr(img2mse(got, images[0])))
    result["arithmetics"][kind] = {"loss_per_250_steps": curve, "psnr_held_out_view": round(psnr_held, 3),
                                   "psnr_train_view": round(psnr_train, 3), "steps_per_s": round(steps / dt, 1),
                                   "marked_tiles": {k: "%d of %d" % tuple(v) for k, v in marked.items()},
                                   "fast_adam_steps": opt.fast_steps}
    print(kind, "loss", curve[0], "->", curve[-1], "PSNR held-out %.2f train %.2f" % (psnr_held, psnr_train),
          "%.1f steps/s" % (steps / dt), result["arithmetics"][kind]["marked_tiles"], flush=True)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
with open(out_path, "w") as fh:
    json.dump(result, fh, indent=1)
