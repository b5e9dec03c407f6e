"""Does a kernel run faster after the GPU has idled?  python3 tools/probe_idle_credit.py
The inference forward (4.19 M points) launched back to back, then with the device left idle for 5, 20 and 100 ms before
every launch: mean time per launch (HIP events around each launch) and the clock and power sampled meanwhile."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from nerf_amd import gpustat, ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
m = models[1]
g = torch.Generator(device=dev)
g.manual_seed(0)
n_rays, S = 32768, 128
pts = (torch.rand(n_rays * S, 3, device=dev, generator=g) - 0.5) * 8
vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1)


def run():
    return ops.mlp_forward_f16x2(pts, vd, m._packed_f16x2(), None, True, 5, S)


for _ in range(20):
    run()
torch.cuda.synchronize()
for idle_ms in (0, 5, 20, 100, 0):
    n = 60 if idle_ms < 100 else 20
    evs = []
    with gpustat.sampler_for(0, 0.02) as st:
        for _ in range(n):
            if idle_ms:
                torch.cuda.synchronize()
                time.sleep(idle_ms * 1e-3)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            run()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    s = st.summary()
    print(f"idle {idle_ms:3d} ms before every launch: mean {sum(ms) / len(ms):.3f} ms, median {ms[len(ms) // 2]:.3f}, min {ms[0]:.3f}; "
          f"sampled {s['clock_ghz_mean']} GHz {s['power_w_mean']} W", flush=True)

# how long the clock takes to come back: ten launches back to back behind one 100 ms pause, each timed
for kernel_pts in (n_rays * S, n_rays * S // 16):
    sub_pts, sub_vd = pts[:kernel_pts], vd[:kernel_pts // S]
    torch.cuda.synchronize()
    time.sleep(0.1)
    evs = []
    for _ in range(24):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.mlp_forward_f16x2(sub_pts, sub_vd, m._packed_f16x2(), None, True, 5, S)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    print(f"behind a 100 ms pause, {kernel_pts} points per launch, ms per launch:", " ".join(f"{a.elapsed_time(b):.2f}" for a, b in evs), flush=True)
