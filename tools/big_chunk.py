"""One 800x800 view rendered with several chunk sizes: same bits, and the time of each (python3 tools/big_chunk.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from nerf_amd import ops
from nerf_amd.run_nerf import render
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
H = W = 800
K = bench.lego_camera(H, W)
pose = bench.orbit_pose(33.0).to(dev)
ref = None
for chunk in (1 << 15, 1 << 16, 1 << 17, 1 << 18, 640000):
    with torch.no_grad():
        render(H, W, K, chunk=chunk, c2w=pose, **kwargs)
        torch.cuda.synchronize()
        ops.PROFILE = []
        t0 = time.perf_counter()
        out = render(H, W, K, chunk=chunk, c2w=pose, **kwargs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof, ops.PROFILE = ops.PROFILE, None
        hbm = {k: (v["mean_us"], v["frac_of_hbm_peak"]) for k, v in bench.hbm_kernels(prof).items()}
    same = True if ref is None else all(bool(((a == b) | (a.isnan() & b.isnan())).all()) for a, b in zip(out[:3], ref[:3]))   # empty rays have disp = NaN, as upstream
    if ref is None:
        ref = out
    print(f"chunk {chunk:7d}: {H * W / dt / 1e3:8.1f} k rays/s  identical to chunk 32768: {same}  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB\n      streaming kernels (mean us, fraction of 8 TB/s): {hbm}", flush=True)
