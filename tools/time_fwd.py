"""Times the fp32 inference kernel on one 32768-ray fine chunk (4.19 M points): python tools/time_fwd.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from nerf_amd import ops
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
g = torch.Generator(device=dev); g.manual_seed(0)
m = models[1]
n_rays, S = 32768, 128
pts = (torch.rand(n_rays * S, 3, device=dev, generator=g) - 0.5) * 8
vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1)
ref = ops.mlp_forward(pts, vd, m._packed(False)[0], True, 5, S)
out = ops.mlp_forward_x4(pts, vd, m._packed_x4(), True, 5, S)
print("max diff vs 32-point kernel", (ref - out).abs().max().item())
for rep in range(3):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.mlp_forward_x4(pts, vd, m._packed_x4(), True, 5, S)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{ms:.3f} ms, {n_rays * S * 1186816 / ms / 1e9:.1f} TFLOP/s", flush=True)
