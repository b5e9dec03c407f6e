import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from nerf_amd import ops
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
g = torch.Generator(device=dev); g.manual_seed(0)
for m in models:
    for n_rays, S in ((1000, 64), (777, 5), (32768, 192)):
        pts = (torch.rand(n_rays * S, 3, device=dev, generator=g) - 0.5) * 8
        vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1)
        a = ops.mlp_forward(pts, vd, m._packed(False)[0], True, 5, S)
        b = ops.mlp_forward_x4(pts, vd, m._packed_x4(), True, 5, S)
        d = (a - b).abs()
        print(n_rays, S, "max diff", d.max().item(), "scale", a.abs().max().item(), flush=True)
# no view branch
from nerf_amd.run_nerf_helpers import NeRF
from oracle import nerf_oracle as O
m2 = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
m2.load_state_dict(O.random_state_dict(33, in_dirs=0, use_viewdirs=False, output_ch=5)); m2 = m2.to(dev)
pts = (torch.rand(5000 * 7, 3, device=dev, generator=g) - 0.5) * 8
a = ops.mlp_forward(pts, None, m2._packed(False)[0], False, 5, 7)
b = ops.mlp_forward_x4(pts, None, m2._packed_x4(), False, 5, 7)
print("no views max diff", (a - b).abs().max().item(), "scale", a.abs().max().item())
