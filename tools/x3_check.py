"""Split-bf16 inference kernel against the fp32 kernels and an fp64 evaluation: python tools/x3_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from nerf_amd import ops
from nerf_amd.run_nerf_helpers import NeRF
from oracle import nerf_oracle as O
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
g = torch.Generator(device=dev); g.manual_seed(0)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


m = models[1]
for n_rays, S in ((1000, 64), (777, 5), (32768, 128)):
    pts = (torch.rand(n_rays * S, 3, device=dev, generator=g) - 0.5) * 8
    vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1)
    a = ops.mlp_forward_x4(pts, vd, m._packed_x4(), True, 5, S)
    b = ops.mlp_forward_bf16x3(pts, vd, m._packed_bf16x3(), True, 5, S)
    line = f"rays={n_rays} S={S}: |x3 - fp32| max {(a - b).abs().max().item():.3e} (scale {a.abs().max().item():.3e})"
    if n_rays <= 1000:   # fp64 truth on the CPU
        sd64 = {k: v.double() for k, v in sds[1].items()}
        t = O.query_field(pts.cpu().double().reshape(n_rays, S, 3), vd.cpu().double(), sd64).reshape(-1, 4)
        line += f"; against fp64: fp32 kernel {(a.cpu().double() - t).abs().max().item():.3e}, x3 kernel {(b.cpu().double() - t).abs().max().item():.3e}"
    else:
        ta = timed(lambda: ops.mlp_forward_x4(pts, vd, m._packed_x4(), True, 5, S))
        tb = timed(lambda: ops.mlp_forward_bf16x3(pts, vd, m._packed_bf16x3(), True, 5, S))
        fl = n_rays * S * 1186816 / 1e9
        line += f"; fp32 {ta:.3f} ms ({fl / ta:.1f} TFLOP/s), x3 {tb:.3f} ms ({fl / tb:.1f} TFLOP/s algorithmic)"
    print(line, flush=True)
m2 = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
sd2 = O.random_state_dict(33, in_dirs=0, use_viewdirs=False, output_ch=5)
m2.load_state_dict(sd2); m2 = m2.to(dev)
pts = (torch.rand(5000 * 7, 3, device=dev, generator=g) - 0.5) * 8
a = ops.mlp_forward_x4(pts, None, m2._packed_x4(), False, 5, 7)
b = ops.mlp_forward_bf16x3(pts, None, m2._packed_bf16x3(), False, 5, 7)
print("no view branch: |x3 - fp32| max", (a - b).abs().max().item(), "scale", a.abs().max().item())
