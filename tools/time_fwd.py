"""Times the forward kernels on one 32768-ray fine chunk (4.19 M points): python3 tools/time_fwd.py [x2|x2only|x3|x4|save] [reps]
(NERF_HIP_LIB selects a variant build).  x2: the two-piece fp16 inference kernel followed by the three-piece kernel on
the tiles it marked (default of rendering); x2only: its first launch alone; x3: the three-piece bf16 kernel; x4: the fp32
matrix instruction; save: the three-piece training forward that stores activations (2.1 M points)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from nerf_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
g = torch.Generator(device=dev); g.manual_seed(0)
m = models[1]
n_rays, S = (32768, 128) if which != "save" else (16384, 128)
pts = (torch.rand(n_rays * S, 3, device=dev, generator=g) - 0.5) * 8
vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1)
act = ops.mlp_act_buffer(n_rays * S, True, dev) if which == "save" else None
ref = ops.mlp_forward_x4(pts, vd, m._packed_x4(), True, 5, S)
def run():
    if which == "x4":
        return ops.mlp_forward_x4(pts, vd, m._packed_x4(), True, 5, S)
    if which in ("x2", "x2only"):
        return ops.mlp_forward_f16x2(pts, vd, m._packed_f16x2(), m._packed_bf16x3() if which == "x2" else None, True, 5, S)
    return ops.mlp_forward_bf16x3(pts, vd, m._packed_bf16x3(), True, 5, S, act)
out = run()
if which.startswith("x2"):
    print("marked tiles:", int(ops.mlp_forward_f16x2(pts, vd, m._packed_f16x2(), None, True, 5, S, want_flags=True)[1].sum()), "of", n_rays * S // 128)
print("max diff vs the fp32 matrix instruction", (ref - out).abs().max().item(), "checksum", float(out.double().sum()))
from nerf_amd import gpustat
for rep in range(3):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with gpustat.sampler_for(0, 0.02) as stat:
        a.record()
        for _ in range(reps if rep < 2 else 8 * reps):     # the last block is long enough for the power reading to settle
            run()
        b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / (reps if rep < 2 else 8 * reps)
    st = stat.summary()
    print(f"{os.environ.get('NERF_HIP_LIB', 'product')} {which}: {ms:.3f} ms, {n_rays * S * 1186816 / ms / 1e9:.1f} TFLOP/s algorithmic, "
          f"{st['clock_ghz_mean']} GHz {st['power_w_mean']} W ({st['samples']} samples)", flush=True)
