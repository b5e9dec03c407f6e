"""Times the backward-data stage alone: python3 tools/time_bwd.py [x2|x2only|x3] [n_rays] [samples] [launches]
x2 = two-piece kernel + the (empty) launch over marked tiles, x2only = without it, x3 = three-piece kernel."""
import sys
import os
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import ops, gpustat
from nerf_amd._lib import lib, check
from nerf_amd.run_nerf_helpers import NeRF

which = sys.argv[1] if len(sys.argv) > 1 else "x2"
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
S = int(sys.argv[3]) if len(sys.argv) > 3 else 192
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True).to(dev)
n = n_rays * S
pts = torch.rand(n, 3, device=dev) * 4 - 2
vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
act = ops.mlp_act_buffer(n, True, dev)
x3 = (m._packed_bf16x3(with_bwd=True), m._packed_bwd_bf16x3())
x2 = (m._packed_f16x2(with_bwd=True), m._packed_bwd_f16x2())
with torch.no_grad():
    ops.mlp_forward_f16x2(pts, vd, x2[0], x3[0], True, 5, S, act)
G = torch.randn(n, 4, device=dev) * 1e-4
scratch = torch.empty(lib.nerf_mlp_scratch_floats(1, n), dtype=torch.float32, device=dev)
flags = torch.zeros((n + 127) // 128, dtype=torch.int32, device=dev)
tile_max = torch.zeros(((n + 127) // 128, 18), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    if which == "x3":
        check(lib.nerf_mlp_bwd_data_bf16x3(n, G.data_ptr(), act.data_ptr(), 1, 5, x3[0].data_ptr(), x3[1].data_ptr(), scratch.data_ptr(), st))
    else:
        fb = (x3[0].data_ptr(), x3[1].data_ptr()) if which == "x2" else (None, None)
        check(lib.nerf_mlp_bwd_data_f16x2(n, G.data_ptr(), act.data_ptr(), 1, 5, x2[0].data_ptr(), x2[1].data_ptr(), fb[0], fb[1],
                                          scratch.data_ptr(), flags.data_ptr(), tile_max.data_ptr() if os.environ.get("NO_TILE_MAX") is None else None, st))


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with gpustat.sampler_for(dev) as smp:
    a.record()
    for _ in range(reps):
        run()
    b.record()
    torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
print(f"{which}: {ms:.3f} ms per launch over {n} points = {1115392 * n / ms / 1e9:.1f} TFLOP/s algorithmic; marked tiles {int(flags.sum())};",
      smp.summary())
