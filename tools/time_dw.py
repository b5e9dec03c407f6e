"""Times the weight-gradient stage alone on random operands: python3 tools/time_dw.py [n_pts] [reps] [x3|x2]
(NERF_HIP_LIB selects a variant build of the library).  Prints ms per call and algorithmic TFLOP/s / TB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_amd._lib import check, lib  # noqa: E402

n_pts = int(sys.argv[1]) if len(sys.argv) > 1 else 8192 * 192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "x3"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
g = torch.Generator(device=dev).manual_seed(0)
act = torch.rand(lib.nerf_mlp_act_floats(1, n_pts), device=dev, generator=g) - 0.3
act.clamp_(min=0)                                       # like post-ReLU activations: about a third zeros
scr = torch.randn(lib.nerf_mlp_scratch_floats(1, n_pts), device=dev, generator=g)
d_params = torch.empty(lib.nerf_mlp_n_params(1, 5), device=dev)
s = torch.cuda.current_stream().cuda_stream


n_tiles = (n_pts + 127) // 128
flags = torch.zeros(n_tiles, dtype=torch.int32, device=dev)
tile_max = torch.full((n_tiles, 18), 127 + 3 + 1, dtype=torch.int32, device=dev)      # every group: magnitudes below 2^3
ctl = torch.zeros(4096, dtype=torch.int32, device=dev)


def call():
    if which == "x2":
        check(lib.nerf_mlp_bwd_weights_f16x2(n_pts, act.data_ptr(), 1, 5, scr.data_ptr(), d_params.data_ptr(), None, flags.data_ptr(),
                                             tile_max.data_ptr(), ctl.data_ptr(), s))
    else:
        check(lib.nerf_mlp_bwd_weights_bf16x3(n_pts, act.data_ptr(), 1, 5, scr.data_ptr(), d_params.data_ptr(), s))


for _ in range(3):
    call()
torch.cuda.synchronize()
from nerf_amd import gpustat  # noqa: E402
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
with gpustat.sampler_for(0, 0.02) as stat:
    ev[0].record()
    for _ in range(reps):
        call()
    ev[1].record()
    torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
st = stat.summary()
nbytes = n_pts * 4 * (2528 + 2432 + 12)
print(f"{which} {os.environ.get('NERF_HIP_LIB', 'product'):40s} n_pts {n_pts}: {ms:.3f} ms  {n_pts * 1186816 / ms / 1e9:.1f} TFLOP/s alg  "
      f"{nbytes / ms / 1e9:.2f} TB/s  checksum {float(d_params.double().abs().sum()):.6e}  "
      f"{st['clock_ghz_mean']} GHz {st['power_w_mean']} W ({st['samples']} samples)")
