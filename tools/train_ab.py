"""Train step with the backward-data stage on the fp32 matrix instruction (16-point tiles) and on the split-operand kernel,
alternating in one process on one box:
python tools/train_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from nerf_amd import ops

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
for rep in range(2):
    for x4 in (False, True):
        ops.BWD_X3 = x4
        for n_rand, steps in ((8192, 10), (1024, 40)):
            r = bench.time_train(models, kwargs, dev, 1, n_rand, steps, 3, 1)
            print(f"bwd_split={int(x4)} n_rand={n_rand}: {r['rays_per_s']} rays/s, {r['ms_per_step']} ms/step, "
                  f"mlp {r['mlp_tflops_algorithmic']} TFLOP/s", flush=True)
