"""Runs only train steps (python3 tools/trace_train.py N_RAND STEPS) so that a kernel trace of the process is a trace of the step."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
print(bench.time_train(models, kwargs, dev, 1, n_rand, steps, 3, seed=1))
