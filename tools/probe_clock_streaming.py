"""What shader clock and board power the box holds under a pure HBM stream (a 4 GiB device-to-device copy, repeated) and
under the library's kernels: python3 tools/probe_clock_streaming.py.  For reading the clocks other tools report."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_amd import gpustat  # noqa: E402

dev = torch.device("cuda", 0)
a = torch.empty(1 << 30, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
for name, fn in (("copy 4 GiB", lambda: b.copy_(a)), ("fill 4 GiB", lambda: b.fill_(1.0)), ("idle", lambda: time.sleep(0.02))):
    fn()
    torch.cuda.synchronize()
    with gpustat.sampler_for(0, 0.02) as st:
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            fn()
            n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    s = st.summary()
    gb = {"copy 4 GiB": 8.59, "fill 4 GiB": 4.29, "idle": 0.0}[name] * n
    print(f"{name:12s} {gb / dt / 1e3:6.2f} TB/s  {s['clock_ghz_mean']} GHz (min {s['clock_ghz_min']})  {s['power_w_mean']} W  ({s['samples']} samples)")
