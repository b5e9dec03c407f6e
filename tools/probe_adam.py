"""What arithmetic the stock Adam implementations use: one parameter, chosen gradients, three steps, exact values."""
import os
import struct
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")


def hexf(x):
    return struct.pack(">f", float(x)).hex()


def run(kind, g_values, steps=3, lr=1e-3):
    p = torch.nn.Parameter(torch.zeros(len(g_values), device=dev))
    kw = dict(fused=True) if kind == "fused" else (dict(fused=True, capturable=True) if kind == "fused_capt" else
                                                   (dict(foreach=True) if kind == "foreach" else dict(foreach=False)))
    if kind == "flat":
        from nerf_amd.optim import FlatAdam
        opt = FlatAdam([p], lr=lr)
    else:
        opt = torch.optim.Adam([p], lr=lr, **kw)
    out = []
    for i in range(steps):
        p.grad = torch.tensor(g_values, device=dev) * (1.0 + i)
        opt.step()
        st = opt.state[p]
        out.append((p.detach().cpu().numpy().copy(), st["exp_avg"].cpu().numpy().copy(), st["exp_avg_sq"].cpu().numpy().copy()))
    return out


g = [1.0, 1e-3, 3e-9, -0.37]
res = {k: run(k, g) for k in ("single", "foreach", "fused", "fused_capt", "flat")}
for i in range(3):
    print("step", i + 1)
    for k, r in res.items():
        p, m, v = r[i]
        print(f"  {k:10s} p " + " ".join(hexf(x) for x in p) + "  m " + " ".join(hexf(x) for x in m) + "  v " + " ".join(hexf(x) for x in v))
print("float(0.1)", hexf(0.1), "1-0.9f", hexf(np.float32(1) - np.float32(0.9)), "float(0.001)", hexf(0.001), "1-0.999f",
      hexf(np.float32(1) - np.float32(0.999)))
