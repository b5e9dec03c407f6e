"""Mean time of the fine-sampling kernel on a 32768-ray chunk (64 coarse + 128 fine, deterministic uniforms)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_amd import ops
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n, Sc, Si = 32768, 64, 128
g = torch.Generator(device=dev); g.manual_seed(0)
z = torch.sort(torch.rand(n, Sc, device=dev, generator=g) * 4 + 2, -1)[0]
w = torch.rand(n, Sc, device=dev, generator=g) ** 4
rays = torch.rand(n, 11, device=dev, generator=g)
for u in (ops.cached_linspace(Si, dev), torch.rand(n, Si, device=dev, generator=g)):
    for _ in range(3):
        ops.sample_fine(z, w, u, rays)
    ops.PROFILE = []
    for _ in range(20):
        ops.sample_fine(z, w, u, rays)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    t = [s.elapsed_time(e) * 1e3 for (_, _, s, e) in prof]
    print(f"{'shared sorted u' if u.dim() == 1 else 'random u       '}: {sum(t) / len(t):7.1f} us", flush=True)
