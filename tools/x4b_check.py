"""Backward on 16-point tiles against the 32-point backward: same saved activations, same d_out; prints the largest
difference of the flat parameter gradient and the time of each.  python tools/x4b_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from nerf_amd import ops
from nerf_amd.run_nerf_helpers import NeRF
from oracle import nerf_oracle as O

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
models, sds, kwargs = bench.build_scene(dev)
g = torch.Generator(device=dev); g.manual_seed(0)
m2 = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
m2.load_state_dict(O.random_state_dict(33, in_dirs=0, use_viewdirs=False, output_ch=5)); m2 = m2.to(dev)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for m, uv in ((models[0], True), (m2, False)):
    for n_rays, S in ((1000, 64), (777, 5), (8192, 192)):
        n_pts = n_rays * S
        pts = (torch.rand(n_pts, 3, device=dev, generator=g) - 0.5) * 8
        vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1) if uv else None
        fwd, bwd, flat = m._packed(True)
        act = ops.mlp_act_buffer(n_pts, uv, dev)
        out = ops.mlp_forward_x4(pts, vd, m._packed_x4(), uv, 5, S, act)
        d_out = torch.randn(out.shape, device=dev, generator=g)
        a = ops.mlp_backward(d_out, act, flat, bwd, fwd, uv, 5, S)
        b = ops.mlp_backward_x4(d_out, act, flat, m._packed_x4(), m._packed_bwd_x4(), uv, 5)
        c = ops.mlp_backward_bf16x3(d_out, act, flat, m._packed_bf16x3(), m._packed_bwd_bf16x3(), uv, 5)
        tc = timed(lambda: ops.mlp_backward_bf16x3(d_out, act, flat, m._packed_bf16x3(), m._packed_bwd_bf16x3(), uv, 5))
        print(f"   split-operand backward: max diff vs 32-pt {(a - c).abs().max().item():.3e}, finite={bool(torch.isfinite(c).all())}, {tc:.3f} ms", flush=True)
        d = (a - b).abs()
        rel = (d / (a.abs() + 1e-3 * a.abs().max())).max().item()
        ta = timed(lambda: ops.mlp_backward(d_out, act, flat, bwd, fwd, uv, 5, S))
        tb = timed(lambda: ops.mlp_backward_x4(d_out, act, flat, m._packed_x4(), m._packed_bwd_x4(), uv, 5))
        print(f"viewdirs={uv} rays={n_rays} S={S}: max diff {d.max().item():.3e} (grad scale {a.abs().max().item():.3e}, "
              f"rel {rel:.2e}) finite={bool(torch.isfinite(b).all())}  32-pt {ta:.3f} ms  16-pt {tb:.3f} ms", flush=True)
