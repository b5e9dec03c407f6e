"""The backward entry point that allocates nothing (nerf_mlp_bwd_bf16x3_static) against nerf_mlp_bwd_bf16x3, through the
C ABI: bit-identical parameter gradients on two calls in a row, the second with fewer points (the built-in workspace
keeps the first call's contents), and a clear refusal beyond its stated maximum."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads(n_pts, seed, dev, use_static, use_viewdirs=True):
    from nerf_amd import ops
    from nerf_amd._lib import check, lib
    from nerf_amd.run_nerf_helpers import NeRF
    torch.manual_seed(seed)
    out_ch = 4 if use_viewdirs else 5
    m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27 if use_viewdirs else 0, output_ch=5, skips=[4],
             use_viewdirs=use_viewdirs).to(dev)
    flat = m._flat()
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    pts_per_ray = 4
    assert n_pts % pts_per_ray == 0
    n_rays = n_pts // pts_per_ray
    pts = (torch.rand(n_rays * pts_per_ray, 3, device=dev, generator=g) * 4 - 2)[:n_pts].contiguous()
    dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1)
    act = ops.mlp_act_buffer(n_pts, use_viewdirs, dev)
    ops.mlp_forward_bf16x3(pts, dirs if use_viewdirs else None, m._packed_bf16x3(), use_viewdirs, m.output_ch, pts_per_ray, act)
    d_out = torch.randn(n_pts, out_ch, device=dev, generator=g)
    d_flat = torch.full_like(flat, float("nan"))
    x3, x3b = m._packed_bf16x3(), m._packed_bwd_bf16x3()
    stream = torch.cuda.current_stream().cuda_stream
    if use_static:
        check(lib.nerf_mlp_bwd_bf16x3_static(n_pts, d_out.data_ptr(), act.data_ptr(), int(use_viewdirs), m.output_ch,
                                             x3.data_ptr(), x3b.data_ptr(), d_flat.data_ptr(), stream))
    else:
        scratch = torch.empty(lib.nerf_mlp_scratch_floats(int(use_viewdirs), n_pts), dtype=torch.float32, device=dev)
        check(lib.nerf_mlp_bwd_bf16x3(n_pts, d_out.data_ptr(), act.data_ptr(), int(use_viewdirs), m.output_ch,
                                      x3.data_ptr(), x3b.data_ptr(), scratch.data_ptr(), d_flat.data_ptr(), stream))
    torch.cuda.synchronize()
    return d_flat.cpu().numpy()


def test_static_workspace_backward_matches_the_caller_workspace_one(device):
    from nerf_amd._lib import NerfHipError, check, lib
    n_max = lib.nerf_mlp_bwd_static_max_points()
    assert n_max == 2048
    # two calls in a row on the shared array: the maximum size, then a smaller, ragged one; then without the view branch
    for n_pts, seed, uv in ((n_max, 11, True), (300, 12, True), (1000, 13, False)):
        want = _grads(n_pts, seed, device, use_static=False, use_viewdirs=uv)
        got = _grads(n_pts, seed, device, use_static=True, use_viewdirs=uv)
        assert np.isfinite(want).all() and np.abs(want).max() > 0
        assert np.array_equal(got, want), (n_pts, float(np.abs(got - want).max()))
    with pytest.raises(NerfHipError, match="exceeds the built-in workspace"):
        check(lib.nerf_mlp_bwd_bf16x3_static(n_max + 1, 16, 16, 1, 4, 16, 16, 16, None))
