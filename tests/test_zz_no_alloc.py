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


def test_static_workspace_backward_can_be_captured_in_a_hip_graph(device):
    """After one eager call on the device (which looks up the built-in workspace and sets the kernels' LDS limits), the
    entry point only enqueues: captured with torch.cuda.graph and replayed on new upstream gradients, it gives the bits
    of the eager caller-workspace call."""
    from nerf_amd import ops
    from nerf_amd._lib import check, lib
    from nerf_amd.run_nerf_helpers import NeRF
    torch.manual_seed(21)
    m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True).to(device)
    flat = m._flat()
    g = torch.Generator(device=device).manual_seed(22)
    n_pts, ppr = 512, 4
    pts = torch.rand(n_pts, 3, device=device, generator=g) * 4 - 2
    dirs = torch.nn.functional.normalize(torch.randn(n_pts // ppr, 3, device=device, generator=g), dim=-1)
    act = ops.mlp_act_buffer(n_pts, True, device)
    ops.mlp_forward_bf16x3(pts, dirs, m._packed_bf16x3(), True, m.output_ch, ppr, act)
    x3, x3b = m._packed_bf16x3(), m._packed_bwd_bf16x3()
    d_out = torch.randn(n_pts, 4, device=device, generator=g)
    d_flat = torch.zeros_like(flat)

    def call(stream):
        check(lib.nerf_mlp_bwd_bf16x3_static(n_pts, d_out.data_ptr(), act.data_ptr(), 1, m.output_ch, x3.data_ptr(), x3b.data_ptr(),
                                             d_flat.data_ptr(), stream))
    call(torch.cuda.current_stream().cuda_stream)            # eager warm-up: first use of the entry point on this device
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream().cuda_stream)
    for seed in (31, 32):
        d_out.copy_(torch.randn(n_pts, 4, device=device, generator=torch.Generator(device=device).manual_seed(seed)))
        d_flat.zero_()
        graph.replay()
        torch.cuda.synchronize()
        got = d_flat.cpu().numpy()
        scratch = torch.empty(lib.nerf_mlp_scratch_floats(1, n_pts), dtype=torch.float32, device=device)
        want = torch.zeros_like(flat)
        check(lib.nerf_mlp_bwd_bf16x3(n_pts, d_out.data_ptr(), act.data_ptr(), 1, m.output_ch, x3.data_ptr(), x3b.data_ptr(),
                                      scratch.data_ptr(), want.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert np.abs(got).max() > 0 and np.array_equal(got, want.cpu().numpy()), seed
