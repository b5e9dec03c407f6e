"""Parity cases at the edges of the render path: frames through render_path (render_factor, 8-bit conversion, PNG files),
per-ray near/far bounds, non-finite and huge activations in the split-operand kernel, weights written behind autograd's
back.  Same oracle and tolerances as test_gpu_parity.py."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from test_gpu_parity import T, _model, _peaky, _query_fn, close, exact

pytestmark = pytest.mark.gpu


def _scene(device, n_coarse=16, n_fine=16):
    sd_c, sd_f = _peaky(O.random_state_dict(71), 25.0), _peaky(O.random_state_dict(72), 25.0)
    kw = dict(network_fn=_model(sd_c, device), network_fine=_model(sd_f, device), network_query_fn=_query_fn(),
              N_samples=n_coarse, N_importance=n_fine, perturb=0.0, raw_noise_std=0.0, white_bkgd=True, use_viewdirs=True,
              ndc=False, near=2.0, far=6.0)
    return sd_c, sd_f, kw


def test_render_path_frames_match_the_oracle(device, tmp_path):
    """render_path (run_nerf.py:163-201) on three poses, whole and with render_factor=2: every frame against the oracle
    render of the same pose (as upstream, the factor shrinks H, W and hwf's focal while K is passed on unchanged), the
    8-bit frames against to8b of the oracle's, the PNG files against the returned frames, pose order kept."""
    from PIL import Image
    from nerf_amd.run_nerf import render_path
    from nerf_amd.run_nerf_helpers import to8b
    sd_c, sd_f, kw = _scene(device)
    H, W, focal = 20, 24, 30.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    poses = torch.stack([O.orbit_pose(th, -30.0, 4.0) for th in (-120.0, 10.0, 95.0)], 0)
    for factor in (0, 2):
        out = tmp_path / f"f{factor}"
        os.makedirs(out)
        with torch.no_grad():
            rgbs, disps = render_path(poses.to(device), [H, W, focal], K, 100, kw, savedir=str(out), render_factor=factor)
        h, w = (H, W) if factor == 0 else (H // factor, W // factor)
        assert rgbs.shape == (3, h, w, 3) and disps.shape == (3, h, w) and rgbs.dtype == np.float32
        assert sorted(os.listdir(out)) == ["000.png", "001.png", "002.png"]
        for i in range(3):
            o, d = O.pixel_rays(h, w, K, poses[i, :3, :4])
            with torch.no_grad():
                want = O.render_chunks(O.pack_rays(o, d, 2.0, 6.0, use_viewdirs=True), 1 << 15, sd_coarse=sd_c, sd_fine=sd_f,
                                       n_coarse=16, n_fine=16, white_bkgd=True)
            close(rgbs[i].reshape(-1, 3), want["rgb_map"], 2e-4, 2e-5, f"frame {i} rgb (factor {factor})")
            close(disps[i].reshape(-1), want["disp_map"], 1e-3, 1e-5, f"frame {i} disp (factor {factor})")
            got8, want8 = to8b(rgbs[i]), to8b(want["rgb_map"].numpy().reshape(h, w, 3))
            diff = np.abs(got8.astype(np.int32) - want8.astype(np.int32))
            # truncation to 8 bits turns a 1e-6 difference into one count where 255 x sits on an integer
            assert diff.max() <= 1 and (diff > 0).mean() < 0.01, (diff.max(), (diff > 0).mean())
            png = np.asarray(Image.open(out / f"{i:03d}.png"))
            assert np.array_equal(png, got8), f"PNG {i} is the 8-bit frame"
        # frames differ from each other (pose order is observable) and the sequence is deterministic
        assert np.abs(rgbs[0] - rgbs[1]).max() > 1e-3
        with torch.no_grad():
            again, _ = render_path(poses.to(device), [H, W, focal], K, 100, kw, render_factor=factor)
        assert np.array_equal(again, rgbs)


def test_per_ray_bounds(device):
    """near / far that differ from ray to ray: through the ray record builder (bit-exact columns) and through
    render_rays against the oracle's march on the same records."""
    from nerf_amd import ops
    from nerf_amd.run_nerf import render_rays
    rng = np.random.Generator(np.random.PCG64(5))
    n = 77
    o = T(rng.normal(size=(n, 3)).astype(np.float32) * 0.3 + np.array([0, 0, 4.0], dtype=np.float32))
    d = T(rng.normal(size=(n, 3)).astype(np.float32) * 0.2 + np.array([0, 0, -1.0], dtype=np.float32))
    near = T(rng.uniform(1.0, 2.5, size=(n,)).astype(np.float32))
    far = T(rng.uniform(5.0, 7.0, size=(n,)).astype(np.float32))
    rec = ops.pack_rays(o.to(device), d.to(device), near.to(device), far.to(device), viewdir_src=d.to(device))
    want = O.pack_rays(o, d, 0.0, 0.0, use_viewdirs=True)
    want[:, 6], want[:, 7] = near, far
    exact(rec[:, :8], want[:, :8], "ray records with per-ray bounds: origins, directions, bounds")
    close(rec[:, 8:], want[:, 8:], 0, 1.2e-7, "unit view directions (the CPU norm sums in another order)")
    # one bound per ray, the other a scalar; 8-column records
    rec8 = ops.pack_rays(o.to(device), d.to(device), 2.0, far.to(device))
    exact(rec8[:, 6], torch.full((n,), 2.0), "scalar near")
    exact(rec8[:, 7], far, "per-ray far")
    sd_c, sd_f, kw = _scene(device)
    for k in ("ndc", "near", "far", "use_viewdirs"):
        kw.pop(k)
    with torch.no_grad():
        got = render_rays(rec, **kw)
        ref = O.march(want, sd_c, sd_f, 16, 16, white_bkgd=True)
    close(got["rgb_map"], ref["rgb_map"], 2e-4, 2e-5, "rgb with per-ray bounds")
    close(got["acc_map"], ref["acc_map"], 2e-4, 2e-5, "acc with per-ray bounds")
    close(got["z_std"], ref["z_std"], 1e-4, 1e-6, "z_std with per-ray bounds")


def test_split_operand_kernel_on_huge_and_non_finite_activations(device):
    """One hidden unit of layer 0 is driven to 1e30, to +inf and to NaN through its bias.  Huge finite activations: the
    split-operand kernel agrees with the fp32 matrix instruction to the usual relative tolerance.  Non-finite ones: the
    reference's ReLU keeps a NaN a NaN, an infinite unit meets weights of both signs in the next layer (inf - inf), so
    upstream returns NaN at every point and channel; so do both kernels (an infinite operand splits into inf, NaN, NaN:
    the split scheme reaches NaN one layer earlier than the fp32 instruction, never a finite number)."""
    from nerf_amd import ops
    rng = np.random.Generator(np.random.PCG64(6))
    n, s = 40, 8
    pts = T(rng.uniform(-2, 2, size=(n * s, 3)).astype(np.float32))
    vd = T(rng.normal(size=(n, 3)).astype(np.float32))
    vd = vd / vd.norm(dim=-1, keepdim=True)

    def both(bias_value):
        sd = O.random_state_dict(81)
        sd["pts_linears.0.bias"][7] = bias_value
        m = _model(sd, device)
        with torch.no_grad():
            a = ops.mlp_forward_bf16x3(pts.to(device), vd.to(device), m._packed_bf16x3(), True, 5, s)
            b = ops.mlp_forward_x4(pts.to(device), vd.to(device), m._packed_x4(), True, 5, s)
            want = O.query_field(pts.reshape(n, s, 3), vd, sd).reshape(-1, 4)
        return a.cpu().numpy(), b.cpu().numpy(), want.numpy()

    a, b, want = both(1e30)
    assert np.isfinite(a).all() and np.isfinite(b).all() and np.abs(b).max() > 1e20
    scale = float(np.abs(want).max())
    np.testing.assert_allclose(a, b, rtol=0, atol=3e-6 * scale)
    np.testing.assert_allclose(a, want, rtol=0, atol=2e-5 * scale)
    for bad in (float("inf"), float("nan")):
        a, b, want = both(bad)
        assert np.isnan(want).all(), "upstream arithmetic gives NaN everywhere"
        assert np.isnan(b).all(), "fp32 matrix instruction"
        assert np.isnan(a).all(), "split operands"


def test_weights_written_through_data_need_invalidate_packs(device):
    """The packed weight streams follow in-place parameter updates that autograd sees (optimiser steps); a write through
    `.data` is invisible to that bookkeeping until invalidate_packs() is called."""
    qf = _query_fn()
    m = _model(O.random_state_dict(91), device)
    rng = np.random.Generator(np.random.PCG64(7))
    pts = T(rng.uniform(-2, 2, size=(6, 4, 3)).astype(np.float32)).to(device)
    vd = torch.nn.functional.normalize(T(rng.normal(size=(6, 3)).astype(np.float32)), dim=-1).to(device)
    with torch.no_grad():
        before = qf(pts, vd, m).clone()
        m.rgb_linear.weight.mul_(2.0)                    # bumps the version counter: picked up by itself
        visible = qf(pts, vd, m).clone()
        m.rgb_linear.weight.data.mul_(0.5)               # back to the start, behind the counter's back
        stale = qf(pts, vd, m).clone()
        m.invalidate_packs()
        fresh = qf(pts, vd, m).clone()
    assert float((visible - before).abs().max()) > 0
    exact(stale, visible, "a .data write alone leaves the packed streams as they were")
    close(fresh, before, 0, 1e-6, "after invalidate_packs the streams follow the parameters")


@pytest.mark.parametrize("use_viewdirs,output_ch", [(True, 5), (False, 4), (False, 5)])
def test_split_streams_from_one_gather_equal_the_packing_kernels(device, use_viewdirs, output_ch):
    """nerf_gather_split3 through the map taken from the packing kernels rebuilds both split streams (and the forward
    stream's fp32 tables) bit for bit, on ordinary weights and on values whose pieces carry signs and zeros."""
    from nerf_amd import ops
    n = ops.mlp_n_params(use_viewdirs, output_ch)
    table = ops.x3_pack_table(use_viewdirs, output_ch, device)
    rng = np.random.Generator(np.random.PCG64(3))
    for scale in (1.0, 1e-3, 1e4):
        flat = T((rng.standard_normal(n) * scale).astype(np.float32)).to(device)
        flat[::7] = 0.0
        flat[1::11] = -flat[1::11].abs()
        a, b = ops.pack_weights_x3_pair(flat, *table)
        exact(a, ops.pack_weights_bf16x3(flat, use_viewdirs, output_ch), "forward split stream")
        exact(b, ops.pack_weights_bwd_bf16x3(flat, use_viewdirs, output_ch), "transposed split stream")


def test_parameters_live_in_one_flat_buffer(device):
    """After the first query the parameters are views of one flat vector: names, shapes and values unchanged, optimiser
    steps and load_state_dict land in the vector the weight streams are packed from, and a parameter whose storage
    was replaced is adopted again with its new values."""
    qf = _query_fn()
    sd = O.random_state_dict(93)
    m = _model(sd, device)
    rng = np.random.Generator(np.random.PCG64(8))
    pts = T(rng.uniform(-2, 2, size=(5, 4, 3)).astype(np.float32)).to(device)
    vd = torch.nn.functional.normalize(T(rng.normal(size=(5, 3)).astype(np.float32)), dim=-1).to(device)
    with torch.no_grad():
        first = qf(pts, vd, m).clone()
    flat = m._flat()
    off = 0
    for (name, p), (name0, v0) in zip(m.named_parameters(), sd.items()):
        assert name == name0 and tuple(p.shape) == tuple(v0.shape)
        assert p.data_ptr() == flat.data_ptr() + 4 * off, name
        exact(p.detach().cpu(), v0, name)
        off += p.numel()
    assert off == flat.numel()
    # an optimiser step moves the flat vector, and the next query sees it without any copy
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, fused=True)
    out = qf(pts, vd, m)
    out.square().mean().backward()
    opt.step()
    assert m._flat() is flat
    exact(flat[:16], m.pts_linears[0].weight.detach().reshape(-1)[:16], "views follow the step")
    with torch.no_grad():
        stepped = qf(pts, vd, m).clone()
    assert float((stepped - first).abs().max()) > 0
    # load_state_dict copies in place; replacing a parameter's storage is noticed
    m.load_state_dict(sd)
    with torch.no_grad():
        close(qf(pts, vd, m), first, 0, 1e-6, "after load_state_dict")
        m.rgb_linear.weight.data = (2.0 * m.rgb_linear.weight.data).clone()
        doubled = qf(pts, vd, m).clone()
    assert m._flat() is not flat and m.rgb_linear.weight.data_ptr() != 0
    m2 = _model(dict(sd, **{"rgb_linear.weight": 2.0 * sd["rgb_linear.weight"]}), device)
    with torch.no_grad():
        close(doubled, qf(pts, vd, m2), 0, 1e-6, "re-adopted after a storage swap")
    # the checkpoint round trip keeps names and values
    sd2 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    assert list(sd2) == list(sd)


def test_embedder_keyword_settings_against_the_reference(device):
    """Embedder under non-default keyword settings (no identity block, linear bands, cos before sin, one or three
    functions, 2 and 5 input dims, no bands at all): the general encoding kernel against outputs of the reference's own
    class; the default setting keeps the kernel that is fused into the MLP; other callables are refused."""
    from test_oracle_golden import embedder_cases
    from nerf_amd.run_nerf_helpers import Embedder, get_embedder
    for name, kw, x, want in embedder_cases():
        e = Embedder(**kw)
        assert e.out_dim == want.shape[-1] and e.n_freq is None, name
        got = e.embed(x.to(device))
        # arguments reach 3 * 32; sinf / cosf of the device library against the CPU's: a few ulp of a value below 1
        close(got, want, 0, 4e-7, name)
        close(e.embed(x.reshape(1, -1, x.shape[-1]).to(device)), want.reshape(1, -1, want.shape[-1]), 0, 4e-7, name + " 3-d input")
    fn, width = get_embedder(10, 0)
    assert width == 63 and fn.n_freq == 10
    with pytest.raises(NotImplementedError):
        Embedder(include_input=True, input_dims=3, max_freq_log2=3, num_freqs=4, log_sampling=True, periodic_fns=[torch.tanh])


@pytest.mark.parametrize("n,n_in,n_out,relu,bias", [(1, 3, 1, False, True), (70, 63, 128, True, True), (1000, 90, 64, True, False),
                                                    (4097, 319, 256, True, True), (33, 5, 200, False, True), (2500, 130, 3, False, True)])
def test_linear_layer_kernel_against_fp64(device, n, n_in, n_out, relu, bias):
    """ops.linear (forward, gradient w.r.t. input, weight and bias) against an fp64 evaluation of torch.nn.functional.linear:
    ragged sizes around the 64 x 64 tiles and the 16-value k-steps, inputs that are column slices of a wider buffer, points
    beyond one slice of the weight-gradient product; errors at the level of the CPU's own fp32 evaluation."""
    from nerf_amd import ops
    rng = np.random.Generator(np.random.PCG64(n + n_in))
    wide = T(rng.normal(size=(n, n_in + 7)).astype(np.float32))
    w = T((rng.normal(size=(n_out, n_in)) / np.sqrt(n_in)).astype(np.float32))
    b = T(rng.normal(size=(n_out,)).astype(np.float32)) if bias else None
    g = T(rng.normal(size=(n, n_out)).astype(np.float32))

    def run(x, w, b, dtype, fn):
        x, w = x.clone().to(dtype).requires_grad_(True), w.clone().to(dtype).requires_grad_(True)
        b = None if b is None else b.clone().to(dtype).requires_grad_(True)
        y = fn(x, w, b)
        (y * g.to(y.device, dtype)).sum().backward()
        return [t.detach().double().cpu() for t in (y, x.grad, w.grad)] + ([b.grad.double().cpu()] if b is not None else [])

    def ref(x, w, b):
        y = torch.nn.functional.linear(x, w, b)
        return torch.relu(y) if relu else y
    x_cpu = wide[:, 3:3 + n_in]
    truth = run(x_cpu, w, b, torch.float64, ref)
    cpu32 = run(x_cpu, w, b, torch.float32, ref)
    xd = wide.to(device)[:, 3:3 + n_in]                      # a column slice: row stride n_in + 7
    got = run(xd, w.to(device), None if b is None else b.to(device), torch.float32, lambda x, w, b: ops.linear(x, w, b, relu))
    # units of the ReLU that sit within rounding of zero may fall on either side in any fp32 evaluation
    y64 = torch.nn.functional.linear(x_cpu.double(), w.double(), None if b is None else b.double())
    safe = bool(((y64.abs() > 1e-4).all())) or not relu
    for name, t, c, h in zip(("y", "dx", "dw", "db"), truth, cpu32, got):
        scale = float(t.abs().max()) + 1e-30
        err_cpu, err_gpu = float((c - t).abs().max()) / scale, float((h - t).abs().max()) / scale
        assert h.shape == t.shape, name
        if safe or name == "y":
            assert err_gpu <= max(2e-6, 3 * err_cpu), (name, n, n_in, n_out, err_gpu, err_cpu)
        else:
            assert err_gpu <= 1e-3, (name, err_gpu)
    again = run(xd, w.to(device), None if b is None else b.to(device), torch.float32, lambda x, w, b: ops.linear(x, w, b, relu))
    for a, c in zip(got, again):
        exact(a, c, "bitwise reproducible")


def test_relu_gradient_lets_the_gradient_of_a_nan_unit_through(device):
    """A unit whose forward value is NaN: torch's threshold_backward zeroes the gradient where the output is <= 0, which a
    NaN is not, so the incoming gradient passes; the library's layer-by-layer backward does the same."""
    from nerf_amd import ops
    x = torch.tensor([[1.0, -2.0, 0.5, 0.0, 3.0]], device=device)
    w = torch.eye(5, device=device)
    b = torch.tensor([0.0, 0.0, float("nan"), 0.0, 0.0], device=device)      # unit 2 is NaN in the forward pass, the others finite
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ops.linear(xr, wr, br, True).backward(torch.ones(1, 5, device=device))
    xt, wt, bt = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    torch.relu(torch.nn.functional.linear(xt, wt, bt)).backward(torch.ones(1, 5, device=device))
    for got, want, what in ((xr.grad, xt.grad, "dx"), (wr.grad, wt.grad, "dw"), (br.grad, bt.grad, "db")):
        exact(got, want, what)
    assert float(xr.grad[0, 2]) == 1.0 and float(xr.grad[0, 1]) == 0.0 and float(xr.grad[0, 3]) == 0.0
