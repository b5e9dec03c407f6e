"""HIP path vs the CPU oracle and vs the committed reference outputs (tests/golden).

Tolerances (fp32 path): elementwise kernels with no libm call are bit-exact; kernels that call
exp/sin/cos differ from the CPU libm by a few ulp; the MLP differs by fp32 summation order
(MFMA k-order vs the CPU GEMM), |d| <= 2e-5 * max(1, |x|) on the raw outputs for these weights;
composited maps inherit that.  sample_pdf indices are exact given identical cdf inputs.
"""
import os

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(got, want, rtol, atol, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, equal_nan=True, err_msg=what)


def exact(got, want, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} differ, max|d|={np.nanmax(np.abs(got - want))}"


def _peaky(sd, gain, shift=0.5):
    sd = {k: v.clone() for k, v in sd.items()}
    sd["alpha_linear.weight"] *= gain
    sd["alpha_linear.bias"] = sd["alpha_linear.bias"] * gain + shift
    return sd


def _peaky_out(sd, gain, shift=0.5):
    """_peaky for the network without view branch: density is row 3 of output_linear."""
    sd = {k: v.clone() for k, v in sd.items()}
    sd["output_linear.weight"][3] *= gain
    sd["output_linear.bias"][3] = sd["output_linear.bias"][3] * gain + shift
    return sd


def _model(sd, device, use_viewdirs=True, output_ch=5):
    from nerf_amd.run_nerf_helpers import NeRF
    m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27 if use_viewdirs else 0, output_ch=output_ch, skips=[4],
             use_viewdirs=use_viewdirs)
    m.load_state_dict(sd)
    return m.to(device)


def _query_fn():
    from nerf_amd.run_nerf import run_network
    from nerf_amd.run_nerf_helpers import get_embedder
    e, _ = get_embedder(10, 0)
    ed, _ = get_embedder(4, 0)
    return lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=e, embeddirs_fn=ed, netchunk=1 << 16)


# ---------------------------------------------------------------------------------------------
def test_device_is_gfx950(device):
    from nerf_amd._lib import lib, check
    check(lib.nerf_check_device())


def test_get_rays_and_ndc(device):
    from nerf_amd.run_nerf_helpers import get_rays, ndc_rays
    g = golden("rays")
    for name in ("lego_half", "lego_full", "fern"):
        H, W, focal = g[name + "_hwf"]
        H, W = int(H), int(W)
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        pose = T(g[name + "_pose"])
        pick = g[name + "_pick"]
        o, d = get_rays(H, W, K, pose.to(device))
        assert o.shape == (H, W, 3) and d.is_cuda
        exact(d.reshape(-1, 3)[pick], g[name + "_d"], name + " rays_d vs reference")
        oo, od = O.pixel_rays(H, W, K, pose)
        exact(d, od, name + " rays_d vs oracle")
        exact(o, oo, name + " rays_o vs oracle")
        if name == "fern":
            no, nd = ndc_rays(H, W, K[0][0], 1.0, o, d)
            exact(no.reshape(-1, 3)[pick], g["fern_ndc_o"], "ndc o")
            exact(nd.reshape(-1, 3)[pick], g["fern_ndc_d"], "ndc d")


def test_encode(device):
    from nerf_amd.run_nerf_helpers import get_embedder
    g = golden("encode")
    x = T(g["x"]).to(device)
    for L in (10, 4):
        fn, width = get_embedder(L, 0)
        e = fn(x)
        assert e.shape == (256, width)
        # sin/cos of arguments up to 512*6: the device libm and the CPU one are both accurate to ~1 ulp
        close(e, g[f"enc{L}"], 0, 4e-7, f"encode L={L}")
        exact(e[:, :3], g["x"], "identity columns")


def test_sample_coarse_bit_exact(device):
    from nerf_amd import ops
    rng = np.random.Generator(np.random.PCG64(5))
    n = 300
    rays = T(rng.normal(size=(n, 11)).astype(np.float32))
    rays[:, 6] = T(rng.uniform(0.5, 2.0, size=n).astype(np.float32))
    rays[:, 7] = rays[:, 6] + T(rng.uniform(1.0, 4.0, size=n).astype(np.float32))
    for S in (64, 7, 256, 1):
        t_rand = T(rng.uniform(size=(n, S)).astype(np.float32))
        for lindisp in (False, True):
            for tr in (None, t_rand):
                z, pts = ops.sample_coarse(rays.to(device), ops.cached_linspace(S, device), None if tr is None else tr.to(device), lindisp)
                zo = O.coarse_depths(rays[:, 6:7], rays[:, 7:8], S, lindisp, tr)
                exact(z, zo, f"z S={S} lindisp={lindisp} perturb={tr is not None}")
                po = rays[:, None, 0:3] + rays[:, None, 3:6] * zo[..., None]
                exact(pts, po, "pts")


def test_composite_forward_backward(device):
    from nerf_amd.run_nerf import raw2outputs
    from nerf_amd import ops
    g = golden("composite")
    z, d, noise = T(g["z"]), T(g["rays_d"]), T(g["noise"])
    names = ("rgb", "disp", "acc", "weights", "depth")
    for name in ("soft", "peaky", "edge"):
        for white in (0, 1):
            for noisy in (0, 1):
                key = f"{name}_w{white}_n{noisy}"
                raw = T(g[name + "_raw"]).to(device).requires_grad_(True)
                out = ops.composite(raw, z.to(device), d.to(device), noise.to(device) if noisy else None, bool(white))
                for t, nm in zip(out, names):
                    # disparity = 1/max(1e-10, depth/acc) amplifies relative error; everything else is O(1)
                    close(t, g[f"{key}_{nm}"], 2e-5 if nm != "disp" else 2e-4, 2e-6, f"{key} {nm}")
                (out[0] * T(g[key + "_G"]).to(device)).sum().backward()
                want = g[key + "_draw"]
                scale = max(1.0, float(np.abs(want).max()))
                close(raw.grad, want, 2e-4, 2e-6 * scale, key + " d_raw")
    # the public wrapper, no noise, [n,3] directions with a row stride (a slice of a ray record)
    rec = torch.zeros(z.shape[0], 11, device=device)
    rec[:, 3:6] = d.to(device)
    out = raw2outputs(T(g["soft_raw"]).to(device), z.to(device), rec[:, 3:6], 0, True)
    close(out[0], g["soft_w1_n0_rgb"], 2e-5, 2e-6, "strided rays_d")
    e = raw2outputs(T(g["edge_raw"]).to(device), z.to(device), d.to(device))
    assert e[2][0].item() == 0.0 and np.isnan(e[1][0].item())


def test_composite_all_gradients_and_long_rays(device):
    """Upstream grads on every output, S spanning several 64-sample tiles, against autograd on the oracle."""
    from nerf_amd import ops
    rng = np.random.Generator(np.random.PCG64(6))
    for S in (192, 65, 700):
        n = 40
        raw = T(rng.normal(scale=2.0, size=(n, S, 4)).astype(np.float32))
        z = T(np.sort(rng.uniform(2, 6, size=(n, S)), -1).astype(np.float32))
        d = T(rng.normal(size=(n, 3)).astype(np.float32))
        G = [T(rng.normal(size=s).astype(np.float32)) for s in ((n, 3), (n,), (n,), (n, S), (n,))]
        ro = raw.clone().requires_grad_(True)
        oo = O.composite(ro, z, d, None, True)
        sum((a * b).sum() for a, b in zip(oo, G)).backward()
        rg = raw.to(device).requires_grad_(True)
        og = ops.composite(rg, z.to(device), d.to(device), None, True)
        sum((a * b.to(device)).sum() for a, b in zip(og, G)).backward()
        for a, b, nm in zip(og, oo, ("rgb", "disp", "acc", "weights", "depth")):
            close(a, b, 3e-5 if nm != "disp" else 3e-4, 3e-6, f"S={S} {nm}")
        scale = float(ro.grad.abs().max())
        close(rg.grad, ro.grad, 5e-4, 5e-6 * scale, f"S={S} d_raw")


def _index_mismatches_are_ulp_ties(inds, want, u, bins, weights):
    """Every index that differs from the reference must sit on a cdf value within 1 ulp of its uniform."""
    bad = np.argwhere(inds != want)
    if len(bad) == 0:
        return 0
    w = weights + np.float32(1e-5)
    cdf = np.concatenate([np.zeros((w.shape[0], 1), np.float32),
                          np.cumsum((w / w.sum(-1, keepdims=True, dtype=np.float64)).astype(np.float32), -1, dtype=np.float64).astype(np.float32)], -1)
    for r, j in bad:
        k = min(inds[r, j], want[r, j])
        assert abs(int(inds[r, j]) - int(want[r, j])) == 1
        assert abs(cdf[r, k] - u[r, j]) <= 2 * np.spacing(np.float32(u[r, j])), (r, j, cdf[r, k], u[r, j])
    return len(bad)


def test_sample_pdf_indices_exact(device):
    from nerf_amd import ops
    g = golden("sample_pdf")
    for bk, wk, uk, sk, ik in (("bins", "weights", "u", "samples", "inds"), ("bins5", "weights5", "u5", "samples5", "inds5")):
        s, inds = ops.sample_pdf(T(g[bk]).to(device), T(g[wk]).to(device), T(g[uk]).to(device), want_inds=True)
        exact(inds, g[ik], ik + " vs reference")
        close(s, g[sk], 2e-6, 2e-7, sk)
    n, ns = g["u"].shape
    s, inds = ops.sample_pdf(T(g["bins"]).to(device), T(g["weights"]).to(device), ops.cached_linspace(ns, device), want_inds=True)
    exact(inds, g["inds_det"], "det inds")
    close(s, g["samples_det"], 2e-6, 2e-7, "det samples")


def test_sample_pdf_indices_exact_at_scale(device):
    """512K uniforms against the oracle on this host: indices identical (ulp ties, if any, are proven to be ties)."""
    from nerf_amd import ops
    rng = np.random.Generator(np.random.PCG64(7))
    n, nb, ns = 4096, 63, 128
    bins = np.sort(rng.uniform(2, 6, size=(n, nb)), -1).astype(np.float32)
    w = (rng.uniform(0, 1, size=(n, nb - 1)) ** 6).astype(np.float32)
    u = rng.uniform(0, 1, size=(n, ns)).astype(np.float32)
    so, io = O.inverse_cdf(T(bins), T(w), T(u), return_inds=True)
    sg, ig = ops.sample_pdf(T(bins).to(device), T(w).to(device), T(u).to(device), want_inds=True)
    n_ties = _index_mismatches_are_ulp_ties(ig.cpu().numpy(), io.numpy(), u, bins, w)
    assert n_ties <= 2, n_ties
    close(sg, so, 1e-5, 1e-6, "samples")
    from nerf_amd.run_nerf_helpers import sample_pdf
    s2 = sample_pdf(T(bins[:16]).to(device), T(w[:16]).to(device), 32, det=True)
    exact(s2, O.inverse_cdf(T(bins[:16]), T(w[:16]), O.det_uniforms(16, 32)), "public sample_pdf det")
    s3 = sample_pdf(T(bins[:16]).to(device), T(w[:16]).to(device), 32, det=False, pytest=True)
    np.random.seed(0)
    exact(s3, O.inverse_cdf(T(bins[:16]), T(w[:16]), torch.Tensor(np.random.rand(16, 32))), "public sample_pdf pytest")


def test_sample_fine_merge_sort_std(device):
    from nerf_amd import ops
    rng = np.random.Generator(np.random.PCG64(8))
    for Sc, Si in ((64, 128), (64, 64), (256, 512), (9, 5), (3, 1)):
        n = 64
        rays = T(rng.normal(size=(n, 11)).astype(np.float32))
        z = T(np.sort(rng.uniform(2, 6, size=(n, Sc)), -1).astype(np.float32))
        w = T((rng.uniform(0, 1, size=(n, Sc)) ** 5).astype(np.float32))
        u = T(rng.uniform(0, 1, size=(n, Si)).astype(np.float32))
        mid = 0.5 * (z[..., 1:] + z[..., :-1])
        smp, inds = O.inverse_cdf(mid, w[..., 1:-1], u, return_inds=True)
        zs, _ = torch.sort(torch.cat([z, smp], -1), -1)
        out = ops.sample_fine(z.to(device), w.to(device), u.to(device), rays.to(device), want_samples=True, want_inds=True)
        exact(out["inds"], inds, f"inds {Sc}+{Si}")
        exact(out["samples"], smp, f"samples {Sc}+{Si}")
        exact(out["z"], zs, f"sorted z {Sc}+{Si}")
        exact(out["pts"], rays[:, None, 0:3] + rays[:, None, 3:6] * zs[..., None], "pts")
        close(out["z_std"], torch.std(smp, dim=-1, unbiased=False), 2e-6, 1e-7, "z_std")
        # shared deterministic uniforms
        ud = ops.cached_linspace(Si, device)
        out = ops.sample_fine(z.to(device), w.to(device), ud, rays.to(device))
        smp = O.inverse_cdf(mid, w[..., 1:-1], O.det_uniforms(n, Si))
        exact(out["z"], torch.sort(torch.cat([z, smp], -1), -1)[0], "det z")


def test_sample_fine_merge_edge_cases(device):
    """The merge must equal sort(cat(.)) also when the coarse depths arrive unordered, when depths tie (flat weights
    with deterministic uniforms put fine samples exactly on coarse ones), with NaN (sorts last) and with -0/+0."""
    from nerf_amd import ops
    rng = np.random.Generator(np.random.PCG64(18))
    n, Sc, Si = 32, 16, 24
    rays = T(rng.normal(size=(n, 11)).astype(np.float32))

    def check(z, w, u, what):
        mid = 0.5 * (z[..., 1:] + z[..., :-1])
        smp = O.inverse_cdf(mid, w[..., 1:-1], u if u.dim() == 2 else u.expand(n, -1).contiguous())
        want, _ = torch.sort(torch.cat([z, smp], -1), -1)
        out = ops.sample_fine(z.to(device), w.to(device), u.to(device), rays.to(device))
        got = out["z"].cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(want)), what
        assert torch.equal(torch.nan_to_num(got, nan=0.0), torch.nan_to_num(want, nan=0.0)), what
        # signed zeros: same multiset of bit patterns per row
        gb = np.sort(got.numpy().view(np.uint32), -1)
        wb = np.sort(want.numpy().view(np.uint32), -1)
        fin = ~np.isnan(want.numpy()).any(-1)
        assert np.array_equal(gb[fin], wb[fin]), what

    w = T((rng.uniform(0, 1, size=(n, Sc)) ** 3).astype(np.float32))
    u = T(rng.uniform(0, 1, size=(n, Si)).astype(np.float32))
    z_desc = T(-np.sort(-rng.uniform(2, 6, size=(n, Sc)), -1).astype(np.float32))
    check(z_desc, w, u, "descending coarse depths")
    z_shuf = T(rng.uniform(2, 6, size=(n, Sc)).astype(np.float32))
    check(z_shuf, w, u, "unordered coarse depths")
    z_ties = T(np.round(np.sort(rng.uniform(0, 4, size=(n, Sc)), -1)).astype(np.float32))
    check(z_ties, torch.ones(n, Sc), ops.cached_linspace(Si, "cpu"), "ties between coarse and fine depths")
    z_nan = T(np.sort(rng.uniform(2, 6, size=(n, Sc)), -1).astype(np.float32))
    z_nan[::3, 5] = float("nan")
    check(z_nan, w, u, "NaN depth")
    w_nan = w.clone()
    w_nan[1::4, 3] = float("nan")
    check(T(np.sort(rng.uniform(2, 6, size=(n, Sc)), -1).astype(np.float32)), w_nan, u, "NaN weights")
    z_zero = T(np.sort(rng.uniform(-1, 1, size=(n, Sc)), -1).astype(np.float32))
    z_zero[:, 7] = -0.0
    z_zero[:, 8] = 0.0
    z_zero = torch.sort(z_zero, -1)[0]
    check(z_zero, w, u, "signed zeros")


def test_mlp_forward(device):
    g = golden("mlp")
    qf = _query_fn()
    m = _model(O.random_state_dict(21), device)
    pts, vd = T(g["vd_pts"]).to(device), T(g["vd_viewdirs"]).to(device)
    with torch.no_grad():
        raw = qf(pts, vd, m)
    assert raw.shape == (24, 8, 4)
    close(raw, g["vd_raw"], 2e-5, 2e-5, "mlp raw vs reference")
    # no view branch
    sd = O.random_state_dict(22, use_viewdirs=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("views_linears")}
    from nerf_amd.run_nerf_helpers import NeRF
    m2 = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
    m2.load_state_dict(sd, strict=False)
    with torch.no_grad():
        raw2 = qf(T(g["novd_pts"]).to(device), None, m2.to(device))
    close(raw2, g["novd_raw"], 2e-5, 2e-5, "mlp raw (no view branch) vs reference")


def test_mlp_forward_ragged_sizes_and_peaky_weights(device):
    """Point counts that are not multiples of the 128-point tile, against the oracle on this host."""
    rng = np.random.Generator(np.random.PCG64(9))
    qf = _query_fn()
    sd = _peaky(O.random_state_dict(33), 30.0)
    m = _model(sd, device)
    for n, s in ((1, 1), (5, 7), (129, 3), (64, 64), (100, 192)):
        pts = T(rng.uniform(-5, 5, size=(n, s, 3)).astype(np.float32))
        vd = T(rng.normal(size=(n, 3)).astype(np.float32))
        vd = vd / vd.norm(dim=-1, keepdim=True)
        with torch.no_grad():
            raw = qf(pts.to(device), vd.to(device), m)
            want = O.query_field(pts, vd, sd)
        scale = float(want.abs().max())
        close(raw, want, 3e-5, 3e-6 * max(1.0, scale), f"mlp {n}x{s}")
        # the fp32-accurate routes to the same numbers: the split-operand kernel (what no_grad queries run, checked
        # above), the fp32 matrix instruction, the kernel for pre-encoded inputs, and the training forwards that save
        # activations
        from nerf_amd import ops
        flat_pts, dirs = pts.reshape(-1, 3).to(device), vd.to(device)
        with torch.no_grad():
            x_enc = torch.cat([ops.encode(flat_pts, 10), ops.encode(dirs[:, None].expand(n, s, 3).reshape(-1, 3), 4)], -1)
            raw32 = ops.mlp_forward_encoded(x_enc, m._packed(), True, 5)
            raw16 = ops.mlp_forward_x4(flat_pts, dirs, m._packed_x4(), True, 5, s)
            raw3 = ops.mlp_forward_bf16x3(flat_pts, dirs, m._packed_bf16x3(), True, 5, s)
            raw2 = ops.mlp_forward_f16x2(flat_pts, dirs, m._packed_f16x2(), m._packed_bf16x3(), True, 5, s)
        close(raw32.reshape(n, s, 4), want, 3e-5, 3e-6 * max(1.0, scale), f"mlp on pre-encoded inputs {n}x{s}")
        close(raw16.reshape(n, s, 4), want, 3e-5, 3e-6 * max(1.0, scale), f"mlp fp32 matrix instruction {n}x{s}")
        close(raw3.reshape(n, s, 4), want, 3e-5, 3e-6 * max(1.0, scale), f"mlp three-piece bf16 kernel {n}x{s}")
        close(raw2.reshape(n, s, 4), raw, 0, 0, f"mlp two-piece fp16 kernel {n}x{s} (same kernel as the query)")
        for dtype in ("f32", "f32_mfma"):
            m.mlp_dtype = dtype
            raw_train = qf(pts.to(device), vd.to(device), m)
            assert raw_train.requires_grad
            close(raw_train, want, 3e-5, 3e-6 * max(1.0, scale), f"mlp training forward ({dtype}) {n}x{s}")
        m.mlp_dtype = "f32"


def test_split_operand_kernel_is_as_accurate_as_the_fp32_matrix_instruction(device):
    """The three-piece kernel (mlp_dtype = 'bf16x3'; training forward of the default) rebuilds every fp32 product from
    exact three-way bf16 splits (six piece products, fp32 sums).  Against an fp64 evaluation its error is that of the fp32 matrix-instruction kernel (within 1.5x, or below
    1e-6 of the outputs' scale), with and without the view branch, on random and on peaky weights; and a model told to
    use the fp32 matrix instruction (mlp_dtype = 'f32_mfma') renders within the usual summation-order distance."""
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import NeRF
    rng = np.random.Generator(np.random.PCG64(77))
    qf = _query_fn()
    for use_vd, peak, n, s in ((True, 1.0, 300, 64), (True, 30.0, 130, 33), (False, 1.0, 257, 17)):
        if use_vd:
            sd = _peaky(O.random_state_dict(400 + n), peak)
            m = _model(sd, device)
        else:
            sd = O.random_state_dict(401, use_viewdirs=False)
            sd = {k: v for k, v in sd.items() if not k.startswith("views_linears")}
            m = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
            m.load_state_dict(sd, strict=False)
            m = m.to(device)
        pts = T(rng.uniform(-4, 4, size=(n, s, 3)).astype(np.float32))
        vd = T(rng.normal(size=(n, 3)).astype(np.float32))
        vd = vd / vd.norm(dim=-1, keepdim=True)
        truth = O.query_field(pts.double(), vd.double() if use_vd else None, {k: v.double() for k, v in sd.items()})
        flat, dirs = pts.reshape(-1, 3).to(device), vd.to(device) if use_vd else None
        with torch.no_grad():
            r3 = ops.mlp_forward_bf16x3(flat, dirs, m._packed_bf16x3(), use_vd, 5, s).cpu().double().reshape(truth.shape)
            r32 = ops.mlp_forward_x4(flat, dirs, m._packed_x4(), use_vd, 5, s).cpu().double().reshape(truth.shape)
            m.mlp_dtype = 'bf16x3'
            default = qf(pts.to(device), dirs, m)
            m.mlp_dtype = 'f32_mfma'
            forced = qf(pts.to(device), dirs, m)
            m.mlp_dtype = 'f32'
        scale = float(truth.abs().max())
        e3, e32 = float((r3 - truth).abs().max()), float((r32 - truth).abs().max())
        assert e3 <= max(1.5 * e32, 1e-6 * scale), (use_vd, peak, e3, e32, scale)
        assert torch.equal(default.cpu().double(), r3) and torch.equal(forced.cpu().double(), r32)
        close(default, forced.cpu(), 3e-5, 3e-6 * max(1.0, scale), "split-operand kernel vs fp32 matrix instruction")


def _rng_inputs(n, sc, si, perturb, std):
    np.random.seed(0)
    t_rand = torch.Tensor(np.random.rand(n, sc)) if perturb > 0 else None
    np.random.seed(0)
    u = torch.Tensor(np.random.rand(n, si)) if (perturb > 0 and si > 0) else None
    nc = nf = None
    if std > 0:
        np.random.seed(0)
        nc = torch.Tensor(np.random.rand(n, sc) * std)
        np.random.seed(0)
        nf = torch.Tensor(np.random.rand(n, sc + si) * std)
    return t_rand, u, nc, nf


def test_render_matches_reference_outputs(device):
    """render() through the public API on the configurations of the benchmark, against the reference's own outputs."""
    from nerf_amd.run_nerf import render
    g = golden("march")
    mc = _model(_peaky(O.random_state_dict(31), 40.0), device)
    mf = _model(_peaky(O.random_state_dict(32), 60.0), device)
    qf = _query_fn()
    for tag in ("lego_train", "lego_test", "fern_train", "lindisp", "coarse_only", "stress"):
        sc, si, perturb, std, white, lindisp, near, far, ndc, H, W, focal = g[tag + ".cfg"]
        sc, si, H, W = int(sc), int(si), int(H), int(W)
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        rays = T(g[tag + ".rays"]).to(device)
        kw = dict(network_fn=mc, network_query_fn=qf, N_samples=sc, N_importance=si, network_fine=mf, perturb=perturb,
                  raw_noise_std=std, white_bkgd=bool(white), lindisp=bool(lindisp), retraw=True, use_viewdirs=True,
                  ndc=bool(ndc), near=near, far=far, pytest=(perturb > 0 or std > 0))
        with torch.no_grad():
            rgb, disp, acc, extras = render(H, W, K, chunk=1 << 15, rays=rays, **kw)
        got = dict(extras, rgb_map=rgb, disp_map=disp, acc_map=acc)
        for k in [k[len(tag) + 1:] for k in g if k.startswith(tag + ".") and k.split(".")[1] not in ("rays", "packed", "cfg")]:
            want = g[f"{tag}.{k}"]
            if k.startswith("disp"):
                close(got[k], want, 1e-3, 1e-5, f"{tag} {k}")
            elif k == "raw":
                # the fine depths inherit ulp-level differences of the coarse weights, and a fine sample that lands
                # on the other side of a density spike changes its raw value by the network's (steep) slope: bound
                # the bulk tightly and the few outliers loosely
                gk, dk = got[k].cpu().numpy(), np.abs(got[k].cpu().numpy() - want)
                ok = dk <= 2e-3 + 1e-3 * np.abs(want)
                assert ok.mean() >= 0.998, (tag, ok.mean())
                assert dk.max() <= 0.05 * max(1.0, np.abs(want).max()), (tag, dk.max())
            else:
                close(got[k], want, 2e-4, 2e-5, f"{tag} {k}")


def test_render_whole_image_and_chunk_independence(device):
    from nerf_amd.run_nerf import render
    g = golden("image")
    mc = _model(_peaky(O.random_state_dict(31), 40.0), device)
    mf = _model(_peaky(O.random_state_dict(32), 60.0), device)
    H, W, f = g["hwf"]
    H, W = int(H), int(W)
    K = np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]])
    kw = dict(network_fn=mc, network_query_fn=_query_fn(), N_samples=64, N_importance=128, network_fine=mf, perturb=0.0,
              white_bkgd=True, use_viewdirs=True, ndc=False, near=2.0, far=6.0)
    with torch.no_grad():
        rgb, disp, acc, extras = render(H, W, K, chunk=50, c2w=T(g["pose"]).to(device), **kw)
        rgb2, disp2, acc2, _ = render(H, W, K, chunk=1 << 15, c2w=T(g["pose"]).to(device), **kw)
    assert rgb.shape == (H, W, 3) and disp.shape == (H, W) and extras["z_std"].shape == (H, W)
    close(rgb, g["rgb"], 2e-4, 2e-5, "image rgb")
    close(acc, g["acc"], 2e-4, 2e-5, "image acc")
    close(disp, g["disp"], 1e-3, 1e-5, "image disp")
    close(extras["z_std"], g["z_std"], 1e-3, 1e-5, "image z_std")
    exact(rgb, rgb2, "chunk size must not change a deterministic render")
    exact(disp, disp2, "chunk size (disp)")


def test_full_size_chunks_are_independent_and_match_the_oracle(device):
    """At the benchmark's chunk size (32768 rays x 64+128 samples) and at the stress shape (256+512 samples, chunk
    65536): a deterministic render does not depend on how the rays are chunked (bit for bit), and a random subset of
    its rays matches the oracle."""
    from nerf_amd.run_nerf import render
    sd_c, sd_f = _peaky(O.random_state_dict(31), 40.0), _peaky(O.random_state_dict(32), 60.0)
    mc, mf = _model(sd_c, device), _model(sd_f, device)
    qf = _query_fn()
    rng = np.random.Generator(np.random.PCG64(21))
    H = W = 400
    K = np.array([[555.5555, 0, 200.0], [0, 555.5555, 200.0], [0, 0, 1]])
    o, d = O.pixel_rays(H, W, K, O.orbit_pose(77.0, -30.0, 4.0)[:3, :4])
    for n, sc, si, big, small in ((40000, 64, 128, 1 << 15, 4096), (3000, 256, 512, 1 << 16, 1000)):
        pick = torch.from_numpy(rng.choice(H * W, size=n, replace=False))
        rays = torch.stack([o.reshape(-1, 3)[pick], d.reshape(-1, 3)[pick]], 0)
        kw = dict(network_fn=mc, network_query_fn=qf, N_samples=sc, N_importance=si, network_fine=mf, perturb=0.0,
                  white_bkgd=True, use_viewdirs=True, ndc=False, near=2.0, far=6.0)
        with torch.no_grad():
            a = render(H, W, K, chunk=big, rays=rays.to(device), **kw)
            b = render(H, W, K, chunk=small, rays=rays.to(device), **kw)
        for x, y, what in zip(a[:3], b[:3], ("rgb", "disp", "acc")):
            exact(x, y, f"{what}: chunk {big} vs {small} at {sc}+{si}")
        exact(a[3]["z_std"], b[3]["z_std"], "z_std chunking")
        sub = torch.from_numpy(rng.choice(n, size=96, replace=False))
        with torch.no_grad():
            want = O.render_chunks(O.pack_rays(rays[0][sub], rays[1][sub], 2.0, 6.0, use_viewdirs=True), 1 << 15, sd_coarse=sd_c,
                                   sd_fine=sd_f, n_coarse=sc, n_fine=si, white_bkgd=True)
        close(a[0][sub.to(device)], want["rgb_map"], 2e-4, 2e-5, f"rgb vs oracle at {sc}+{si}")
        close(a[2][sub.to(device)], want["acc_map"], 2e-4, 2e-5, f"acc vs oracle at {sc}+{si}")


# ---------------------------------------------------------------------------------------------
# gradients
# ---------------------------------------------------------------------------------------------
def _grad_close(got, want, what, rtol=2e-3, atol_frac=2e-5):
    """Gradients are sums over points of products of O(1) terms: compare against the tensor's own scale."""
    got = got.detach().cpu().numpy()
    want = want.detach().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol_frac * scale, err_msg=what)


# the arithmetics a training step can run: split operands on the 16-bit matrix instructions (the default: two fp16 pieces
# in the forward and backward-data stages, three bf16 pieces in the weight-gradient stage; 'bf16x3': three pieces
# throughout), or the fp32 matrix instruction throughout
TRAIN_DTYPES = pytest.mark.parametrize("mlp_dtype", ["f32", "f32_mfma"], ids=["split", "mfma32"])
TRAIN_DTYPES_ALL = pytest.mark.parametrize("mlp_dtype", ["f32", "bf16x3", "f32_mfma"], ids=["split", "split3", "mfma32"])


@TRAIN_DTYPES_ALL
def test_mlp_backward_matches_reference_and_oracle(device, mlp_dtype):
    g = golden("mlp")
    qf = _query_fn()
    sd = O.random_state_dict(21)
    m = _model(sd, device)
    m.mlp_dtype = mlp_dtype
    pts, vd, G = T(g["vd_pts"]), T(g["vd_viewdirs"]), T(g["vd_G"])
    raw = qf(pts.to(device), vd.to(device), m)
    (raw * G.to(device)).sum().backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (O.query_field(pts, vd, params) * G).sum().backward()
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        _grad_close(p.grad, params[k].grad, "grad vs oracle " + k)
        flat = p.grad.reshape(-1)
        ref = g[f"vd_gsample.{k}"]
        scale = max(float(np.abs(params[k].grad.numpy()).max()), 1e-12)
        np.testing.assert_allclose(flat[:: max(1, flat.numel() // 64)][:64].cpu().numpy(), ref, rtol=2e-3, atol=2e-5 * scale,
                                   err_msg="grad vs reference " + k)
        sums = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
        np.testing.assert_allclose(sums[1], g[f"vd_gsum.{k}"][1], rtol=1e-4)


def _relu_margin(sd, pts, vd, use_vd):
    """Per point, the smallest |pre-activation| over every ReLU unit of the network, from an fp64 evaluation.  A unit
    within fp32 rounding of zero may land on either side in any fp32 evaluation (this kernel, another tiling of it, the
    CPU), which moves the gradients by that point's whole contribution: strict gradient comparisons leave such points
    out by giving them a zero upstream gradient."""
    F = torch.nn.functional
    sd = {k: v.double() for k, v in sd.items()}
    flat = pts.double().reshape(-1, 3)
    x = O.encode(flat, 10)
    h, margin = x, torch.full((flat.shape[0],), float("inf"), dtype=torch.float64)
    for i in range(8):
        z = F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"])
        margin = torch.minimum(margin, z.abs().amin(-1))
        h = torch.relu(z)
        if i == 4:
            h = torch.cat([x, h], -1)
    if use_vd:
        dirs = O.encode(vd.double()[:, None].expand(pts.shape).reshape(-1, 3), 4)
        feat = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
        z = F.linear(torch.cat([feat, dirs], -1), sd["views_linears.0.weight"], sd["views_linears.0.bias"])
        margin = torch.minimum(margin, z.abs().amin(-1))
    return margin.reshape(pts.shape[:-1])


@TRAIN_DTYPES_ALL
def test_mlp_backward_sizes_and_no_view_branch(device, mlp_dtype):
    """Ragged point counts (tile padding, empty K slices), the output_linear head, and batches large enough for the
    weight-gradient stage to cut every task into its full number of slices (57,600 and 65,536 points), against oracle
    autograd in fp64.  Points with a ReLU unit within 1e-5 of zero get a zero upstream gradient (see _relu_margin)."""
    rng = np.random.Generator(np.random.PCG64(10))
    qf = _query_fn()
    for use_vd, n, s in ((True, 3, 5), (True, 70, 64), (False, 33, 8), (True, 300, 192), (True, 1024, 64)):
        if use_vd:
            sd = _peaky(O.random_state_dict(50 + n), 5.0)
            m = _model(sd, device)
        else:
            sd = O.random_state_dict(60, use_viewdirs=False)
            sd = {k: v for k, v in sd.items() if not k.startswith("views_linears")}
            from nerf_amd.run_nerf_helpers import NeRF
            m = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
            m.load_state_dict(sd, strict=False)
            m = m.to(device)
        m.mlp_dtype = mlp_dtype
        pts = T(rng.uniform(-3, 3, size=(n, s, 3)).astype(np.float32))
        vd = T(rng.normal(size=(n, 3)).astype(np.float32))
        vd = vd / vd.norm(dim=-1, keepdim=True)
        raw = qf(pts.to(device), vd.to(device) if use_vd else None, m)
        G = T(rng.normal(size=tuple(raw.shape)).astype(np.float32))
        G = G * (_relu_margin(sd, pts, vd, use_vd) > 1e-5)[..., None].float()     # (a handful of points per million units)
        (raw * G.to(device)).sum().backward()
        # truth in fp64: with thousands of points the fp32 CPU reduction is itself ~1e-4 of the scale off
        params = {k: v.double().requires_grad_(True) for k, v in sd.items()}
        (O.query_field(pts.double(), vd.double() if use_vd else None, params) * G.double()).sum().backward()
        p32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        (O.query_field(pts, vd if use_vd else None, p32) * G).sum().backward()
        for k, p in m.named_parameters():
            if k.startswith("views_linears") and not use_vd:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0
                continue
            truth = params[k].grad
            scale = float(truth.abs().max())
            err_gpu = float((p.grad.cpu().double() - truth).abs().max()) / scale
            err_cpu = float((p32[k].grad.double() - truth).abs().max()) / scale
            assert err_gpu <= max(3e-5, 2.0 * err_cpu), (k, n, s, err_gpu, err_cpu)


@TRAIN_DTYPES
def test_mlp_backward_at_the_data_parallel_batch(device, mlp_dtype):
    """The per-GPU batch of the data-parallel lego configuration: 8192 rays x 64 coarse samples = 524,288 points through
    one network in ONE backward pass, where every slice of the weight-gradient stage is 8 times longer than at the
    largest size above.  Truth = oracle autograd in fp64, accumulated over chunks of 32,768 points (gradients are sums
    over points); the fp32 CPU gradients are accumulated the same way in fp32.  Same ReLU-margin mask and the same
    bound as at the smaller sizes (run_nerf.py:862-874, run_nerf_helpers.py:109-133)."""
    rng = np.random.Generator(np.random.PCG64(12))
    qf = _query_fn()
    n, s, step = 8192, 64, 512   # 512 rays x 64 samples = 32,768 points per CPU chunk
    sd = _peaky(O.random_state_dict(8192), 5.0)
    m = _model(sd, device)
    m.mlp_dtype = mlp_dtype
    pts = T(rng.uniform(-3, 3, size=(n, s, 3)).astype(np.float32))
    vd = T(rng.normal(size=(n, 3)).astype(np.float32))
    vd = vd / vd.norm(dim=-1, keepdim=True)
    G = T(rng.normal(size=(n, s, 4)).astype(np.float32))
    keep = torch.cat([_relu_margin(sd, pts[a:a + step], vd[a:a + step], True) > 1e-5 for a in range(0, n, step)], 0)
    assert float(keep.float().mean()) > 0.75, "the mask must leave most of the batch in (about 2,300 ReLU units per point)"
    G = G * keep[..., None].float()
    raw = qf(pts.to(device), vd.to(device), m)
    assert raw.shape == (n, s, 4)
    (raw * G.to(device)).sum().backward()
    p64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    acc32 = {k: torch.zeros_like(v) for k, v in sd.items()}
    for a in range(0, n, step):
        (O.query_field(pts[a:a + step].double(), vd[a:a + step].double(), p64) * G[a:a + step].double()).sum().backward()
        p32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        (O.query_field(pts[a:a + step], vd[a:a + step], p32) * G[a:a + step]).sum().backward()
        for k in acc32:
            acc32[k] += p32[k].grad
    for k, p in m.named_parameters():
        truth = p64[k].grad
        scale = float(truth.abs().max())
        err_gpu = float((p.grad.cpu().double() - truth).abs().max()) / scale
        err_cpu = float((acc32[k].double() - truth).abs().max()) / scale
        assert err_gpu <= max(3e-5, 2.0 * err_cpu), (k, err_gpu, err_cpu)


def test_mlp_backward_at_scale(device):
    """57,600 points = 1.2e8 ReLU units: a handful of them sit within fp32 rounding of zero, and which side they fall on
    differs between ANY two fp32 evaluations (this kernel, the CPU, fp64), each flip moving the gradients by ~1e-3 of
    their norm.  So against the fp64 truth only a flip-tolerant bound holds (the CPU fp32 path misses it by as much);
    the strict check at this size is additivity: the gradient of the batch equals the sum over its thirds, computed by
    the same kernels from the same forward values, which exercises the sliced weight-gradient reduction at scale."""
    rng = np.random.Generator(np.random.PCG64(11))
    qf = _query_fn()
    n, s = 300, 192
    sd = _peaky(O.random_state_dict(350), 5.0)
    m = _model(sd, device)
    pts = T(rng.uniform(-3, 3, size=(n, s, 3)).astype(np.float32)).to(device)
    vd = T(rng.normal(size=(n, 3)).astype(np.float32))
    vd = (vd / vd.norm(dim=-1, keepdim=True)).to(device)
    G = T(rng.normal(size=(n, s, 4)).astype(np.float32)).to(device)
    (qf(pts, vd, m) * G).sum().backward()
    whole = {k: p.grad.clone() for k, p in m.named_parameters()}
    parts = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in whole.items()}
    for a in range(0, n, 100):
        m.zero_grad()
        (qf(pts[a:a + 100], vd[a:a + 100], m) * G[a:a + 100]).sum().backward()
        for k, p in m.named_parameters():
            parts[k] += p.grad.double()
    for k in whole:
        scale = float(parts[k].abs().max())
        assert float((whole[k].double() - parts[k]).abs().max()) <= 2e-5 * scale, k
    params = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    (O.query_field(pts.cpu().double(), vd.cpu().double(), params) * G.cpu().double()).sum().backward()
    for k in whole:
        truth = params[k].grad
        rel = float((whole[k].cpu().double() - truth).norm() / truth.norm())
        assert rel <= 1e-2, (k, rel)


def test_train_step_matches_reference(device):
    """One lego-shaped optimisation step through the public API: loss, every gradient, post-Adam weights."""
    from nerf_amd.run_nerf import render
    from nerf_amd.run_nerf_helpers import img2mse, mse2psnr
    g = golden("train_step")
    mc = _model(_peaky(O.random_state_dict(41), 20.0), device)
    mf = _model(_peaky(O.random_state_dict(42), 20.0), device)
    rays, target = T(g["rays"]).to(device), T(g["target"]).to(device)
    H = W = 400
    focal = 555.5555155968841
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    params = list(mc.parameters()) + list(mf.parameters())
    opt = torch.optim.Adam(params=params, lr=5e-4, betas=(0.9, 0.999))
    kw = dict(network_fn=mc, network_query_fn=_query_fn(), N_samples=64, N_importance=128, network_fine=mf, perturb=1.0,
              raw_noise_std=0.0, white_bkgd=True, use_viewdirs=True, ndc=False, near=2.0, far=6.0, pytest=True, retraw=True)
    rgb, disp, acc, extras = render(H, W, K, chunk=1 << 15, rays=rays, verbose=True, **kw)
    opt.zero_grad()
    img_loss = img2mse(rgb, target)
    loss = img_loss + img2mse(extras["rgb0"], target)
    close(loss, g["loss"], 1e-5, 0, "loss")
    close(mse2psnr(img_loss), g["psnr"], 1e-5, 0, "psnr")
    loss.backward()
    names = [("c." + k) for k, _ in mc.named_parameters()] + [("f." + k) for k, _ in mf.named_parameters()]
    for k, p in zip(names, params):
        flat = p.grad.reshape(-1)
        want = g["gsample." + k]
        got = flat[:: max(1, flat.numel() // 64)][:64].cpu().numpy()
        # The fine network sees sample depths that inherit ulp-level differences of the coarse weights, and first-layer
        # gradients are sums of high-frequency terms that cancel almost completely: compare each tensor in the L2
        # sense against its own magnitude (kernel-level accuracy is pinned against fp64 in the tests above).
        rel = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
        assert rel <= 2e-2, (k, rel)
        np.testing.assert_allclose(flat.double().abs().sum().item(), g["gsum." + k][1], rtol=2e-3, err_msg="|grad| " + k)
    before = [p.detach().reshape(-1)[:: max(1, p.numel() // 64)][:64].cpu().numpy().copy() for p in params]
    opt.step()
    checked = 0
    for (k, p), w0 in zip(zip(names, params), before):
        flat = p.detach().reshape(-1)
        got = flat[:: max(1, flat.numel() // 64)][:64].cpu().numpy()
        want, g_ref = g["wsample." + k], g["gsample." + k]
        # Adam's first step moves a weight by -lr * g / (|g| + eps); these gradients are of the order of eps = 1e-8, so
        # the step is a fraction of lr that follows g.  Compare the UPDATE (not the weight, whose own size would hide
        # it): a skipped step gives zero, a wrong sign the negative.  Entries whose reference update is small against
        # the tensor's largest are left out of the sign check; weights of ~0.05 resolve an update to ~4e-9.
        upd, upd_ref = (got - w0).astype(np.float64), (want - w0).astype(np.float64)
        big = float(np.abs(upd_ref).max())
        assert big > 2e-8, (k, big)
        rel = np.linalg.norm(upd - upd_ref) / np.linalg.norm(upd_ref)
        assert rel <= 5e-2, (k, rel)
        keep = np.abs(upd_ref) >= 0.2 * big
        assert np.array_equal(np.sign(upd[keep]), np.sign(upd_ref[keep])), k
        checked += int(keep.sum())
    assert checked > 100
    # the repacked streams must follow the optimiser step
    with torch.no_grad():
        rgb2, _, _, _ = render(H, W, K, chunk=1 << 15, rays=rays, **dict(kw, retraw=False))
    assert float((rgb2 - rgb.detach()).abs().max()) > 0.0


def test_fern_train_step_matches_oracle(device):
    """Forward-facing configuration (NDC warp, density noise, 64+64 samples, no white background): loss and every
    gradient of one step against autograd through the oracle on the same rays and the same random draws."""
    from nerf_amd.run_nerf import render
    from nerf_amd.run_nerf_helpers import img2mse
    g = golden("march")
    sc, si, perturb, std, white, lindisp, near, far, ndc, H, W, focal = g["fern_train.cfg"]
    sc, si, H, W = int(sc), int(si), int(H), int(W)
    assert ndc == 1.0 and std == 1.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    sd_c, sd_f = _peaky(O.random_state_dict(31), 40.0), _peaky(O.random_state_dict(32), 60.0)
    mc, mf = _model(sd_c, device), _model(sd_f, device)
    rays = T(g["fern_train.rays"])
    n = rays.shape[1]
    target = T(np.random.Generator(np.random.PCG64(3)).uniform(0, 1, size=(n, 3)).astype(np.float32))
    kw = dict(network_fn=mc, network_query_fn=_query_fn(), N_samples=sc, N_importance=si, network_fine=mf, perturb=perturb,
              raw_noise_std=std, white_bkgd=bool(white), lindisp=False, retraw=True, use_viewdirs=True, ndc=True,
              near=near, far=far, pytest=True)
    rgb, disp, acc, extras = render(H, W, K, chunk=1 << 15, rays=rays.to(device), **kw)
    loss = img2mse(rgb, target.to(device)) + img2mse(extras["rgb0"], target.to(device))
    loss.backward()

    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    t_rand, u, nc, nf = _rng_inputs(n, sc, si, perturb, std)
    packed = O.pack_rays(rays[0], rays[1], near, far, use_viewdirs=True, ndc=(H, W, K[0][0]))
    want = O.render_chunks(packed, 1 << 15, sd_coarse=pc, sd_fine=pf, n_coarse=sc, n_fine=si, perturb=perturb,
                           white_bkgd=bool(white), t_rand=t_rand, u=u, noise_coarse=nc, noise_fine=nf, retraw=True)
    # (bit-for-bit on the machine that made the fixture; another host's thread count changes the GEMM blocking)
    close(want["rgb_map"], g["fern_train.rgb_map"], 2e-4, 2e-5, "oracle reproduces the reference's fern forward")
    loss_o = O.mse(want["rgb_map"], target) + O.mse(want["rgb0"], target)
    loss_o.backward()
    close(loss, loss_o.detach(), 2e-5, 0, "fern loss")
    for net, model, ref in (("c", mc, pc), ("f", mf, pf)):
        for k, p in model.named_parameters():
            a, b = p.grad.cpu().double().reshape(-1), ref[k].grad.double().reshape(-1)
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert rel <= 2e-2, (net, k, rel)
            np.testing.assert_allclose(float(a.abs().sum()), float(b.abs().sum()), rtol=3e-3, err_msg=f"|grad| {net}.{k}")


# ---------------------------------------------------------------------------------------------
# the remaining public surface
# ---------------------------------------------------------------------------------------------
def test_nerf_forward_on_encoded_inputs_and_generic_run_network(device):
    """NeRF.forward(x) as upstream calls it, and run_network with a callable that is not a NeRF module."""
    from nerf_amd.run_nerf import run_network
    from nerf_amd.run_nerf_helpers import get_embedder
    g = golden("mlp")
    sd = O.random_state_dict(21)
    m = _model(sd, device)
    pts, vd = T(g["vd_pts"]), T(g["vd_viewdirs"])
    flat = pts.reshape(-1, 3)
    dirs = vd[:, None].expand(pts.shape).reshape(-1, 3)
    x = torch.cat([O.encode(flat, 10), O.encode(dirs, 4)], -1)
    with torch.no_grad():
        out = m(x.to(device))
    assert out.shape == (flat.shape[0], 4)
    close(out.reshape(24, 8, 4), g["vd_raw"], 2e-5, 2e-5, "NeRF.forward(encoded)")
    e, _ = get_embedder(10, 0)
    ed, _ = get_embedder(4, 0)
    calls = []

    def wrapped(z):  # not a NeRF instance: takes the generic (encode, concatenate, slice by netchunk) route
        calls.append(z.shape)
        return m(z)
    with torch.no_grad():
        out2 = run_network(pts.to(device), vd.to(device), wrapped, e, ed, netchunk=50)
    assert len(calls) == 4 and calls[0] == (50, 90)
    close(out2, g["vd_raw"], 2e-5, 2e-5, "generic run_network")
    # gradients through the encoded entry
    xg = x.to(device)
    (m(xg) * T(g["vd_G"]).reshape(-1, 4).to(device)).sum().backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (O.query_field(pts, vd, params) * T(g["vd_G"])).sum().backward()
    for k, p in m.named_parameters():
        _grad_close(p.grad, params[k].grad, "encoded-entry grad " + k)


def test_side_stream_and_graph_capture(device):
    """Every entry point only enqueues on the caller's current stream: a render issued on a side stream, and one
    captured into a HIP graph and replayed on new rays, give the bits of the plain call."""
    from nerf_amd.run_nerf import render
    mc = _model(_peaky(O.random_state_dict(31), 40.0), device)
    mf = _model(_peaky(O.random_state_dict(32), 60.0), device)
    kw = dict(network_fn=mc, network_query_fn=_query_fn(), N_samples=64, N_importance=128, network_fine=mf, perturb=0.0,
              white_bkgd=True, use_viewdirs=True, ndc=False, near=2.0, far=6.0)
    H = W = 40
    K = np.array([[55.0, 0, 20.0], [0, 55.0, 20.0], [0, 0, 1]])

    def rays_of(theta):
        o, d = O.pixel_rays(H, W, K, O.orbit_pose(theta, -30.0, 4.0)[:3, :4])
        return torch.stack([o.reshape(-1, 3), d.reshape(-1, 3)], 0).to(device)
    r1, r2 = rays_of(10.0), rays_of(200.0)
    with torch.no_grad():
        want1 = render(H, W, K, chunk=1024, rays=r1, **kw)
        want2 = render(H, W, K, chunk=1024, rays=r2, **kw)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            got1 = render(H, W, K, chunk=1024, rays=r1, **kw)
        side.synchronize()
        for a, b, what in zip(got1[:3], want1[:3], ("rgb", "disp", "acc")):
            exact(a, b, "side stream " + what)

        static = r1.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = render(H, W, K, chunk=1024, rays=static, **kw)
        static.copy_(r2)
        graph.replay()
        torch.cuda.synchronize(device)
        for a, b, what in zip(out[:3], want2[:3], ("rgb", "disp", "acc")):
            exact(a, b, "graph replay " + what)
        exact(out[3]["z_std"], want2[3]["z_std"], "graph replay z_std")


def test_gradient_with_respect_to_inputs_is_never_silently_dropped(device):
    """NeRF.forward(x) with x.requires_grad gives d/dx (layer-by-layer route, checked against the oracle); raw positions
    that require a gradient are refused by the fused kernels and by the encoder instead of getting none."""
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import get_embedder
    sd = O.random_state_dict(21)
    m = _model(sd, device)
    rng = np.random.Generator(np.random.PCG64(5))
    x = T(rng.normal(size=(50, 90)).astype(np.float32))
    xg = x.to(device).requires_grad_(True)
    m(xg).square().sum().backward()
    xo = x.clone().requires_grad_(True)
    par = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.mlp_apply(par, xo, 63, 27).square().sum().backward()
    _grad_close(xg.grad, xo.grad, "d/dx of NeRF.forward", rtol=1e-3, atol_frac=1e-4)
    for k, p in m.named_parameters():
        _grad_close(p.grad, par[k].grad, "param grad on the d/dx route " + k, rtol=1e-3, atol_frac=1e-4)
    pts = torch.zeros(8, 4, 3, device=device, requires_grad=True)
    vd = torch.zeros(8, 3, device=device)
    with pytest.raises(NotImplementedError):
        m.query(pts.reshape(-1, 3), vd, 4)
    with pytest.raises(NotImplementedError):
        get_embedder(10, 0)[0](pts)
    with torch.no_grad():
        assert m.query(pts.reshape(-1, 3), vd, 4).shape == (32, 4)


def test_other_network_shapes_run_layer_by_layer(device):
    """--netdepth / --netwidth / --multires other than the shipped configs: encoding kernel + the library's generic
    linear-layer kernel (ops.linear), same numbers as the oracle, gradients included, through the same public entry points."""
    from nerf_amd.run_nerf import render, run_network
    from nerf_amd.run_nerf_helpers import NeRF, get_embedder
    rng = np.random.Generator(np.random.PCG64(14))
    e, ch = get_embedder(6, 0)
    ed, chd = get_embedder(2, 0)
    sd = O.random_state_dict(70, depth=4, width=128, in_pts=ch, in_dirs=chd, skips=(2,))
    m = NeRF(D=4, W=128, input_ch=ch, input_ch_views=chd, output_ch=5, skips=[2], use_viewdirs=True)
    m.load_state_dict(sd)
    m = m.to(device)
    assert not m.fused_arch()
    qf = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=e, embeddirs_fn=ed, netchunk=1 << 16)
    pts = T(rng.uniform(-2, 2, size=(40, 16, 3)).astype(np.float32))
    vd = T(rng.normal(size=(40, 3)).astype(np.float32))
    vd = vd / vd.norm(dim=-1, keepdim=True)
    raw = qf(pts.to(device), vd.to(device), m)
    want_p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want = O.query_field(pts, vd, want_p, n_freq_pts=6, n_freq_dirs=2, skips=(2,))
    close(raw, want, 1e-4, 1e-5, "raw of a 4x128 network")
    raw.square().sum().backward()
    want.square().sum().backward()
    for k, p in m.named_parameters():
        _grad_close(p.grad, want_p[k].grad, "grad " + k, rtol=1e-3, atol_frac=1e-4)
    rays = torch.stack([T(rng.normal(size=(30, 3)).astype(np.float32)) * 0.1 + torch.tensor([0.0, 0.0, 4.0]),
                        T(rng.normal(size=(30, 3)).astype(np.float32)) * 0.1 - torch.tensor([0.0, 0.0, 1.0])], 0)
    with torch.no_grad():
        rgb, disp, acc, _ = render(8, 8, np.eye(3), chunk=1 << 15, rays=rays.to(device), network_fn=m, network_query_fn=qf,
                                   N_samples=16, N_importance=8, network_fine=m, perturb=0.0, white_bkgd=True,
                                   use_viewdirs=True, ndc=False, near=2.0, far=6.0)
        got = O.render_chunks(O.pack_rays(rays[0], rays[1], 2.0, 6.0, use_viewdirs=True), 1 << 15, sd_coarse=sd, sd_fine=sd,
                              n_coarse=16, n_fine=8, white_bkgd=True, n_freq_pts=6, n_freq_dirs=2, skips=(2,))
    close(rgb, got["rgb_map"], 2e-4, 2e-5, "render with a 4x128 network")


def test_render_option_paths(device):
    """Branches of render()/render_rays() the shipped configs do not take: c2w_staticcam (view directions of one
    camera, rays of another: run_nerf.py:120-125), network_fine=None with N_importance > 0 (the coarse network
    answers both passes: run_nerf.py:472), rays without view directions (8 columns), and i_embed=-1 (identity
    encoding: run_nerf_helpers.py:51-52)."""
    from nerf_amd.run_nerf import render, run_network
    from nerf_amd.run_nerf_helpers import NeRF, get_embedder
    sd_c, sd_f = _peaky(O.random_state_dict(31), 6.0, 1.0), _peaky(O.random_state_dict(32), 6.0, 1.0)  # soft density: no spikes between samples
    mc, mf = _model(sd_c, device), _model(sd_f, device)
    qf = _query_fn()
    H = W = 12
    K = np.array([[16.0, 0, 6.0], [0, 16.0, 6.0], [0, 0, 1]])
    cam, fixed = O.orbit_pose(40.0, -30.0, 4.0)[:3, :4], O.orbit_pose(95.0, -20.0, 4.0)[:3, :4]
    kw = dict(network_fn=mc, network_query_fn=qf, N_samples=32, N_importance=24, perturb=0.0, white_bkgd=True,
              ndc=False, near=2.0, far=6.0)
    okw = dict(n_coarse=32, n_fine=24, white_bkgd=True)

    with torch.no_grad():
        rgb, disp, acc, ex = render(H, W, K, c2w=cam.to(device), c2w_staticcam=fixed.to(device), use_viewdirs=True,
                                    network_fine=mf, **kw)
        o_f, d_f = O.pixel_rays(H, W, K, fixed)
        _, d_c = O.pixel_rays(H, W, K, cam)
        want = O.render_chunks(O.pack_rays(o_f, d_f, 2.0, 6.0, use_viewdirs=True, viewdirs_from=d_c), 1 << 15,
                               sd_coarse=sd_c, sd_fine=sd_f, **okw)
    assert rgb.shape == (H, W, 3)
    close(rgb.reshape(-1, 3), want["rgb_map"], 2e-4, 2e-5, "staticcam rgb")
    close(ex["rgb0"].reshape(-1, 3), want["rgb0"], 2e-4, 2e-5, "staticcam rgb0")
    with torch.no_grad():
        plain = render(H, W, K, c2w=fixed.to(device), use_viewdirs=True, network_fine=mf, **kw)[0]
    assert (plain - rgb).abs().max() > 1e-4, "the view directions of the other camera must matter"

    with torch.no_grad():
        rgb, disp, acc, ex = render(H, W, K, c2w=cam.to(device), use_viewdirs=True, network_fine=None, **kw)
        o_c, _ = O.pixel_rays(H, W, K, cam)
        want = O.render_chunks(O.pack_rays(o_c, d_c, 2.0, 6.0, use_viewdirs=True), 1 << 15, sd_coarse=sd_c, sd_fine=None, **okw)
    close(rgb.reshape(-1, 3), want["rgb_map"], 2e-4, 2e-5, "network_fine=None rgb")
    close(acc.reshape(-1), want["acc_map"], 2e-4, 2e-5, "network_fine=None acc")

    # no view directions: 8-column ray records, output_linear head
    sd_n = _peaky_out(O.random_state_dict(33, in_dirs=0, use_viewdirs=False, output_ch=5), 6.0, 1.0)
    mn = _model(sd_n, device, use_viewdirs=False)
    with torch.no_grad():
        rgb, disp, acc, ex = render(H, W, K, c2w=cam.to(device), use_viewdirs=False, network_fine=mn,
                                    **dict(kw, network_fn=mn))
        want = O.render_chunks(O.pack_rays(o_c, d_c, 2.0, 6.0, use_viewdirs=False), 1 << 15, sd_coarse=sd_n, sd_fine=sd_n, **okw)
    close(rgb.reshape(-1, 3), want["rgb_map"], 2e-4, 2e-5, "no-viewdirs rgb")
    close(disp.reshape(-1), want["disp_map"], 1e-3, 1e-5, "no-viewdirs disp")
    # ... and its train step: the fifth raw channel is carried along and gets a zero gradient
    rgb, _, _, ex = render(H, W, K, c2w=cam.to(device), use_viewdirs=False, network_fine=mn, retraw=True, **dict(kw, network_fn=mn))
    assert ex["raw"].shape == (H, W, 32 + 24, 5)
    (rgb.square().mean() + ex["rgb0"].square().mean()).backward()
    par = {k: v.clone().requires_grad_(True) for k, v in sd_n.items()}
    want = O.render_chunks(O.pack_rays(o_c, d_c, 2.0, 6.0, use_viewdirs=False), 1 << 15, sd_coarse=par, sd_fine=par, **okw)
    (want["rgb_map"].square().mean() + want["rgb0"].square().mean()).backward()
    for k, p in mn.named_parameters():
        if par[k].grad is None:       # views_linears exists in the module but takes no part without view directions
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        # (against an fp32 CPU evaluation, itself ~1e-4 of the scale away from the fp64 value on sums over 16,000 points)
        _grad_close(p.grad, par[k].grad, "no-viewdirs grad " + k, rtol=2e-3, atol_frac=4e-4)
    assert float(mn.output_linear.weight.grad[4].abs().max()) == 0.0

    # identity encoding: the network sees raw xyz and raw directions
    e, ch = get_embedder(10, -1)
    ed, chd = get_embedder(4, -1)
    assert (ch, chd) == (3, 3)
    sd_i = _peaky(O.random_state_dict(34, in_pts=3, in_dirs=3), 6.0, 1.0)
    mi = NeRF(D=8, W=256, input_ch=3, input_ch_views=3, output_ch=5, skips=[4], use_viewdirs=True)
    mi.load_state_dict(sd_i)
    mi = mi.to(device)
    qi = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=e, embeddirs_fn=ed, netchunk=1 << 16)
    with torch.no_grad():
        rgb = render(H, W, K, c2w=cam.to(device), use_viewdirs=True, network_fine=mi, **dict(kw, network_fn=mi, network_query_fn=qi))[0]
        want = O.render_chunks(O.pack_rays(o_c, d_c, 2.0, 6.0, use_viewdirs=True), 1 << 15, sd_coarse=sd_i, sd_fine=sd_i,
                               n_freq_pts=-1, n_freq_dirs=-1, **okw)
    close(rgb.reshape(-1, 3), want["rgb_map"], 2e-4, 2e-5, "identity-encoding rgb")


def test_create_nerf_checkpoint_roundtrip_and_render_path(device, tmp_path):
    from nerf_amd.run_nerf import config_parser, create_nerf, render_path, save_checkpoint
    (tmp_path / "exp").mkdir()
    argv = ["--expname", "exp", "--basedir", str(tmp_path), "--dataset_type", "blender", "--use_viewdirs", "--white_bkgd",
            "--N_samples", "16", "--N_importance", "16", "--N_rand", "64"]
    args = config_parser().parse_args(argv)
    kw_train, kw_test, start, grad_vars, opt = create_nerf(args)
    assert start == 0 and len(grad_vars) == 48
    assert set(kw_train) == {"network_query_fn", "perturb", "N_importance", "network_fine", "N_samples", "network_fn",
                             "use_viewdirs", "white_bkgd", "raw_noise_std", "ndc", "lindisp"}
    assert kw_test["perturb"] is False and kw_test["raw_noise_std"] == 0.0 and kw_train["perturb"] == 1.0
    for kw in (kw_train, kw_test):
        kw.update(near=2.0, far=6.0)
    H = W = 8
    K = np.array([[10.0, 0, 4], [0, 10.0, 4], [0, 0, 1]])
    poses = torch.stack([O.orbit_pose(t, -30.0, 4.0) for t in (0.0, 90.0)]).to(device)
    with torch.no_grad():
        rgbs, disps = render_path(poses, [H, W, 10.0], K, 32, kw_test)
    assert rgbs.shape == (2, H, W, 3) and disps.shape == (2, H, W) and np.isfinite(rgbs).all()
    # one optimisation step, save, reload into fresh models: same step counter, same weights, same image
    from nerf_amd.run_nerf import render
    from nerf_amd.run_nerf_helpers import get_rays, img2mse
    o, d = get_rays(H, W, K, poses[0][:3, :4])
    rgb, _, _, extras = render(H, W, K, chunk=32, rays=torch.stack([o.reshape(-1, 3), d.reshape(-1, 3)]), retraw=True, **kw_train)
    loss = img2mse(rgb, torch.zeros_like(rgb)) + img2mse(extras["rgb0"], torch.zeros_like(rgb))
    opt.zero_grad()
    loss.backward()
    opt.step()
    save_checkpoint(str(tmp_path / "exp" / "000001.tar"), 1, kw_train, opt)
    ck = torch.load(str(tmp_path / "exp" / "000001.tar"), map_location="cpu")
    assert set(ck) == {"global_step", "network_fn_state_dict", "network_fine_state_dict", "optimizer_state_dict"}
    assert ck["network_fn_state_dict"]["pts_linears.5.weight"].shape == (256, 319)
    kw2, kw2_test, start2, _, opt2 = create_nerf(args)
    assert start2 == 1
    for a, b in zip(kw_train["network_fn"].parameters(), kw2["network_fn"].parameters()):
        assert torch.equal(a, b)
    assert opt2.state_dict()["state"][0]["step"] == opt.state_dict()["state"][0]["step"]
    kw2_test.update(near=2.0, far=6.0)
    with torch.no_grad():
        a, _ = render_path(poses[:1], [H, W, 10.0], K, 32, kw_test)
        b, _ = render_path(poses[:1], [H, W, 10.0], K, 32, kw2_test)
    assert np.array_equal(a, b)
    # no_reload ignores the checkpoint
    args2 = config_parser().parse_args(argv + ["--no_reload"])
    assert create_nerf(args2)[2] == 0


def test_data_parallel_gradient_equals_full_batch(device):
    """Mean-loss gradients of equal ray shards, averaged, equal the gradient of the whole batch (what the
    bucket all-reduce computes across ranks)."""
    from nerf_amd.run_nerf import render
    from nerf_amd.run_nerf_helpers import img2mse
    rng = np.random.Generator(np.random.PCG64(11))
    mc = _model(_peaky(O.random_state_dict(71), 10.0), device)
    mf = _model(_peaky(O.random_state_dict(72), 10.0), device)
    params = list(mc.parameters()) + list(mf.parameters())
    n = 64
    o = T(rng.normal(size=(n, 3)).astype(np.float32) * 0.1 + np.array([0, 0, 4], np.float32)).to(device)
    d = T(rng.normal(size=(n, 3)).astype(np.float32) * 0.2 + np.array([0, 0, -1], np.float32)).to(device)
    target = T(rng.uniform(size=(n, 3)).astype(np.float32)).to(device)
    t_rand = torch.rand(n, 32, device=device)
    kw = dict(network_fn=mc, network_query_fn=_query_fn(), N_samples=32, N_importance=32, network_fine=mf, perturb=0.0,
              white_bkgd=True, use_viewdirs=True, ndc=False, near=2.0, far=6.0, retraw=True)
    K = np.eye(3)

    def grads(sl):
        for p in params:
            p.grad = None
        rgb, _, _, ex = render(1, 1, K, chunk=1 << 15, rays=torch.stack([o[sl], d[sl]]), **kw)
        (img2mse(rgb, target[sl]) + img2mse(ex["rgb0"], target[sl])).backward()
        return torch.cat([p.grad.reshape(-1) for p in params]).clone()
    full = grads(slice(0, n))
    world = 4
    shards = [grads(slice(r * n // world, (r + 1) * n // world)) for r in range(world)]
    avg = sum(shards) / world
    scale = float(full.abs().max())
    assert float((avg - full).abs().max()) <= 5e-5 * scale


def test_bf16_variant_within_stated_tolerance(device):
    """mlp_dtype='bf16': every product's inputs rounded to bf16, fp32 sums.  Stated tolerance against the fp32
    oracle: raw outputs within 2e-2 of the tensor's scale (max |x|), rendered rgb/acc within 1e-2 absolute."""
    from nerf_amd.run_nerf import render
    rng = np.random.Generator(np.random.PCG64(12))
    qf = _query_fn()
    sd = _peaky(O.random_state_dict(33), 30.0)
    m = _model(sd, device)
    m.mlp_dtype = 'bf16'
    for n, s in ((1, 1), (129, 3), (100, 192), (37, 64)):
        pts = T(rng.uniform(-5, 5, size=(n, s, 3)).astype(np.float32))
        vd = T(rng.normal(size=(n, 3)).astype(np.float32))
        vd = vd / vd.norm(dim=-1, keepdim=True)
        with torch.no_grad():
            raw = qf(pts.to(device), vd.to(device), m).cpu()
            want = O.query_field(pts, vd, sd)
        for ch in range(4):
            scale = float(want[..., ch].abs().max())
            err = float((raw[..., ch] - want[..., ch]).abs().max())
            assert err <= 2e-2 * max(scale, 1e-3), (n, s, ch, err, scale)
    # no view branch
    sd2 = O.random_state_dict(22, use_viewdirs=False)
    sd2 = {k: v for k, v in sd2.items() if not k.startswith("views_linears")}
    from nerf_amd.run_nerf_helpers import NeRF
    m2 = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
    m2.load_state_dict(sd2, strict=False)
    m2 = m2.to(device)
    m2.mlp_dtype = 'bf16'
    pts = T(rng.uniform(-4, 4, size=(50, 9, 3)).astype(np.float32))
    with torch.no_grad():
        raw2 = qf(pts.to(device), None, m2).cpu()
        want2 = O.query_field(pts, None, sd2)
    assert float((raw2 - want2).abs().max()) <= 2e-2 * float(want2.abs().max())
    # rendered image vs the reference fixture
    g = golden("image")
    mc = _model(_peaky(O.random_state_dict(31), 40.0), device)
    mf = _model(_peaky(O.random_state_dict(32), 60.0), device)
    mc.mlp_dtype = mf.mlp_dtype = 'bf16'
    H, W, f = g["hwf"]
    H, W = int(H), int(W)
    K = np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]])
    kw = dict(network_fn=mc, network_query_fn=qf, N_samples=64, N_importance=128, network_fine=mf, perturb=0.0,
              white_bkgd=True, use_viewdirs=True, ndc=False, near=2.0, far=6.0)
    with torch.no_grad():
        rgb, disp, acc, _ = render(H, W, K, chunk=1 << 15, c2w=T(g["pose"]).to(device), **kw)
    assert float(np.abs(rgb.cpu().numpy() - g["rgb"]).max()) <= 1e-2
    assert float(np.abs(acc.cpu().numpy() - g["acc"]).max()) <= 1e-2
    # gradients still flow through the fp32 path when the model is in bf16 mode
    raw = qf(pts.to(device)[:4], None, m2)
    raw.sum().backward()
    assert m2.pts_linears[0].weight.grad is not None


def test_ragged_shapes_against_the_oracle(device):
    """120 random shapes (1..1000 rays; 2..768 samples; flat, empty, single-spike and peaky weights; shared and own
    uniforms) through compositing (forward, gradient), coarse depths, inverse-CDF sampling and fused fine sampling:
    bit-exact wherever the path is (depths, positions, indices, samples, merged depths), stated tolerances elsewhere."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_edges", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_edges.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(120, 7) == 0


def test_single_sample_rays_match_upstream(device):
    """raw2outputs with one sample per ray: upstream's appended 1e10 gap has the empty shape of the other gaps
    (run_nerf.py:309-311), so every sum runs over zero samples: no weights, acc 0, rgb = background, disp NaN."""
    from nerf_amd.run_nerf import raw2outputs
    raw, z, d = torch.randn(4, 1, 4), torch.ones(4, 1), torch.ones(4, 3)
    for white in (False, True):
        ref = O.composite(raw, z, d, None, white)
        got = raw2outputs(raw.to(device), z.to(device), d.to(device), 0, white)
        for a, b in zip(got, ref):
            assert a.shape == b.shape
            assert torch.equal(a.cpu().nan_to_num(nan=-7.0), b.nan_to_num(nan=-7.0))
