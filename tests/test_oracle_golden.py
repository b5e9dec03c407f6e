"""The CPU oracle against outputs of the reference itself (tests/golden, made by oracle/gen_golden.py)."""
import numpy as np
import torch

from conftest import assert_pinned, golden
from oracle import nerf_oracle as O


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_rays_and_ndc():
    g = golden("rays")
    for name in ("lego_half", "lego_full", "fern"):
        H, W, focal = g[name + "_hwf"]
        H, W = int(H), int(W)
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        pose = T(g[name + "_pose"])
        pick = g[name + "_pick"]
        o, d = O.pixel_rays(H, W, K, pose)
        assert_pinned(d.reshape(-1, 3)[pick], g[name + "_d"], name + " d")
        assert torch.equal(o.reshape(-1, 3)[0], pose[:, 3])
        _, dn = O.pixel_rays_np(H, W, K, pose.numpy())
        assert dn.dtype == np.float64
        assert_pinned(dn.reshape(-1, 3)[pick], g[name + "_d_np"], name + " d np")
        if name == "fern":
            no, nd = O.ndc_warp(H, W, K[0][0], 1.0, o, d)
            assert_pinned(no.reshape(-1, 3)[pick], g["fern_ndc_o"], "ndc o")
            assert_pinned(nd.reshape(-1, 3)[pick], g["fern_ndc_d"], "ndc d")


def test_encode():
    g = golden("encode")
    x = T(g["x"])
    assert_pinned(O.encode(x, 10), g["enc10"], "enc10")
    assert_pinned(O.encode(x, 4), g["enc4"], "enc4")
    assert O.encode(x, -1) is x
    assert O.encoded_width(10) == 63 and O.encoded_width(4) == 27 and O.encoded_width(-1) == 3


EMBED_FNS = {ord("s"): torch.sin, ord("c"): torch.cos}


def embedder_cases():
    """The keyword settings of tests/golden/embedder_general.npz -> (name, kwargs, x, reference output)."""
    g = golden("embedder_general")
    for name in sorted(k[:-3] for k in g if k.endswith("_kw")):
        inc, d, mf, nf, log = g[name + "_kw"]
        kw = dict(include_input=bool(inc), input_dims=int(d), max_freq_log2=float(mf), num_freqs=int(nf),
                  log_sampling=bool(log), periodic_fns=[EMBED_FNS[int(c)] for c in g[name + "_fns"]])
        yield name, kw, T(g[name + "_x"]), g[name + "_out"]


def test_embedder_keyword_settings():
    """Embedder with include_input off, linear bands, other function orders / counts, other input widths
    (run_nerf_helpers.py:13-47): the oracle against outputs of the reference's own class."""
    n = 0
    for name, kw, x, want in embedder_cases():
        assert_pinned(O.embed_general(x, **kw), want, name)
        n += 1
    assert n == 7


def test_mlp_forward_and_grads():
    g = golden("mlp")
    sd = O.random_state_dict(21)
    pts, vd = T(g["vd_pts"]), T(g["vd_viewdirs"])
    assert_pinned(O.query_field(pts, vd, sd, netchunk=64), g["vd_raw"], "mlp raw")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (O.query_field(pts, vd, params, netchunk=64) * T(g["vd_G"])).sum().backward()
    for k, p in params.items():
        flat = p.grad.reshape(-1)
        assert_pinned(flat[:: max(1, flat.numel() // 64)][:64], g[f"vd_gsample.{k}"], "grad " + k)
        sums = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
        np.testing.assert_allclose(sums, g[f"vd_gsum.{k}"], rtol=1e-6, atol=1e-9)
    # n_params of the two-net model: 2 x 595,844
    assert sum(v.numel() for v in sd.values()) == 595844


def test_composite_forward_and_grad():
    g = golden("composite")
    z, d, noise = T(g["z"]), T(g["rays_d"]), T(g["noise"])
    for name in ("soft", "peaky", "edge"):
        for white in (0, 1):
            for noisy in (0, 1):
                key = f"{name}_w{white}_n{noisy}"
                raw = T(g[name + "_raw"]).requires_grad_(True)
                out = O.composite(raw, z, d, noise if noisy else None, bool(white))
                for t, nm in zip(out, ("rgb", "disp", "acc", "weights", "depth")):
                    assert_pinned(t, g[f"{key}_{nm}"], f"{key} {nm}")
                (out[0] * T(g[key + "_G"])).sum().backward()
                assert_pinned(raw.grad, g[key + "_draw"], key + " d_raw")
    # the documented edge semantics: an empty ray has acc 0 and NaN disparity, an opaque first sample leaves T = 1e-10
    e = O.composite(T(g["edge_raw"]), z, d, None, False)
    assert e[2][0].item() == 0.0 and np.isnan(e[1][0].item())
    assert e[3][2, 1].item() <= 1e-10


def test_inverse_cdf_indices_and_samples():
    g = golden("sample_pdf")
    s, inds = O.inverse_cdf(T(g["bins"]), T(g["weights"]), T(g["u"]), return_inds=True)
    assert np.array_equal(inds.numpy(), g["inds"])
    assert_pinned(s, g["samples"], "samples")
    n, ns = g["u"].shape
    s, inds = O.inverse_cdf(T(g["bins"]), T(g["weights"]), O.det_uniforms(n, ns), return_inds=True)
    assert np.array_equal(inds.numpy(), g["inds_det"])
    assert_pinned(s, g["samples_det"], "samples det")
    # u == 1 runs off the end of the cdf (index == len) and is clamped back onto the last bin edge
    off_end = inds[:, -1] == g["bins"].shape[1]
    assert off_end.any()
    assert torch.equal(s[off_end, -1], T(g["bins"])[off_end, -1])
    s5, inds5 = O.inverse_cdf(T(g["bins5"]), T(g["weights5"]), T(g["u5"]), return_inds=True)
    assert np.array_equal(inds5.numpy(), g["inds5"])
    assert_pinned(s5, g["samples5"], "samples stress")


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


def _peaky(sd, gain):
    sd = {k: v.clone() for k, v in sd.items()}
    sd["alpha_linear.weight"] *= gain
    sd["alpha_linear.bias"] = sd["alpha_linear.bias"] * gain + 0.5
    return sd


def test_march_all_configs():
    g = golden("march")
    sd_c, sd_f = _peaky(O.random_state_dict(31), 40.0), _peaky(O.random_state_dict(32), 60.0)
    for tag in ("lego_train", "lego_test", "fern_train", "lindisp", "coarse_only", "stress"):
        sc, si, perturb, std, white, lindisp, near, far, ndc, H, W, focal = g[tag + ".cfg"]
        sc, si = int(sc), int(si)
        rays = T(g[tag + ".rays"])
        packed = O.pack_rays(rays[0], rays[1], near, far, use_viewdirs=True,
                             ndc=(int(H), int(W), focal) if ndc else None)
        assert_pinned(packed, g[tag + ".packed"], tag + " packed")
        t_rand, u, nc, nf = _rng_inputs(packed.shape[0], sc, si, perturb, std)
        with torch.no_grad():
            out = O.render_chunks(packed, 1 << 15, sd_coarse=sd_c, sd_fine=sd_f, n_coarse=sc, n_fine=si,
                                  perturb=perturb, lindisp=bool(lindisp), white_bkgd=bool(white), t_rand=t_rand,
                                  u=u, noise_coarse=nc, noise_fine=nf, retraw=True)
        keys = [k[len(tag) + 1:] for k in g if k.startswith(tag + ".") and k.split(".")[1] not in ("rays", "packed", "cfg")]
        assert keys
        for k in keys:
            assert_pinned(out[k], g[f"{tag}.{k}"], f"{tag} {k}")


def test_image_render():
    g = golden("image")
    sd_c, sd_f = _peaky(O.random_state_dict(31), 40.0), _peaky(O.random_state_dict(32), 60.0)
    H, W, f = g["hwf"]
    H, W = int(H), int(W)
    K = np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]])
    o, d = O.pixel_rays(H, W, K, T(g["pose"]))
    with torch.no_grad():
        out = O.render_chunks(O.pack_rays(o, d, 2.0, 6.0, use_viewdirs=True), 50, sd_coarse=sd_c, sd_fine=sd_f,
                              n_coarse=64, n_fine=128, white_bkgd=True)
    assert_pinned(out["rgb_map"].reshape(H, W, 3), g["rgb"], "image rgb")
    assert_pinned(out["disp_map"].reshape(H, W), g["disp"], "image disp")
    assert_pinned(out["acc_map"].reshape(H, W), g["acc"], "image acc")
    assert_pinned(out["z_std"].reshape(H, W), g["z_std"], "image z_std")


def test_train_step():
    g = golden("train_step")
    pc = {k: v.clone().requires_grad_(True) for k, v in _peaky(O.random_state_dict(41), 20.0).items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in _peaky(O.random_state_dict(42), 20.0).items()}
    rays, target = T(g["rays"]), T(g["target"])
    n = rays.shape[1]
    t_rand, u, _, _ = _rng_inputs(n, 64, 128, 1.0, 0.0)
    packed = O.pack_rays(rays[0], rays[1], 2.0, 6.0, use_viewdirs=True)
    out = O.render_chunks(packed, 1 << 15, sd_coarse=pc, sd_fine=pf, n_coarse=64, n_fine=128, perturb=1.0,
                          white_bkgd=True, t_rand=t_rand, u=u, retraw=True)
    loss = O.mse(out["rgb_map"], target) + O.mse(out["rgb0"], target)
    assert_pinned(loss, g["loss"], "loss")
    assert_pinned(O.psnr(O.mse(out["rgb_map"], target)), g["psnr"], "psnr")
    loss.backward()
    params = [("c." + k, p) for k, p in pc.items()] + [("f." + k, p) for k, p in pf.items()]
    opt = torch.optim.Adam(params=[p for _, p in params], lr=5e-4, betas=(0.9, 0.999))
    for k, p in params:
        flat = p.grad.reshape(-1)
        assert_pinned(flat[:: max(1, flat.numel() // 64)][:64], g["gsample." + k], "grad " + k)
    opt.step()
    for k, p in params:
        flat = p.detach().reshape(-1)
        assert_pinned(flat[:: max(1, flat.numel() // 64)][:64], g["wsample." + k], "weight " + k)
    assert O.decayed_lr(5e-4, 500, 1234) == g["lr_1234"][0]
