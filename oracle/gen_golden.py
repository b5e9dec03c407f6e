"""Generate tests/golden/*.npz from the reference itself and pin the oracle to it.

Runs only in a container where the upstream tree is mounted at /root/reference.
The reference is imported (never copied): its dataset loaders and image
libraries, which are not on the render path and are not installed, are
replaced by empty modules before `import run_nerf`.  For every fixture the
oracle (oracle/nerf_oracle.py) is evaluated on the same inputs and must agree
with the reference bit for bit, otherwise this script aborts.

Random draws: the reference's `pytest=True` hook reseeds numpy's legacy
generator with 0 before each draw (run_nerf.py:446-449, 328-331;
run_nerf_helpers.py:275-283); the same arrays are rebuilt here and passed to
the oracle explicitly.  Where a specific tensor must be injected into the
reference (chosen uniforms for sample_pdf), torch.rand is temporarily replaced
inside this process only.

Usage:  python oracle/gen_golden.py      (writes tests/golden/)
"""
import os
import sys
import types
from contextlib import contextmanager

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

for _name, _attr in [("imageio", None), ("cv2", None), ("load_llff", "load_llff_data"),
                     ("load_deepvoxels", "load_dv_data"), ("load_blender", "load_blender_data"),
                     ("load_LINEMOD", "load_LINEMOD_data")]:
    _m = types.ModuleType(_name)
    if _attr:
        setattr(_m, _attr, None)
    sys.modules[_name] = _m
sys.path.insert(0, REF)
import run_nerf as R  # noqa: E402
import run_nerf_helpers as RH  # noqa: E402

from oracle import nerf_oracle as O  # noqa: E402


def same(a, b, what):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().numpy() if torch.is_tensor(b) else np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b, equal_nan=True):
        raise SystemExit(f"oracle != reference for {what}: max|d|={np.nanmax(np.abs(a - b))}")


@contextmanager
def injected_rand(*tensors):
    queue = list(tensors)
    real = torch.rand

    def fake(*a, **k):
        t = queue.pop(0)
        return t.clone()
    torch.rand = fake
    try:
        yield
    finally:
        torch.rand = real


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    arrays = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(f"{name}: {sum(v.nbytes for v in arrays.values()) / 1024:.1f} KiB raw")


def ref_model(sd, use_viewdirs=True, output_ch=5):
    m = RH.NeRF(D=8, W=256, input_ch=63, input_ch_views=27 if use_viewdirs else 0, output_ch=output_ch,
                skips=[4], use_viewdirs=use_viewdirs)
    m.load_state_dict(sd)
    return m


def ref_query_fn(netchunk=1024 * 64):
    embed_fn, _ = RH.get_embedder(10, 0)
    embeddirs_fn, _ = RH.get_embedder(4, 0)
    return lambda inputs, viewdirs, fn: R.run_network(inputs, viewdirs, fn, embed_fn=embed_fn,
                                                      embeddirs_fn=embeddirs_fn, netchunk=netchunk)


def peaky(sd, gain):
    sd = {k: v.clone() for k, v in sd.items()}
    sd["alpha_linear.weight"] *= gain
    sd["alpha_linear.bias"] = sd["alpha_linear.bias"] * gain + 0.5
    return sd


# -------------------------------------------------------------------- rays
def gen_rays():
    rng = np.random.Generator(np.random.PCG64(11))
    out = {}
    cases = {
        "lego_half": (400, 400, 555.5555155968841, O.orbit_pose(36.0, -30.0, 4.0)[:3, :4]),
        "lego_full": (800, 800, 1111.1110311937682, O.orbit_pose(-117.0, -30.0, 4.0)[:3, :4]),
        "fern": (378, 504, 407.5658, None),
    }
    fern_pose = np.eye(4, dtype=np.float32)[:3]
    rot = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
    fern_pose[:, :3] = (0.97 * np.eye(3) + 0.03 * rot).astype(np.float32)
    fern_pose[:, 3] = [0.12, -0.07, 0.05]
    for name, (H, W, focal, pose) in cases.items():
        pose = T(fern_pose) if pose is None else pose
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        o, d = RH.get_rays(H, W, K, pose)
        oo, od = O.pixel_rays(H, W, K, pose)
        same(o, oo, name + " rays_o")
        same(d, od, name + " rays_d")
        on, dn = RH.get_rays_np(H, W, K, pose.numpy())
        oon, odn = O.pixel_rays_np(H, W, K, pose.numpy())
        same(on, oon, name + " np rays_o")
        same(dn, odn, name + " np rays_d")
        pick = rng.choice(H * W, size=512, replace=False)
        pick[:4] = [0, W - 1, (H - 1) * W, H * W - 1]
        out[name + "_hwf"] = np.array([H, W, focal], dtype=np.float64)
        out[name + "_pose"] = pose.numpy()
        out[name + "_pick"] = pick.astype(np.int64)
        out[name + "_d"] = d.reshape(-1, 3)[pick].numpy()
        out[name + "_d_np"] = dn.reshape(-1, 3)[pick]
        if name == "fern":
            no, nd = RH.ndc_rays(H, W, K[0][0], 1.0, o, d)
            oo2, od2 = O.ndc_warp(H, W, K[0][0], 1.0, o, d)
            same(no, oo2, "ndc o")
            same(nd, od2, "ndc d")
            out[name + "_ndc_o"] = no.reshape(-1, 3)[pick].numpy()
            out[name + "_ndc_d"] = nd.reshape(-1, 3)[pick].numpy()
    save("rays", **out)


# ----------------------------------------------------------------- encoding
def gen_encode():
    rng = np.random.Generator(np.random.PCG64(12))
    x = T(rng.uniform(-6, 6, size=(256, 3)))
    x[0] = 0.0
    x[1] = torch.tensor([6.0, -6.0, 1e-8])
    out = {"x": x}
    for L in (10, 4):
        fn, width = RH.get_embedder(L, 0)
        e = fn(x)
        assert e.shape[-1] == width == O.encoded_width(L)
        same(e, O.encode(x, L), f"encode L={L}")
        out[f"enc{L}"] = e
    fn, width = RH.get_embedder(10, -1)
    same(fn(x), O.encode(x, -1), "identity encode")
    save("encode", **out)


# ---------------------------------------------------------------------- MLP
def gen_mlp():
    rng = np.random.Generator(np.random.PCG64(13))
    out = {}
    for tag, use_viewdirs in (("vd", True), ("novd", False)):
        sd = O.random_state_dict(21 if use_viewdirs else 22, use_viewdirs=use_viewdirs)
        if not use_viewdirs:
            sd = {k: v for k, v in sd.items() if not k.startswith("views_linears")}
            full = RH.NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
            extra = {k: v for k, v in full.state_dict().items() if k.startswith("views_linears")}
            sd.update(extra)  # unused head kept by the reference's state_dict
        model = ref_model(sd, use_viewdirs)
        n, s = 24, 8
        pts = T(rng.uniform(-4, 4, size=(n, s, 3)))
        vd = T(rng.normal(size=(n, 3)))
        vd = vd / vd.norm(dim=-1, keepdim=True)
        # netchunk 64 forces three slices
        raw = ref_query_fn(64)(pts, vd if use_viewdirs else None, model)
        if use_viewdirs:
            raw_o = O.query_field(pts, vd, sd, netchunk=64)
        else:
            raw_o = O.query_field(pts, None, sd, netchunk=64)
        same(raw, raw_o, f"mlp forward {tag}")
        out[f"{tag}_pts"], out[f"{tag}_viewdirs"], out[f"{tag}_raw"] = pts, vd, raw
        if use_viewdirs:
            G = T(rng.normal(size=tuple(raw.shape)))
            (raw * G).sum().backward()
            params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
            (O.query_field(pts, vd, params, netchunk=64) * G).sum().backward()
            for k, p in model.named_parameters():
                same(p.grad, params[k].grad, f"grad {k}")
                g = p.grad.reshape(-1)
                out[f"{tag}_gsum.{k}"] = np.array([g.double().sum().item(), g.double().abs().sum().item()])
                out[f"{tag}_gsample.{k}"] = g[:: max(1, g.numel() // 64)][:64].numpy().copy()
            out[f"{tag}_G"] = G
    save("mlp", **out)


# ---------------------------------------------------------------- composite
def gen_composite():
    rng = np.random.Generator(np.random.PCG64(14))
    out = {}
    n, s = 48, 64
    z = np.sort(rng.uniform(2, 6, size=(n, s)), -1)
    d = rng.normal(size=(n, 3))
    cases = {
        "soft": rng.normal(scale=0.3, size=(n, s, 4)),
        "peaky": rng.normal(scale=10.0, size=(n, s, 4)),
    }
    edge = rng.normal(scale=1.0, size=(n, s, 4))
    edge[0, :, 3] = -1.0                    # no density anywhere: acc = 0, disp = NaN
    edge[1, :, 3] = 0.0
    edge[2, 0, 3] = 1e6                     # first sample opaque: T collapses to 1e-10
    edge[3, :, 3] = 1e4                     # everything opaque
    edge[4, :-1, 3] = -5.0
    edge[4, -1, 3] = 0.1                    # only the last (1e10-long) interval contributes
    edge[5, :, :3] = 40.0                   # saturated colour
    cases["edge"] = edge
    for name, raw in cases.items():
        for white in (False, True):
            for noisy in (False, True):
                rawt = T(raw).requires_grad_(True)
                zt, dt = T(z), T(d)
                if noisy:
                    np.random.seed(0)
                    noise = torch.Tensor(np.random.rand(n, s) * 0.7)
                    r = R.raw2outputs(rawt, zt, dt, 0.7, white, pytest=True)
                else:
                    noise = None
                    r = R.raw2outputs(rawt, zt, dt, 0.0, white)
                G = T(rng.normal(size=(n, 3)))
                (r[0] * G).sum().backward()
                rawo = T(raw).requires_grad_(True)
                ro = O.composite(rawo, zt, dt, noise, white)
                (ro[0] * G).sum().backward()
                key = f"{name}_w{int(white)}_n{int(noisy)}"
                for a, b, nm in zip(r, ro, ("rgb", "disp", "acc", "weights", "depth")):
                    same(a, b, f"composite {key} {nm}")
                    out[f"{key}_{nm}"] = a
                same(rawt.grad, rawo.grad, f"composite {key} grad")
                out[f"{key}_G"] = G
                out[f"{key}_draw"] = rawt.grad
        out[f"{name}_raw"] = T(raw)
    out["z"], out["rays_d"] = T(z), T(d)
    np.random.seed(0)
    out["noise"] = torch.Tensor(np.random.rand(n, s) * 0.7)
    save("composite", **out)


# --------------------------------------------------------------- sample_pdf
def gen_sample_pdf():
    rng = np.random.Generator(np.random.PCG64(15))
    out = {}
    n, nb, ns = 64, 63, 128
    bins = np.sort(rng.uniform(2, 6, size=(n, nb)), -1)
    w = rng.uniform(0, 1, size=(n, nb - 1)) ** 4
    w[0] = 0.0                               # flat pdf
    w[1] = 0.0
    w[1, 17] = 1.0                           # one spike
    w[2] = 1e-9                              # every cdf step below the 1e-5 guard
    w[3, :30] = 0.0                          # long empty prefix
    w[4] = rng.uniform(0, 1e-3, size=nb - 1)
    u = rng.uniform(0, 1, size=(n, ns))
    u[:, 0] = 0.0
    u[:, 1] = 1.0                            # right=True index runs off the end and is clamped
    u[:, 2] = np.nextafter(np.float32(1.0), np.float32(0.0))
    bt, wt, ut = T(bins), T(w), T(u)
    with injected_rand(ut):
        s_ref = RH.sample_pdf(bt, wt, ns, det=False)
    s_or, inds = O.inverse_cdf(bt, wt, ut, return_inds=True)
    same(s_ref, s_or, "sample_pdf random u")
    s_det = RH.sample_pdf(bt, wt, ns, det=True)
    s_det_o, inds_det = O.inverse_cdf(bt, wt, O.det_uniforms(n, ns), return_inds=True)
    same(s_det, s_det_o, "sample_pdf det")
    np.random.seed(0)
    u_py = torch.Tensor(np.random.rand(n, ns))
    s_py = RH.sample_pdf(bt, wt, ns, det=False, pytest=True)
    same(s_py, O.inverse_cdf(bt, wt, u_py), "sample_pdf pytest")
    # the stress shape: 255 bins, 512 samples
    bins5 = np.sort(rng.uniform(0, 1, size=(8, 255)), -1)
    w5 = rng.uniform(0, 1, size=(8, 254)) ** 8
    u5 = rng.uniform(0, 1, size=(8, 512))
    with injected_rand(T(u5)):
        s5 = RH.sample_pdf(T(bins5), T(w5), 512, det=False)
    s5o, inds5 = O.inverse_cdf(T(bins5), T(w5), T(u5), return_inds=True)
    same(s5, s5o, "sample_pdf stress")
    save("sample_pdf", bins=bt, weights=wt, u=ut, samples=s_ref, inds=inds.numpy(), samples_det=s_det,
         inds_det=inds_det.numpy(), bins5=T(bins5), weights5=T(w5), u5=T(u5), samples5=s5, inds5=inds5.numpy())


# ------------------------------------------------------------- render_rays
def make_rays(kind, n, rng):
    if kind == "lego":
        H = W = 400
        focal = 555.5555155968841
        pose = O.orbit_pose(54.0, -30.0, 4.0)[:3, :4]
        near, far, ndc = 2.0, 6.0, False
    else:
        H, W, focal = 378, 504, 407.5658
        pose = torch.eye(4)[:3, :4].clone()
        pose[:, 3] = torch.tensor([0.1, -0.05, 0.02])
        near, far, ndc = 0.0, 1.0, True
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    o, d = RH.get_rays(H, W, K, pose)
    pick = torch.from_numpy(rng.choice(H * W, size=n, replace=False))
    rays = torch.stack([o.reshape(-1, 3)[pick], d.reshape(-1, 3)[pick]], 0)
    return H, W, K, rays, near, far, ndc


def gen_march():
    rng = np.random.Generator(np.random.PCG64(16))
    sd_c = peaky(O.random_state_dict(31), 40.0)
    sd_f = peaky(O.random_state_dict(32), 60.0)
    mc, mf = ref_model(sd_c), ref_model(sd_f)
    qf = ref_query_fn()
    out = {}
    cases = [
        # tag, rays, n, Sc, Si, perturb, noise_std, white, lindisp
        ("lego_train", "lego", 32, 64, 128, 1.0, 0.0, True, False),
        ("lego_test", "lego", 32, 64, 128, 0.0, 0.0, True, False),
        ("fern_train", "fern", 32, 64, 64, 1.0, 1.0, False, False),
        ("lindisp", "lego", 16, 32, 16, 1.0, 0.0, False, True),
        ("coarse_only", "lego", 16, 64, 0, 1.0, 0.0, True, False),
        ("stress", "lego", 4, 256, 512, 0.0, 0.0, True, False),
    ]
    for tag, kind, n, sc, si, perturb, std, white, lindisp in cases:
        H, W, K, rays, near, far, ndc = make_rays(kind, n, rng)
        kw = dict(network_fn=mc, network_query_fn=qf, N_samples=sc, N_importance=si, network_fine=mf,
                  perturb=perturb, raw_noise_std=std, white_bkgd=white, lindisp=lindisp, retraw=True,
                  use_viewdirs=True, ndc=ndc, near=near, far=far, pytest=(perturb > 0 or std > 0))
        with torch.no_grad():
            rgb, disp, acc, extras = R.render(H, W, K, chunk=1024 * 32, rays=rays, **kw)
        packed = O.pack_rays(rays[0], rays[1], near, far, use_viewdirs=True, ndc=(H, W, K[0][0]) if ndc else None)
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
        with torch.no_grad():
            got = O.render_chunks(packed, 1024 * 32, sd_coarse=sd_c, sd_fine=sd_f, n_coarse=sc, n_fine=si,
                                  perturb=perturb, lindisp=lindisp, white_bkgd=white, t_rand=t_rand, u=u,
                                  noise_coarse=nc, noise_fine=nf, retraw=True)
        ref = dict(extras, rgb_map=rgb, disp_map=disp, acc_map=acc)
        assert set(ref) == set(got), (set(ref), set(got))
        for k in ref:
            same(ref[k], got[k], f"march {tag} {k}")
            if k != "raw" or tag == "lego_train":
                out[f"{tag}.{k}"] = ref[k]
        out[f"{tag}.rays"] = rays
        out[f"{tag}.packed"] = packed
        out[f"{tag}.cfg"] = np.array([sc, si, perturb, std, float(white), float(lindisp), near, far, float(ndc),
                                      H, W, K[0][0]], dtype=np.float64)
    save("march", **out)

    # chunking must not change a deterministic render (run_nerf.py:96-97)
    H, W, K, rays, near, far, ndc = make_rays("lego", 40, rng)
    packed = O.pack_rays(rays[0], rays[1], near, far, use_viewdirs=True)
    with torch.no_grad():
        a = O.render_chunks(packed, 16, sd_coarse=sd_c, sd_fine=sd_f, n_coarse=64, n_fine=128, white_bkgd=True)
        kw = dict(network_fn=mc, network_query_fn=qf, N_samples=64, N_importance=128, network_fine=mf,
                  perturb=0.0, white_bkgd=True, use_viewdirs=True, ndc=False, near=near, far=far)
        rgb, disp, acc, _ = R.render(H, W, K, chunk=16, rays=rays, **kw)
    same(rgb, a["rgb_map"], "chunked render")

    # whole-image entry (c2w=...) on a tiny camera
    Hs, Ws, fs = 12, 10, 14.0
    Ks = np.array([[fs, 0, 0.5 * Ws], [0, fs, 0.5 * Hs], [0, 0, 1]])
    pose = O.orbit_pose(10.0, -30.0, 4.0)[:3, :4]
    with torch.no_grad():
        rgb, disp, acc, extras = R.render(Hs, Ws, Ks, chunk=50, c2w=pose, **kw)
        o, d = O.pixel_rays(Hs, Ws, Ks, pose)
        got = O.render_chunks(O.pack_rays(o, d, near, far, use_viewdirs=True), 50, sd_coarse=sd_c, sd_fine=sd_f,
                              n_coarse=64, n_fine=128, white_bkgd=True)
    same(rgb.reshape(-1, 3), got["rgb_map"], "image render rgb")
    same(disp.reshape(-1), got["disp_map"], "image render disp")
    save("image", pose=pose, hwf=np.array([Hs, Ws, fs]), rgb=rgb, disp=disp, acc=acc, z_std=extras["z_std"])


# --------------------------------------------------------------- train step
def gen_train_step():
    rng = np.random.Generator(np.random.PCG64(17))
    sd_c = peaky(O.random_state_dict(41), 20.0)
    sd_f = peaky(O.random_state_dict(42), 20.0)
    mc, mf = ref_model(sd_c), ref_model(sd_f)
    n, sc, si = 48, 64, 128
    H, W, K, rays, near, far, ndc = make_rays("lego", n, rng)
    target = T(rng.uniform(0, 1, size=(n, 3)))
    params = list(mc.parameters()) + list(mf.parameters())
    opt = torch.optim.Adam(params=params, lr=5e-4, betas=(0.9, 0.999))
    kw = dict(network_fn=mc, network_query_fn=ref_query_fn(), N_samples=sc, N_importance=si, network_fine=mf,
              perturb=1.0, raw_noise_std=0.0, white_bkgd=True, use_viewdirs=True, ndc=False, near=near, far=far,
              pytest=True, retraw=True)
    rgb, disp, acc, extras = R.render(H, W, K, chunk=1024 * 32, rays=rays, verbose=True, **kw)
    opt.zero_grad()
    loss = RH.img2mse(rgb, target) + RH.img2mse(extras["rgb0"], target)
    loss.backward()
    ref_grads = {("c." + k): p.grad.clone() for k, p in mc.named_parameters()}
    ref_grads.update({("f." + k): p.grad.clone() for k, p in mf.named_parameters()})
    opt.step()

    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    np.random.seed(0)
    t_rand = torch.Tensor(np.random.rand(n, sc))
    np.random.seed(0)
    u = torch.Tensor(np.random.rand(n, si))
    packed = O.pack_rays(rays[0], rays[1], near, far, use_viewdirs=True)
    got = O.render_chunks(packed, 1024 * 32, sd_coarse=pc, sd_fine=pf, n_coarse=sc, n_fine=si, perturb=1.0,
                          white_bkgd=True, t_rand=t_rand, u=u, retraw=True)
    loss_o = O.mse(got["rgb_map"], target) + O.mse(got["rgb0"], target)
    loss_o.backward()
    same(loss, loss_o, "train loss")
    names = list(pc) + list(pf)
    oparams = list(pc.values()) + list(pf.values())
    opt_o = torch.optim.Adam(params=oparams, lr=5e-4, betas=(0.9, 0.999))
    out = {"rays": rays, "target": target, "loss": loss.detach(), "psnr": RH.mse2psnr(RH.img2mse(rgb, target)).detach()}
    same(RH.mse2psnr(RH.img2mse(rgb, target)), O.psnr(O.mse(got["rgb_map"], target)), "psnr")
    for (k, g), p in zip(ref_grads.items(), oparams):
        same(g, p.grad, f"train grad {k}")
        flat = g.reshape(-1)
        out[f"gsum.{k}"] = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
        out[f"gsample.{k}"] = flat[:: max(1, flat.numel() // 64)][:64].numpy().copy()
    opt_o.step()
    for (k, _), p, q in zip(ref_grads.items(), params, oparams):
        same(p, q, f"post-Adam {k}")
        flat = p.detach().reshape(-1)
        out[f"wsample.{k}"] = flat[:: max(1, flat.numel() // 64)][:64].numpy().copy()
    lr = 5e-4 * (0.1 ** (1234 / (500 * 1000)))
    assert lr == O.decayed_lr(5e-4, 500, 1234)
    out["lr_1234"] = np.array([lr])
    save("train_step", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    gen_rays()
    gen_encode()
    gen_mlp()
    gen_composite()
    gen_sample_pdf()
    gen_march()
    gen_train_step()
    import json
    with open(os.path.join(OUT, "meta.json"), "w") as fh:
        json.dump({"torch": torch.__version__, "numpy": np.__version__,
                   "cpu_capability": torch.backends.cpu.get_cpu_capability(),
                   "threads": torch.get_num_threads()}, fh, indent=1)
    print("all fixtures written; oracle == reference on every one")
