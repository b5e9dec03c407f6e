"""Random ragged shapes through the streaming kernels against the CPU oracle: compositing (forward + gradient), coarse
depths, inverse-CDF sampling and the fused fine sampling.  python tools/fuzz_edges.py [cases] [seed]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_amd import ops
from oracle import nerf_oracle as O

T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def run(cases: int, seed: int) -> int:
    """Number of mismatches over `cases` random shapes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    dev = torch.device("cuda", torch.cuda.current_device())
    bad = 0

    def report(what, ok, detail):
        nonlocal bad
        if not ok:
            bad += 1
            print("MISMATCH", what, detail, flush=True)

    for case in range(cases):
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 200, 1000]))
        S = int(rng.choice([2, 3, 4, 5, 63, 64, 65, 128, 192, 193, 300, 768]))
        # ---- compositing
        scale = float(rng.choice([0.1, 1.0, 10.0, 100.0]))
        raw = T(rng.normal(size=(n, S, 4)) * scale)
        near = rng.uniform(0.0, 3.0, size=(n, 1))
        z = T(np.sort(near + rng.uniform(0, 4, size=(n, S)), axis=-1))
        d = T(rng.normal(size=(n, 3)))
        noise = T(rng.normal(size=(n, S))) if rng.random() < 0.3 else None
        white = bool(rng.random() < 0.5)
        raw32 = raw.clone().requires_grad_(True)
        ref = O.composite(raw32, z, d, noise, white)
        rg = raw.clone().to(dev).requires_grad_(True)
        got = ops.composite(rg, z.to(dev), d.to(dev), None if noise is None else noise.to(dev), white)
        names = ("rgb", "disp", "acc", "weights", "depth")
        for k, (a, b) in enumerate(zip(got, ref)):
            a = a.detach().cpu()
            b = b.detach()
            tol = 2e-4 if names[k] == "disp" else 2e-5
            ok = torch.allclose(a, b, rtol=tol, atol=tol * max(1.0, float(b[torch.isfinite(b)].abs().max()) if torch.isfinite(b).any() else 1.0), equal_nan=True)
            report(f"composite {names[k]}", ok, (n, S, scale, white, noise is not None, float((a - b).abs().nan_to_num().max())))
        G = [T(rng.normal(size=tuple(t.shape))) for t in ref]
        G[1] = torch.zeros_like(G[1])          # disparity: 1/max(1e-10, depth/acc) is ill-conditioned for empty rays
        raw_ref = raw.clone().double().requires_grad_(True)
        ref64 = O.composite(raw_ref, z.double(), d.double(), None if noise is None else noise.double(), white)
        sum((r * g.double()).sum() for r, g in zip(ref64, G)).backward()
        sum((a * g.to(dev)).sum() for a, g in zip(got, G)).backward()
        sum((r * g).sum() for r, g in zip(ref, G)).backward()
        gs = float(raw_ref.grad.abs().max())
        err = float((rg.grad.cpu().double() - raw_ref.grad).abs().max())
        err_cpu = float((raw32.grad.double() - raw_ref.grad).abs().max())   # the fp32 CPU evaluation against the same truth
        # (floor: a third of an fp32 rounding of the O(1) colours and upstream gradients -- with raw values of scale 100 every
        #  sample saturates, the true gradients are 1e-8 and smaller, and weights below 1e-45 are zero in fp32)
        report("composite grad", err <= max(1e-4 * gs, 2.0 * err_cpu, 2e-8), (n, S, scale, white, err, err_cpu, gs))

        # ---- coarse depths
        Sc = int(rng.choice([3, 4, 9, 64, 65, 128, 256]))
        rays = T(np.concatenate([rng.normal(size=(n, 6)), rng.uniform(0.5, 2, size=(n, 1)), rng.uniform(3, 6, size=(n, 1)),
                                 rng.normal(size=(n, 3))], -1))
        lindisp = bool(rng.random() < 0.3)
        t_rand = T(rng.uniform(size=(n, Sc))) if rng.random() < 0.5 else None
        zc, pts = ops.sample_coarse(rays.to(dev), ops.cached_linspace(Sc, dev), None if t_rand is None else t_rand.to(dev), lindisp)
        zr = O.coarse_depths(rays[:, 6:7], rays[:, 7:8], Sc, lindisp, t_rand)
        report("coarse z", torch.equal(zc.cpu(), zr), (n, Sc, lindisp, t_rand is not None))
        pr = rays[:, None, 0:3] + rays[:, None, 3:6] * zr[..., :, None]
        report("coarse pts", torch.equal(pts.cpu(), pr), (n, Sc))

        # ---- fine sampling on those depths
        Si = int(rng.choice([1, 2, 5, 64, 100, 128, 512]))
        kind = rng.integers(0, 4)
        w = rng.uniform(size=(n, Sc))
        if kind == 1:
            w = np.zeros((n, Sc))
        elif kind == 2:
            w = np.zeros((n, Sc)); w[np.arange(n), rng.integers(0, Sc, size=n)] = 1.0
        elif kind == 3:
            w = w ** 8
        w = T(w)
        det = bool(rng.random() < 0.5)
        u = O.det_uniforms(n, Si).contiguous() if det else T(rng.uniform(size=(n, Si)))
        mids = 0.5 * (zr[:, 1:] + zr[:, :-1])
        smp_ref, inds_ref = O.inverse_cdf(mids, w[:, 1:-1], u, return_inds=True)
        zf_ref = torch.sort(torch.cat([zr, smp_ref], -1), -1)[0]
        out = ops.sample_fine(zr.to(dev), w.to(dev), u.to(dev), rays.to(dev), want_samples=True, want_inds=True)
        same_inds = torch.equal(out["inds"].cpu(), inds_ref)
        if not same_inds:   # a uniform within an ulp of a cdf value may fall either side
            diff = (out["inds"].cpu() != inds_ref)
            report("fine inds", int(diff.sum()) <= 2, (n, Sc, Si, kind, det, int(diff.sum())))
        else:
            report("fine samples", torch.equal(out["samples"].cpu(), smp_ref), (n, Sc, Si, kind, det))
            report("fine merged z", torch.equal(out["z"].cpu(), zf_ref), (n, Sc, Si, kind, det))
            std_ref = torch.std(smp_ref, dim=-1, unbiased=False)
            report("fine z_std", torch.allclose(out["z_std"].cpu(), std_ref, rtol=1e-4, atol=1e-6), (n, Sc, Si, kind, det,
                   float((out["z_std"].cpu() - std_ref).abs().max())))
            report("fine pts", torch.equal(out["pts"].cpu(), rays[:, None, 0:3] + rays[:, None, 3:6] * zf_ref[..., :, None]), (n, Sc, Si))
        s2 = ops.sample_pdf(mids.to(dev), w[:, 1:-1].contiguous().to(dev), u.to(dev))
        if same_inds:
            report("sample_pdf", torch.equal(s2.cpu(), smp_ref), (n, Sc, Si, kind, det))
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    torch.cuda.set_device(0)
    n_bad = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{n_cases} cases, {n_bad} mismatches")
    sys.exit(1 if n_bad else 0)
