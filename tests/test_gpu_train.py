"""Ray batchers and the training driver on the GPU (synthetic scenes written by the tests)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden  # noqa: F401
from test_datasets import write_blender_scene

pytestmark = pytest.mark.gpu


def _camera(H, W, focal):
    return np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])


def test_rays_at_equals_get_rays_gather(device):
    from nerf_amd import ops
    from nerf_amd.datasets import pose_spherical
    from nerf_amd.run_nerf_helpers import get_rays
    H, W = 37, 53
    K = _camera(H, W, 61.7)
    pose = pose_spherical(40.0, -25.0, 3.5)[:3, :4].to(device)
    image = torch.rand(H, W, 3, device=device)
    ro, rd = get_rays(H, W, K, pose)
    g = torch.Generator(device=device)
    g.manual_seed(1)
    for win in ((0, 0, H, W), (H // 2 - 9, W // 2 - 13, 18, 26)):
        n_pix = win[2] * win[3]
        pick = torch.randperm(n_pix, device=device, generator=g)[:200]
        o, d, t = ops.rays_at(pick, H, W, win, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), pose, image)
        rows = win[0] + pick // win[3]
        cols = win[1] + pick % win[3]
        assert torch.equal(o, ro[rows, cols]) and torch.equal(d, rd[rows, cols]) and torch.equal(t, image[rows, cols])
    with pytest.raises(RuntimeError):
        ops.rays_at(pick, H, W, (0, 0, H + 1, W), 1.0, 1.0, 0.0, 0.0, pose)


def test_view_sampler_crop_window_and_distinct_pixels(device):
    from nerf_amd.datasets import pose_spherical
    from nerf_amd.raybatch import ViewSampler
    H, W = 40, 60
    K = _camera(H, W, 70.0)
    images = np.random.default_rng(0).uniform(0, 1, (3, H, W, 3)).astype(np.float32)
    # make every pixel's colour identify it: r = row / H, g = col / W
    images[..., 0] = (np.arange(H)[:, None] / H)
    images[..., 1] = (np.arange(W)[None, :] / W)
    poses = np.stack([pose_spherical(20.0 * k, -30.0, 4.0).numpy() for k in range(3)], 0)
    vs = ViewSampler(images, poses, [0, 2], H, W, K, N_rand=150, precrop_iters=5, precrop_frac=0.5, device=device, seed=3)
    assert vs.window(0) == (10, 15, 20, 30) and vs.window(5) == (0, 0, H, W)
    rays, target = vs.next(0)
    assert rays.shape == (2, 150, 3) and target.shape == (150, 3)
    rows = torch.round(target[:, 0] * H).long().cpu().numpy()
    cols = torch.round(target[:, 1] * W).long().cpu().numpy()
    assert rows.min() >= 10 and rows.max() < 30 and cols.min() >= 15 and cols.max() < 45
    assert len(set(zip(rows.tolist(), cols.tolist()))) == 150
    rays, target = vs.next(7, img_i=2)
    rows = torch.round(target[:, 0] * H).long()
    cols = torch.round(target[:, 1] * W).long()
    from nerf_amd.run_nerf_helpers import get_rays
    ro, rd = get_rays(H, W, K, torch.from_numpy(poses[2, :3, :4]))
    assert torch.equal(rays[1], rd[rows, cols]) and torch.equal(rays[0], ro[rows, cols])


def test_rays_random_draws_distinct_pixels_and_equals_rays_at(device):
    """nerf_rays_random: n distinct pixels of the window from a keyed permutation, rays and colours equal to
    nerf_rays_at on the pixels it reports; whole windows are permutations; seeds give different, reproducible draws."""
    from nerf_amd import ops
    from nerf_amd.datasets import pose_spherical
    rng = np.random.default_rng(4)
    for (H, W, win, n) in [(40, 60, (10, 15, 20, 30), 150), (400, 400, (0, 0, 400, 400), 1024), (7, 5, (0, 0, 7, 5), 35),
                           (3, 3, (1, 1, 1, 1), 1), (64, 64, (0, 0, 64, 64), 4096), (33, 17, (2, 3, 30, 11), 330)]:
        K = _camera(H, W, 50.0)
        intr = (float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))
        c2w = pose_spherical(33.0, -30.0, 4.0)[:3, :4].to(device)
        image = torch.from_numpy(rng.uniform(0, 1, (H * W, 3)).astype(np.float32)).to(device)
        rays, target, pick = ops.rays_random(n, 12345, H, W, win, *intr, c2w, image, want_pick=True)
        q = pick.cpu().numpy()
        n_pix = win[2] * win[3]
        assert q.min() >= 0 and q.max() < n_pix and len(set(q.tolist())) == n, (H, W, win, n)
        if n == n_pix:
            assert sorted(q.tolist()) == list(range(n_pix))
        o, d, t = ops.rays_at(pick, H, W, win, *intr, c2w, image)
        assert torch.equal(rays[0], o) and torch.equal(rays[1], d) and torch.equal(target, t)
        again = ops.rays_random(n, 12345, H, W, win, *intr, c2w, image, want_pick=True)[2]
        other = ops.rays_random(n, 12346, H, W, win, *intr, c2w, image, want_pick=True)[2]
        assert torch.equal(again, pick)
        if n_pix > 8:
            assert not torch.equal(other, pick)
    # no visible structure: over many seeds every pixel of a small window is drawn about equally often
    H, W, win, n = 16, 16, (0, 0, 16, 16), 64
    K = _camera(H, W, 20.0)
    intr = (float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))
    c2w = pose_spherical(0.0, -30.0, 4.0)[:3, :4].to(device)
    hits = np.zeros(256)
    first = np.zeros(256)
    for seed in range(400):
        q = ops.rays_random(n, seed * 7919 + 1, H, W, win, *intr, c2w, None, want_pick=True)[2].cpu().numpy()
        hits[q] += 1
        first[q[0]] += 1
    assert abs(hits.mean() - 100.0) < 1e-9 and hits.min() > 60 and hits.max() < 145, (hits.min(), hits.max())
    assert first.max() <= 9     # 400 first draws over 256 pixels
    with pytest.raises(RuntimeError):
        ops.rays_random(17, 1, 4, 4, (0, 0, 4, 4), *intr, c2w, None)


def test_ray_shuffler_epoch_and_rank_slices(device):
    from nerf_amd.datasets import pose_spherical
    from nerf_amd.raybatch import RayShuffler
    from nerf_amd.run_nerf_helpers import get_rays_np
    H, W = 6, 8
    K = _camera(H, W, 9.0)
    rng = np.random.default_rng(1)
    images = rng.uniform(0, 1, (4, H, W, 3)).astype(np.float32)
    poses = np.stack([pose_spherical(25.0 * k, -30.0, 4.0).numpy() for k in range(4)], 0)
    i_train = [0, 1, 3]
    want = []
    for i in i_train:
        ro, rd = get_rays_np(H, W, K, poses[i, :3, :4])
        want.append(np.stack([ro, rd, images[i]], -2).reshape(-1, 3, 3).astype(np.float32))
    want = np.concatenate(want, 0)
    a = RayShuffler(images, poses, i_train, H, W, K, N_rand=16, device=device, seed=5, rank=0, world_size=2)
    b = RayShuffler(images, poses, i_train, H, W, K, N_rand=16, device=device, seed=5, rank=1, world_size=2)
    seen = []
    for _ in range(want.shape[0] // 32):
        (ra, ta), (rb, tb) = a.next(), b.next()
        assert ra.shape == (2, 16, 3) and ta.shape == (16, 3)
        seen += [torch.cat([ra.transpose(0, 1), ta[:, None]], 1), torch.cat([rb.transpose(0, 1), tb[:, None]], 1)]
    seen = torch.cat(seen, 0).cpu().numpy()
    # one epoch of both ranks = distinct records of the training set, every one exactly as the host builds them
    key = lambda r: r.reshape(r.shape[0], -1)
    all_rows = {row.tobytes() for row in key(want)}
    got_rows = [row.tobytes() for row in key(seen)]
    assert len(set(got_rows)) == len(got_rows) and set(got_rows) <= all_rows
    assert a.epochs == 1 and b.epochs == 1


def test_train_driver_runs_checkpoints_and_resumes(device, tmp_path, capsys):
    from nerf_amd.run_nerf import train
    scene = os.path.join(tmp_path, "scene")
    write_blender_scene(scene, n=(3, 1, 1), size=16)
    cfg = os.path.join(tmp_path, "tiny.txt")
    with open(cfg, "w") as fh:
        fh.write("expname = tiny\nbasedir = %s\ndatadir = %s\ndataset_type = blender\nno_batching = True\n"
                 "use_viewdirs = True\nwhite_bkgd = True\nlrate_decay = 500\nN_samples = 8\nN_importance = 8\nN_rand = 64\n"
                 "precrop_iters = 2\nprecrop_frac = 0.5\n" % (os.path.join(tmp_path, "logs"), scene))
    common = ["--config", cfg, "--i_print", "2", "--i_weights", "4", "--i_testset", "4", "--i_video", "1000", "--testskip", "1"]
    loss = train(common + ["--N_iters", "5"])
    assert np.isfinite(loss)
    log = os.path.join(tmp_path, "logs", "tiny")
    assert os.path.exists(os.path.join(log, "args.txt")) and os.path.exists(os.path.join(log, "config.txt"))
    assert os.path.exists(os.path.join(log, "000004.tar")) and os.path.exists(os.path.join(log, "testset_000004", "000.png"))
    ck = torch.load(os.path.join(log, "000004.tar"), map_location="cpu", weights_only=True)
    assert set(ck) == {"global_step", "network_fn_state_dict", "network_fine_state_dict", "optimizer_state_dict"}
    assert ck["global_step"] == 3                       # saved before the step counter moves (run_nerf.py:890-898, 971)
    out = capsys.readouterr().out
    assert "Center cropping of size 8 x 8" in out and "[TRAIN] Iter: 4" in out
    # resume: picks the checkpoint up and continues from its step
    train(common + ["--N_iters", "7", "--no_batching"])
    out = capsys.readouterr().out
    assert "Reloading from" in out and "[TRAIN] Iter: 6" in out
    # batching mode (all training pixels shuffled) in its own experiment, then render_only from the first one
    cfg_b = os.path.join(tmp_path, "tiny_b.txt")
    with open(cfg_b, "w") as fh:
        fh.write(open(cfg).read().replace("no_batching = True\n", "").replace("expname = tiny", "expname = tiny_b"))
    assert np.isfinite(train(["--config", cfg_b, "--N_iters", "4", "--testskip", "1", "--i_print", "1"]))
    assert "[TRAIN] Iter: 3" in capsys.readouterr().out
    rgbs = train(common + ["--render_only", "--render_factor", "2"])
    assert rgbs.shape == (40, 8, 8, 3)
    assert os.path.exists(os.path.join(log, "renderonly_path_000003", "039.png"))


def test_graphed_train_step_matches_eager(device):
    """The optimisation step recorded into HIP graphs and replayed on new rays walks the same weights as the eager loop
    (deterministic settings: no jitter, no noise), and a render after replays sees the updated weights."""
    import copy
    from nerf_amd.graph_step import GraphedTrainStep
    from nerf_amd.run_nerf import render, run_network
    from nerf_amd.run_nerf_helpers import NeRF, get_embedder, img2mse
    from oracle import nerf_oracle as O
    e, _ = get_embedder(10, 0)
    ed, _ = get_embedder(4, 0)
    qf = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=e, embeddirs_fn=ed, netchunk=1 << 16)

    def make():
        ms = []
        for seed in (41, 42):
            m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
            m.load_state_dict(O.random_state_dict(seed))
            ms.append(m.to(device))
        params = list(ms[0].parameters()) + list(ms[1].parameters())
        opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.999), capturable=True)
        kw = dict(network_fn=ms[0], network_query_fn=qf, N_samples=16, N_importance=24, network_fine=ms[1], perturb=0.0,
                  raw_noise_std=0.0, white_bkgd=True, use_viewdirs=True, ndc=False, near=2.0, far=6.0)
        return ms, params, opt, kw
    H = W = 32
    K = np.array([[40.0, 0, 16.0], [0, 40.0, 16.0], [0, 0, 1]])
    n_rand, n_steps = 200, 7
    rng = np.random.Generator(np.random.PCG64(3))
    batches = []
    for i in range(n_steps):
        o, d = O.pixel_rays(H, W, K, O.orbit_pose(30.0 * i, -30.0, 4.0)[:3, :4])
        pick = torch.from_numpy(rng.choice(H * W, size=n_rand, replace=False))
        batches.append((torch.stack([o.reshape(-1, 3)[pick], d.reshape(-1, 3)[pick]], 0).to(device),
                        torch.from_numpy(rng.uniform(0, 1, size=(n_rand, 3)).astype(np.float32)).to(device)))
    lrs = [5e-4 * (0.1 ** (i / 50.0)) for i in range(n_steps)]

    ms_a, params_a, opt_a, kw_a = make()
    lr_a = torch.tensor(5e-4, device=device)
    for g in opt_a.param_groups:
        g["lr"] = lr_a
    losses_a = []
    for (rays, target), lr in zip(batches, lrs):
        lr_a.fill_(lr)
        rgb, _, _, ex = render(H, W, K, chunk=1 << 15, rays=rays, retraw=True, **kw_a)
        opt_a.zero_grad()
        loss = img2mse(rgb, target) + img2mse(ex["rgb0"], target)
        loss.backward()
        opt_a.step()
        losses_a.append(float(loss))

    ms_b, params_b, opt_b, kw_b = make()
    stepper = GraphedTrainStep(H, W, K, kw_b, params_b, opt_b, n_rand, eager_steps=2)
    losses_b = []
    for (rays, target), lr in zip(batches, lrs):
        stepper.set_lr(lr)
        losses_b.append(float(stepper.step(rays, target)))
    assert stepper.g_grad is not None, "steps after the eager ones must be replays"
    np.testing.assert_allclose(losses_b, losses_a, rtol=1e-6)
    for pa, pb in zip(params_a, params_b):
        assert torch.equal(pa, pb), "graphed and eager steps must walk the same weights"
    assert np.isfinite(float(stepper.psnr))
    with torch.no_grad():
        ra = render(H, W, K, chunk=1 << 15, rays=batches[0][0], **kw_a)[0]
        rb = render(H, W, K, chunk=1 << 15, rays=batches[0][0], **kw_b)[0]
    assert torch.equal(ra, rb), "a render after replays must use the weights of the last step"


def test_optimisation_recovers_a_teacher_scene(device):
    """End to end: images rendered from one pair of networks are the training targets of another pair started from
    different weights; 250 steps of the upstream loop (jittered samples, both losses, Adam) must cut the loss several
    times over and raise the PSNR -- a check of the gradient path as a whole that per-step parity cannot give.  Along
    the way the two-piece stages must keep what real gradients look like (transmittance-weighted, zero behind opaque
    samples, density and colour terms of different size) inside their range: at most 1 % of the 128-point tiles handed to
    the three-piece kernels, forward and backward."""
    from nerf_amd import ops
    from nerf_amd.raybatch import ViewSampler
    from nerf_amd.run_nerf import render, run_network
    from nerf_amd.run_nerf_helpers import NeRF, get_embedder, img2mse, mse2psnr
    from nerf_amd.datasets import pose_spherical
    from oracle import nerf_oracle as O
    e, _ = get_embedder(10, 0)
    ed, _ = get_embedder(4, 0)
    qf = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=e, embeddirs_fn=ed, netchunk=1 << 16)

    def pair(seed_c, seed_f, gain):
        out = []
        for seed in (seed_c, seed_f):
            sd = O.random_state_dict(seed)
            sd["alpha_linear.weight"] *= gain
            sd["alpha_linear.bias"] = sd["alpha_linear.bias"] * gain + 1.0
            sd["rgb_linear.weight"] *= 8.0          # colours that vary over the scene
            m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
            m.load_state_dict(sd)
            out.append(m.to(device))
        return out
    H = W = 48
    K = _camera(H, W, 60.0)
    common = dict(network_query_fn=qf, N_samples=32, N_importance=32, use_viewdirs=True, white_bkgd=True, ndc=False,
                  near=2.0, far=6.0)
    teacher = pair(101, 102, 6.0)
    poses = torch.stack([pose_spherical(th, -30.0, 4.0) for th in (0.0, 70.0, 150.0, 230.0, 300.0)], 0).to(device)
    with torch.no_grad():
        images = torch.stack([render(H, W, K, chunk=1 << 15, c2w=p[:3, :4], network_fn=teacher[0], network_fine=teacher[1],
                                     perturb=0.0, raw_noise_std=0.0, **common)[0] for p in poses], 0)
    assert float(images.std()) > 0.02, "the teacher scene must not be flat"
    student = pair(201, 202, 1.0)
    params = list(student[0].parameters()) + list(student[1].parameters())
    opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.999))
    sampler = ViewSampler(images, poses, list(range(5)), H, W, K, 512, device=device, seed=3)
    losses = []
    ops.MARKED = []
    try:
        for i in range(250):
            rays, target = sampler.next(i)
            rgb, _, _, ex = render(H, W, K, chunk=1 << 15, rays=rays, retraw=True, network_fn=student[0], network_fine=student[1],
                                   perturb=1.0, raw_noise_std=0.0, **common)
            opt.zero_grad()
            loss = img2mse(rgb, target) + img2mse(ex["rgb0"], target)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        marked = ops.MARKED
    finally:
        ops.MARKED = None
    for stage in ("forward", "backward_data"):
        flags = torch.cat([f for st, f in marked if st == stage])
        assert flags.numel() == 250 * (512 * 32 + 512 * 64) // 128, stage
        share = float(flags.float().mean())
        print(f"marked tiles, {stage}: {share:.5f}")
        assert share <= 0.01, (stage, share)
    first, last = np.mean(losses[:10]), np.mean(losses[-10:])
    assert np.isfinite(losses).all()
    assert last < 0.25 * first, (first, last)
    with torch.no_grad():
        got = render(H, W, K, chunk=1 << 15, c2w=poses[1][:3, :4], network_fn=student[0], network_fine=student[1],
                     perturb=0.0, raw_noise_std=0.0, **common)[0]
    assert float(mse2psnr(img2mse(got, images[1]))) > 20.0


def test_command_line_entry_in_a_fresh_interpreter(device, tmp_path):
    """`python -m nerf_amd --config ...` as users start it: a new interpreter, where the package (and with it the HIP
    library) is imported before anything has touched the GPU."""
    import subprocess
    import sys
    scene = os.path.join(tmp_path, "scene")
    write_blender_scene(scene, n=(3, 1, 1), size=16)
    cfg = os.path.join(tmp_path, "cli.txt")
    with open(cfg, "w") as fh:
        fh.write("expname = cli\nbasedir = %s\ndatadir = %s\ndataset_type = blender\nno_batching = True\nuse_viewdirs = True\n"
                 "white_bkgd = True\nN_samples = 8\nN_importance = 8\nN_rand = 64\n" % (os.path.join(tmp_path, "logs"), scene))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    run = subprocess.run([sys.executable, "-m", "nerf_amd", "--config", cfg, "--N_iters", "4", "--i_print", "1", "--i_weights", "3",
                          "--i_testset", "1000", "--i_video", "1000", "--testskip", "1"], cwd=root, env=env, capture_output=True,
                         text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "[TRAIN] Iter: 3" in run.stdout
    assert os.path.exists(os.path.join(tmp_path, "logs", "cli", "000003.tar"))


def test_train_driver_on_a_forward_facing_scene(device, tmp_path, capsys):
    """The LLFF branch of the driver (configs/fern.txt: NDC rays, near 0 / far 1, density noise, every llffhold-th view held
    out, batching over all rays) on a synthetic capture: trains, checkpoints, renders the held-out views and the spiral."""
    from nerf_amd.run_nerf import train
    from test_datasets import write_llff_scene
    scene = os.path.join(tmp_path, "capture")
    write_llff_scene(scene, n=6, h=12, w=16)
    cfg = os.path.join(tmp_path, "ff.txt")
    with open(cfg, "w") as fh:
        fh.write("expname = ff\nbasedir = %s\ndatadir = %s\ndataset_type = llff\nfactor = 2\nllffhold = 3\nN_rand = 32\n"
                 "N_samples = 8\nN_importance = 8\nuse_viewdirs = True\nraw_noise_std = 1e0\n" % (os.path.join(tmp_path, "logs"), scene))
    common = ["--config", cfg, "--i_print", "1", "--i_weights", "3", "--i_testset", "3", "--i_video", "1000"]
    loss = train(common + ["--N_iters", "4"])
    assert np.isfinite(loss)
    out = capsys.readouterr().out
    assert "NEAR FAR 0.0 1.0" in out or "[TRAIN] Iter: 3" in out
    log = os.path.join(tmp_path, "logs", "ff")
    assert os.path.exists(os.path.join(log, "000003.tar"))
    assert sorted(os.listdir(os.path.join(log, "testset_000003"))) == ["000.png", "001.png"]     # views 0 and 3 of six
    rgbs = train(common + ["--render_only", "--render_factor", "2"])
    assert rgbs.shape == (120, 3, 4, 3) and np.isfinite(rgbs).all()
    # inward-facing capture flags (run_nerf.py:563-572): spherified poses, no NDC, samples linear in disparity
    cfg2 = os.path.join(tmp_path, "ff360.txt")
    with open(cfg2, "w") as fh:
        fh.write(open(cfg).read().replace("expname = ff", "expname = ff360") + "no_ndc = True\nspherify = True\nlindisp = True\n")
    assert np.isfinite(train(["--config", cfg2, "--N_iters", "3", "--i_print", "1", "--i_weights", "1000", "--i_testset", "1000",
                              "--i_video", "1000"]))
    assert "[TRAIN] Iter: 2" in capsys.readouterr().out
