"""world_size-2 checks of the multi-GPU helpers on CPU (gloo): ray sharding + gather, gradient bucket all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def test_shard_bounds_cover_and_balance():
    from nerf_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 640000, 160001):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_render(H, W, K, chunk, rays, **kw):
    o, d = rays
    rgb = o * 2.0 + d
    disp = (o * d).sum(-1)
    acc = d[:, 0] - o[:, 1]
    return [rgb, disp, acc, {}]


def _sharded(rank, world):
    from nerf_amd.parallel import render_sharded
    g = torch.Generator().manual_seed(3)
    o, d = torch.randn(101, 3, generator=g), torch.randn(101, 3, generator=g)
    rgb, disp, acc = render_sharded(4, 4, None, 16, (o, d), render_fn=_fake_render)
    want = _fake_render(4, 4, None, 16, (o, d))
    return bool(torch.equal(rgb, want[0]) and torch.equal(disp, want[1]) and torch.equal(acc, want[2]))


def test_render_sharded_reassembles_image_on_every_rank():
    assert _run(_sharded) == [True, True]
    assert _run(_sharded, world=3) == [True, True, True]


def _fake_frame(H, W, K, chunk, c2w, **kw):
    base = c2w[:3, 3].sum() + torch.arange(H * W, dtype=torch.float32).reshape(H, W)
    return [torch.stack([base, base * 2, base * 3], -1), base * 0.5, None, {}]


def _frames(rank, world, savedir=None):
    from nerf_amd.parallel import render_path_sharded
    ok = True
    for n in (5, 1, 4):          # more, fewer and as many frames as a multiple of the ranks
        poses = torch.zeros(n, 4, 4)
        poses[:, 0, 3] = torch.arange(n, dtype=torch.float32) * 100.0
        sub = None if savedir is None else os.path.join(savedir, str(n))
        if sub is not None:
            os.makedirs(sub, exist_ok=True)
        rgbs, disps = render_path_sharded(poses, [6, 8, 10.0], None, 16, {}, render_factor=2, render_fn=_fake_frame, savedir=sub)
        want = [_fake_frame(3, 4, None, 16, poses[i]) for i in range(n)]
        ok = ok and rgbs.shape == (n, 3, 4, 3) and disps.shape == (n, 3, 4)
        ok = ok and all(torch.equal(rgbs[i], want[i][0]) and torch.equal(disps[i], want[i][1]) for i in range(n))
    return bool(ok)


def _frames_saved(rank, world):
    return _frames(rank, world, savedir=os.environ["NERF_TEST_FRAMES_DIR"])


def test_render_path_sharded_writes_every_frame_once_in_pose_order(tmp_path, monkeypatch):
    """Each rank writes the PNGs of the frames it rendered; together they are the whole sequence, named by pose."""
    from PIL import Image
    from nerf_amd.run_nerf_helpers import to8b
    monkeypatch.setenv("NERF_TEST_FRAMES_DIR", str(tmp_path))
    assert _run(_frames_saved) == [True, True]
    for n in (5, 1, 4):
        names = sorted(os.listdir(tmp_path / str(n)))
        assert names == ['{:03d}.png'.format(i) for i in range(n)]
        for i in range(n):
            pose = torch.zeros(4, 4)
            pose[0, 3] = 100.0 * i
            want = to8b(_fake_frame(3, 4, None, 16, pose)[0].numpy())
            got = np.asarray(Image.open(tmp_path / str(n) / '{:03d}.png'.format(i)))
            assert np.array_equal(got, want), (n, i)


def test_render_path_sharded_returns_frames_in_pose_order():
    assert _run(_frames) == [True, True]
    assert _run(_frames, world=3) == [True, True, True]


def _dp(rank, world):
    from nerf_amd.parallel import allreduce_gradients, broadcast_parameters
    torch.manual_seed(100 + rank)  # ranks start apart, as independently initialised processes would
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    frozen = torch.nn.Linear(2, 2)
    for p in frozen.parameters():
        p.requires_grad_(False)
    params = list(net.parameters()) + list(frozen.parameters())
    broadcast_parameters(params, src=0)
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(8 * world, 6, generator=g), torch.randn(8 * world, 3, generator=g)
    xs, ys = x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8]
    torch.mean((net(xs) - ys) ** 2).backward()
    allreduce_gradients(params)
    got = [p.grad.clone() for p in net.parameters()]
    # the same weights on the full batch in one process
    net.zero_grad()
    torch.mean((net(x) - y) ** 2).backward()
    ok = all(torch.allclose(a, p.grad, rtol=1e-5, atol=1e-7) for a, p in zip(got, net.parameters()))
    return ok, [float(p.detach().sum()) for p in net.parameters()]


def test_gradient_bucket_allreduce_equals_full_batch_gradient():
    res = _run(_dp)
    assert all(ok for ok, _ in res)
    assert res[0][1] == res[1][1], "broadcast_parameters must leave identical weights on every rank"


def test_single_process_is_a_no_op():
    from nerf_amd.parallel import allreduce_gradients, all_gather_rows
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    allreduce_gradients([p])
    assert torch.equal(p.grad, torch.full((3,), 2.0))
    t = torch.arange(6.).reshape(3, 2)
    assert all_gather_rows(t, 3) is t


def _dp_flat(rank, world):
    """Gradients that are views of one flat vector per network (what the fused MLP backward hands autograd) plus one
    separately allocated gradient and one parameter without a gradient."""
    from nerf_amd.parallel import allreduce_gradients, gradient_runs
    shapes = [(4, 3), (4,), (2, 4), (2,)]
    nets = []
    for n in range(2):
        ps = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        flat = torch.arange(sum(p.numel() for p in ps), dtype=torch.float32) * (rank + 1) + 100.0 * n
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        nets.append((ps, flat))
    lone = torch.nn.Parameter(torch.zeros(5))
    lone.grad = torch.full((5,), float(rank + 1))
    empty = torch.nn.Parameter(torch.zeros(3))          # never touched by the loss: no gradient
    params = nets[0][0] + [lone] + nets[1][0] + [empty]
    runs, rest = gradient_runs(params)
    shape_ok = [r.numel() for r in runs] == [26, 5, 26] and rest == [empty]
    before = [p.grad.data_ptr() for p in params if p.grad is not None]
    info = allreduce_gradients(params)
    same_memory = before == [p.grad.data_ptr() for p in params[:-1]]
    mean_scale = sum(r + 1 for r in range(world)) / world
    ok = True
    for n, (ps, flat) in enumerate(nets):
        want = torch.arange(flat.numel(), dtype=torch.float32) * mean_scale + 100.0 * n
        ok = ok and torch.allclose(flat, want) and torch.allclose(torch.cat([p.grad.reshape(-1) for p in ps]), want)
    ok = ok and torch.allclose(lone.grad, torch.full((5,), mean_scale)) and torch.equal(empty.grad, torch.zeros(3))
    return bool(ok and shape_ok and same_memory), info


def test_gradients_laid_out_back_to_back_are_reduced_in_place():
    for world in (2, 3):
        res = _run(_dp_flat, world=world)
        assert all(ok for ok, _ in res), res
        info = res[0][1]
        assert info["collectives"] == 4 and info["in_place_bytes"] == 4 * (26 + 5 + 26) and info["bucket_bytes"] == 12


def _forced(rank, world):
    from nerf_amd.parallel import all_gather_rows, allreduce_gradients, broadcast_parameters, render_sharded
    p = torch.nn.Parameter(torch.arange(6.).reshape(2, 3))
    p.grad = torch.full((2, 3), 3.0)
    info = allreduce_gradients([p], force_collective=True)
    broadcast_parameters([p], force_collective=True)
    t = torch.arange(10.).reshape(5, 2)
    g = all_gather_rows(t, 5, force_collective=True)
    o, d = torch.randn(9, 3), torch.randn(9, 3)
    rgb, disp, acc = render_sharded(3, 3, None, 4, (o, d), render_fn=_fake_render, force_collective=True)
    want = _fake_render(3, 3, None, 4, (o, d))
    return bool(info["collectives"] == 1 and torch.equal(p.grad, torch.full((2, 3), 3.0)) and torch.equal(g, t) and g is not t
                and torch.equal(p.detach(), torch.arange(6.).reshape(2, 3)) and torch.equal(rgb, want[0])
                and torch.equal(disp, want[1]) and torch.equal(acc, want[2]))


def test_a_group_of_one_rank_can_be_made_to_run_every_collective():
    assert _run(_forced, world=1) == [True]
