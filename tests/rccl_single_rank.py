"""Child process of tests/test_gpu_rccl.py: ONE rank on cuda:0 that initialises RCCL (backend "nccl") and drives every
collective helper of nerf_amd.parallel through its device-tensor branch -- the calls of the 8-GPU run, in a group of
one, where each result must equal its input bit for bit.  Started as a fresh interpreter so that the process group
exists before anything else has touched the GPU.  Prints one JSON object; exit status 0 = every check held."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from datetime import timedelta

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=timedelta(seconds=120))
    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "checks": {}}
    try:
        res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as exc:   # informational only
        res["rccl_version"] = repr(exc)
    ck = res["checks"]

    def same(a, b):   # bit-equal, an empty ray's NaN disparity included
        return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b.reshape(a.shape), nan=-7.0))

    from nerf_amd import parallel
    from nerf_amd.run_nerf import render, run_network
    from nerf_amd.run_nerf_helpers import NeRF, get_embedder, get_rays, img2mse
    from oracle import nerf_oracle as O

    # the patterns bench.py uses between its timed legs
    dist.barrier()
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ck["barrier_and_max"] = float(t.item()) == 1.25

    models = []
    for seed in (71, 72):
        m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict(O.random_state_dict(seed))
        models.append(m.to(dev))
    e, _ = get_embedder(10, 0)
    ed, _ = get_embedder(4, 0)
    qf = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=e, embeddirs_fn=ed, netchunk=1 << 16)
    kw = dict(network_fn=models[0], network_fine=models[1], network_query_fn=qf, N_samples=16, N_importance=16, perturb=0.0,
              raw_noise_std=0.0, white_bkgd=True, use_viewdirs=True, ndc=False, near=2.0, far=6.0)
    H = W = 24
    K = np.array([[30.0, 0, 12.0], [0, 30.0, 12.0], [0, 0, 1]])
    pose = O.orbit_pose(40.0, -30.0, 4.0)[:3, :4].to(dev)
    params = [p for m in models for p in m.parameters()]

    # data-parallel step: backward, then the gradient exchange exactly as the training loop calls it
    o, d = get_rays(H, W, K, pose)
    rays = torch.stack([o.reshape(-1, 3)[:256], d.reshape(-1, 3)[:256]], 0)
    target = torch.rand(256, 3, device=dev)
    rgb, _, _, extras = render(H, W, K, chunk=1 << 15, rays=rays, retraw=True, **kw)
    (img2mse(rgb, target) + img2mse(extras["rgb0"], target)).backward()
    before = [p.grad.clone() for p in params]
    addrs = [p.grad.data_ptr() for p in params]
    info = parallel.allreduce_gradients(params, force_collective=True)
    torch.cuda.synchronize()
    n_bytes = 4 * sum(p.numel() for p in params)
    res["allreduce"] = info
    ck["allreduce_in_place_two_runs"] = (info["collectives"] == 2 and info["in_place_bytes"] == n_bytes and info["bucket_bytes"] == 0
                                         and addrs == [p.grad.data_ptr() for p in params])
    ck["allreduce_identity"] = all(torch.equal(a, p.grad) for a, p in zip(before, params))
    # the bucket route (gradients that are not laid out back to back), also on device tensors
    for p in params:
        p.grad = p.grad.clone()
    info2 = parallel.allreduce_gradients(params, force_collective=True)
    ck["allreduce_bucket_identity"] = (info2["collectives"] == 1 and info2["bucket_bytes"] == n_bytes
                                       and all(torch.equal(a, p.grad) for a, p in zip(before, params)))

    w0 = [p.detach().clone() for p in params]
    parallel.broadcast_parameters(params, src=0, force_collective=True)
    ck["broadcast_identity"] = all(torch.equal(a, p.detach()) for a, p in zip(w0, params))

    rows = torch.arange(35, device=dev, dtype=torch.float32).reshape(7, 5)
    got = parallel.all_gather_rows(rows, 7, force_collective=True)
    ck["all_gather_rows"] = bool(torch.equal(got, rows) and got.data_ptr() != rows.data_ptr())

    with torch.no_grad():
        want = render(H, W, K, chunk=128, rays=(o.reshape(-1, 3), d.reshape(-1, 3)), **kw)
        got = parallel.render_sharded(H, W, K, 128, (o, d), force_collective=True, **kw)
        ck["render_sharded"] = all(same(a, b) for a, b in zip(got, want[:3]))
        poses = torch.stack([O.orbit_pose(a, -30.0, 4.0) for a in (0.0, 90.0, 200.0)], 0).to(dev)
        rgbs, disps = parallel.render_path_sharded(poses, [H, W, 30.0], K, 128, kw, force_collective=True)
        ok = rgbs.shape == (3, H, W, 3) and disps.shape == (3, H, W)
        for i in range(3):
            a = render(H, W, K, chunk=128, c2w=poses[i][:3, :4], **kw)
            ok = ok and same(rgbs[i], a[0]) and same(disps[i], a[1])
        ck["render_path_sharded"] = bool(ok)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    res["ok"] = all(ck.values())
    print(json.dumps(res), flush=True)
    return 0 if res["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
