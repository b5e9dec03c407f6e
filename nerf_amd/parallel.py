"""Multi-GPU use of the render path: one process per GPU, torch.distributed over RCCL (backend "nccl").

The upstream program is single-process; rays are independent, so the two extensions are
  * render_sharded: the flat ray list of one view is cut into world_size contiguous slices, every
    rank runs the unchanged single-GPU path on its slice, and ONE all_gather of rgb/disp/acc
    (20 bytes per ray) reassembles the image on every rank;
  * allreduce_gradients: data-parallel training.  Each rank draws its own rays, then the gradients
    are summed over the ranks and scaled by 1/world_size.  The fused MLP's backward leaves the
    gradients of a network as views of ONE flat vector (595,844 floats = 2.38 MB), so each network's
    vector is all-reduced where it lies (one collective per network, no concatenation, no copy back);
    gradients laid out any other way travel as one concatenated bucket.  Because the loss is a mean
    over equally sized shards, the result equals the gradient of the world_size-times larger batch.
    At this size the collectives are latency-bound over xGMI.
Both work on any backend (the CPU tests use gloo).
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of `n` items for `rank`: the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def _world(group=None) -> Tuple[int, int]:
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def rank_world(group=None) -> Tuple[int, int]:
    """(rank, world size); (0, 1) when torch.distributed is not initialised."""
    return _world(group)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int]:
    """Join the process group that torch.distributed.run describes in the environment (one process per GPU): take the
    device of LOCAL_RANK, initialise RCCL (backend "nccl"; NERF_DIST_BACKEND or `backend` overrides it, e.g. gloo to
    rehearse several ranks on one GPU).  Does nothing in a single process or when a group already exists.
    -> (rank, world size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and dist.is_available() and not dist.is_initialized():
        backend = backend or os.environ.get("NERF_DIST_BACKEND", "nccl")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            index = local % torch.cuda.device_count()
            torch.cuda.set_device(index)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", index))
                return _world()
        dist.init_process_group(backend)
    return _world()


def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo moves host memory: device tensors go through a host copy on that backend (rehearsals of the multi-GPU
    path on a box without RCCL peers); RCCL takes device tensors as they are."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_rows(local: torch.Tensor, n_total: int, group=None, force_collective: bool = False) -> torch.Tensor:
    """Concatenate per-rank row blocks cut by shard_bounds (blocks differ by at most one row).  force_collective: go
    through the collective even in a group of one rank."""
    rank, world = _world(group)
    if world == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        return local
    rows = (n_total + world - 1) // world
    pad = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if _host_staged(pad, group):
        pieces = [torch.empty(pad.shape, dtype=pad.dtype) for _ in range(world)]
        dist.all_gather(pieces, pad.cpu(), group=group)
        out = torch.cat(pieces, 0).to(local.device)
    else:
        out = torch.empty((world * rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad, group=group)
    parts = []
    for r in range(world):
        b, e = shard_bounds(n_total, r, world)
        parts.append(out[r * rows: r * rows + (e - b)])
    return torch.cat(parts, 0)


def render_sharded(H: int, W: int, K, chunk: int, rays, render_fn: Optional[Callable] = None, group=None,
                   force_collective: bool = False, **render_kwargs):
    """Render `rays` = (rays_o [N,3], rays_d [N,3]) with the ray list split across ranks.

    Returns (rgb [N,3], disp [N], acc [N]) complete on every rank.  render_fn defaults to
    nerf_amd.run_nerf.render and is called as render_fn(H, W, K, chunk=chunk, rays=(o, d), **render_kwargs)."""
    if render_fn is None:
        from .run_nerf import render as render_fn
    rank, world = _world(group)
    rays_o, rays_d = rays
    rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    n = rays_o.shape[0]
    b, e = shard_bounds(n, rank, world)
    rgb, disp, acc, _ = render_fn(H, W, K, chunk=chunk, rays=(rays_o[b:e], rays_d[b:e]), **render_kwargs)
    packed = torch.cat([rgb.reshape(-1, 3), disp.reshape(-1, 1), acc.reshape(-1, 1)], -1)  # one collective, 20 B/ray
    full = all_gather_rows(packed.contiguous(), n, group, force_collective)
    return full[:, 0:3], full[:, 3], full[:, 4]


def render_path_sharded(render_poses, hwf, K, chunk: int, render_kwargs, render_factor: int = 0,
                        render_fn: Optional[Callable] = None, group=None, savedir: Optional[str] = None,
                        force_collective: bool = False):
    """The pose loop of render_path (run_nerf.py:163-201) with the FRAMES dealt round-robin to the ranks: rank r
    renders poses r, r + world, ...; one all_gather of the finished frames returns the whole sequence, in pose
    order, on every rank as (rgbs [N,H,W,3], disps [N,H,W]) tensors.  With savedir every rank writes the PNGs of its
    own frames ('{pose index:03d}.png'), each copied to the host and encoded while the rank's next frame renders."""
    if render_fn is None:
        from .run_nerf import render as render_fn
    rank, world = _world(group)
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    H, W = int(H), int(W)
    n = len(render_poses)
    per = (n + world - 1) // world
    mine = []
    frames, pending = None, None
    if savedir is not None:
        from .run_nerf import FrameDownloader, _write_png
        from .run_nerf_helpers import to8b
        frames = FrameDownloader()

        def write(i, handle):
            (rgb8,) = frames.finish(handle)
            _write_png(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgb8))
    for i in range(rank, n, world):
        rgb, disp, _, _ = render_fn(H, W, K, chunk=chunk, c2w=render_poses[i][:3, :4], **render_kwargs)
        mine.append(torch.cat([rgb.reshape(H, W, 3), disp.reshape(H, W, 1)], -1))
        if frames is not None:
            handle = frames.start(rgb.reshape(H, W, 3))
            if pending is not None:
                write(*pending)
            pending = (i, handle)
    if pending is not None:
        write(*pending)
    dev, dtype = (mine[0].device, mine[0].dtype) if mine else (None, torch.float32)
    if world == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        frames = torch.stack(mine, 0)
    else:
        if dev is None:   # more ranks than frames: this rank has nothing to render but still joins the collective
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        block = torch.zeros((per, H, W, 4), dtype=dtype, device=dev)
        if mine:
            block[: len(mine)] = torch.stack(mine, 0)
        if _host_staged(block, group):
            pieces = [torch.empty(block.shape, dtype=dtype) for _ in range(world)]
            dist.all_gather(pieces, block.cpu(), group=group)
            out = torch.cat(pieces, 0).to(dev)
        else:
            out = torch.empty((world * per, H, W, 4), dtype=dtype, device=dev)
            dist.all_gather_into_tensor(out, block, group=group)
        # frame i was rendered by rank i % world as its (i // world)-th
        order = torch.tensor([(i % world) * per + i // world for i in range(n)], device=dev)
        frames = out[order]
    return frames[..., :3], frames[..., 3]


def flatten_gradients(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """One contiguous fp32 bucket holding every gradient (zeros where a parameter has none)."""
    chunks = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]
    return torch.cat(chunks)


def unflatten_into_gradients(bucket: torch.Tensor, params: Sequence[torch.Tensor]) -> None:
    """Copy the bucket back into the .grad tensors: one multi-tensor copy, not one launch per parameter."""
    off, dst, src = 0, [], []
    for p in params:
        g = bucket[off: off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = torch.empty_like(p)
        dst.append(p.grad)
        src.append(g)
        off += p.numel()
    torch._foreach_copy_(dst, src)


MAX_IN_PLACE_RUNS = 8


def gradient_runs(params: Sequence[torch.Tensor]) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """(runs, rest): the gradients of `params`, in order, grouped into maximal runs that already lie back to back in
    one storage, each run returned as ONE flat tensor that aliases those .grad tensors; `rest` = the parameters whose
    gradient is missing, not contiguous or not fp32.  The fused MLP backward hands autograd views of one flat
    gradient vector per network (ops._FusedMLP.backward), which autograd keeps as the .grad tensors, so a NeRF pair
    gives two runs and an empty rest: the exchange can then work in place, with no concatenation and no copy back."""
    runs, rest = [], []
    first, count, end = None, 0, None

    def close():
        if first is not None:
            runs.append(first.as_strided((count,), (1,), first.storage_offset()))
    for p in params:
        g = p.grad
        if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.numel() == 0:
            rest.append(p)
            continue
        here = g.data_ptr()
        if (first is not None and here == end and g.device == first.device
                and g.untyped_storage().data_ptr() == first.untyped_storage().data_ptr()):
            count += g.numel()
        else:
            close()
            first, count = g, g.numel()
        end = here + 4 * g.numel()
    close()
    return runs, rest


def _all_reduce_sum(t: torch.Tensor, group) -> None:
    if _host_staged(t, group):
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def allreduce_gradients(params: Iterable[torch.Tensor], group=None, force_collective: bool = False) -> dict:
    """Average the gradients of `params` over all ranks.  Gradients that already lie back to back in memory (one run
    per NeRF network, see gradient_runs) are all-reduced IN PLACE, one collective per run and one multi-tensor scale;
    whatever is left over travels as one concatenated bucket that is copied back.  force_collective: issue the
    collectives even in a group of one rank (a one-GPU check that the calls of the multi-GPU run execute).
    -> {"collectives", "in_place_bytes", "bucket_bytes"} of this call."""
    params = [p for p in params if p.requires_grad]
    rank, world = _world(group)
    info = {"collectives": 0, "in_place_bytes": 0, "bucket_bytes": 0}
    grouped = dist.is_available() and dist.is_initialized()
    if not params or not grouped or (world == 1 and not force_collective):
        return info
    runs, rest = gradient_runs(params)
    if len(runs) > MAX_IN_PLACE_RUNS:   # separately allocated gradients (stock autograd): one bucket beats a collective each
        runs, rest = [], params
    for r in runs:
        _all_reduce_sum(r, group)
        info["collectives"] += 1
        info["in_place_bytes"] += 4 * r.numel()
    if runs:
        torch._foreach_mul_(runs, 1.0 / world)
    if rest:
        bucket = flatten_gradients(rest)
        _all_reduce_sum(bucket, group)
        bucket.mul_(1.0 / world)
        unflatten_into_gradients(bucket, rest)
        info["collectives"] += 1
        info["bucket_bytes"] += 4 * bucket.numel()
    return info


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0, group=None, force_collective: bool = False) -> None:
    """Make every rank start from rank `src`'s weights (one flat broadcast)."""
    params = list(params)
    rank, world = _world(group)
    grouped = dist.is_available() and dist.is_initialized()
    if not params or not grouped or (world == 1 and not force_collective):
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        if _host_staged(flat, group):
            host = flat.cpu()
            dist.broadcast(host, src=src, group=group)
            flat = host.to(flat.device)
        else:
            dist.broadcast(flat, src=src, group=group)
        off = 0
        for p in params:
            p.copy_(flat[off: off + p.numel()].view_as(p))
            off += p.numel()
