"""Drop-in for the render path of the upstream run_nerf.py (render / batchify_rays / render_rays /
raw2outputs / run_network / create_nerf / render_path / config_parser): same signatures, keyword
names, return structure and checkpoint layout; the arithmetic runs in libnerf_hip.so.
Citations are file:line of the upstream tree."""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from . import ops
from .run_nerf_helpers import *  # noqa: F401,F403  (upstream does the same, run_nerf.py:14)
from .run_nerf_helpers import NeRF, get_embedder, get_rays

DEBUG = False


def batchify(fn, chunk):
    """run_nerf.py:28-38."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """run_nerf.py:40-66.  With the module's own NeRF and embedders the encoding, the concatenation and
    every layer run as ONE fused kernel over all points; netchunk (a memory knob upstream) then has
    nothing to bound.  Any other `fn` gets the encoded features in netchunk slices, as upstream."""
    fused = (isinstance(fn, NeRF) and fn.fused_arch() and getattr(embed_fn, 'n_freq', None) == 10
             and (viewdirs is None or getattr(embeddirs_fn, 'n_freq', None) == 4)
             and fn.use_viewdirs == (viewdirs is not None))
    if fused:
        pts = ops.to_dev(inputs)
        S = int(np.prod(pts.shape[1:-1])) if pts.dim() > 2 else 1
        out = fn.query(pts, None if viewdirs is None else ops.to_dev(viewdirs), S)
        return torch.reshape(out, list(inputs.shape[:-1]) + [out.shape[-1]])
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    embedded = embed_fn(inputs_flat)
    if viewdirs is not None:
        input_dirs = viewdirs[:, None].expand(inputs.shape)
        input_dirs_flat = torch.reshape(input_dirs, [-1, input_dirs.shape[-1]])
        embedded = torch.cat([embedded, embeddirs_fn(input_dirs_flat)], -1)
    outputs_flat = batchify(fn, netchunk)(embedded)
    return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """run_nerf.py:71-85."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            if k not in all_ret:
                all_ret[k] = []
            all_ret[k].append(ret[k])
    # (a single chunk -- every training batch -- is returned as it is: concatenating one tensor is a copy per key)
    all_ret = {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}
    return all_ret


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """run_nerf.py:87-160.  Returns [rgb_map, disp_map, acc_map, extras]."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, K, c2w)
    else:
        rays_o, rays_d = rays
        rays_o, rays_d = ops.to_dev(rays_o), ops.to_dev(rays_d)

    viewdir_src = None
    if use_viewdirs:
        viewdir_src = rays_d
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)

    sh = rays_d.shape
    ndc_par = None
    if ndc:
        focal = K[0][0]
        ndc_par = (float(-1. / (W / (2. * focal))), float(-1. / (H / (2. * focal))), 1.0)
    rays = ops.pack_rays(rays_o, rays_d, near, far, viewdir_src, ndc_par)

    all_ret = batchify_rays(rays, chunk, **kwargs)
    for k in all_ret:
        k_sh = list(sh[:-1]) + list(all_ret[k].shape[1:])
        all_ret[k] = torch.reshape(all_ret[k], k_sh)

    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    ret_list = [all_ret[k] for k in k_extract]
    ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
    return ret_list + [ret_dict]


class FrameDownloader:
    """Finished frames -> host arrays without stalling the stream that renders the next frame.

    start(*tensors) records an event on the current stream, lets a side stream wait for it and copy the tensors into
    one of two sets of pinned host buffers; finish(handle) waits for that copy alone and returns numpy copies.  Used
    as start(frame i); finish(frame i-1): the device-to-host copy of a frame and its 8-bit conversion / PNG encoding on
    the host overlap the rendering of the next one, and the host never synchronises the render stream (upstream's
    `.cpu()` per frame, run_nerf.py:181-182, does).  CPU tensors pass straight through."""

    def __init__(self, slots: int = 2):
        self.slots = [None] * slots
        self.count = 0
        self.side = None

    def start(self, *tensors):
        if not tensors[0].is_cuda:
            return ("host", [t.detach().numpy().copy() for t in tensors])
        dev = tensors[0].device
        if self.side is None:
            self.side = torch.cuda.Stream(device=dev)
        slot = self.count % len(self.slots)
        self.count += 1
        host = self.slots[slot]
        if host is None or any(h.shape != t.shape or h.dtype != t.dtype for h, t in zip(host, tensors)):
            host = self.slots[slot] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors]
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        self.side.wait_event(ready)
        with torch.cuda.stream(self.side):
            for h, t in zip(host, tensors):
                h.copy_(t.detach(), non_blocking=True)
                t.record_stream(self.side)       # the allocator must not hand the frame's memory out before the copy ran
            done = torch.cuda.Event()
            done.record(self.side)
        return ("pinned", slot, done)

    def finish(self, handle):
        if handle[0] == "host":
            return handle[1]
        _, slot, done = handle
        done.synchronize()
        return [h.numpy().copy() for h in self.slots[slot]]


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """run_nerf.py:163-201: one render() per pose, frames returned as numpy arrays and written as PNG (PIL) when savedir
    is given.  As upstream, render_factor shrinks H, W and the focal of `hwf` but K is passed on unchanged.  Frame i is
    copied to the host and encoded while frame i+1 renders (FrameDownloader)."""
    H, W, focal = hwf
    if render_factor != 0:
        H = H // render_factor
        W = W // render_factor
        focal = focal / render_factor

    rgbs, disps = [], []
    frames = FrameDownloader()

    def collect(i, handle):
        rgb, disp = frames.finish(handle)
        rgbs.append(rgb)
        disps.append(disp)
        if i == 0:
            print(rgb.shape, disp.shape)
        if savedir is not None:
            _write_png(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgb))  # noqa: F405

    t = time.time()
    pending = None
    for i, c2w in enumerate(render_poses):
        print(i, time.time() - t)
        t = time.time()
        rgb, disp, acc, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        handle = frames.start(rgb, disp)
        if pending is not None:
            collect(*pending)
        pending = (i, handle)
    if pending is not None:
        collect(*pending)
    return np.stack(rgbs, 0), np.stack(disps, 0)


def create_nerf(args):
    """run_nerf.py:204-289: models, query function, Adam, checkpoint reload, render kwargs."""
    dev = ops.device()
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views = 0
    embeddirs_fn = None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(dev)
    grad_vars = list(model.parameters())

    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                          skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(dev)
        grad_vars += list(model_fine.parameters())

    network_query_fn = lambda inputs, viewdirs, network_fn: run_network(inputs, viewdirs, network_fn,
                                                                        embed_fn=embed_fn,
                                                                        embeddirs_fn=embeddirs_fn,
                                                                        netchunk=args.netchunk)

    # the parameters move into their flat buffers NOW, before the optimiser (and any HIP graph or shadow copy a caller
    # makes afterwards) takes their addresses: from here on data_ptr() of every parameter is stable
    for net in (model, model_fine):
        if net is not None and net.fused_arch():
            net.adopt_flat()

    # (torch.optim.Adam with the same state and checkpoint format; the parameters of each network lie back to back, and so
    # do their gradients, so a step is one launch of the library's Adam kernel over both networks; step counters and,
    # if it is made a tensor, the learning rate live on the device, so the optimiser can be recorded into a HIP graph)
    from .optim import FlatAdam
    optimizer = FlatAdam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))

    start = 0
    basedir = args.basedir
    expname = args.expname

    if args.ft_path is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        ckpts = [os.path.join(basedir, expname, f) for f in sorted(os.listdir(os.path.join(basedir, expname)))
                 if 'tar' in f]

    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not args.no_reload:
        ckpt_path = ckpts[-1]
        print('Reloading from', ckpt_path)
        ckpt = torch.load(ckpt_path, map_location=dev)
        start = ckpt['global_step']
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        model.load_state_dict(ckpt['network_fn_state_dict'])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt['network_fine_state_dict'])

    render_kwargs_train = {
        'network_query_fn': network_query_fn,
        'perturb': args.perturb,
        'N_importance': args.N_importance,
        'network_fine': model_fine,
        'N_samples': args.N_samples,
        'network_fn': model,
        'use_viewdirs': args.use_viewdirs,
        'white_bkgd': args.white_bkgd,
        'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        print('Not ndc!')
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp

    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer


def save_checkpoint(path, global_step, render_kwargs_train, optimizer):
    """The checkpoint dict of the upstream train loop (run_nerf.py:890-898)."""
    torch.save({
        'global_step': global_step,
        'network_fn_state_dict': render_kwargs_train['network_fn'].state_dict(),
        'network_fine_state_dict': render_kwargs_train['network_fine'].state_dict(),
        'optimizer_state_dict': optimizer.state_dict(),
    }, path)


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """run_nerf.py:292-362 -> rgb_map, disp_map, acc_map, weights, depth_map."""
    raw, z_vals = ops.to_dev(raw), ops.to_dev(z_vals)
    if not (torch.is_tensor(rays_d) and rays_d.is_cuda and rays_d.dtype == torch.float32):
        rays_d = ops.to_dev(rays_d)
    noise = None
    if raw_noise_std > 0.:
        if pytest:
            np.random.seed(0)
            noise = torch.Tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std).to(raw.device)
        else:
            noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std
    return ops.composite(raw, z_vals, rays_d, noise, white_bkgd)


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False):
    """run_nerf.py:366-491."""
    ray_batch = ops.to_dev(ray_batch)
    N_rays = ray_batch.shape[0]
    dev = ray_batch.device
    rays_d = ray_batch[:, 3:6]
    # (a column slice of the batch: made contiguous once here, not once per network inside the MLP call)
    viewdirs = ray_batch[:, -3:].contiguous() if ray_batch.shape[-1] > 8 else None

    t_vals = ops.cached_linspace(N_samples, dev)
    t_rand = None
    if perturb > 0.:
        if pytest:
            np.random.seed(0)
            t_rand = torch.Tensor(np.random.rand(N_rays, N_samples)).to(dev)
        else:
            t_rand = torch.rand(N_rays, N_samples, device=dev)
    z_vals, pts = ops.sample_coarse(ray_batch, t_vals, t_rand, lindisp)

    raw = network_query_fn(pts, viewdirs, network_fn)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd,
                                                                 pytest=pytest)

    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0 = rgb_map, disp_map, acc_map
        if pytest:
            np.random.seed(0)
            if perturb == 0.:
                u = torch.Tensor(np.linspace(0., 1., N_importance)).to(dev)
            else:
                u = torch.Tensor(np.random.rand(N_rays, N_importance)).to(dev)
        elif perturb == 0.:
            u = ops.cached_linspace(N_importance, dev)
        else:
            u = torch.rand(N_rays, N_importance, device=dev)
        fine = ops.sample_fine(z_vals, weights.detach(), u, ray_batch)
        z_vals, pts = fine["z"], fine["pts"]

        run_fn = network_fn if network_fine is None else network_fine
        raw = network_query_fn(pts, viewdirs, run_fn)
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd,
                                                                     pytest=pytest)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'] = rgb_map_0
        ret['disp0'] = disp_map_0
        ret['acc0'] = acc_map_0
        ret['z_std'] = fine["z_std"]

    if DEBUG:
        for k in ret:
            if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret


# ------------------------------------------------------------------------------------------------
# flags (run_nerf.py:494-604).  configargparse is not a dependency here: --config names a file of
# `key = value` lines whose entries act as defaults that the command line overrides.
# ------------------------------------------------------------------------------------------------
class _ConfigParser(argparse.ArgumentParser):
    def parse_known_args(self, args=None, namespace=None):
        import sys
        argv = list(sys.argv[1:] if args is None else args)
        cfg = None
        for i, a in enumerate(argv):
            if a == '--config' and i + 1 < len(argv):
                cfg = argv[i + 1]
            elif a.startswith('--config='):
                cfg = a.split('=', 1)[1]
        file_args = []
        if cfg is not None:
            with open(cfg, 'r') as fh:
                for line in fh:
                    line = line.split('#', 1)[0].strip()
                    if not line or '=' not in line:
                        continue
                    key, val = [s.strip() for s in line.split('=', 1)]
                    action = next((a for a in self._actions if ('--' + key) in a.option_strings), None)
                    if action is None:
                        raise SystemExit(f"unknown option '{key}' in {cfg}")
                    if isinstance(action, argparse._StoreTrueAction):
                        if val.lower() in ('true', '1', 'yes'):
                            file_args.append('--' + key)
                    else:
                        file_args += ['--' + key, val]
        return super().parse_known_args(file_args + argv, namespace)


def config_parser():
    parser = _ConfigParser()
    parser.add_argument('--config', type=str, default=None, help='config file path')
    parser.add_argument("--expname", type=str, help='experiment name')
    parser.add_argument("--basedir", type=str, default='./logs/', help='where to store ckpts and logs')
    parser.add_argument("--datadir", type=str, default='./data/llff/fern', help='input data directory')
    # training options
    parser.add_argument("--netdepth", type=int, default=8, help='layers in network')
    parser.add_argument("--netwidth", type=int, default=256, help='channels per layer')
    parser.add_argument("--netdepth_fine", type=int, default=8, help='layers in fine network')
    parser.add_argument("--netwidth_fine", type=int, default=256, help='channels per layer in fine network')
    parser.add_argument("--N_rand", type=int, default=32 * 32 * 4, help='batch size (number of random rays per gradient step)')
    parser.add_argument("--lrate", type=float, default=5e-4, help='learning rate')
    parser.add_argument("--lrate_decay", type=int, default=250, help='exponential learning rate decay (in 1000 steps)')
    parser.add_argument("--chunk", type=int, default=1024 * 32, help='number of rays processed in parallel')
    parser.add_argument("--netchunk", type=int, default=1024 * 64, help='number of pts sent through network in parallel')
    parser.add_argument("--no_batching", action='store_true', help='only take random rays from 1 image at a time')
    parser.add_argument("--no_reload", action='store_true', help='do not reload weights from saved ckpt')
    parser.add_argument("--ft_path", type=str, default=None, help='specific weights file to reload for coarse network')
    # rendering options
    parser.add_argument("--N_samples", type=int, default=64, help='number of coarse samples per ray')
    parser.add_argument("--N_importance", type=int, default=0, help='number of additional fine samples per ray')
    parser.add_argument("--perturb", type=float, default=1., help='set to 0. for no jitter, 1. for jitter')
    parser.add_argument("--use_viewdirs", action='store_true', help='use full 5D input instead of 3D')
    parser.add_argument("--i_embed", type=int, default=0, help='set 0 for default positional encoding, -1 for none')
    parser.add_argument("--multires", type=int, default=10, help='log2 of max freq for positional encoding (3D location)')
    parser.add_argument("--multires_views", type=int, default=4, help='log2 of max freq for positional encoding (2D direction)')
    parser.add_argument("--raw_noise_std", type=float, default=0., help='std dev of noise added to regularize sigma_a output')
    parser.add_argument("--render_only", action='store_true', help='do not optimize, reload weights and render out render_poses path')
    parser.add_argument("--render_test", action='store_true', help='render the test set instead of render_poses path')
    parser.add_argument("--render_factor", type=int, default=0, help='downsampling factor to speed up rendering')
    # training options
    parser.add_argument("--precrop_iters", type=int, default=0, help='number of steps to train on central crops')
    parser.add_argument("--precrop_frac", type=float, default=.5, help='fraction of img taken for central crops')
    # dataset options
    parser.add_argument("--dataset_type", type=str, default='llff', help='options: llff / blender / deepvoxels')
    parser.add_argument("--testskip", type=int, default=8, help='will load 1/N images from test/val sets')
    parser.add_argument("--shape", type=str, default='greek', help='options : armchair / cube / greek / vase')
    parser.add_argument("--white_bkgd", action='store_true', help='set to render synthetic data on a white bkgd')
    parser.add_argument("--half_res", action='store_true', help='load blender synthetic data at 400x400 instead of 800x800')
    parser.add_argument("--factor", type=int, default=8, help='downsample factor for LLFF images')
    parser.add_argument("--no_ndc", action='store_true', help='do not use normalized device coordinates')
    parser.add_argument("--lindisp", action='store_true', help='sampling linearly in disparity rather than depth')
    parser.add_argument("--spherify", action='store_true', help='set for spherical 360 scenes')
    parser.add_argument("--llffhold", type=int, default=8, help='will take every 1/N images as LLFF test set')
    # logging/saving options
    parser.add_argument("--i_print", type=int, default=100, help='frequency of console printout and metric loggin')
    parser.add_argument("--i_img", type=int, default=500, help='frequency of tensorboard image logging')
    parser.add_argument("--i_weights", type=int, default=10000, help='frequency of weight ckpt saving')
    parser.add_argument("--i_testset", type=int, default=50000, help='frequency of testset saving')
    parser.add_argument("--i_video", type=int, default=50000, help='frequency of render_poses video saving')
    # not upstream (the upstream loop is hard-wired to 200001 iterations)
    parser.add_argument("--N_iters", type=int, default=200000 + 1, help='last iteration + 1')
    return parser


# ------------------------------------------------------------------------------------------------
# training driver (run_nerf.py:607-971)
# ------------------------------------------------------------------------------------------------
def _write_png(path, rgb8):
    from PIL import Image
    Image.fromarray(rgb8).save(path)


def _save_frames(prefix, frames8):
    """Video frames as numbered PNGs (there is no mp4 writer among this package's dependencies)."""
    os.makedirs(prefix, exist_ok=True)
    for k, fr in enumerate(frames8):
        _write_png(os.path.join(prefix, '{:03d}.png'.format(k)), fr)


def _render_frames(rank, world, render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """render_path for the driver: alone, or with the frames dealt to all ranks (every rank must call it).  With
    savedir each rank writes the PNGs of the frames it rendered, so the directory must be on storage that all ranks
    share (one node, or a shared file system); the complete sequence is returned on every rank either way.
    -> (rgbs, disps) numpy arrays, complete on every rank."""
    from . import parallel
    if world == 1:
        return render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=gt_imgs, savedir=savedir,
                           render_factor=render_factor)
    rgbs, disps = parallel.render_path_sharded(render_poses, hwf, K, chunk, render_kwargs, render_factor=render_factor,
                                               savedir=savedir)   # every rank writes the PNGs of its own frames
    return rgbs.cpu().numpy(), disps.cpu().numpy()


def load_dataset(args):
    """-> images, poses, render_poses, hwf, K, (i_train, i_val, i_test), near, far  (run_nerf.py:612-701)."""
    from . import datasets
    K = None
    if args.dataset_type == 'llff':
        images, poses, bds, render_poses, i_test = datasets.load_llff_data(args.datadir, args.factor, recenter=True,
                                                                           bd_factor=.75, spherify=args.spherify)
        hwf = poses[0, :3, -1]
        poses = poses[:, :3, :4]
        print('Loaded llff', images.shape, render_poses.shape, hwf, args.datadir)
        if not isinstance(i_test, list):
            i_test = [i_test]
        if args.llffhold > 0:
            print('Auto LLFF holdout,', args.llffhold)
            i_test = np.arange(images.shape[0])[::args.llffhold]
        i_val = i_test
        i_train = np.array([i for i in np.arange(int(images.shape[0])) if (i not in i_test and i not in i_val)])
        print('DEFINING BOUNDS')
        if args.no_ndc:
            near, far = np.ndarray.min(bds) * .9, np.ndarray.max(bds) * 1.
        else:
            near, far = 0., 1.
        print('NEAR FAR', near, far)
    elif args.dataset_type == 'blender':
        images, poses, render_poses, hwf, i_split = datasets.load_blender_data(args.datadir, args.half_res, args.testskip)
        print('Loaded blender', images.shape, render_poses.shape, hwf, args.datadir)
        i_train, i_val, i_test = i_split
        near, far = 2., 6.
        if args.white_bkgd:
            images = images[..., :3] * images[..., -1:] + (1. - images[..., -1:])
        else:
            images = images[..., :3]
    else:
        raise NotImplementedError(f"dataset_type '{args.dataset_type}': this package reads 'blender' and 'llff' scenes")
    H, W, focal = hwf
    H, W = int(H), int(W)
    hwf = [H, W, focal]
    if K is None:
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    return images, poses, render_poses, hwf, K, (i_train, i_val, i_test), near, far


def train(argv=None):
    """The upstream optimisation driver: data, log directory, models, ray batches, the step, checkpoints and
    periodic test renders.  Under torch.distributed (one process per GPU) the step is data parallel: every rank
    draws its own N_rand rays, gradients are averaged with one all-reduce, rank 0 writes the files."""
    from . import parallel
    from .raybatch import RayShuffler, ViewSampler
    args = config_parser().parse_args(argv)
    rank, world = parallel.init_from_env()
    dev = ops.device()
    images, poses, render_poses, hwf, K, (i_train, i_val, i_test), near, far = load_dataset(args)
    H, W, focal = hwf
    if args.render_test:
        render_poses = np.array(poses[i_test])

    basedir, expname = args.basedir, args.expname
    os.makedirs(os.path.join(basedir, expname), exist_ok=True)
    if rank == 0:
        with open(os.path.join(basedir, expname, 'args.txt'), 'w') as fh:
            for arg in sorted(vars(args)):
                fh.write('{} = {}\n'.format(arg, getattr(args, arg)))
        if args.config is not None:
            with open(os.path.join(basedir, expname, 'config.txt'), 'w') as fh:
                fh.write(open(args.config, 'r').read())

    render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer = create_nerf(args)
    global_step = start
    bds_dict = {'near': near, 'far': far}
    render_kwargs_train.update(bds_dict)
    render_kwargs_test.update(bds_dict)
    if world > 1:
        parallel.broadcast_parameters(grad_vars)
    render_poses = torch.Tensor(np.asarray(render_poses)).to(dev)

    if args.render_only:
        print('RENDER ONLY')
        with torch.no_grad():
            gt = images[i_test] if args.render_test else None
            testsavedir = os.path.join(basedir, expname, 'renderonly_{}_{:06d}'.format('test' if args.render_test else 'path', start))
            os.makedirs(testsavedir, exist_ok=True)
            print('test poses shape', render_poses.shape)
            rgbs, _ = _render_frames(rank, world, render_poses, hwf, K, args.chunk, render_kwargs_test, gt_imgs=gt,
                                     savedir=testsavedir, render_factor=args.render_factor)
            print('Done rendering', testsavedir)
        return rgbs

    N_rand = args.N_rand
    if not args.no_batching:
        print('get rays')
        batcher = RayShuffler(images, poses, i_train, H, W, K, N_rand, device=dev, seed=0, rank=rank, world_size=world)
        print('done')
    else:
        batcher = ViewSampler(images, poses, i_train, H, W, K, N_rand, args.precrop_iters, args.precrop_frac, device=dev,
                              seed=rank)
        if args.precrop_iters > 0:
            w = batcher.window(0)
            print(f"[Config] Center cropping of size {w[2]} x {w[3]} is enabled until iter {args.precrop_iters}")
    poses_dev = torch.Tensor(np.asarray(poses)).to(dev)

    # A full pass of Python's cyclic collector over everything imported and loaded so far holds the launching thread for
    # some 80 ms every few dozen steps (a 1024-ray step is under 5 ms).  What exists now lives until the process ends:
    # set it aside, so that later passes look at the loop's own objects only.
    import gc
    gc.collect()
    gc.freeze()

    print('Begin')
    print('TRAIN views are', i_train)
    print('TEST views are', i_test)
    print('VAL views are', i_val)
    loss = None
    for i in range(start + 1, args.N_iters):
        batch_rays, target_s = batcher.next() if not args.no_batching else batcher.next(i)

        rgb, disp, acc, extras = render(H, W, K, chunk=args.chunk, rays=batch_rays, verbose=i < 10, retraw=True,
                                        **render_kwargs_train)
        optimizer.zero_grad()
        if 'rgb0' in extras:
            img_loss, _, loss = img2mse_pair(rgb, extras['rgb0'], target_s)  # noqa: F405
        else:
            loss = img_loss = img2mse(rgb, target_s)  # noqa: F405
        loss.backward()
        if world > 1:
            parallel.allreduce_gradients(grad_vars)
        optimizer.step()

        new_lrate = args.lrate * (0.1 ** (global_step / (args.lrate_decay * 1000)))
        for param_group in optimizer.param_groups:
            param_group['lr'] = new_lrate

        if rank == 0 and i % args.i_weights == 0:
            path = os.path.join(basedir, expname, '{:06d}.tar'.format(i))
            save_checkpoint(path, global_step, render_kwargs_train, optimizer)
            print('Saved checkpoints at', path)
        # the periodic renders involve every rank (frames are dealt round-robin); rank 0 writes the movie frames, every
        # rank the test-set PNGs of the frames it rendered (the log directory is shared by the ranks of a node)
        if i % args.i_video == 0 and i > 0:
            with torch.no_grad():
                rgbs, disps = _render_frames(rank, world, render_poses, hwf, K, args.chunk, render_kwargs_test)
            if rank == 0:
                print('Done, saving', rgbs.shape, disps.shape)
                moviebase = os.path.join(basedir, expname, '{}_spiral_{:06d}_'.format(expname, i))
                _save_frames(moviebase + 'rgb', to8b(rgbs))  # noqa: F405
                _save_frames(moviebase + 'disp', to8b(disps / np.max(disps)))  # noqa: F405
        if i % args.i_testset == 0 and i > 0:
            testsavedir = os.path.join(basedir, expname, 'testset_{:06d}'.format(i))
            os.makedirs(testsavedir, exist_ok=True)
            if rank == 0:
                print('test poses shape', poses_dev[i_test].shape)
            with torch.no_grad():
                _render_frames(rank, world, poses_dev[i_test], hwf, K, args.chunk, render_kwargs_test, gt_imgs=images[i_test],
                               savedir=testsavedir)
            if rank == 0:
                print('Saved test set')
        if rank == 0 and i % args.i_print == 0:
            psnr = mse2psnr(img_loss.detach())  # noqa: F405  (three launches: only when it is printed)
            print(f"[TRAIN] Iter: {i} Loss: {loss.item()}  PSNR: {psnr.item()}")
        global_step += 1
    return None if loss is None else float(loss.detach())
