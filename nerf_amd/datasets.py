"""Dataset readers with the return values of the upstream loaders (load_blender.py:34-97, load_llff.py:65-385).

    load_blender_data(basedir, half_res, testskip) -> images [N,H,W,4], poses [N,4,4], render_poses [40,4,4],
                                                     [H, W, focal], [i_train, i_val, i_test]
    load_llff_data(basedir, factor, recenter, bd_factor, spherify, path_zflat)
                                                  -> images [N,H,W,3], poses [N,3,5], bds [N,2], render_poses, i_test

On-disk formats: transforms_{train,val,test}.json + RGBA PNG; poses_bounds.npy [N,17] + images[_factor]/.
PNG/JPEG decoding uses PIL (imageio / OpenCV are not dependencies here).  The camera-axis convention of the
LLFF poses is this fork's (load_llff.py:293): [x, -y, -z | t | hwf].
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------
# Blender synthetic scenes
# ------------------------------------------------------------------------------------------------
def pose_spherical(theta, phi, radius):
    """Camera on a sphere of `radius` looking at the origin, angles in degrees (load_blender.py:9-32).
    The factors are float32 matrices multiplied in float32, like upstream."""
    ph, th = phi / 180. * np.pi, theta / 180. * np.pi
    lift = torch.Tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]]).float()
    tilt = torch.Tensor([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]]).float()
    turn = torch.Tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]]).float()
    swap = torch.Tensor(np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]))
    return swap @ (turn @ (tilt @ lift))


def _decode(path, mode, size=None):
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert(mode)
        if size is not None and im.size != size:
            im = im.resize(size, Image.LANCZOS)
        return np.asarray(im)


def load_blender_data(basedir, half_res=False, testskip=1):
    images, poses, bounds = [], [], [0]
    for split in ('train', 'val', 'test'):
        with open(os.path.join(basedir, f'transforms_{split}.json'), 'r') as fh:
            meta = json.load(fh)
        step = 1 if (split == 'train' or testskip == 0) else testskip
        frames = meta['frames'][::step]
        images.append((np.array([_decode(os.path.join(basedir, f['file_path'] + '.png'), "RGBA") for f in frames])
                       / 255.).astype(np.float32))
        poses.append(np.array([f['transform_matrix'] for f in frames]).astype(np.float32))
        bounds.append(bounds[-1] + len(frames))
    i_split = [np.arange(bounds[k], bounds[k + 1]) for k in range(3)]
    images, poses = np.concatenate(images, 0), np.concatenate(poses, 0)

    H, W = images.shape[1:3]
    focal = .5 * W / np.tan(.5 * float(meta['camera_angle_x']))       # the last split's field of view, as upstream
    render_poses = torch.stack([pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 40 + 1)[:-1]], 0)
    if half_res:
        # area interpolation at an exact factor of two = the mean of each 2x2 block; upstream's result is float64
        H, W, focal = H // 2, W // 2, focal / 2.
        images = images[:, :2 * H, :2 * W].astype(np.float64).reshape(-1, H, 2, W, 2, 4).mean((2, 4))
    return images, poses, render_poses, [H, W, focal], i_split


# ------------------------------------------------------------------------------------------------
# LLFF scenes
# ------------------------------------------------------------------------------------------------
def _unit(v):
    return v / np.linalg.norm(v)


def _look_along(z, up, pos):
    """[right | up' | forward | position] with forward = z/|z| (load_llff.py:141-149)."""
    fwd = _unit(z)
    right = _unit(np.cross(up, fwd))
    return np.stack([right, _unit(np.cross(fwd, right)), fwd, pos], 1)


def _mean_pose(poses):
    """load_llff.py:156-166: mean position, summed forward and up axes, hwf of the first pose."""
    # (the summed forward axis is normalised here and once more inside _look_along, as upstream does)
    return np.concatenate([_look_along(_unit(poses[:, :3, 2].sum(0)), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0)),
                           poses[0, :3, -1:]], 1)


def _homogeneous(p34):
    last = np.broadcast_to(np.array([0, 0, 0, 1.]), p34.shape[:-2] + (1, 4))
    return np.concatenate([p34, last], -2)


def _recentre(poses):
    """Express every pose in the frame of the mean pose (load_llff.py:200-217)."""
    out = poses + 0
    out[:, :3, :4] = (np.linalg.inv(_homogeneous(_mean_pose(poses)[:3, :4])) @ _homogeneous(poses[:, :3, :4]))[:, :3, :4]
    return out


def _spiral_path(c2w, up, rads, focal, zrate, rots, n_views):
    """load_llff.py:170-196: n_views poses on a spiral around c2w, all looking at a point `focal` ahead."""
    rads = np.array(list(rads) + [1.])
    target = np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.]))
    out = []
    for theta in np.linspace(0., 2. * np.pi * rots, int(n_views) + 1)[:-1]:
        eye = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        out.append(np.concatenate([_look_along(eye - target, up, eye), c2w[:, 4:5]], 1))
    return out


def _spherify(poses, bds):
    """360-degree captures (load_llff.py:219-275): centre on the point nearest to all optical axes, scale the
    cameras onto the unit sphere, render path = a circle at the cameras' mean height."""
    axis, origin = poses[:, :3, 2:3], poses[:, :3, 3:4]
    A = np.eye(3) - axis * np.transpose(axis, [0, 2, 1])
    b = -A @ origin
    centre = np.squeeze(-np.linalg.inv((np.transpose(A, [0, 2, 1]) @ A).mean(0)) @ b.mean(0))
    v0 = _unit((poses[:, :3, 3] - centre).mean(0))
    v1 = _unit(np.cross([.1, .2, .3], v0))
    v2 = _unit(np.cross(v0, v1))
    frame = np.stack([v1, v2, v0, centre], 1)
    reset = np.linalg.inv(_homogeneous(frame[None])) @ _homogeneous(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    sc = 1. / rad
    reset[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    zh = np.mean(reset[:, :3, 3], 0)[2]
    ring = np.sqrt(rad ** 2 - zh ** 2)
    circle = []
    for th in np.linspace(0., 2. * np.pi, 120):
        eye = np.array([ring * np.cos(th), ring * np.sin(th), zh])
        fwd = _unit(eye)
        right = _unit(np.cross(fwd, np.array([0, 0, -1.])))
        circle.append(np.stack([right, _unit(np.cross(fwd, right)), fwd, eye], 1))
    circle = np.stack(circle, 0)
    hwf = poses[0, :3, -1:]
    circle = np.concatenate([circle, np.broadcast_to(hwf, circle[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, reset[:, :3, -1:].shape)], -1)
    return reset, circle, bds


def _image_files(d):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(('JPG', 'jpg', 'png'))]


def _read_llff(basedir, factor):
    """poses [3,5,N], bds [2,N], images [H,W,3,N] (load_llff.py:65-126, the `factor` route).  Without a
    pre-shrunk images_{factor}/ directory the full-size images are reduced in memory (Lanczos)."""
    arr = np.load(os.path.join(basedir, 'poses_bounds.npy'))
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = arr[:, -2:].transpose([1, 0])
    shrunk = os.path.join(basedir, 'images' + ('' if factor is None else f'_{factor}'))
    factor = 1 if factor is None else factor
    size = None
    if os.path.isdir(shrunk):
        files = _image_files(shrunk)
    else:
        files = _image_files(os.path.join(basedir, 'images'))
        h0, w0 = _decode(files[0], "RGB").shape[:2]
        size = (int(round(w0 / factor)), int(round(h0 / factor)))
    if poses.shape[-1] != len(files):
        raise ValueError(f'Mismatch between imgs {len(files)} and poses {poses.shape[-1]} !!!!')
    imgs = np.stack([_decode(f, "RGB", size) / 255. for f in files], -1)
    poses[:2, 4, :] = np.array(imgs.shape[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    print('Loaded image data', imgs.shape, poses[:, -1, 0])
    return poses, bds, imgs


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    poses, bds, imgs = _read_llff(basedir, factor)
    return _prepare_llff(poses, bds, imgs, recenter, bd_factor, spherify, path_zflat)


def _prepare_llff(poses, bds, imgs, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """Everything load_llff_data does after reading the files (load_llff.py:287-385)."""
    poses = np.concatenate([poses[:, 0:1, :], -poses[:, 1:2, :], -poses[:, 2:3, :], poses[:, 3:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)

    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = _recentre(poses)
    if spherify:
        poses, render_poses, bds = _spherify(poses, bds)
    else:
        c2w = _mean_pose(poses)
        up = _unit(poses[:, :3, 1].sum(0))
        close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
        dt = .75
        focal = 1. / (((1. - dt) / close_depth + dt / inf_depth))     # depth the spiral cameras look at
        rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
        n_views, n_rots = 120, 2
        if path_zflat:
            c2w[:3, 3] = c2w[:3, 3] + (-close_depth * .1) * c2w[:3, 2]
            rads[2] = 0.
            n_rots, n_views = 1, n_views / 2
        render_poses = _spiral_path(c2w, up, rads, focal, zrate=.5, rots=n_rots, n_views=n_views)
    render_poses = np.array(render_poses).astype(np.float32)

    centre = _mean_pose(poses)
    print('Data:')
    print(poses.shape, images.shape, bds.shape)
    i_test = np.argmin(np.sum(np.square(centre[:3, 3] - poses[:, :3, 3]), -1))
    print('HOLDOUT view is', i_test)
    return images.astype(np.float32), poses.astype(np.float32), bds, render_poses, i_test
