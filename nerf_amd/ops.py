"""Tensor-level wrappers around the C-ABI kernels.

PyTorch is used for device memory, the current HIP stream and autograd graph edges only; every
number is produced by libnerf_hip.so.  All wrappers require fp32 tensors on a HIP device and raise
if the library reports an error.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch

from ._lib import check, lib


# When set to a list, kernel launches append (name, units, start_event, end_event) recorded on the launch stream.
# units: sample points for the MLP kernels, algorithmic HBM bytes for the streaming kernels.
PROFILE = None
# When set to a list, the two-piece MLP stages append (stage, flag words) -- one word per 128-point tile, 1 = the tile left
# fp16's range and was redone by the three-piece kernel -- so that a caller can count marked tiles after a step.
MARKED = None


class _timed:
    """Brackets one launch with HIP events on the current stream when PROFILE is a list; free otherwise."""
    __slots__ = ("name", "units", "ev0")

    def __init__(self, name, units):
        self.name, self.units, self.ev0 = name, units, None

    def __enter__(self):
        if PROFILE is not None:
            self.ev0 = torch.cuda.Event(enable_timing=True)
            self.ev0.record()
        return self

    def __exit__(self, *exc):
        if self.ev0 is not None and PROFILE is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            PROFILE.append((self.name, self.units, self.ev0, ev1))
        return False


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("nerf_amd needs a HIP device: torch.cuda.is_available() is False and there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(x, dtype=torch.float32) -> torch.Tensor:
    """Host arrays / CPU tensors / other dtypes -> contiguous fp32 tensor on the current HIP device."""
    if not torch.is_tensor(x):
        x = torch.as_tensor(x)
    if not x.is_cuda:
        x = x.to(device())
    if x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError(f"{what}: expected an fp32 tensor on a HIP device, got {t.dtype} on {t.device}")
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
# rays
# ------------------------------------------------------------------------------------------------
def get_rays(H: int, W: int, fx: float, fy: float, cx: float, cy: float, c2w: torch.Tensor):
    c2w = _f32c(to_dev(c2w)[:3, :4], "c2w")
    o = torch.empty((H, W, 3), dtype=torch.float32, device=c2w.device)
    d = torch.empty_like(o)
    with torch.cuda.device(c2w.device), _timed("get_rays", 24 * H * W):
        check(lib.nerf_get_rays(H, W, fx, fy, cx, cy, c2w.data_ptr(), o.data_ptr(), d.data_ptr(), _stream()))
    return o, d


def rays_at(pick: torch.Tensor, H: int, W: int, window, fx: float, fy: float, cx: float, cy: float, c2w: torch.Tensor,
            image: Optional[torch.Tensor] = None):
    """Rays (and target colours) of the pixels pick[k] of a window (row0, col0, rows, cols) of one view."""
    c2w = _f32c(to_dev(c2w)[:3, :4], "c2w")
    if not (pick.is_cuda and pick.dtype == torch.int64):
        raise TypeError(f"pick: expected an int64 tensor on a HIP device, got {pick.dtype} on {pick.device}")
    pick = pick.contiguous()
    n = pick.numel()
    row0, col0, rows, cols = [int(v) for v in window]
    o = torch.empty((n, 3), dtype=torch.float32, device=c2w.device)
    d = torch.empty_like(o)
    target = None
    if image is not None:
        image = _f32c(image, "image")
        assert image.numel() == H * W * 3, (image.shape, H, W)
        target = torch.empty_like(o)
    with torch.cuda.device(c2w.device), _timed("rays_at", n * (8 + 24 + (24 if image is not None else 0))):
        check(lib.nerf_rays_at(n, pick.data_ptr(), H, W, row0, col0, rows, cols, fx, fy, cx, cy, c2w.data_ptr(),
                               _ptr(image), o.data_ptr(), d.data_ptr(), _ptr(target), _stream()))
    return o, d, target


def rays_random(n: int, seed: int, H: int, W: int, window, fx: float, fy: float, cx: float, cy: float, c2w: torch.Tensor,
                image: Optional[torch.Tensor] = None, want_pick: bool = False):
    """Rays (and target colours) of n distinct pseudo-random pixels of a window (row0, col0, rows, cols) of one view,
    drawn inside the kernel from `seed` -> rays [2, n, 3] (origins, directions), target, pick (None unless asked for)."""
    c2w = _f32c(to_dev(c2w)[:3, :4], "c2w")
    n = int(n)
    row0, col0, rows, cols = [int(v) for v in window]
    both = torch.empty((2, n, 3), dtype=torch.float32, device=c2w.device)
    o, d = both[0], both[1]
    target = None
    if image is not None:
        image = _f32c(image, "image")
        assert image.numel() == H * W * 3, (image.shape, H, W)
        target = torch.empty_like(o)
    pick = torch.empty((n,), dtype=torch.int64, device=c2w.device) if want_pick else None
    with torch.cuda.device(c2w.device), _timed("rays_random", n * (24 + (24 if image is not None else 0))):
        check(lib.nerf_rays_random(n, int(seed) & 0xffffffffffffffff, H, W, row0, col0, rows, cols, fx, fy, cx, cy,
                                   c2w.data_ptr(), _ptr(image), o.data_ptr(), d.data_ptr(), _ptr(target), _ptr(pick), _stream()))
    return both, target, pick


def ndc_rays(sx: float, sy: float, near: float, rays_o: torch.Tensor, rays_d: torch.Tensor):
    o, d = _f32c(rays_o, "rays_o"), _f32c(rays_d, "rays_d")
    oo, od = torch.empty_like(o), torch.empty_like(d)
    n = o.numel() // 3
    with torch.cuda.device(o.device):
        check(lib.nerf_ndc_rays(n, sx, sy, near, o.data_ptr(), d.data_ptr(), oo.data_ptr(), od.data_ptr(), _stream()))
    return oo, od


_LINSPACE = {}


def cached_linspace(steps: int, dev) -> torch.Tensor:
    """linspace(0, 1, steps) evaluated by the CPU kernel (the values the reference's CPU run uses) and kept on `dev`."""
    key = (int(steps), str(dev))
    t = _LINSPACE.get(key)
    if t is None:
        t = torch.linspace(0., 1., steps=int(steps), device="cpu").to(dev)
        _LINSPACE[key] = t
    return t


def pack_rays(rays_o, rays_d, near, far, viewdir_src=None, ndc=None):
    """-> [n, 8|11] ray records.  ndc = (sx, sy, near_plane) or None; near/far: floats or per-ray tensors."""
    o = _f32c(rays_o.reshape(-1, 3), "rays_o")
    d = _f32c(rays_d.reshape(-1, 3), "rays_d")
    n = o.shape[0]
    vd = None if viewdir_src is None else _f32c(viewdir_src.reshape(-1, 3), "viewdirs")
    ld = 11 if vd is not None else 8
    out = torch.empty((n, ld), dtype=torch.float32, device=o.device)

    def bound(b):
        if torch.is_tensor(b) and b.numel() > 1:
            return 0.0, _f32c(to_dev(b).reshape(-1), "near/far").expand(n).contiguous()
        return float(b), None
    near_s, near_v = bound(near)
    far_s, far_v = bound(far)
    sx, sy, nn = ndc if ndc is not None else (0.0, 0.0, 0.0)
    with torch.cuda.device(o.device), _timed("pack_rays", n * (24 + 4 * ld + (48 if ndc is not None else 0))):
        check(lib.nerf_pack_rays(n, o.data_ptr(), d.data_ptr(), _ptr(vd), int(ndc is not None), sx, sy, nn, near_s, far_s,
                                 _ptr(near_v), _ptr(far_v), out.data_ptr(), ld, _stream()))
    return out


def _no_input_grad(t, what: str) -> None:
    """The kernels differentiate with respect to the network parameters (and, in compositing, the raw network
    outputs); an input that asks for a gradient would silently get none, so that is an error."""
    if torch.is_tensor(t) and t.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError(f"{what} requires a gradient: the HIP render path has no backward with respect to "
                                  "positions, directions or depths (upstream detaches them too: run_nerf.py:467)")


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, u: torch.Tensor, want_inds: bool = False):
    _no_input_grad(bins, "sample_pdf: bins")
    _no_input_grad(weights, "sample_pdf: weights (detach them, as run_nerf.py:467 does with the result)")
    bins, weights, u = _f32c(bins, "bins"), _f32c(weights, "weights"), _f32c(u, "u")
    n, nb = bins.shape
    assert weights.shape == (n, nb - 1), (bins.shape, weights.shape)
    shared = u.dim() == 1
    Si = u.shape[-1]
    samples = torch.empty((n, Si), dtype=torch.float32, device=bins.device)
    inds = torch.empty((n, Si), dtype=torch.int64, device=bins.device) if want_inds else None
    with torch.cuda.device(bins.device):
        check(lib.nerf_sample_pdf(n, nb, Si, bins.data_ptr(), weights.data_ptr(), u.data_ptr(), int(shared),
                                  samples.data_ptr(), _ptr(inds), _stream()))
    return (samples, inds) if want_inds else samples


def sample_coarse(rays: torch.Tensor, t_vals: torch.Tensor, t_rand: Optional[torch.Tensor], lindisp: bool):
    rays = _f32c(rays, "rays")
    n, ld = rays.shape
    S = t_vals.numel()
    z = torch.empty((n, S), dtype=torch.float32, device=rays.device)
    pts = torch.empty((n, S, 3), dtype=torch.float32, device=rays.device)
    if t_rand is not None:
        t_rand = _f32c(t_rand, "t_rand")
        assert t_rand.shape == (n, S)
    # ray record in; z (4 B) + position (12 B) per sample out (+ 4 B per sample of jitter in)
    with torch.cuda.device(rays.device), _timed("sample_coarse", n * (4 * ld + S * (16 + (4 if t_rand is not None else 0)))):
        check(lib.nerf_sample_coarse(n, S, rays.data_ptr(), ld, _f32c(t_vals, "t_vals").data_ptr(), _ptr(t_rand),
                                     int(bool(lindisp)), z.data_ptr(), pts.data_ptr(), _stream()))
    return z, pts


def encode(x: torch.Tensor, n_freq: int) -> torch.Tensor:
    _no_input_grad(x, "encode: x")
    x = _f32c(x, "x")
    assert x.shape[-1] == 3
    out = torch.empty(x.shape[:-1] + (3 + 6 * n_freq,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.nerf_encode(x.numel() // 3, n_freq, x.data_ptr(), out.data_ptr(), _stream()))
    return out


def encode_general(x: torch.Tensor, include_input: bool, freqs: torch.Tensor, cos_flags: Sequence[bool]) -> torch.Tensor:
    """[x (include_input)] ++ [fn_j(x * freqs[b])]: the reference's Embedder for any band table and any sequence of
    sines / cosines (cos_flags[j] True = cosine); x: [..., d]."""
    _no_input_grad(x, "encode: x")
    x = _f32c(x, "x")
    d = x.shape[-1]
    freqs = _f32c(to_dev(freqs).reshape(-1), "freqs")
    n_fn = len(cos_flags)
    mask = sum(1 << j for j, c in enumerate(cos_flags) if c)
    width = (d if include_input else 0) + freqs.numel() * n_fn * d
    out = torch.empty(x.shape[:-1] + (width,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.nerf_encode_general(x.numel() // d, d, int(bool(include_input)), freqs.numel(), freqs.data_ptr(), n_fn, mask,
                                      x.data_ptr(), out.data_ptr(), _stream()))
    return out


def sample_fine(z: torch.Tensor, weights: torch.Tensor, u: torch.Tensor, rays: Optional[torch.Tensor],
                want_pts: bool = True, want_samples: bool = False, want_inds: bool = False):
    """-> dict(z, pts, z_std, samples, inds).  u is [n, Si], or [Si] shared by all rays."""
    z, weights, u = _f32c(z, "z"), _f32c(weights, "weights"), _f32c(u, "u")
    n, Sc = z.shape
    shared = u.dim() == 1
    Si = u.shape[-1]
    dev = z.device
    out = {"z": torch.empty((n, Sc + Si), dtype=torch.float32, device=dev),
           "z_std": torch.empty((n,), dtype=torch.float32, device=dev),
           "pts": torch.empty((n, Sc + Si, 3), dtype=torch.float32, device=dev) if want_pts else None,
           "samples": torch.empty((n, Si), dtype=torch.float32, device=dev) if want_samples else None,
           "inds": torch.empty((n, Si), dtype=torch.int64, device=dev) if want_inds else None}
    ld = 0
    if want_pts:
        rays = _f32c(rays, "rays")
        ld = rays.shape[1]
    # coarse z + weights in (8 B per coarse sample), own uniforms in, merged z out (+ 12 B positions per merged sample)
    nbytes = n * (8 * Sc + (0 if shared else 4 * Si) + 4 * (Sc + Si) + 4 + ((4 * ld + 12 * (Sc + Si)) if want_pts else 0))
    with torch.cuda.device(dev), _timed("sample_fine", nbytes):
        check(lib.nerf_sample_fine(n, Sc, Si, z.data_ptr(), weights.data_ptr(), u.data_ptr(), int(shared),
                                   _ptr(rays) if want_pts else None, ld, out["z"].data_ptr(), _ptr(out["pts"]),
                                   out["z_std"].data_ptr(), _ptr(out["samples"]), _ptr(out["inds"]), _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# compositing
# ------------------------------------------------------------------------------------------------
class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z, rays_d, noise, white_bkgd):
        raw, z = _f32c(raw, "raw"), _f32c(z, "z")
        n, S = z.shape
        rays = rays_d
        if not (rays.is_cuda and rays.dtype == torch.float32 and rays.dim() == 2 and rays.shape == (n, 3)
                and rays.stride(1) == 1):
            rays = _f32c(rays.reshape(n, 3), "rays_d")
        assert raw.shape == (n, S, 4)
        if noise is not None:
            noise = _f32c(noise, "noise")
            assert noise.shape == (n, S)
        dev = raw.device
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
        disp = torch.empty((n,), dtype=torch.float32, device=dev)
        acc = torch.empty_like(disp)
        depth = torch.empty_like(disp)
        weights = torch.empty((n, S), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev), _timed("composite_fwd", n * (S * (24 + (4 if noise is not None else 0)) + 36)):
            check(lib.nerf_composite_fwd(n, S, raw.data_ptr(), z.data_ptr(), rays.data_ptr(), rays.stride(0), _ptr(noise),
                                         int(bool(white_bkgd)), rgb.data_ptr(), disp.data_ptr(), acc.data_ptr(),
                                         weights.data_ptr(), depth.data_ptr(), _stream()))
        ctx.save_for_backward(raw, z, rays, noise, weights)
        ctx.white = bool(white_bkgd)
        # outputs the loss does not touch arrive as None in backward (not as zero tensors the kernel would have to read)
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_weights, g_depth):
        raw, z, rays, noise, weights = ctx.saved_tensors
        n, S = z.shape
        if all(t is None for t in (g_rgb, g_disp, g_acc, g_weights, g_depth)):
            return None, None, None, None, None
        d_raw = torch.empty_like(raw)

        def g(t):
            return None if t is None else _f32c(t, "grad")
        g_rgb, g_disp, g_acc, g_weights, g_depth = g(g_rgb), g(g_disp), g(g_acc), g(g_weights), g(g_depth)
        with torch.cuda.device(raw.device), _timed("composite_bwd", n * (S * (36 + (4 if noise is not None else 0)) + 24)):
            check(lib.nerf_composite_bwd(n, S, raw.data_ptr(), z.data_ptr(), rays.data_ptr(), rays.stride(0), _ptr(noise),
                                         int(ctx.white), weights.data_ptr(), _ptr(g_rgb), _ptr(g_disp), _ptr(g_acc),
                                         _ptr(g_weights), _ptr(g_depth), d_raw.data_ptr(), _stream()))
        return d_raw, None, None, None, None


def composite(raw, z, rays_d, noise=None, white_bkgd=False):
    """-> rgb_map, disp_map, acc_map, weights, depth_map; differentiable w.r.t. raw only.
    rays_d: [n,3], any row stride (e.g. a column slice of the packed ray records).
    raw may carry more than 4 channels (output_ch = 5 without view directions): like run_nerf.py:323,333 only
    channels 0..3 are read, and the others receive a zero gradient."""
    _no_input_grad(z, "composite: z")
    _no_input_grad(rays_d, "composite: rays_d")
    if z.shape[-1] == 1:
        # One sample per ray: upstream's appended 1e10 gap takes the (empty) shape of the other gaps
        # (run_nerf.py:309-311), every per-sample product broadcasts to zero samples, and the sums run over nothing:
        # no weights, acc = 0, rgb = background, disp = 1 / max(1e-10, 0 / 0) = NaN.  No kernel has anything to do.
        n, dev = z.shape[0], z.device
        acc = torch.zeros(n, dtype=torch.float32, device=dev)
        rgb = torch.full((n, 3), 1.0 if white_bkgd else 0.0, dtype=torch.float32, device=dev)
        return (rgb, torch.full_like(acc, float("nan")), acc, torch.zeros((n, 0), dtype=torch.float32, device=dev),
                torch.zeros_like(acc))
    if raw.shape[-1] > 4:
        raw = raw[..., :4]
    return _Composite.apply(raw, z, rays_d, noise, white_bkgd)


# ------------------------------------------------------------------------------------------------
# MLP
# ------------------------------------------------------------------------------------------------
def mlp_n_params(use_viewdirs: bool, output_ch: int) -> int:
    return lib.nerf_mlp_n_params(int(use_viewdirs), output_ch)


def pack_weights(flat: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    """flat parameters -> the weight stream of the forward on pre-encoded inputs."""
    flat = _f32c(flat, "params")
    assert flat.numel() == mlp_n_params(use_viewdirs, output_ch)
    fwd = torch.empty(lib.nerf_mlp_packed_floats(int(use_viewdirs)), dtype=torch.float32, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_pack_weights(int(use_viewdirs), output_ch, flat.data_ptr(), fwd.data_ptr(), _stream()))
    return fwd


def mlp_forward_encoded(x: torch.Tensor, packed_fwd: torch.Tensor, use_viewdirs: bool, output_ch: int,
                        act: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NeRF.forward on pre-encoded rows [P, 63 (+27)], as upstream calls it."""
    x = _f32c(x, "x")
    width = 63 + (27 if use_viewdirs else 0)
    assert x.shape[-1] == width, (x.shape, width)
    n_pts = x.numel() // width
    out = torch.empty((n_pts, 4 if use_viewdirs else output_ch), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.nerf_mlp_fwd_encoded(n_pts, x.data_ptr(), int(use_viewdirs), output_ch, packed_fwd.data_ptr(),
                                       out.data_ptr(), _ptr(act), _stream()))
    return out


def pack_weights_x4(flat: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    flat = _f32c(flat, "params")
    assert flat.numel() == mlp_n_params(use_viewdirs, output_ch)
    out = torch.empty(lib.nerf_mlp_packed_x4_floats(int(use_viewdirs)), dtype=torch.float32, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_pack_weights_x4(int(use_viewdirs), output_ch, flat.data_ptr(), out.data_ptr(), _stream()))
    return out


def mlp_forward_x4(pts: torch.Tensor, viewdirs: Optional[torch.Tensor], packed_x4: torch.Tensor, use_viewdirs: bool,
                   output_ch: int, pts_per_ray: int, act: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 forward on raw positions [P,3] (+ per-ray viewdirs): the 16-point-tile kernel; `act` as in mlp_forward."""
    pts = _f32c(pts, "pts")
    n_pts = pts.numel() // 3
    out = torch.empty((n_pts, 4 if use_viewdirs else output_ch), dtype=torch.float32, device=pts.device)
    if use_viewdirs:
        viewdirs = _f32c(viewdirs, "viewdirs")
        assert viewdirs.shape[0] * pts_per_ray == n_pts, (viewdirs.shape, pts_per_ray, n_pts)
    with torch.cuda.device(pts.device), _timed("mlp_fwd" if act is None else "mlp_fwd_save_f32", n_pts):
        check(lib.nerf_mlp_fwd_x4(n_pts, pts_per_ray, pts.data_ptr(), _ptr(viewdirs) if use_viewdirs else None,
                                  int(use_viewdirs), output_ch, packed_x4.data_ptr(), out.data_ptr(), _ptr(act), _stream()))
    return out


def pack_weights_bf16(flat: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    flat = _f32c(flat, "params")
    assert flat.numel() == mlp_n_params(use_viewdirs, output_ch)
    out = torch.empty(lib.nerf_mlp_packed_bf16_bytes(int(use_viewdirs)), dtype=torch.uint8, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_pack_weights_bf16(int(use_viewdirs), output_ch, flat.data_ptr(), out.data_ptr(), _stream()))
    return out


def mlp_forward_bf16(pts: torch.Tensor, viewdirs: Optional[torch.Tensor], packed_bf16: torch.Tensor, use_viewdirs: bool,
                     output_ch: int, pts_per_ray: int) -> torch.Tensor:
    """bf16-operand / fp32-accumulate inference variant (raw outputs within ~1e-2 of their scale of the fp32 path)."""
    pts = _f32c(pts, "pts")
    n_pts = pts.numel() // 3
    out = torch.empty((n_pts, 4 if use_viewdirs else output_ch), dtype=torch.float32, device=pts.device)
    if use_viewdirs:
        viewdirs = _f32c(viewdirs, "viewdirs")
        assert viewdirs.shape[0] * pts_per_ray == n_pts, (viewdirs.shape, pts_per_ray, n_pts)
    with torch.cuda.device(pts.device), _timed("mlp_fwd_bf16", n_pts):
        check(lib.nerf_mlp_fwd_bf16(n_pts, pts_per_ray, pts.data_ptr(), _ptr(viewdirs) if use_viewdirs else None,
                                    int(use_viewdirs), output_ch, packed_bf16.data_ptr(), out.data_ptr(), _stream()))
    return out


def pack_weights_bf16x3(flat: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    flat = _f32c(flat, "params")
    assert flat.numel() == mlp_n_params(use_viewdirs, output_ch)
    out = torch.empty(lib.nerf_mlp_packed_bf16x3_bytes(int(use_viewdirs)), dtype=torch.uint8, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_pack_weights_bf16x3(int(use_viewdirs), output_ch, flat.data_ptr(), out.data_ptr(), _stream()))
    return out


def mlp_forward_bf16x3(pts: torch.Tensor, viewdirs: Optional[torch.Tensor], packed: torch.Tensor, use_viewdirs: bool,
                       output_ch: int, pts_per_ray: int, act: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32-accurate inference on the bf16 matrix instruction: three-way exact bf16 splits, six products, fp32 sums."""
    pts = _f32c(pts, "pts")
    n_pts = pts.numel() // 3
    out = torch.empty((n_pts, 4 if use_viewdirs else output_ch), dtype=torch.float32, device=pts.device)
    if use_viewdirs:
        viewdirs = _f32c(viewdirs, "viewdirs")
        assert viewdirs.shape[0] * pts_per_ray == n_pts, (viewdirs.shape, pts_per_ray, n_pts)
    with torch.cuda.device(pts.device), _timed("mlp_fwd_bf16x3" if act is None else "mlp_fwd_save", n_pts):
        check(lib.nerf_mlp_fwd_bf16x3(n_pts, pts_per_ray, pts.data_ptr(), _ptr(viewdirs) if use_viewdirs else None,
                                      int(use_viewdirs), output_ch, packed.data_ptr(), out.data_ptr(), _ptr(act), _stream()))
    return out


def pack_weights_f16x2(flat: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    flat = _f32c(flat, "params")
    assert flat.numel() == mlp_n_params(use_viewdirs, output_ch)
    out = torch.empty(lib.nerf_mlp_packed_f16x2_bytes(int(use_viewdirs)), dtype=torch.uint8, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_pack_weights_f16x2(int(use_viewdirs), output_ch, flat.data_ptr(), out.data_ptr(), _stream()))
    return out


def pack_weights_bwd_f16x2(flat: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    flat = _f32c(flat, "params")
    assert flat.numel() == mlp_n_params(use_viewdirs, output_ch)
    out = torch.empty(lib.nerf_mlp_packed_bwd_f16x2_bytes(int(use_viewdirs)), dtype=torch.uint8, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_pack_weights_bwd_f16x2(int(use_viewdirs), output_ch, flat.data_ptr(), out.data_ptr(), _stream()))
    return out


def x2_pack_table(use_viewdirs: bool, output_ch: int, dev):
    """(int32 source map of the fp16-pair stream, its fp32 tables and the transposed fp16-pair stream; pair blocks of the
    first; fp32 table entries; pair blocks of the second): the packing kernels run once on the vector 1/64, 2/64, 3/64,
    ... -- a network has under 2^20 parameters, so every value has at most 20 significant bits and stays below fp16's
    largest: its two pieces add up to it exactly."""
    n = mlp_n_params(use_viewdirs, output_ch)
    assert n < (1 << 20)
    ramp = torch.arange(1, n + 1, dtype=torch.float32, device=dev) / 64.0
    raw = pack_weights_f16x2(ramp, use_viewdirs, output_ch)
    raw_b = pack_weights_bwd_f16x2(ramp, use_viewdirs, output_ch)
    tile = 2048
    total = lib.nerf_mlp_packed_f16x2_bytes(int(use_viewdirs))
    pairs = lib.nerf_mlp_bf16x3_triples(int(use_viewdirs))   # one pair block per weight block, as many as triples
    n_aux = (total - 64 - pairs * tile) // 4                 # behind the pair blocks: the fp32 tables and 16 range words
    assert pairs * tile + n_aux * 4 + 64 == total and n_aux > 0 and n_aux % 4 == 0 and raw_b.numel() % tile == 0
    pairs1 = raw_b.numel() // tile

    def sources(blocks, count):
        halves = blocks.view(torch.float16).reshape(count, 2, 512).double()
        return ((halves[:, 0] + halves[:, 1] / 2048.0) * 64.0).reshape(-1)
    src = torch.cat([sources(raw[:pairs * tile], pairs),
                     raw[pairs * tile: pairs * tile + n_aux * 4].view(torch.float32).double() * 64.0,
                     sources(raw_b, pairs1)])
    return (src.round().to(torch.int32) - 1).contiguous(), pairs, n_aux, pairs1


def pack_weights_f16x2_gather(flat: torch.Tensor, table: torch.Tensor, pairs: int, n_aux: int, pairs1: int = 0):
    """-> (the fp16-pair stream with its fp32 tables and range words, the transposed fp16-pair stream or None when
    pairs1 = 0) in one gather-and-split launch."""
    flat = _f32c(flat, "params")
    n0 = pairs * 2048 + n_aux * 4 + 64
    out = torch.empty(n0 + pairs1 * 2048, dtype=torch.uint8, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_gather_split2(table.data_ptr(), flat.data_ptr(), out.data_ptr(), pairs, n_aux, pairs1, _stream()))
    return out[:n0], (out[n0:] if pairs1 else None)


def mlp_forward_f16x2(pts: torch.Tensor, viewdirs: Optional[torch.Tensor], packed: torch.Tensor, packed_x3: Optional[torch.Tensor],
                      use_viewdirs: bool, output_ch: int, pts_per_ray: int, act: Optional[torch.Tensor] = None,
                      want_flags: bool = False):
    """fp32-accurate inference on the fp16 matrix instruction: two fp16 pieces per operand, three products, fp32 sums;
    tiles whose operands leave fp16's normal range are redone by the three-piece bf16 kernel (packed_x3; None = leave
    them, for tests).  want_flags: also return the per-tile flag words."""
    pts = _f32c(pts, "pts")
    n_pts = pts.numel() // 3
    out = torch.empty((n_pts, 4 if use_viewdirs else output_ch), dtype=torch.float32, device=pts.device)
    if use_viewdirs:
        viewdirs = _f32c(viewdirs, "viewdirs")
        assert viewdirs.shape[0] * pts_per_ray == n_pts, (viewdirs.shape, pts_per_ray, n_pts)
    tile = lib.nerf_mlp_f16x2_tile_points()
    flags = torch.empty(((n_pts + tile - 1) // tile,), dtype=torch.int32, device=pts.device)
    with torch.cuda.device(pts.device), _timed("mlp_fwd_f16x2" if act is None else "mlp_fwd_save", n_pts):
        check(lib.nerf_mlp_fwd_f16x2(n_pts, pts_per_ray, pts.data_ptr(), _ptr(viewdirs) if use_viewdirs else None,
                                     int(use_viewdirs), output_ch, packed.data_ptr(), _ptr(packed_x3), out.data_ptr(), _ptr(act),
                                     flags.data_ptr(), _stream()))
    if MARKED is not None:
        MARKED.append(("forward", flags))
    return (out, flags) if want_flags else out


def mlp_act_buffer(n_pts: int, use_viewdirs: bool, dev) -> torch.Tensor:
    return torch.empty(lib.nerf_mlp_act_floats(int(use_viewdirs), n_pts), dtype=torch.float32, device=dev)


def pack_weights_bwd_x4(flat: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    flat = _f32c(flat, "params")
    assert flat.numel() == mlp_n_params(use_viewdirs, output_ch)
    out = torch.empty(lib.nerf_mlp_packed_bwd_x4_floats(int(use_viewdirs)), dtype=torch.float32, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_pack_weights_bwd_x4(int(use_viewdirs), output_ch, flat.data_ptr(), out.data_ptr(), _stream()))
    return out


def x4_pack_table(use_viewdirs: bool, output_ch: int, dev):
    """(int32 source map of the forward 16-point stream followed by the transposed one, length of the first): the
    packing kernels run once on the vector 1, 2, 3, ... (exact in fp32: a network has under 2^24 parameters)."""
    n = mlp_n_params(use_viewdirs, output_ch)
    assert n < (1 << 24)
    ramp = torch.arange(1, n + 1, dtype=torch.float32, device=dev)
    a = pack_weights_x4(ramp, use_viewdirs, output_ch)
    b = pack_weights_bwd_x4(ramp, use_viewdirs, output_ch)
    assert a.numel() % 4 == 0   # the second stream starts 16-byte aligned
    return (torch.cat([a, b]).to(torch.int32) - 1).contiguous(), a.numel()


def pack_weights_x4_pair(flat: torch.Tensor, table: torch.Tensor, n_fwd: int):
    flat = _f32c(flat, "params")
    out = torch.empty(table.numel(), dtype=torch.float32, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_gather(table.data_ptr(), flat.data_ptr(), out.data_ptr(), table.numel(), _stream()))
    return out[:n_fwd], out[n_fwd:]


def mlp_backward_x4(d_out: torch.Tensor, act: torch.Tensor, flat: torch.Tensor, packed_x4: torch.Tensor,
                    packed_bwd_x4: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    """Parameter gradients on the fp32 matrix instruction (backward-data on 16-point tiles, fp32 weight-gradient GEMM)."""
    d_out = _f32c(d_out, "d_out")
    n_pts = d_out.shape[0]
    scratch = torch.empty(lib.nerf_mlp_scratch_floats(int(use_viewdirs), n_pts), dtype=torch.float32, device=d_out.device)
    d_flat = torch.empty_like(flat)
    with torch.cuda.device(d_out.device), _timed("mlp_bwd_f32", n_pts):
        check(lib.nerf_mlp_bwd_x4(n_pts, d_out.data_ptr(), act.data_ptr(), int(use_viewdirs), output_ch,
                                  packed_x4.data_ptr(), packed_bwd_x4.data_ptr(), scratch.data_ptr(), d_flat.data_ptr(),
                                  _stream()))
    return d_flat


def pack_weights_bwd_bf16x3(flat: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    flat = _f32c(flat, "params")
    assert flat.numel() == mlp_n_params(use_viewdirs, output_ch)
    out = torch.empty(lib.nerf_mlp_packed_bwd_bf16x3_bytes(int(use_viewdirs)), dtype=torch.uint8, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_pack_weights_bwd_bf16x3(int(use_viewdirs), output_ch, flat.data_ptr(), out.data_ptr(), _stream()))
    return out


def x3_pack_table(use_viewdirs: bool, output_ch: int, dev):
    """(int32 source map of the forward split stream, its fp32 tables and the transposed split stream; triples in the
    first; fp32 table entries; triples in the second): the packing kernels run once on the vector 1, 2, 3, ... -- the
    three pieces of a split add up to the value exactly, and a network has under 2^24 parameters."""
    n = mlp_n_params(use_viewdirs, output_ch)
    assert n < (1 << 24)
    ramp = torch.arange(1, n + 1, dtype=torch.float32, device=dev)
    a = pack_weights_bf16x3(ramp, use_viewdirs, output_ch)
    b = pack_weights_bwd_bf16x3(ramp, use_viewdirs, output_ch)
    t0 = lib.nerf_mlp_bf16x3_triples(int(use_viewdirs))
    assert (a.numel() - t0 * 3072) % 16 == 0 and b.numel() % 3072 == 0
    n_aux, t1 = (a.numel() - t0 * 3072) // 4, b.numel() // 3072

    def triples(raw, t):
        return raw[:t * 3072].view(torch.bfloat16).reshape(t, 3, 512).float().sum(1).reshape(-1)
    src = torch.cat([triples(a, t0), a[t0 * 3072:].view(torch.float32), triples(b, t1)])
    return (src.to(torch.int32) - 1).contiguous(), t0, n_aux, t1


def pack_weights_train_streams(flat: torch.Tensor, x2_table, x3_table):
    """(fp16-pair stream, its transposed stream, bf16-triple stream, its transposed stream) of one network from ONE
    gather-and-split launch; x2_table / x3_table as x2_pack_table / x3_pack_table return them."""
    flat = _f32c(flat, "params")
    table2, pairs, n_aux2, pairs1 = x2_table
    table3, t0, n_aux3, t1 = x3_table
    n2 = pairs * 2048 + n_aux2 * 4 + 64
    n3 = t0 * 3072 + n_aux3 * 4
    out2 = torch.empty(n2 + pairs1 * 2048, dtype=torch.uint8, device=flat.device)
    out3 = torch.empty(n3 + t1 * 3072, dtype=torch.uint8, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_gather_split23(table2.data_ptr(), table3.data_ptr(), flat.data_ptr(), out2.data_ptr(), out3.data_ptr(),
                                      pairs, n_aux2, pairs1, t0, n_aux3, t1, _stream()))
    return out2[:n2], out2[n2:], out3[:n3], out3[n3:]


def pack_weights_x3_pair(flat: torch.Tensor, table: torch.Tensor, t0: int, n_aux: int, t1: int):
    """-> (forward split stream with its fp32 tables, transposed split stream): one gather-and-split launch."""
    flat = _f32c(flat, "params")
    n0 = t0 * 3072 + n_aux * 4
    out = torch.empty(n0 + t1 * 3072, dtype=torch.uint8, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_gather_split3(table.data_ptr(), flat.data_ptr(), out.data_ptr(), t0, n_aux, t1, _stream()))
    return out[:n0], out[n0:]


def mlp_backward_bf16x3(d_out: torch.Tensor, act: torch.Tensor, flat: torch.Tensor, packed_x3: torch.Tensor,
                        packed_bwd_x3: torch.Tensor, use_viewdirs: bool, output_ch: int) -> torch.Tensor:
    """Parameter gradients on the split-operand kernels (backward-data, then the weight-gradient stage)."""
    d_out = _f32c(d_out, "d_out")
    n_pts = d_out.shape[0]
    scratch = torch.empty(lib.nerf_mlp_scratch_floats(int(use_viewdirs), n_pts), dtype=torch.float32, device=d_out.device)
    d_flat = torch.empty_like(flat)
    with torch.cuda.device(d_out.device):
        with _timed("mlp_bwd_data", n_pts):
            check(lib.nerf_mlp_bwd_data_bf16x3(n_pts, d_out.data_ptr(), act.data_ptr(), int(use_viewdirs), output_ch,
                                               packed_x3.data_ptr(), packed_bwd_x3.data_ptr(), scratch.data_ptr(), _stream()))
        with _timed("mlp_bwd_dw", n_pts):
            check(lib.nerf_mlp_bwd_weights_bf16x3(n_pts, act.data_ptr(), int(use_viewdirs), output_ch, scratch.data_ptr(),
                                                  d_flat.data_ptr(), _stream()))
    return d_flat


DY_STAGES, DW_CTL_WORDS = 18, 4096     # NERF_DY_STAGES, NERF_DW_CTL_WORDS of include/nerf_hip.h


def mlp_backward_f16x2(d_out: torch.Tensor, act: torch.Tensor, flat: torch.Tensor, packed_x2: torch.Tensor,
                       packed_bwd_x2: torch.Tensor, packed_x3: Optional[torch.Tensor], packed_bwd_x3: Optional[torch.Tensor],
                       use_viewdirs: bool, output_ch: int, want_flags: bool = False, fwd_flags: Optional[torch.Tensor] = None,
                       dw: str = "f16x2"):
    """Parameter gradients with backward-data on the two-piece fp16 kernel (tiles that leave its range are redone by the
    three-piece kernel; both three-piece streams None = leave them, for tests), then the weight-gradient stage: on two
    pieces as well (dw='f16x2'; fwd_flags: the tile flags of the forward that saved `act` -- any marked tile, forward
    or backward, and the three-piece weight-gradient kernel takes the batch), or on three (dw='bf16x3')."""
    d_out = _f32c(d_out, "d_out")
    n_pts = d_out.shape[0]
    dev = d_out.device
    scratch = torch.empty(lib.nerf_mlp_scratch_floats(int(use_viewdirs), n_pts), dtype=torch.float32, device=dev)
    tile = lib.nerf_mlp_f16x2_tile_points()
    n_tiles = (n_pts + tile - 1) // tile
    two = dw == "f16x2"
    # one int32 buffer: tile flags | per-tile maxima | control words
    words = torch.empty(n_tiles * (1 + (DY_STAGES if two else 0)) + (DW_CTL_WORDS if two else 0), dtype=torch.int32, device=dev)
    flags = words[:n_tiles]
    tile_max = words[n_tiles:n_tiles * (1 + DY_STAGES)] if two else None
    ctl = words[n_tiles * (1 + DY_STAGES):] if two else None
    d_flat = torch.empty_like(flat)
    with torch.cuda.device(dev):
        with _timed("mlp_bwd_data", n_pts):
            check(lib.nerf_mlp_bwd_data_f16x2(n_pts, d_out.data_ptr(), act.data_ptr(), int(use_viewdirs), output_ch,
                                              packed_x2.data_ptr(), packed_bwd_x2.data_ptr(), _ptr(packed_x3), _ptr(packed_bwd_x3),
                                              scratch.data_ptr(), flags.data_ptr(), _ptr(tile_max), _stream()))
        with _timed("mlp_bwd_dw", n_pts):
            if two:
                check(lib.nerf_mlp_bwd_weights_f16x2(n_pts, act.data_ptr(), int(use_viewdirs), output_ch, scratch.data_ptr(),
                                                     d_flat.data_ptr(), _ptr(fwd_flags), flags.data_ptr(), tile_max.data_ptr(),
                                                     ctl.data_ptr(), _stream()))
            else:
                check(lib.nerf_mlp_bwd_weights_bf16x3(n_pts, act.data_ptr(), int(use_viewdirs), output_ch, scratch.data_ptr(),
                                                      d_flat.data_ptr(), _stream()))
    if MARKED is not None:
        MARKED.append(("backward_data", flags))
    return (d_flat, flags) if want_flags else d_flat


def train_kernels(model) -> str:
    """Which kernels a training step of `model` runs: 'mfma32' (forward, backward-data and weight gradients on the fp32
    matrix instruction) for mlp_dtype 'f32_mfma', else 'bf16x3' (all three on split operands: forward and backward-data
    on two fp16 pieces with the three-piece fall-back, unless mlp_dtype is 'bf16x3'; weight gradients as model.dw_dtype says)."""
    return "mfma32" if getattr(model, "mlp_dtype", "f32") == "f32_mfma" else "bf16x3"


class _FusedMLP(torch.autograd.Function):
    """raw = NeRF(encode(pts), encode(viewdirs)); gradients flow to the parameters only."""

    @staticmethod
    def forward(ctx, pts, viewdirs, model, pts_per_ray, need_grad, *params):
        kind = train_kernels(model)
        two_piece = kind == "bf16x3" and pts_per_ray > 0 and getattr(model, "mlp_dtype", "f32") != "bf16x3"
        streams = model._train_streams() if (two_piece and need_grad) else None
        flat = streams[0] if streams is not None else model._flat()
        n_pts = pts.numel() // (3 if pts_per_ray > 0 else pts.shape[-1])
        act = mlp_act_buffer(n_pts, model.use_viewdirs, pts.device) if need_grad else None
        if pts_per_ray == 0:      # pre-encoded rows (NeRF.forward(x)); every backward reads what this forward saves
            out = mlp_forward_encoded(pts, model._packed(), model.use_viewdirs, model.output_ch, act)
        elif two_piece:
            # the two-piece fp16 kernel, saving activations; marked tiles are redone (and re-saved) by the three-piece kernel
            x2f, x3f = (streams[1], streams[3]) if streams is not None else (model._packed_f16x2(), model._packed_bf16x3())
            out, fwd_flags = mlp_forward_f16x2(pts, viewdirs, x2f, x3f, model.use_viewdirs, model.output_ch, pts_per_ray, act,
                                               want_flags=True)
        elif kind == "bf16x3":
            out = mlp_forward_bf16x3(pts, viewdirs, model._packed_bf16x3(with_bwd=need_grad), model.use_viewdirs, model.output_ch,
                                     pts_per_ray, act)
        else:
            out = mlp_forward_x4(pts, viewdirs, model._packed_x4(), model.use_viewdirs, model.output_ch, pts_per_ray, act)
        if need_grad:
            if two_piece:
                ctx.save_for_backward(act, streams[3], streams[4], flat, streams[1], streams[2], fwd_flags)
            else:
                if kind == "bf16x3":
                    fwd, bwd = model._packed_bf16x3(), model._packed_bwd_bf16x3()
                else:
                    fwd, bwd = model._packed_x4(), model._packed_bwd_x4()
                ctx.save_for_backward(act, fwd, bwd, flat)
            ctx.kind = "f16x2" if two_piece else kind
            ctx.model = model
        return out

    @staticmethod
    def backward(ctx, d_out):
        act, fwd, bwd, flat = ctx.saved_tensors[:4]
        model = ctx.model
        if ctx.kind == "f16x2":
            x2f, x2b, fwd_flags = ctx.saved_tensors[4:]
            d_flat = mlp_backward_f16x2(d_out, act, flat, x2f, x2b, fwd, bwd, model.use_viewdirs, model.output_ch,
                                        fwd_flags=fwd_flags, dw=getattr(model, "dw_dtype", "f16x2"))
        elif ctx.kind == "bf16x3":
            d_flat = mlp_backward_bf16x3(d_out, act, flat, fwd, bwd, model.use_viewdirs, model.output_ch)
        else:
            d_flat = mlp_backward_x4(d_out, act, flat, fwd, bwd, model.use_viewdirs, model.output_ch)
        grads = model._split_flat(d_flat)
        return (None, None, None, None, None) + tuple(grads)


def fused_mlp(pts: torch.Tensor, viewdirs: Optional[torch.Tensor], model, pts_per_ray: int) -> torch.Tensor:
    params = model._params() if hasattr(model, "_params") else list(model.parameters())
    _no_input_grad(pts, "fused MLP: pts")
    _no_input_grad(viewdirs, "fused MLP: viewdirs")
    # decided here, where the caller's grad mode is visible: under no_grad (rendering) nothing is saved
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if not need_grad and pts_per_ray > 0:   # no-grad queries on raw positions: the inference kernels
        dtype = getattr(model, "mlp_dtype", "f32")
        if dtype == "bf16":
            return mlp_forward_bf16(pts, viewdirs, model._packed_bf16(), model.use_viewdirs, model.output_ch, pts_per_ray)
        if dtype == "f32_mfma":
            return mlp_forward_x4(pts, viewdirs, model._packed_x4(), model.use_viewdirs, model.output_ch, pts_per_ray)
        if dtype == "bf16x3":
            return mlp_forward_bf16x3(pts, viewdirs, model._packed_bf16x3(), model.use_viewdirs, model.output_ch, pts_per_ray)
        return mlp_forward_f16x2(pts, viewdirs, model._packed_f16x2(), model._packed_bf16x3(), model.use_viewdirs, model.output_ch,
                                 pts_per_ray)
    return _FusedMLP.apply(pts, viewdirs, model, pts_per_ray, need_grad, *params)


# ------------------------------------------------------------------------------------------------
# the photometric loss
# ------------------------------------------------------------------------------------------------
class _MSE(torch.autograd.Function):
    """mean((x - y) ** 2) of two fp32 device tensors of one shape; the gradient flows to x only."""

    @staticmethod
    def forward(ctx, x, y):
        out = torch.empty((), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(lib.nerf_mse_fwd(x.numel(), x.data_ptr(), y.data_ptr(), out.data_ptr(), _stream()))
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib.nerf_mse_bwd(x.numel(), x.data_ptr(), y.data_ptr(), g.data_ptr(), dx.data_ptr(), _stream()))
        return dx, None


class _MSEPair(torch.autograd.Function):
    """(mean((x0 - y) ** 2), mean((x1 - y) ** 2), their sum) in one launch; the gradients of both inputs in one more."""

    @staticmethod
    def forward(ctx, x0, x1, y):
        out = [torch.empty((), dtype=torch.float32, device=x0.device) for _ in range(3)]
        with torch.cuda.device(x0.device):
            check(lib.nerf_mse_pair_fwd(x0.numel(), x0.data_ptr(), x1.data_ptr(), y.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                        out[2].data_ptr(), _stream()))
        ctx.save_for_backward(x0, x1, y)
        ctx.set_materialize_grads(False)
        return tuple(out)

    @staticmethod
    def backward(ctx, g0, g1, g_sum):
        x0, x1, y = ctx.saved_tensors
        if g0 is None and g1 is None and g_sum is None:
            return None, None, None
        gs = [None if g is None else g.to(torch.float32).contiguous() for g in (g0, g1, g_sum)]
        d0, d1 = torch.empty_like(x0), torch.empty_like(x1)
        with torch.cuda.device(x0.device):
            check(lib.nerf_mse_pair_bwd(x0.numel(), x0.data_ptr(), x1.data_ptr(), y.data_ptr(), _ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]),
                                        d0.data_ptr(), d1.data_ptr(), _stream()))
        return d0, d1, None


_MSE_MAX = None


def mse(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """img2mse (run_nerf_helpers.py:8): mean((x - y) ** 2).  A training batch on the device (fp32, one shape, contiguous,
    no gradient wanted for y, at most nerf_mse_max_elems() elements) takes one launch forward and one backward; anything
    else the stock fused mse_loss, which computes the same expression."""
    global _MSE_MAX
    if _MSE_MAX is None:
        _MSE_MAX = int(lib.nerf_mse_max_elems())
    if (torch.is_tensor(x) and torch.is_tensor(y) and x.is_cuda and y.is_cuda and x.device == y.device
            and x.dtype == torch.float32 and y.dtype == torch.float32 and x.shape == y.shape and 0 < x.numel() <= _MSE_MAX
            and x.is_contiguous() and y.is_contiguous() and not y.requires_grad):
        return _MSE.apply(x, y)
    return torch.nn.functional.mse_loss(x, y)


def mse_pair(x0: torch.Tensor, x1: torch.Tensor, y: torch.Tensor):
    """Both losses of a training step against one target (run_nerf.py:863-870) -> (mse(x0, y), mse(x1, y), their sum):
    one launch forward and one backward where `mse` would take one each per term, plus the addition."""
    global _MSE_MAX
    if _MSE_MAX is None:
        _MSE_MAX = int(lib.nerf_mse_max_elems())
    ts = (x0, x1, y)
    if (all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts)
            and x0.device == y.device and x1.device == y.device and x0.shape == y.shape and x1.shape == y.shape
            and 0 < y.numel() <= _MSE_MAX and not y.requires_grad):
        return _MSEPair.apply(x0, x1, y)
    a, b = mse(x0, y), mse(x1, y)
    return a, b, a + b


# ------------------------------------------------------------------------------------------------
# nn.Linear for shapes the fused kernels do not cover
# ------------------------------------------------------------------------------------------------
def _rows(t: torch.Tensor, what: str) -> torch.Tensor:
    """[..., c] -> a 2-d fp32 view whose rows are contiguous (row stride >= c): column slices of a wider buffer pass as they are."""
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError(f"{what}: expected an fp32 tensor on a HIP device, got {t.dtype} on {t.device}")
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1 or (t2.shape[0] > 1 and t2.stride(0) < t2.shape[1]):
        t2 = t2.contiguous()
    return t2


def _ld(t2: torch.Tensor) -> int:
    return t2.stride(0) if t2.shape[0] > 1 else t2.shape[1]


class _Linear(torch.autograd.Function):
    """y = act(x w^T + b) with act = ReLU or identity, every product in libnerf_hip.so (linear.hip)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x2, w2 = _rows(x, "x"), _f32c(w, "weight")
        n, n_in = x2.shape
        n_out = w2.shape[0]
        assert w2.shape[1] == n_in, (x.shape, w.shape)
        bias = None if b is None else _f32c(b, "bias")
        y = torch.empty((n, n_out), dtype=torch.float32, device=x2.device)
        with torch.cuda.device(x2.device):
            check(lib.nerf_linear_fwd(n, n_in, n_out, x2.data_ptr(), _ld(x2), w2.data_ptr(), _ptr(bias), int(bool(relu)),
                                      y.data_ptr(), n_out, _stream()))
        ctx.save_for_backward(x2, w2, y if relu else None)
        ctx.has_bias = b is not None
        ctx.lead = tuple(x.shape[:-1])
        return y.reshape(ctx.lead + (n_out,))

    @staticmethod
    def backward(ctx, dy):
        x2, w2, y = ctx.saved_tensors
        n, n_in = x2.shape
        n_out = w2.shape[0]
        dy2 = _rows(dy, "grad")
        dev = x2.device
        with torch.cuda.device(dev):
            if y is not None:
                masked = torch.empty((n, n_out), dtype=torch.float32, device=dev)
                dyc = dy2.contiguous()
                check(lib.nerf_relu_grad(n * n_out, dyc.data_ptr(), y.data_ptr(), masked.data_ptr(), _stream()))
                dy2 = masked
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty((n, n_in), dtype=torch.float32, device=dev)
                check(lib.nerf_linear_bwd_data(n, n_in, n_out, dy2.data_ptr(), _ld(dy2), w2.data_ptr(), dx.data_ptr(), n_in, _stream()))
                dx = dx.reshape(ctx.lead + (n_in,))
            if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(w2)
                db = torch.empty((n_out,), dtype=torch.float32, device=dev) if ctx.has_bias else None
                ws = torch.empty(lib.nerf_linear_workspace_floats(n_in, n_out), dtype=torch.float32, device=dev)
                check(lib.nerf_linear_bwd_weights(n, n_in, n_out, dy2.data_ptr(), _ld(dy2), x2.data_ptr(), _ld(x2), dw.data_ptr(),
                                                  _ptr(db), ws.data_ptr(), _stream()))
        return dx, dw, db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = False) -> torch.Tensor:
    """torch.nn.functional.linear (+ optional ReLU) on the library's own matrix kernel, differentiable in x, weight, bias."""
    return _Linear.apply(x, weight, bias, relu)
