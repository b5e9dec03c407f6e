"""Drop-in for the upstream run_nerf_helpers.py: same names, arguments and results, computed by the
HIP kernels of libnerf_hip.so.  Citations are file:line of the upstream tree."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops

# Misc (run_nerf_helpers.py:8-10) -- the train-step tail stays in PyTorch.
# (img2mse: mean((x - y) ** 2); a training batch on the device takes one launch forward and one backward -- ops.mse --,
# anything else torch's fused mse_loss)
img2mse = lambda x, y: ops.mse(x, y)
# (the two losses of a step, img2mse(rgb, target) and img2mse(rgb0, target), and their sum from one launch:
#  -> (img_loss, img_loss0, loss); upstream computes them at run_nerf.py:863-870)
img2mse_pair = lambda x, x0, y: ops.mse_pair(x, x0, y)
_LOG10 = {}


def _log10(dev):
    """log(10) as upstream forms it, a one-element fp32 tensor, kept per device: building it inside every call is a
    blocking host-to-device copy per optimisation step."""
    t = _LOG10.get(dev)
    if t is None:
        t = _LOG10[dev] = torch.log(torch.tensor([10.])).to(dev)
    return t


mse2psnr = lambda x: -10. * torch.log(x) / _log10(x.device)
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


# ------------------------------------------------------------------------------------------------
# Positional encoding (run_nerf_helpers.py:13-66)
# ------------------------------------------------------------------------------------------------
class Embedder:
    """[x, fn(freq x) ...] with the upstream keyword arguments (run_nerf_helpers.py:13-47).  The default setting
    (include_input, 3 dims, log-spaced powers of two, sin then cos) is the one fused into the MLP kernels; any other
    setting whose periodic_fns are torch.sin / torch.cos runs the general encoding kernel; other callables cannot
    run inside a kernel and raise."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.create_embedding_fn()

    def create_embedding_fn(self):
        kw = self.kwargs
        fns = list(kw.get('periodic_fns', [torch.sin, torch.cos]))
        d = int(kw.get('input_dims', 3))
        n_freq = int(kw['num_freqs'])
        self.standard = bool(kw.get('include_input', True) and d == 3 and kw.get('log_sampling', True)
                             and fns == [torch.sin, torch.cos] and n_freq >= 1 and kw['max_freq_log2'] == n_freq - 1)
        self.n_freq = n_freq if self.standard else None     # run_network fuses the encoding only when this is set
        self.include_input = bool(kw.get('include_input', True))
        self.out_dim = (d if self.include_input else 0) + d * n_freq * len(fns)
        if self.standard:
            return
        bad = [f for f in fns if f not in (torch.sin, torch.cos)]
        if bad:
            raise NotImplementedError(f"Embedder: periodic_fns must be torch.sin / torch.cos for the HIP kernels, got {bad}")
        self.cos_flags = [f is torch.cos for f in fns]
        # the band values exactly as upstream evaluates them (fp32, on the CPU)
        max_freq = kw['max_freq_log2']
        if kw.get('log_sampling', True):
            self.freq_bands = 2. ** torch.linspace(0., max_freq, steps=n_freq)
        else:
            self.freq_bands = torch.linspace(2. ** 0., 2. ** max_freq, steps=n_freq)
        self._bands_dev = None

    def embed(self, inputs):
        x = ops.to_dev(inputs)
        if self.standard:
            return ops.encode(x, self.n_freq)
        if self._bands_dev is None or self._bands_dev.device != x.device:
            self._bands_dev = self.freq_bands.to(x.device)
        return ops.encode_general(x, self.include_input, self._bands_dev, self.cos_flags)


class _EmbedFn:
    """The callable get_embedder returns; carries n_freq so that run_network can fuse it into the MLP kernel."""

    def __init__(self, embedder: Embedder):
        self.embedder = embedder
        self.n_freq = embedder.n_freq

    def __call__(self, x):
        return self.embedder.embed(x)


def get_embedder(multires, i=0):
    if i == -1:
        return nn.Identity(), 3
    embedder_obj = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                            log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return _EmbedFn(embedder_obj), embedder_obj.out_dim


# ------------------------------------------------------------------------------------------------
# Model (run_nerf_helpers.py:73-133)
# ------------------------------------------------------------------------------------------------
# Every optimiser step in the process advances this count; the packed weight streams are keyed on it as well as on the
# parameters' version counters, because fused optimiser kernels (torch.optim.Adam(..., fused=True)) update the
# parameters without touching those counters.
_OPT_STEPS = [0]


def _count_optimizer_step(optimizer, args, kwargs):
    _OPT_STEPS[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_count_optimizer_step)


class NeRF(nn.Module):
    """Same constructor, attributes and state_dict as upstream; forward runs the fused MFMA kernel.

    The kernel exists for the architecture every upstream config uses: D=8, W=256, skips=[4],
    input_ch=63 (L=10), input_ch_views=27 (L=4) or no view branch.  Any other shape (other --netdepth /
    --netwidth / --multires, --i_embed -1) runs layer by layer through the library's generic matrix kernel
    (ops.linear) -- same results, autograd included (inputs too), none of the fusion."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super(NeRF, self).__init__()
        self.D = D
        self.W = W
        self.input_ch = input_ch
        self.input_ch_views = input_ch_views
        self.skips = skips
        self.use_viewdirs = use_viewdirs
        self.output_ch = output_ch

        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W)
                                        for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        self._pack_key = None
        self._pack = None
        self._pack16 = None
        self._packx4 = None
        self._x4_table = None
        self._x3_table = None
        self._x2_table = None
        self._flat_vec = None
        self._flat_addrs = ()
        self._param_list = None
        self._pack16x3 = None
        self._pack16x3b = None
        self._pack16x2 = None
        self._pack16x2b = None
        # Which kernels evaluate the network:
        #   'f32' (default)  fp32 accuracy on the 16-bit matrix instructions.  No-grad queries (rendering): two fp16 pieces
        #                    per operand, three piece products per product, fp32 sums; tiles of 128 points whose operands
        #                    leave fp16's normal range are redone on the three-piece kernel.  Training: the same two-piece
        #                    kernel for the forward and its mirror for backward-data (every point's gradient carries a
        #                    power-of-two scale), weight gradients as dw_dtype below says.  All measure the error of the
        #                    fp32 instruction against fp64
        #   'bf16x3'         the three-piece kernels everywhere, rendering included
        #   'f32_mfma'       the fp32 matrix instruction, rendering and all of training
        #   'bf16'           no-grad queries only, reduced precision: bf16 operands, fp32 sums, raw outputs within ~1e-2
        #                    of their scale (training with this setting runs the 'f32' kernels)
        self.mlp_dtype = 'f32'
        # Weight-gradient stage of the default ('f32') training path: 'f16x2' = two fp16 pieces per operand, three matrix
        # instructions per product into one accumulator, the gradients under one power-of-two scale per layer and batch
        # (the three-piece kernel takes any batch in which the two-piece forward or backward-data marked a tile);
        # 'bf16x3' = exact three-way bf16 splits, six instructions.  The first is exact to an fp32 sum's rounding relative
        # to each weight matrix's largest gradient; the second also relative to every single entry
        self.dw_dtype = 'f16x2'

    # -- kernel plumbing ---------------------------------------------------------------------------
    def fused_arch(self) -> bool:
        """Whether the hand-written kernels cover this module's shape."""
        return (self.D == 8 and self.W == 256 and list(self.skips) == [4] and self.input_ch == 63
                and ((self.use_viewdirs and self.input_ch_views == 27) or
                     (not self.use_viewdirs and self.input_ch_views == 0 and 1 <= self.output_ch <= 8)))

    def _check_arch(self):
        if not self.fused_arch():
            raise NotImplementedError(
                f"NeRF(D={self.D}, W={self.W}, skips={self.skips}, input_ch={self.input_ch}, "
                f"input_ch_views={self.input_ch_views}, use_viewdirs={self.use_viewdirs}): the fused HIP kernel "
                "implements D=8, W=256, skips=[4], input_ch=63 and input_ch_views=27 (or 0 without view directions)")

    def _flat(self) -> torch.Tensor:
        """The parameters as one flat vector (state_dict order).  On first use the module ADOPTS one flat buffer: every
        parameter's `.data` becomes a view into it (same values, shapes, names and state_dict), so that optimiser steps
        and load_state_dict write the vector the kernels' weight streams are packed from and no per-step concatenation
        is needed.  A parameter whose storage was replaced since (`.to()`, `p.data = ...`) is noticed by its address and
        the buffer is rebuilt from the current values.  Every packed stream is dropped when a parameter changed
        (in-place updates bump the tensors' version counters)."""
        self._check_arch()
        params = self._params()
        store = self._flat_vec
        if store is None or any(p.data_ptr() != a for p, a in zip(params, self._flat_addrs)):
            with torch.no_grad():
                store = torch.cat([p.detach().reshape(-1) for p in params])
            if not store.is_cuda:
                raise RuntimeError("NeRF parameters must live on a HIP device (call .to('cuda'))")
            off, addrs = 0, []
            for p in params:
                n = p.numel()
                p.data = store[off:off + n].view(p.shape)
                addrs.append(p.data_ptr())
                off += n
            self._flat_vec, self._flat_addrs, self._pack_key = store, addrs, None
        key = (_OPT_STEPS[0],) + tuple(p._version for p in params)
        if self._pack_key != key:
            self._pack_key = key
            self._pack = self._pack16 = self._packx4 = self._pack16x3 = self._pack16x3b = self._pack16x2 = self._pack16x2b = None
        return store

    def _params(self):
        """list(self.parameters()), kept: walking the module tree costs more than everything else a query does on the
        host (dropped by _apply; a module whose layers are REPLACED afterwards must set _param_list = None itself)."""
        if self._param_list is None:
            self._param_list = list(self.parameters())
        return self._param_list

    def _apply(self, fn, recurse=True):
        """`.to()`, `.cuda()`, `.float()`, ...: the parameters may be replaced, so the cached list of them is dropped."""
        self._param_list = None
        return super()._apply(fn, recurse)

    def adopt_flat(self) -> None:
        """Move the parameters into the module's flat buffer now (see _flat) instead of at the first query: call it after
        `.to(device)` and BEFORE anything keeps a parameter's address or storage -- a HIP graph, an EMA or shadow copy
        made through `p.data`, an external loader holding data_ptr() -- because adoption rebinds every `.data` to a view
        of the new buffer and whoever still points at the old storage is not told.  create_nerf does this for the models
        it builds; a module constructed by hand adopts at its first query."""
        self._flat()

    def invalidate_packs(self) -> None:
        """Forget every packed weight stream; the next query rebuilds them from the parameters.  The cache key is each
        parameter's version counter, which in-place autograd-visible updates bump, and the number of optimiser steps taken
        in the process; writes that bypass both -- `p.data.copy_(...)`, `p.data.mul_(...)`, a HIP-graph replay of an optimiser step, an external loader
        writing through data_ptr() -- must be followed by this call."""
        self._pack_key = None

    def _packed(self):
        """Weight stream of the forward on pre-encoded inputs (NeRF.forward(x))."""
        flat = self._flat()
        if self._pack is None:
            self._pack = ops.pack_weights(flat, self.use_viewdirs, self.output_ch)
        return self._pack

    def _packed_bf16(self):
        flat = self._flat()
        if self._pack16 is None:
            self._pack16 = ops.pack_weights_bf16(flat, self.use_viewdirs, self.output_ch)
        return self._pack16

    def _packed_bf16x3(self, with_bwd: bool = False):
        """Forward split stream.  with_bwd (a training step): it comes together with the transposed one out of one
        gather-and-split launch through a map built once; otherwise the forward stream is packed alone."""
        flat = self._flat()
        if self._pack16x3 is None:
            if with_bwd:
                self._packed_x3_pair()
            else:
                self._pack16x3 = ops.pack_weights_bf16x3(flat, self.use_viewdirs, self.output_ch)
        return self._pack16x3

    def _packed_f16x2(self, with_bwd: bool = False):
        """Stream of fp16 pairs of the default query.  with_bwd (a training step): the transposed stream of the
        backward-data kernel comes out of the same gather-and-split launch."""
        flat = self._flat()
        if self._pack16x2 is None or (with_bwd and self._pack16x2b is None):
            if self._x2_table is None or self._x2_table[0].device != flat.device:
                self._x2_table = ops.x2_pack_table(self.use_viewdirs, self.output_ch, flat.device)
            table, pairs, n_aux, pairs1 = self._x2_table
            self._pack16x2, self._pack16x2b = ops.pack_weights_f16x2_gather(flat, table, pairs, n_aux, pairs1 if with_bwd else 0)
        return self._pack16x2

    def _packed_bwd_f16x2(self):
        self._packed_f16x2(with_bwd=True)
        return self._pack16x2b

    def _train_streams(self):
        """Everything a training step on the default kernels reads, with ONE look at the parameters: (flat parameters,
        fp16-pair stream, its transposed stream, bf16-triple stream, its transposed stream)."""
        flat = self._flat()
        need2 = self._pack16x2 is None or self._pack16x2b is None
        need3 = self._pack16x3 is None or self._pack16x3b is None
        if need2 and need3:      # (the case of every training step: all four streams from one launch)
            if self._x2_table is None or self._x2_table[0].device != flat.device:
                self._x2_table = ops.x2_pack_table(self.use_viewdirs, self.output_ch, flat.device)
            if self._x3_table is None or self._x3_table[0].device != flat.device:
                self._x3_table = ops.x3_pack_table(self.use_viewdirs, self.output_ch, flat.device)
            self._pack16x2, self._pack16x2b, self._pack16x3, self._pack16x3b = ops.pack_weights_train_streams(
                flat, self._x2_table, self._x3_table)
        else:
            if need2:
                self._packed_f16x2(with_bwd=True)
            if need3:
                self._packed_x3_pair()
        return flat, self._pack16x2, self._pack16x2b, self._pack16x3, self._pack16x3b

    def _packed_bwd_bf16x3(self):
        self._flat()
        if self._pack16x3b is None:
            self._packed_x3_pair()
        return self._pack16x3b

    def _packed_x3_pair(self):
        flat = self._flat()
        if self._x3_table is None or self._x3_table[0].device != flat.device:
            self._x3_table = ops.x3_pack_table(self.use_viewdirs, self.output_ch, flat.device)
        self._pack16x3, self._pack16x3b = ops.pack_weights_x3_pair(flat, *self._x3_table)

    def _packed_x4_pair(self):
        """(forward stream, transposed stream) of the 16-point-tile kernels: one gather through a map built once."""
        flat = self._flat()
        if self._packx4 is None:
            if self._x4_table is None or self._x4_table[0].device != flat.device:
                self._x4_table = ops.x4_pack_table(self.use_viewdirs, self.output_ch, flat.device)
            self._packx4 = ops.pack_weights_x4_pair(flat, *self._x4_table)
        return self._packx4

    def _packed_x4(self):
        return self._packed_x4_pair()[0]

    def _packed_bwd_x4(self):
        return self._packed_x4_pair()[1]

    def _split_flat(self, flat: torch.Tensor):
        out, off = [], 0
        for p in self._params():
            out.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return out

    def query(self, pts: torch.Tensor, viewdirs, pts_per_ray: int) -> torch.Tensor:
        """Fused encode + forward on raw positions [P,3] and per-ray unit directions [R,3]."""
        return ops.fused_mlp(pts, viewdirs if self.use_viewdirs else None, self, pts_per_ray)

    def forward(self, x):
        """x: already-encoded inputs [..., input_ch + input_ch_views], as upstream (run_nerf_helpers.py:109-133).
        The render path does not come through here (run_network fuses the encoding).  The fused kernels give
        gradients to the parameters only; an input that itself requires a gradient takes the layer-by-layer route,
        whose autograd covers it."""
        x = ops.to_dev(x)
        if not self.fused_arch() or (torch.is_grad_enabled() and x.requires_grad):
            return self._forward_layers(x)
        out = ops.fused_mlp(x.reshape(-1, x.shape[-1]), None, self, 0)
        return out.reshape(list(x.shape[:-1]) + [out.shape[-1]])

    def _forward_layers(self, x):
        """The module's definition layer by layer (run_nerf_helpers.py:109-133) for shapes the fused kernels do not
        cover: every nn.Linear (+ ReLU) is ops.linear, the library's own fp32 matrix kernel with its own backward;
        concatenations and splits are PyTorch views and copies."""
        def lin(layer, h, relu):
            return ops.linear(h, layer.weight, layer.bias, relu)
        pts, views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        h = pts
        for i, layer in enumerate(self.pts_linears):
            h = lin(layer, h, True)
            if i in self.skips:
                h = torch.cat([pts, h], -1)
        if not self.use_viewdirs:
            return lin(self.output_linear, h, False)
        alpha = lin(self.alpha_linear, h, False)
        h = torch.cat([lin(self.feature_linear, h, False), views], -1)
        for layer in self.views_linears:
            h = lin(layer, h, True)
        return torch.cat([lin(self.rgb_linear, h, False), alpha], -1)

    def load_weights_from_keras(self, weights):
        """run_nerf_helpers.py:136-163."""
        assert self.use_viewdirs, "Not implemented if use_viewdirs=False"
        dev = self.alpha_linear.weight.device

        def put(lin, iw):
            lin.weight.data = torch.from_numpy(np.transpose(weights[iw])).to(dev)
            lin.bias.data = torch.from_numpy(np.transpose(weights[iw + 1])).to(dev)
        for i in range(self.D):
            put(self.pts_linears[i], 2 * i)
        put(self.feature_linear, 2 * self.D)
        put(self.views_linears[0], 2 * self.D + 2)
        put(self.rgb_linear, 2 * self.D + 4)
        put(self.alpha_linear, 2 * self.D + 6)


# ------------------------------------------------------------------------------------------------
# Ray helpers (run_nerf_helpers.py:169-255)
# ------------------------------------------------------------------------------------------------
def get_rays(H, W, K, c2w):
    return ops.get_rays(int(H), int(W), float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), c2w)


def get_rays_np(H, W, K, c2w):
    """Host-side float64 rays used once at start-up to precompute training rays (run_nerf_helpers.py:217-226)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    sx = -1. / (W / (2. * focal))
    sy = -1. / (H / (2. * focal))
    return ops.ndc_rays(float(sx), float(sy), float(near), ops.to_dev(rays_o), ops.to_dev(rays_d))


# ------------------------------------------------------------------------------------------------
# Hierarchical sampling (run_nerf_helpers.py:260-303)
# ------------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    bins, weights = ops.to_dev(bins), ops.to_dev(weights)
    lead = bins.shape[:-1]
    bins2, w2 = bins.reshape(-1, bins.shape[-1]), weights.reshape(-1, weights.shape[-1])
    n = bins2.shape[0]
    if pytest:
        np.random.seed(0)
        if det:
            u = torch.Tensor(np.linspace(0., 1., N_samples)).to(bins.device)
        else:
            u = torch.Tensor(np.random.rand(*(list(lead) + [N_samples]))).to(bins.device).reshape(n, N_samples)
    elif det:
        u = ops.cached_linspace(N_samples, bins.device)
    else:
        u = torch.rand(n, N_samples, device=bins.device)
    return ops.sample_pdf(bins2, w2, u).reshape(list(lead) + [N_samples])
