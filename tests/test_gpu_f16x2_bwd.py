"""Backward-data and the weight-gradient stage on two fp16 pieces per operand (the default training path): agreement
with the three-piece kernels and with fp64, the power-of-two scales (per point; per layer and batch), and the range rules
with their fall-backs."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from test_gpu_parity import T, _model, _peaky
from test_gpu_f16x2 import _inputs

pytestmark = pytest.mark.gpu


class _Case:
    """One network, one batch of points, its saved activations; the parameter gradients for an upstream gradient G on
    the two-piece stage (with and without the fall-back) and on the three-piece stage, all behind the SAME
    weight-gradient stage, so every difference comes from the dY rows."""

    def __init__(self, device, sd, n=48, s=16, seed=5):
        from nerf_amd import ops
        self.ops, self.n, self.s = ops, n, s
        rng = np.random.Generator(np.random.PCG64(seed))
        pts, vd = _inputs(rng, n, s, span=2.0)
        self.pts, self.vd = pts, vd
        self.m = m = _model(sd, device)
        self.flat_pts, self.dirs = pts.reshape(-1, 3).to(device), vd.to(device)
        self.act = ops.mlp_act_buffer(n * s, True, device)
        self.x3 = (m._packed_bf16x3(with_bwd=True), m._packed_bwd_bf16x3())
        self.x2 = (m._packed_f16x2(with_bwd=True), m._packed_bwd_f16x2())
        with torch.no_grad():
            self.raw = ops.mlp_forward_f16x2(self.flat_pts, self.dirs, self.x2[0], self.x3[0], True, 5, s, self.act)
        self.G = T(rng.normal(size=(n * s, 4)).astype(np.float32)).to(device)

    def two(self, G, fallback=True, dw="f16x2"):
        x3 = self.x3 if fallback else (None, None)
        d, flags = self.ops.mlp_backward_f16x2(G, self.act, self.m._flat(), *self.x2, *x3, True, 5, want_flags=True, dw=dw)
        return d.cpu().numpy(), flags.cpu().numpy()

    def three(self, G):
        return self.ops.mlp_backward_bf16x3(G, self.act, self.m._flat(), *self.x3, True, 5).cpu().numpy()


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_two_piece_backward_agrees_with_the_three_piece_kernel_and_fp64(device):
    """Ordinary gradients: no tile is marked, the parameter gradients behind the two-piece stage are those behind the
    three-piece stage to a few fp32 roundings of each tensor's scale, and neither is further from an fp64 evaluation
    than the other by more than that."""
    sd = _peaky(O.random_state_dict(91), 5.0)
    c = _Case(device, sd, n=70, s=64)
    d2, flags = c.two(c.G)
    d3 = c.three(c.G)
    assert flags.sum() == 0 and flags.size == (70 * 64 + 127) // 128
    params = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    (O.query_field(c.pts.double(), c.vd.double(), params) * c.G.cpu().double().reshape(70, 64, 4)).sum().backward()
    off = 0
    for k, p in c.m.named_parameters():
        n = p.numel()
        a, b = d2[off:off + n].astype(np.float64), d3[off:off + n].astype(np.float64)
        truth = params[k].grad.reshape(-1).numpy()
        scale = float(np.abs(truth).max())
        assert np.abs(a - b).max() <= 4e-6 * scale, (k, np.abs(a - b).max() / scale)
        e2, e3 = np.abs(a - truth).max() / scale, np.abs(b - truth).max() / scale
        assert e2 <= max(1.5 * e3, 3e-6), (k, e2, e3)
        off += n
    assert off == d2.size


def test_every_point_carries_its_own_power_of_two_scale(device):
    """The stage is linear in each point's upstream gradient and scales it by an exact power of two, so multiplying G by
    2^-60 or 2^40 multiplies every parameter gradient by exactly that (bit for bit) and marks nothing, and a batch whose
    points' gradients span sixty orders of magnitude agrees with the three-piece kernel on every tensor."""
    c = _Case(device, O.random_state_dict(92))
    base, flags = c.two(c.G)
    assert flags.sum() == 0
    for k in (-60, 40):
        d, flags = c.two(c.G * 2.0 ** k)
        assert flags.sum() == 0, k
        assert _same(d, base * np.float32(2.0 ** k)), k
    rng = np.random.Generator(np.random.PCG64(6))
    spread = torch.from_numpy((10.0 ** rng.uniform(-35, 25, size=(c.n * c.s, 1))).astype(np.float32)).to(device)
    d2, flags = c.two(c.G * spread)
    d3 = c.three(c.G * spread)
    assert flags.sum() == 0
    off = 0
    for k, p in c.m.named_parameters():
        n = p.numel()
        scale = float(np.abs(d3[off:off + n]).max())
        assert np.abs(d2[off:off + n] - d3[off:off + n]).max() <= 4e-6 * scale, k
        off += n
    # points without a gradient (and a batch without any) are exact zeros and mark nothing
    G = c.G.clone()
    G[128:384] = 0.0
    d, flags = c.two(G)
    assert flags.sum() == 0 and np.isfinite(d).all()
    d, flags = c.two(torch.zeros_like(c.G))
    assert flags.sum() == 0 and not d.any()


def test_gradients_that_leave_the_range_are_redone_by_the_three_piece_kernel(device):
    """Marked tiles: an infinite or NaN upstream gradient marks its tile (and only that one), a subnormal one too; a
    layer that grows every gradient by more than 2^7, or a weight outside the range, marks every tile, and the result
    is then the three-piece stage's bit for bit."""
    c = _Case(device, O.random_state_dict(93))
    tiles = c.n * c.s // 128
    for value in (float("inf"), float("nan")):
        G = c.G.clone()
        G[3 * 128 + 5, 1] = value
        d2, flags = c.two(G)
        d3 = c.three(G)
        assert flags.tolist() == [0, 0, 0, 1, 0, 0], value
        assert np.array_equal(np.isnan(d2), np.isnan(d3)) and np.array_equal(np.isinf(d2), np.isinf(d3)), value
    G = c.G.clone()
    G[128:256] *= 1e-42      # subnormal: the scale cannot lift them into range
    d2, flags = c.two(G)
    assert flags.tolist() == [0, 1, 0, 0, 0, 0]
    d3 = c.three(G)
    scale = np.abs(d3).max()
    assert np.abs(d2 - d3).max() <= 4e-6 * scale

    for name, edit in (("pts_linears.3.weight", lambda w: w.mul_(4000.0)),
                       ("pts_linears.5.weight", lambda w: w.__setitem__((5, 70), 1e5)),
                       ("pts_linears.2.weight", lambda w: w.mul_(1e-6))):
        sd = O.random_state_dict(93)
        edit(sd[name])
        c = _Case(device, sd)
        d2, flags = c.two(c.G)
        assert flags.all() and flags.size == tiles, name
        assert _same(d2, c.three(c.G)), name
        alone, flags = c.two(c.G, fallback=False)
        assert flags.all(), name


def test_a_default_model_trains_on_the_two_piece_stage(device):
    """What autograd of a default model's query runs is this stage (and the three-piece one for mlp_dtype 'bf16x3')."""
    from test_gpu_parity import _query_fn
    qf = _query_fn()
    c = _Case(device, O.random_state_dict(94))
    G = c.G.reshape(c.n, c.s, 4)
    act3 = c.ops.mlp_act_buffer(c.n * c.s, True, device)     # the three-piece path saves its own forward's activations
    with torch.no_grad():
        c.ops.mlp_forward_bf16x3(c.flat_pts, c.dirs, c.x3[0], True, 5, c.s, act3)
    three = c.ops.mlp_backward_bf16x3(c.G, act3, c.m._flat(), *c.x3, True, 5).cpu().numpy()
    for dtype, want in (("f32", c.two(c.G)[0]), ("bf16x3", three)):
        c.m.mlp_dtype = dtype
        c.m.zero_grad()
        (qf(c.pts.to(device), c.dirs, c.m) * G).sum().backward()
        got = torch.cat([p.grad.reshape(-1) for p in c.m.parameters()]).cpu().numpy()
        assert _same(got, want), dtype


def test_two_piece_weight_gradients_against_three_pieces_and_fp64(device):
    """The weight-gradient stage alone (same dY rows, which the two-piece backward-data stage left): per tensor, the
    two-piece products are as far from an fp64 evaluation as the three-piece ones (1.5x, or 1e-6 of the tensor's largest
    entry), with gradients whose sizes differ by six orders of magnitude from layer group to layer group (colour and
    density terms) and from point to point."""
    sd = _peaky(O.random_state_dict(95), 5.0)
    c = _Case(device, sd, n=70, s=64)
    rng = np.random.Generator(np.random.PCG64(9))
    spread = torch.from_numpy((10.0 ** rng.uniform(-6, 0, size=(c.n * c.s, 1))).astype(np.float32)).to(device)
    for G in (c.G, c.G * spread, c.G * torch.tensor([1e-6, 1e-6, 1e-6, 1.0], device=device), c.G * torch.tensor([1.0, 1.0, 1.0, 1e-6], device=device)):
        d2, flags = c.two(G, dw="f16x2")
        d3, _ = c.two(G, dw="bf16x3")
        assert flags.sum() == 0
        params = {k: v.double().requires_grad_(True) for k, v in sd.items()}
        (O.query_field(c.pts.double(), c.vd.double(), params) * G.cpu().double().reshape(70, 64, 4)).sum().backward()
        off = 0
        for k, p in c.m.named_parameters():
            n = p.numel()
            truth = params[k].grad.reshape(-1).numpy()
            scale = float(np.abs(truth).max())
            e2 = np.abs(d2[off:off + n] - truth).max() / scale
            e3 = np.abs(d3[off:off + n] - truth).max() / scale
            assert e2 <= max(1.5 * e3, 1e-6), (k, e2, e3)
            off += n


def test_a_marked_tile_sends_the_weight_gradients_to_the_three_piece_kernel(device):
    """The choice is made on the device: one marked tile, in the backward-data stage (a subnormal upstream gradient) or
    in the forward that saved the activations (positions far outside), and the result is the three-piece weight-gradient
    kernel's bit for bit; nothing marked, and it is not (the two forms round differently)."""
    from test_gpu_parity import _query_fn
    c = _Case(device, O.random_state_dict(96))
    G = c.G.clone()
    G[128:256] *= 1e-42
    d2, flags = c.two(G, dw="f16x2")
    d3, _ = c.two(G, dw="bf16x3")
    assert flags.sum() == 1 and _same(d2, d3)
    d2, flags = c.two(c.G, dw="f16x2")
    d3, _ = c.two(c.G, dw="bf16x3")
    assert flags.sum() == 0 and not _same(d2, d3)
    # through a model's autograd: the forward's flags reach the weight-gradient stage
    qf = _query_fn()
    far = c.pts.clone()
    far.reshape(-1, 3)[2 * 128: 3 * 128] *= 1e5
    grads = {}
    for dw in ("f16x2", "bf16x3"):
        c.m.dw_dtype = dw
        c.m.zero_grad()
        (qf(far.to(device), c.dirs, c.m) * c.G.reshape(c.n, c.s, 4)).sum().backward()
        grads[dw] = torch.cat([p.grad.reshape(-1) for p in c.m.parameters()]).cpu().numpy()
    assert _same(grads["f16x2"], grads["bf16x3"])


def test_what_the_two_piece_weight_gradients_keep_of_small_rows_and_columns(device):
    """Activations keep 22 bits of their own magnitude: the weights behind an input unit that stays around 1e-5 for the
    whole batch (layer 3's unit 7, through its weights and bias) get gradients as accurate, relative to that column, as
    from three pieces.  The gradient side is exact to 2^-39 of its layer's largest gradient in the batch: a unit whose
    gradients stay 1e-7 of that (layer 4's unit 9, through the weights that read it) keeps its row of the weight
    gradient to 1e-3 of the row's largest entry; every tensor as a whole stays within 1e-6 of its largest entry."""
    sd = O.random_state_dict(97)
    sd["pts_linears.3.weight"][7] *= 1e-5
    sd["pts_linears.3.bias"][7] = 1e-5
    sd["pts_linears.5.weight"][:, 63 + 9] *= 1e-7          # (layer 5 reads [encoding, h4])
    c = _Case(device, sd, n=70, s=64)
    d2, flags = c.two(c.G, dw="f16x2")
    d3, _ = c.two(c.G, dw="bf16x3")
    assert flags.sum() == 0
    params = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    (O.query_field(c.pts.double(), c.vd.double(), params) * c.G.cpu().double().reshape(70, 64, 4)).sum().backward()
    off = 0
    for k, p in c.m.named_parameters():
        n = p.numel()
        if k == "pts_linears.4.weight":
            truth = params[k].grad.numpy()
            g2, g3 = d2[off:off + n].reshape(truth.shape), d3[off:off + n].reshape(truth.shape)
            whole = np.abs(truth).max()
            col = np.abs(truth[:, 7]).max()
            assert col < 1e-3 * whole
            assert np.abs(g3[:, 7] - truth[:, 7]).max() <= 1e-5 * col
            assert np.abs(g2[:, 7] - truth[:, 7]).max() <= 1e-5 * col
            # (the row's dY comes out of the two-piece backward-data stage, whose weights are exact to 2^-36 absolute: for
            # weights of 1e-8 that is its own loss; here the weight-gradient stages are compared on the SAME dY rows)
            row = np.abs(g3[9]).max()
            assert 0 < row < 1e-5 * whole
            assert np.abs(g2[9] - g3[9]).max() <= 1e-3 * row
            assert np.abs(g2 - truth).max() <= 1e-6 * whole
        off += n


def test_two_piece_gradient_stages_on_ragged_batches_and_every_head_width(device):
    """Many small batches of odd sizes (1 .. 700 points: partial tiles, slices of the point range shorter than a stage
    pair), with and without the view branch, every output width 1..8 of the output_linear head, upstream gradients of
    random overall size: from the SAME saved activations (two forwards would differ by ReLU units that sit on zero),
    the parameter gradients of the two-piece stages agree with those of the three-piece stages per tensor to 5e-6 of
    the tensor's largest entry, nothing is marked, and parameters no stage touches get exact zeros."""
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import NeRF
    rng = np.random.Generator(np.random.PCG64(31))
    cases = [(True, 5, 1, 1), (True, 5, 3, 43), (True, 5, 7, 100), (True, 5, 129, 1), (True, 5, 11, 64)]
    cases += [(False, och, int(rng.integers(1, 40)), int(rng.integers(1, 18))) for och in range(1, 9)]
    for use_vd, och, n, s in cases:
        if use_vd:
            m = _model(O.random_state_dict(500 + n), device)
        else:
            m = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=och, skips=[4], use_viewdirs=False)
            torch.manual_seed(600 + och)
            for p in m.parameters():
                torch.nn.init.uniform_(p, -0.08, 0.08)
            m = m.to(device)
        pts, vd = _inputs(rng, n, s, span=2.5)
        flat_pts, dirs = pts.reshape(-1, 3).to(device), (vd.to(device) if use_vd else None)
        G = T(rng.normal(size=(n * s, 4 if use_vd else och)).astype(np.float32)).to(device) * float(10.0 ** rng.uniform(-8, 3))
        act = ops.mlp_act_buffer(n * s, use_vd, device)
        x3 = (m._packed_bf16x3(with_bwd=True), m._packed_bwd_bf16x3())
        x2 = (m._packed_f16x2(with_bwd=True), m._packed_bwd_f16x2())
        with torch.no_grad():
            _, fwd_flags = ops.mlp_forward_f16x2(flat_pts, dirs, x2[0], x3[0], use_vd, och, s, act, want_flags=True)
        g2, flags = ops.mlp_backward_f16x2(G, act, m._flat(), *x2, *x3, use_vd, och, want_flags=True, fwd_flags=fwd_flags)
        g3 = ops.mlp_backward_bf16x3(G, act, m._flat(), *x3, use_vd, och)
        assert int(fwd_flags.sum()) == 0 and int(flags.sum()) == 0, (use_vd, och, n, s)
        g2, g3 = g2.cpu().numpy(), g3.cpu().numpy()
        off = 0
        for k, p in m.named_parameters():
            nn = p.numel()
            a, b = g2[off:off + nn], g3[off:off + nn]
            off += nn
            scale = float(np.abs(b).max())
            if scale == 0.0:
                assert not a.any(), (use_vd, och, n, s, k)
            else:
                assert np.abs(a - b).max() <= 5e-6 * scale, (use_vd, och, n, s, k, np.abs(a - b).max() / scale)
