"""The default no-grad MLP kernel: two fp16 pieces per operand on v_mfma_f32_16x16x32_f16, with out-of-range tiles
handed to the three-piece bf16 kernel.  Accuracy against fp64 next to the fp32 matrix instruction, and the range rules."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from test_gpu_parity import T, _model, _peaky, _query_fn, close

pytestmark = pytest.mark.gpu


def _inputs(rng, n, s, span=4.0):
    pts = T(rng.uniform(-span, span, size=(n, s, 3)).astype(np.float32))
    vd = T(rng.normal(size=(n, 3)).astype(np.float32))
    return pts, vd / vd.norm(dim=-1, keepdim=True)


def test_two_piece_kernel_is_as_accurate_as_the_fp32_matrix_instruction(device):
    """Against an fp64 evaluation the two-piece kernel's error is that of the fp32 matrix-instruction kernel (within 1.5x,
    or below 1e-6 of the outputs' scale) -- the bound the three-piece kernel is held to -- with and without the view
    branch, on random and on peaky weights; no tile of these inputs leaves the range, so the numbers are the two-piece
    kernel's own; and it is what a default model's no-grad query runs."""
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import NeRF
    rng = np.random.Generator(np.random.PCG64(77))
    qf = _query_fn()
    for use_vd, peak, n, s in ((True, 1.0, 300, 64), (True, 30.0, 130, 33), (False, 1.0, 257, 17)):
        if use_vd:
            sd = _peaky(O.random_state_dict(400 + n), peak)
            m = _model(sd, device)
        else:
            sd = O.random_state_dict(401, use_viewdirs=False)
            sd = {k: v for k, v in sd.items() if not k.startswith("views_linears")}
            m = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
            m.load_state_dict(sd, strict=False)
            m = m.to(device)
        pts, vd = _inputs(rng, n, s)
        truth = O.query_field(pts.double(), vd.double() if use_vd else None, {k: v.double() for k, v in sd.items()})
        flat, dirs = pts.reshape(-1, 3).to(device), vd.to(device) if use_vd else None
        with torch.no_grad():
            r2, flags = ops.mlp_forward_f16x2(flat, dirs, m._packed_f16x2(), None, use_vd, 5, s, want_flags=True)
            r32 = ops.mlp_forward_x4(flat, dirs, m._packed_x4(), use_vd, 5, s)
            assert m.mlp_dtype == 'f32'
            default = qf(pts.to(device), dirs, m)
        assert int(flags.sum()) == 0, "ordinary activations stay inside fp16's normal range"
        r2d, r32d = r2.cpu().double().reshape(truth.shape), r32.cpu().double().reshape(truth.shape)
        scale = float(truth.abs().max())
        e2, e32 = float((r2d - truth).abs().max()), float((r32d - truth).abs().max())
        assert e2 <= max(1.5 * e32, 1e-6 * scale), (use_vd, peak, e2, e32, scale)
        rms2, rms32 = float((r2d - truth).pow(2).mean().sqrt()), float((r32d - truth).pow(2).mean().sqrt())
        assert rms2 <= max(1.5 * rms32, 2e-7 * scale), (use_vd, peak, rms2, rms32, scale)
        assert torch.equal(default.reshape(r2.shape), r2)
        close(r2, r32.cpu(), 3e-5, 3e-6 * max(1.0, scale), "two-piece kernel vs fp32 matrix instruction")


def test_tiles_that_leave_the_fp16_range_are_redone_by_the_three_piece_kernel(device):
    """Range rules of the two-piece kernel.  A tile (128 points) is marked when a point's largest operand of some layer is
    >= 2^15, infinite, or < 2^-11, or when a weight section is; marked tiles come out with the three-piece kernel's bits,
    the others keep the two-piece kernel's."""
    from nerf_amd import ops
    rng = np.random.Generator(np.random.PCG64(8))
    n, s = 48, 16                                    # 768 points = 6 tiles
    pts, vd = _inputs(rng, n, s, span=2.0)
    flat, dirs = pts.reshape(-1, 3).to(device), vd.to(device)

    def run(sd, flat_pts=flat):
        m = _model(sd, device)
        with torch.no_grad():
            fixed, flags = ops.mlp_forward_f16x2(flat_pts, dirs, m._packed_f16x2(), m._packed_bf16x3(), True, 5, s, want_flags=True)
            alone = ops.mlp_forward_f16x2(flat_pts, dirs, m._packed_f16x2(), None, True, 5, s)
            three = ops.mlp_forward_bf16x3(flat_pts, dirs, m._packed_bf16x3(), True, 5, s)
            mfma32 = ops.mlp_forward_x4(flat_pts, dirs, m._packed_x4(), True, 5, s)
        return fixed.cpu().numpy(), flags.cpu().numpy(), alone.cpu().numpy(), three.cpu().numpy(), mfma32.cpu().numpy()

    def same(a, b):
        return np.array_equal(a, b, equal_nan=True)

    # ordinary case: nothing marked
    fixed, flags, alone, three, _ = run(O.random_state_dict(81))
    assert flags.sum() == 0 and same(fixed, alone)

    # positions of tile 2 far outside: a raw coordinate of 1e5 is an operand of layer 0
    far = flat.clone()
    far[2 * 128: 3 * 128] *= 1e5
    fixed, flags, alone, three, mfma32 = run(O.random_state_dict(81), far)
    assert flags.tolist() == [0, 0, 1, 0, 0, 0]
    assert same(fixed[256:384], three[256:384]) and same(fixed[:256], alone[:256]) and same(fixed[384:], alone[384:])
    scale = float(np.abs(mfma32).max())
    np.testing.assert_allclose(fixed, mfma32, rtol=0, atol=3e-6 * scale)

    # one hidden unit of layer 0 driven to 1e30 through its bias: every tile marked, results those of the three-piece
    # kernel, which agree with the fp32 instruction; an infinite or NaN unit: NaN everywhere, as upstream
    for value in (1e30, float("inf"), float("nan")):
        sd = O.random_state_dict(81)
        sd["pts_linears.0.bias"][7] = value
        fixed, flags, alone, three, mfma32 = run(sd)
        if np.isfinite(value):
            assert flags.all() and same(fixed, three)
            np.testing.assert_allclose(fixed, mfma32, rtol=0, atol=3e-6 * float(np.abs(mfma32).max()))
        else:
            assert np.isnan(fixed).all() and np.isnan(mfma32).all()

    # a dead layer: every unit of layer 1 switched off by its bias.  All operands of layer 2 are exactly 0, which any
    # scheme multiplies exactly: nothing is marked.  A layer that is merely tiny (outputs of layer 1 scaled to ~1e-5 through
    # its weights and bias... here: its bias alone, weights zero) is below 2^-11 and goes to the three-piece kernel
    sd = O.random_state_dict(81)
    sd["pts_linears.1.bias"][:] = -1e3
    fixed, flags, alone, three, mfma32 = run(sd)
    assert flags.sum() == 0 and same(fixed, alone)
    np.testing.assert_allclose(fixed, mfma32, rtol=0, atol=3e-6 * float(np.abs(mfma32).max()))
    sd = O.random_state_dict(81)
    sd["pts_linears.1.weight"][:] = 0.0
    sd["pts_linears.1.bias"][:] = 1e-5
    fixed, flags, alone, three, mfma32 = run(sd)
    assert flags.all() and same(fixed, three)

    # weights outside the range: one weight of 1e5 (layer 3), and a whole layer of tiny weights (layer 6 x 1e-6)
    for name, edit in (("pts_linears.3.weight", lambda w: w.__setitem__((5, 9), 1e5)), ("pts_linears.6.weight", lambda w: w.mul_(1e-6))):
        sd = O.random_state_dict(81)
        edit(sd[name])
        fixed, flags, alone, three, mfma32 = run(sd)
        assert flags.all() and same(fixed, three), name
        if name.endswith("6.weight"):   # (behind a weight of 1e5 the sums cancel, and any two fp32 summation orders part ways)
            np.testing.assert_allclose(fixed, mfma32, rtol=0, atol=3e-6 * float(np.abs(mfma32).max()), err_msg=name)


@pytest.mark.parametrize("use_viewdirs,output_ch", [(True, 5), (False, 4), (False, 5)])
def test_pair_stream_from_one_gather_equals_the_packing_kernel(device, use_viewdirs, output_ch):
    """nerf_gather_split2 through the map taken from the packing kernels rebuilds the fp16-pair stream, its fp32 tables,
    its range words and the transposed fp16-pair stream bit for bit, on ordinary weights and on values whose pieces carry
    signs, zeros and fp16 overflow; without the second stream the first comes out the same."""
    from nerf_amd import ops
    n = ops.mlp_n_params(use_viewdirs, output_ch)
    table = ops.x2_pack_table(use_viewdirs, output_ch, device)
    rng = np.random.Generator(np.random.PCG64(3))
    for scale in (1.0, 1e-3, 1e5):
        flat = T((rng.standard_normal(n) * scale).astype(np.float32)).to(device)
        flat[::7] = 0.0
        flat[1::11] = -flat[1::11].abs()
        a, at = ops.pack_weights_f16x2_gather(flat, *table)
        b = ops.pack_weights_f16x2(flat, use_viewdirs, output_ch)
        assert torch.equal(a, b), scale
        assert torch.equal(at, ops.pack_weights_bwd_f16x2(flat, use_viewdirs, output_ch)), scale
        a0, none = ops.pack_weights_f16x2_gather(flat, *table[:3])
        assert none is None and torch.equal(a0, b), scale
        # all four streams of a training step from ONE launch: the bits of the two separate gathers
        table3 = ops.x3_pack_table(use_viewdirs, output_ch, device)
        f2, b2, f3, b3 = ops.pack_weights_train_streams(flat, table, table3)
        g3, g3t = ops.pack_weights_x3_pair(flat, *table3)
        assert torch.equal(f2, a) and torch.equal(b2, at) and torch.equal(f3, g3) and torch.equal(b3, g3t), scale
    # argument errors of the one-launch form: counts that are not whole slabs, a missing table
    from nerf_amd._lib import NerfHipError, check, lib
    t2, pairs, n_aux2, pairs1 = table
    t3, t0, n_aux3, t1 = table3
    out = torch.empty(64, dtype=torch.uint8, device=device)
    with pytest.raises(NerfHipError, match="whole slabs"):
        check(lib.nerf_gather_split23(t2.data_ptr(), t3.data_ptr(), flat.data_ptr(), out.data_ptr(), out.data_ptr(), pairs + 1, n_aux2,
                                      pairs1, t0, n_aux3, t1, None))
    with pytest.raises(NerfHipError, match="null pointer"):
        check(lib.nerf_gather_split23(t2.data_ptr(), None, flat.data_ptr(), out.data_ptr(), out.data_ptr(), pairs, n_aux2, pairs1, t0,
                                      n_aux3, t1, None))
