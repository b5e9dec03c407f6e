"""Random ragged batches through the fused MLP on every arithmetic of the training step, forward and parameter gradients,
against an fp64 evaluation of the module's definition (run_nerf_helpers.py:109-133 and autograd of it) in plain torch
on the same device: python3 tools/fuzz_mlp.py [cases] [seed] [case,case,...]

Per case: a network with random weights (gain 0.25 .. 4 over the hidden layers, with or without the view branch), a ragged
batch (1 .. 3000 rays x 1 .. 130 samples, positions of scale 0.1 .. 30), an upstream gradient whose colour and density
parts differ by up to 1e6.  Points with a ReLU unit within 1e-5 of zero in the fp64 evaluation get a zero upstream
gradient (any fp32 evaluation may put such a unit on either side, which moves the gradients by that point's whole
contribution).  The default path (two fp16 pieces, marked tiles on three bf16 pieces), the three-piece path and the fp32
matrix instruction must give outputs within 2e-5 of the output scale and every parameter's gradient within 5e-5 of its
largest entry, or of twice the error plain fp32 torch makes on the same module.  Prints mismatches, the largest errors seen and the share of marked tiles; exit status = mismatches."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_amd import ops  # noqa: E402
from nerf_amd.run_nerf_helpers import NeRF  # noqa: E402

F = torch.nn.functional


def encode64(x, n_freq):
    """The positional encoding (run_nerf_helpers.py:13-60): x, then sin and cos of x * 2^k for k < n_freq."""
    out = [x]
    for k in range(n_freq):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


def evaluate(m, pts, vd, n_rays, S, dtype):
    """The module's definition in plain torch at `dtype` -> (output with graph, per-point smallest |pre-activation| of a ReLU unit)."""
    viewdirs = vd is not None
    x = encode64(pts.to(dtype), 10)
    h, margin = x, torch.full((n_rays * S,), float("inf"), dtype=dtype, device=pts.device)
    for i, lin in enumerate(m.pts_linears):
        z = F.linear(h, lin.weight, lin.bias)
        margin = torch.minimum(margin, z.detach().abs().amin(-1))
        h = torch.relu(z)
        if i == 4:
            h = torch.cat([x, h], -1)
    if viewdirs:
        dirs = encode64(vd.to(dtype)[:, None].expand(n_rays, S, 3).reshape(-1, 3), 4)
        alpha = F.linear(h, m.alpha_linear.weight, m.alpha_linear.bias)
        feat = F.linear(h, m.feature_linear.weight, m.feature_linear.bias)
        z = F.linear(torch.cat([feat, dirs], -1), m.views_linears[0].weight, m.views_linears[0].bias)
        margin = torch.minimum(margin, z.detach().abs().amin(-1))
        return torch.cat([F.linear(torch.relu(z), m.rgb_linear.weight, m.rgb_linear.bias), alpha], -1), margin
    return F.linear(h, m.output_linear.weight, m.output_linear.bias), margin


def run(cases: int, seed: int, only=None) -> int:
    rng = np.random.Generator(np.random.PCG64(seed))
    dev = torch.device("cuda", torch.cuda.current_device())
    bad = 0
    marked = [0, 0]
    worst = {k: [0.0, 0.0] for k in ("f32", "bf16x3", "f32_mfma")}     # largest output / gradient error seen
    for case in range(cases):
        viewdirs = bool(rng.random() < 0.8)
        och = 5 if viewdirs else int(rng.choice([4, 5, 8, 1]))
        torch.manual_seed(int(rng.integers(1 << 30)))
        m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27 if viewdirs else 0, output_ch=och, skips=[4], use_viewdirs=viewdirs)
        gain = float(rng.choice([0.25, 0.5, 1.0, 1.0, 2.0, 4.0]))
        with torch.no_grad():
            for lin in m.pts_linears:
                lin.weight.mul_(gain ** 0.25)
        m = m.to(dev)
        n_rays = int(rng.choice([1, 2, 3, 17, 100, 257, 1000, 3000]))
        S = int(rng.choice([1, 2, 5, 64, 129, 130]))
        scale = float(rng.choice([0.1, 1.0, 3.0, 30.0]))
        g = torch.Generator(device=dev)
        g.manual_seed(int(rng.integers(1 << 30)))
        pts = (torch.rand(n_rays * S, 3, device=dev, generator=g) - 0.5) * 2 * scale
        vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1) if viewdirs else None
        n_out = 4 if viewdirs else och
        d_out = torch.randn(n_rays * S, n_out, device=dev, generator=g)
        d_out[..., :3] *= float(rng.choice([1.0, 1e-3, 1e-6, 1e3]))
        d_out *= float(rng.choice([1.0, 1e-4, 1e2]))
        if only is not None and case not in only:      # (the case's draws are made: the following cases stay what they were)
            continue
        # fp64 truth, the per-point margin, and what plain fp32 torch makes of the same module (its error is the yardstick
        # for sums that cancel: a bias gradient is a column sum of the upstream gradient)
        m64, m32 = copy.deepcopy(m).double(), copy.deepcopy(m)
        out64, margin = evaluate(m64, pts, vd, n_rays, S, torch.float64)
        d_out = d_out * (margin > 1e-5)[:, None].float()
        (out64 * d_out.double()).sum().backward()
        truth = {k: q.grad for k, q in m64.named_parameters()}
        (evaluate(m32, pts, vd, n_rays, S, torch.float32)[0] * d_out).sum().backward()
        err32 = {k: float((q.grad.double() - truth[k]).abs().max()) for k, q in m32.named_parameters() if q.grad is not None and truth.get(k) is not None}
        oscale = float(out64.detach().abs().max()) + 1e-300
        ops.MARKED = []
        for kind in ("f32", "bf16x3", "f32_mfma"):
            m.mlp_dtype = kind
            for q in m.parameters():
                q.grad = None
            out = m.query(pts, vd, S)
            (out.reshape(n_rays * S, n_out) * d_out).sum().backward()
            if kind == "f32":
                for stage, f in ops.MARKED:
                    marked[0] += int(f.sum())
                    marked[1] += f.numel()
            e = float((out.detach().double().reshape(n_rays * S, n_out) - out64.detach()).abs().max()) / oscale
            worst[kind][0] = max(worst[kind][0], e)
            what = dict(n_rays=n_rays, S=S, scale=scale, gain=gain, viewdirs=viewdirs, och=och)
            if not e <= 2e-5:
                bad += 1
                print("MISMATCH out", kind, case, what, e, flush=True)
            for name, q in m.named_parameters():
                t = truth.get(name)
                if t is None or q.grad is None:
                    continue
                ts = float(t.abs().max())
                err = float((q.grad.double() - t).abs().max()) / (ts + 1e-300) if ts > 0 else float(q.grad.abs().max())
                worst[kind][1] = max(worst[kind][1], err)
                # (the gradient of a head's bias is a plain column sum of the upstream gradient: when it cancels to 1e-5 of
                #  the sum of its terms' magnitudes, an error of 1e-9 of that sum is large in the units above; allow 1e-7 of it)
                cancel = 1e-7 * float(d_out.double().abs().sum(0).max()) / (ts + 1e-300) if name.endswith("_linear.bias") else 0.0
                if not err <= max(5e-5, 2.0 * err32.get(name, 0.0) / (ts + 1e-300), cancel):
                    bad += 1
                    print("MISMATCH grad", kind, name, case, what, err, flush=True)
                    if only is not None and t.numel() <= 8:
                        print("   truth", t.flatten().tolist(), "got", q.grad.flatten().tolist(), "plain fp32 torch error", err32.get(name),
                              "sum |terms| of the last column", float(d_out[:, -1].double().abs().sum()), flush=True)
        ops.MARKED = None
        if (case + 1) % 20 == 0:
            print(f"{case + 1} cases, {bad} mismatches, marked tiles {marked[0]} of {marked[1]}, worst (out, grad) {worst}", flush=True)
    print(f"done: {cases} cases, {bad} mismatches, marked tiles {marked[0]} of {marked[1]}, worst (out, grad) {worst}")
    return bad


if __name__ == "__main__":
    torch.cuda.set_device(0)
    only = set(int(c) for c in sys.argv[3].split(",")) if len(sys.argv) > 3 else None      # evaluate these cases only
    sys.exit(min(run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0, only), 100))
