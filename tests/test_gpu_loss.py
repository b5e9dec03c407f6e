"""img2mse (upstream: run_nerf_helpers.py:8, called at run_nerf.py:863-869) through the library's one-launch loss and
its one-launch gradient, against fp64 and against the stock expression."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_rays", [1, 7, 1024, 8192, 43690])
def test_img2mse_matches_fp64_and_autograd(device, n_rays):
    """Loss within one fp32 rounding of the fp64 mean of the fp32 squares (tolerance 1.2e-7 relative); the gradient
    equals upstream * 2 / n * (x - y) to two roundings (2.4e-7 relative); both repeat bit for bit."""
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import img2mse
    g = torch.Generator(device=device)
    g.manual_seed(n_rays)
    x = torch.rand(n_rays, 3, device=device, generator=g).requires_grad_(True)
    y = torch.rand(n_rays, 3, device=device, generator=g)
    loss = img2mse(x, y)
    assert loss.grad_fn is not None and type(loss.grad_fn).__name__ == "_MSEBackward", "the library's loss did not run"
    d = (x.detach() - y)
    want = float((d * d).double().mean())
    assert abs(float(loss.detach()) - want) <= 1.2e-7 * want + 1e-45
    (loss * 3.0).backward()
    want_g = (3.0 * 2.0 / x.numel()) * d.double()
    err = float((x.grad.double() - want_g).abs().max())
    assert err <= 2.4e-7 * float(want_g.abs().max()) + 1e-45
    # the stock expression, for reference
    x2 = x.detach().clone().requires_grad_(True)
    (torch.mean((x2 - y) ** 2) * 3.0).backward()
    assert float((x2.grad - x.grad).abs().max()) <= 2.4e-7 * float(want_g.abs().max())
    again = img2mse(x, y)
    assert torch.equal(again.detach(), loss.detach())
    x.grad = None
    (again * 3.0).backward()
    assert torch.equal(x.grad, x2.grad * 0 + x.grad) and float((x.grad.double() - want_g).abs().max()) == err
    assert ops.mse(x.detach(), y).shape == ()


def test_img2mse_falls_back_where_the_kernel_does_not_apply(device):
    """Broadcast shapes, strided views, a target that wants a gradient, more elements than one workgroup takes, CPU
    tensors: the stock expression, same value."""
    from nerf_amd._lib import lib
    from nerf_amd.run_nerf_helpers import img2mse
    torch.manual_seed(0)
    big = int(lib.nerf_mse_max_elems()) // 3 + 1
    x = torch.rand(big, 3, device=device, requires_grad=True)
    y = torch.rand(big, 3, device=device)
    cases = [(x, y), (x[:100, :2], y[:100, :2]), (x[:64], y[:64].clone().requires_grad_(True)), (x[:5].cpu(), y[:5].cpu())]
    for a, b in cases:
        got = img2mse(a, b)
        assert type(got.grad_fn).__name__ != "_MSEBackward"
        want = float(((a.detach().double() - b.detach().double()) ** 2).mean())
        assert abs(float(got.detach()) - want) <= 1e-6 * want
    with pytest.raises(RuntimeError):
        from nerf_amd._lib import check
        check(lib.nerf_mse_fwd(0, x.data_ptr(), y.data_ptr(), x.data_ptr(), None))


@pytest.mark.parametrize("n_rays", [1, 1024, 8192])
def test_img2mse_pair_equals_the_two_losses_and_their_sum(device, n_rays):
    """One launch for both losses of a step: the same bits as img2mse on each input, their fp32 sum, and the gradients
    autograd gives the expression img2mse(x, t) + img2mse(x0, t) -- also when a gradient arrives for one loss alone or
    for all three outputs, and when none arrives for one of the inputs."""
    from nerf_amd.run_nerf_helpers import img2mse, img2mse_pair
    g = torch.Generator(device=device)
    g.manual_seed(7 + n_rays)
    x = torch.rand(n_rays, 3, device=device, generator=g).requires_grad_(True)
    x0 = torch.rand(n_rays, 3, device=device, generator=g).requires_grad_(True)
    t = torch.rand(n_rays, 3, device=device, generator=g)
    a, b, total = img2mse_pair(x, x0, t)
    assert type(total.grad_fn).__name__ == "_MSEPairBackward", "the library's paired loss did not run"
    ra, rb = img2mse(x.detach().requires_grad_(True), t), img2mse(x0.detach().requires_grad_(True), t)
    assert torch.equal(a.detach(), ra.detach()) and torch.equal(b.detach(), rb.detach())
    assert torch.equal(total.detach(), (ra + rb).detach())
    total.backward()
    xr, x0r = x.detach().clone().requires_grad_(True), x0.detach().clone().requires_grad_(True)
    (img2mse(xr, t) + img2mse(x0r, t)).backward()
    assert torch.equal(x.grad, xr.grad) and torch.equal(x0.grad, x0r.grad)
    # gradients for single outputs and for a mix
    x.grad = x0.grad = None
    a, b, total = img2mse_pair(x, x0, t)
    (2.0 * a).backward()
    assert float(x0.grad.abs().max()) == 0.0
    assert float((x.grad - 2.0 * xr.grad).abs().max()) <= 2.4e-7 * float(xr.grad.abs().max()) * 2.0
    x.grad = x0.grad = None
    a, b, total = img2mse_pair(x, x0, t)
    (a + 3.0 * b + total).backward()
    assert float((x.grad - 2.0 * xr.grad).abs().max()) <= 5e-7 * float(xr.grad.abs().max())
    assert float((x0.grad - 4.0 * x0r.grad).abs().max()) <= 1e-6 * float(x0r.grad.abs().max())
    # where the kernel does not apply: the same three values from the single losses
    a2, b2, t2 = img2mse_pair(x[:, ::2], x0[:, ::2], t[:, ::2])      # (every other column: not contiguous, even for one row)
    assert type(t2.grad_fn).__name__ != "_MSEPairBackward"
    want = float(((x.detach()[:, ::2].double() - t[:, ::2].double()) ** 2).mean() + ((x0.detach()[:, ::2].double() - t[:, ::2].double()) ** 2).mean())
    assert abs(float(t2.detach()) - want) <= 1e-6 * want
