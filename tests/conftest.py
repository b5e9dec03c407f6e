import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_meta():
    with open(os.path.join(GOLDEN, "meta.json")) as fh:
        return json.load(fh)


def same_cpu_as_golden():
    m = golden_meta()
    return m["cpu_capability"] == torch.backends.cpu.get_cpu_capability() and m["torch"] == torch.__version__


def assert_pinned(got, want, what, rtol=2e-5, atol=2e-6):
    """Oracle output vs a committed reference output.

    Bit-for-bit when this CPU dispatches the same ATen kernels as the one that
    made the fixtures (same torch build and vector ISA); otherwise fp32
    reduction order may differ in the last bits and a tight tolerance applies.
    """
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if same_cpu_as_golden():
        assert np.array_equal(got, want, equal_nan=True), f"{what}: not bit-identical, max|d|={np.nanmax(np.abs(got - want))}"
    else:
        np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, equal_nan=True, err_msg=what)


@pytest.fixture(scope="session")
def device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
