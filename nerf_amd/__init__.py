"""nerf_amd: the NeRF ray-marching render path on MI355X (gfx950).

`nerf_amd.run_nerf` and `nerf_amd.run_nerf_helpers` mirror the upstream modules of the same names;
`nerf_amd.ops` wraps the C ABI of libnerf_hip.so; `nerf_amd.parallel` shards rays across GPUs.
Importing the package loads the shared library and fails if it has not been built.
"""
from . import _lib  # noqa: F401  (raises ImportError when libnerf_hip.so is missing)
from . import ops, run_nerf, run_nerf_helpers  # noqa: F401

__all__ = ["ops", "run_nerf", "run_nerf_helpers"]
