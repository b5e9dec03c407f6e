"""ctypes binding of libnerf_hip.so (declared in include/nerf_hip.h).

The library is the only implementation of the render-path arithmetic: importing this module
fails loudly when it has not been built, and nothing here falls back to PyTorch ops.
"""
import ctypes
import os

# PyTorch first: it brings its own copy of the HIP runtime, and the library below must bind to that same copy (the
# loader reuses an already loaded libamdhip64).  Loaded the other way round the process ends up with two runtimes,
# and kernels launched through the second one find no device.
import torch  # noqa: F401  isort: skip

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERF_HIP_LIB names another build of the same library (A/B runs of kernel variants on one box)
LIB_PATH = os.environ.get("NERF_HIP_LIB") or os.path.join(_HERE, "libnerf_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback path")

lib = ctypes.CDLL(LIB_PATH)

_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_int64
_f = ctypes.c_float
_d = ctypes.c_double

SIGNATURES = {
    "nerf_version": (_i, []),
    "nerf_arch": (ctypes.c_char_p, []),
    "nerf_last_error": (ctypes.c_char_p, []),
    "nerf_check_device": (_i, []),
    "nerf_get_rays": (_i, [_i, _i, _f, _f, _f, _f, _p, _p, _p, _p]),
    "nerf_rays_at": (_i, [_l, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p]),
    "nerf_rays_random": (_i, [_l, ctypes.c_uint64, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    "nerf_ndc_rays": (_i, [_l, _f, _f, _f, _p, _p, _p, _p, _p]),
    "nerf_pack_rays": (_i, [_l, _p, _p, _p, _i, _f, _f, _f, _f, _f, _p, _p, _p, _i, _p]),
    "nerf_sample_coarse": (_i, [_l, _i, _p, _i, _p, _p, _i, _p, _p, _p]),
    "nerf_encode": (_i, [_l, _i, _p, _p, _p]),
    "nerf_encode_general": (_i, [_l, _i, _i, _i, _p, _i, ctypes.c_uint, _p, _p, _p]),
    "nerf_sample_fine": (_i, [_l, _i, _i, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "nerf_sample_pdf": (_i, [_l, _i, _i, _p, _p, _p, _i, _p, _p, _p]),
    "nerf_composite_fwd": (_i, [_l, _i, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "nerf_composite_bwd": (_i, [_l, _i, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nerf_mlp_n_params": (_l, [_i, _i]),
    "nerf_mlp_packed_floats": (_l, [_i]),
    "nerf_mlp_act_floats": (_l, [_i, _l]),
    "nerf_mlp_scratch_floats": (_l, [_i, _l]),
    "nerf_pack_source": (_l, [_i, _i, _l]),
    "nerf_pack_source_table": (_i, [_i, _i, _p, _l]),
    "nerf_pack_weights": (_i, [_i, _i, _p, _p, _p]),
    "nerf_mlp_fwd_encoded": (_i, [_l, _p, _i, _i, _p, _p, _p, _p]),
    "nerf_gather": (_i, [_p, _p, _p, _l, _p]),
    "nerf_mlp_packed_x4_floats": (_l, [_i]),
    "nerf_pack_weights_x4": (_i, [_i, _i, _p, _p, _p]),
    "nerf_mlp_fwd_x4": (_i, [_l, _i, _p, _p, _i, _i, _p, _p, _p, _p]),
    "nerf_mlp_packed_bwd_x4_floats": (_l, [_i]),
    "nerf_pack_weights_bwd_x4": (_i, [_i, _i, _p, _p, _p]),
    "nerf_mlp_bwd_x4": (_i, [_l, _p, _p, _i, _i, _p, _p, _p, _p, _p]),
    "nerf_mlp_packed_bf16x3_bytes": (_l, [_i]),
    "nerf_pack_weights_bf16x3": (_i, [_i, _i, _p, _p, _p]),
    "nerf_mlp_fwd_bf16x3": (_i, [_l, _i, _p, _p, _i, _i, _p, _p, _p, _p]),
    "nerf_mlp_packed_f16x2_bytes": (_l, [_i]),
    "nerf_mlp_f16x2_tile_points": (_l, []),
    "nerf_pack_weights_f16x2": (_i, [_i, _i, _p, _p, _p]),
    "nerf_mse_max_elems": (_l, []),
    "nerf_mse_fwd": (_i, [_l, _p, _p, _p, _p]),
    "nerf_mse_bwd": (_i, [_l, _p, _p, _p, _p, _p]),
    "nerf_mse_pair_fwd": (_i, [_l, _p, _p, _p, _p, _p, _p, _p]),
    "nerf_mse_pair_bwd": (_i, [_l, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nerf_adam_max_runs": (_i, []),
    "nerf_adam_step": (_i, [_i, _p, _p, _p, _p, _p, _p, _f, _p, _d, _d, _d, _d, _d, _p]),
    "nerf_gather_split2": (_i, [_p, _p, _p, _l, _l, _l, _p]),
    "nerf_gather_split23": (_i, [_p, _p, _p, _p, _p, _l, _l, _l, _l, _l, _l, _p]),
    "nerf_mlp_packed_bwd_f16x2_bytes": (_l, [_i]),
    "nerf_pack_weights_bwd_f16x2": (_i, [_i, _i, _p, _p, _p]),
    "nerf_mlp_bwd_data_f16x2": (_i, [_l, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nerf_mlp_bwd_weights_f16x2": (_i, [_l, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "nerf_mlp_fwd_f16x2": (_i, [_l, _i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "nerf_mlp_packed_bwd_bf16x3_bytes": (_l, [_i]),
    "nerf_pack_weights_bwd_bf16x3": (_i, [_i, _i, _p, _p, _p]),
    "nerf_mlp_bwd_bf16x3": (_i, [_l, _p, _p, _i, _i, _p, _p, _p, _p, _p]),
    "nerf_mlp_bf16x3_triples": (_l, [_i]),
    "nerf_gather_split3": (_i, [_p, _p, _p, _l, _l, _l, _p]),
    "nerf_mlp_bwd_data_bf16x3": (_i, [_l, _p, _p, _i, _i, _p, _p, _p, _p]),
    "nerf_mlp_bwd_weights_bf16x3": (_i, [_l, _p, _i, _i, _p, _p, _p]),
    "nerf_linear_fwd": (_i, [_l, _i, _i, _p, _l, _p, _p, _i, _p, _l, _p]),
    "nerf_relu_grad": (_i, [_l, _p, _p, _p, _p]),
    "nerf_linear_bwd_data": (_i, [_l, _i, _i, _p, _l, _p, _p, _l, _p]),
    "nerf_linear_workspace_floats": (_l, [_i, _i]),
    "nerf_linear_bwd_weights": (_i, [_l, _i, _i, _p, _l, _p, _l, _p, _p, _p, _p]),
    "nerf_mlp_bwd_static_max_points": (_l, []),
    "nerf_mlp_bwd_bf16x3_static": (_i, [_l, _p, _p, _i, _i, _p, _p, _p, _p]),
    "nerf_mlp_packed_bf16_bytes": (_l, [_i]),
    "nerf_pack_weights_bf16": (_i, [_i, _i, _p, _p, _p]),
    "nerf_mlp_fwd_bf16": (_i, [_l, _i, _p, _p, _i, _i, _p, _p, _p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the built library is older than the header
    _fn.restype = _res
    _fn.argtypes = _args


class NerfHipError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        raise NerfHipError(f"libnerf_hip error {rc}: {lib.nerf_last_error().decode()}")
