"""Generate tests/golden/embedder_general.npz: the reference's Embedder (run_nerf_helpers.py:13-47) under keyword settings
other than the default one, and pin oracle.embed_general to it bit for bit.

Runs only where the upstream tree is mounted at /root/reference; the reference is imported, never copied (stand-ins for
the modules it imports but the encoder does not use, as in gen_golden.py).

Usage:  python oracle/gen_golden_embedder.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
for _name in ("imageio", "cv2"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.path.insert(0, "/root/reference")
import run_nerf_helpers as RH  # noqa: E402

from oracle import nerf_oracle as O  # noqa: E402

FNS = {"s": torch.sin, "c": torch.cos}
# name -> (include_input, input_dims, max_freq_log2, num_freqs, log_sampling, functions)
CASES = {
    "no_input": (False, 3, 9, 10, True, "sc"),
    "linear_bands": (True, 3, 4, 6, False, "sc"),
    "cos_first": (True, 3, 5, 6, True, "cs"),
    "sin_only_2d": (True, 2, 3, 4, True, "s"),
    "fractional_log": (True, 3, 4.5, 4, True, "sc"),
    "wide_5d_three_fns": (False, 5, 2, 3, False, "scs"),
    "zero_freqs": (True, 3, -1, 0, True, "sc"),
}


def main():
    rng = np.random.Generator(np.random.PCG64(2024))
    out = {}
    for name, (inc, d, mf, nf, log, fns) in CASES.items():
        x = torch.from_numpy(rng.uniform(-3.0, 3.0, size=(37, d)).astype(np.float32))
        kw = dict(include_input=inc, input_dims=d, max_freq_log2=mf, num_freqs=nf, log_sampling=log,
                  periodic_fns=[FNS[c] for c in fns])
        ref = RH.Embedder(**kw)
        want = ref.embed(x)
        assert want.shape[-1] == ref.out_dim
        got = O.embed_general(x, **kw)
        if got.shape != want.shape or not torch.equal(got, want):
            raise SystemExit(f"oracle != reference for {name}")
        out[name + "_x"] = x.numpy()
        out[name + "_out"] = want.numpy()
        out[name + "_kw"] = np.array([int(inc), d, float(mf), nf, int(log)], dtype=np.float64)
        out[name + "_fns"] = np.array([ord(c) for c in fns], dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "embedder_general.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
