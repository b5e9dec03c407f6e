"""Pin nerf_amd/datasets.py to the reference's loaders: writes tests/golden/datasets.npz.

Runs only where the upstream tree is mounted at /root/reference.  Nothing is copied: the reference's
load_blender.py is imported as it is (with empty stand-in modules for imageio and cv2, which pose_spherical
does not touch), and of load_llff.py -- whose module body loads a dataset that is not there -- only the
function definitions are compiled, from the file where it lies, and run on synthetic camera poses through
load_llff_data with its file reader replaced by one that returns those poses.

Usage:  python oracle/gen_golden_datasets.py
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def synthetic_llff(seed, n=9, h=6, w=8):
    """Camera poses scattered around a forward-facing rig, in the poses_bounds.npy conventions."""
    rng = np.random.Generator(np.random.PCG64(seed))
    poses = np.zeros((3, 5, n))
    for i in range(n):
        q, _ = np.linalg.qr(np.eye(3) + 0.15 * rng.normal(size=(3, 3)))
        q = q * np.sign(np.diag(q))          # keep the rig roughly aligned with the axes
        poses[:, :3, i] = q
        poses[:, 3, i] = rng.normal(size=3) * np.array([1.0, 0.4, 0.2])
        poses[:, 4, i] = [h, w, 11.5]
    bds = np.stack([rng.uniform(1.2, 2.0, n), rng.uniform(6.0, 9.0, n)], 0)
    imgs = rng.uniform(0, 1, (h, w, 3, n))
    return poses, bds, imgs


def main():
    from nerf_amd import datasets as D
    out = {}

    for name in ("imageio", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    spec = importlib.util.spec_from_file_location("ref_load_blender", os.path.join(REF, "load_blender.py"))
    ref_blender = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_blender)
    angles = [(-180.0, -30.0, 4.0), (-171.0, -30.0, 4.0), (54.0, -30.0, 4.0), (12.5, 40.0, 2.5)]
    ref = np.stack([ref_blender.pose_spherical(*a).numpy() for a in angles], 0)
    mine = np.stack([D.pose_spherical(*a).numpy() for a in angles], 0)
    assert np.array_equal(ref, mine), "pose_spherical differs from the reference"
    out["sph_args"], out["sph_pose"] = np.array(angles), ref

    with open(os.path.join(REF, "load_llff.py"), "r") as fh:
        tree = ast.parse(fh.read())
    tree.body = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
    ns = {"np": np, "os": os}
    exec(compile(tree, os.path.join(REF, "load_llff.py"), "exec"), ns)
    # (path_zflat=True cannot be pinned: with numpy 2 the reference itself stops in np.linspace on its float view count)
    for tag, seed, kw in (("fwd", 1, {}), ("sph", 2, {"spherify": True}), ("raw", 4, {"recenter": False, "bd_factor": None})):
        poses, bds, imgs = synthetic_llff(seed)
        ns["_load_data"] = lambda basedir, factor=None, **k: (poses.copy(), bds.copy(), imgs.copy())
        r_img, r_poses, r_bds, r_render, r_itest = ns["load_llff_data"]("unused", factor=8, **kw)
        m_img, m_poses, m_bds, m_render, m_itest = D._prepare_llff(poses.copy(), bds.copy(), imgs.copy(), **kw)
        for a, b, what in ((r_img, m_img, "images"), (r_poses, m_poses, "poses"), (r_bds, m_bds, "bds"),
                           (r_render, m_render, "render_poses")):
            assert a.dtype == b.dtype and np.array_equal(a, b), f"{tag} {what}: nerf_amd.datasets differs from the reference"
        assert int(r_itest) == int(m_itest), tag
        out[f"{tag}.seed"] = np.array(seed)
        out[f"{tag}.poses"], out[f"{tag}.bds"], out[f"{tag}.render_poses"] = r_poses, r_bds, r_render
        out[f"{tag}.i_test"] = np.array(int(r_itest))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "datasets.npz"), **out)
    print("datasets.npz written; nerf_amd.datasets == reference on every case")


if __name__ == "__main__":
    main()
