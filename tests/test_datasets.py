"""Dataset readers (CPU): LLFF pose processing and pose_spherical against outputs of the reference's own
functions (tests/golden/datasets.npz, made by oracle/gen_golden_datasets.py), file plumbing on synthetic scenes
written here.  The half_res path (2x2 block mean in place of OpenCV's area interpolation) and the in-memory
Lanczos reduction are parity unpinned: OpenCV / ImageMagick are not available to produce reference outputs."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import golden

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))


def _datasets():
    # the package import needs the built library; the readers themselves are pure numpy
    return pytest.importorskip("nerf_amd.datasets")


def test_pose_spherical_matches_reference():
    D = _datasets()
    g = golden("datasets")
    for args, want in zip(g["sph_args"], g["sph_pose"]):
        np.testing.assert_array_equal(D.pose_spherical(*[float(a) for a in args]).numpy(), want)


def test_llff_pose_processing_matches_reference():
    D = _datasets()
    from gen_golden_datasets import synthetic_llff
    g = golden("datasets")
    for tag, kw in (("fwd", {}), ("sph", {"spherify": True}), ("raw", {"recenter": False, "bd_factor": None})):
        poses, bds, imgs = synthetic_llff(int(g[f"{tag}.seed"]))
        images, p, b, render_poses, i_test = D._prepare_llff(poses, bds, imgs, **kw)
        assert images.shape == (9, 6, 8, 3) and images.dtype == np.float32
        np.testing.assert_allclose(p, g[f"{tag}.poses"], rtol=1e-5, atol=1e-6, err_msg=tag)
        np.testing.assert_allclose(b, g[f"{tag}.bds"], rtol=1e-6, err_msg=tag)
        np.testing.assert_allclose(render_poses, g[f"{tag}.render_poses"], rtol=1e-5, atol=1e-6, err_msg=tag)
        assert int(i_test) == int(g[f"{tag}.i_test"])
    # the flat path of the reference stops inside numpy 2 (float view count); here it yields 60 views
    poses, bds, imgs = synthetic_llff(3)
    assert D._prepare_llff(poses, bds, imgs, path_zflat=True)[3].shape == (60, 3, 5)


def write_blender_scene(root, n=(3, 2, 2), size=8, seed=0):
    from PIL import Image
    D = _datasets()
    rng = np.random.Generator(np.random.PCG64(seed))
    pix = {}
    for split, count in zip(("train", "val", "test"), n):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for k in range(count):
            rgba = rng.integers(0, 256, size=(size, size, 4), dtype=np.uint8)
            Image.fromarray(rgba, "RGBA").save(os.path.join(root, split, f"r_{k}.png"))
            pix[(split, k)] = rgba
            frames.append({"file_path": f"./{split}/r_{k}", "transform_matrix": D.pose_spherical(30.0 * k, -30.0, 4.0).tolist()})
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as fh:
            json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, fh)
    return pix


def test_blender_reader_on_a_synthetic_scene(tmp_path):
    D = _datasets()
    pix = write_blender_scene(str(tmp_path), n=(3, 4, 4), size=8)
    images, poses, render_poses, (H, W, focal), (i_train, i_val, i_test) = D.load_blender_data(str(tmp_path), False, 2)
    assert images.shape == (3 + 2 + 2, 8, 8, 4) and images.dtype == np.float32 and poses.shape == (7, 4, 4)
    assert list(i_train) == [0, 1, 2] and list(i_val) == [3, 4] and list(i_test) == [5, 6]
    np.testing.assert_array_equal(images[1], (pix[("train", 1)] / 255.).astype(np.float32))
    np.testing.assert_array_equal(images[4], (pix[("val", 2)] / 255.).astype(np.float32))      # testskip = 2
    assert render_poses.shape == (40, 4, 4) and (H, W) == (8, 8)
    assert focal == .5 * 8 / np.tan(.5 * 0.6911112070083618)
    half, _, _, (H2, W2, f2), _ = D.load_blender_data(str(tmp_path), True, 2)
    assert half.shape == (7, 4, 4, 4) and (H2, W2) == (4, 4) and f2 == focal / 2.
    np.testing.assert_allclose(half[0, 0, 0], images[0, :2, :2].astype(np.float64).mean((0, 1)), rtol=1e-12)


def write_llff_scene(root, n=4, h=12, w=16, seed=5):
    """poses_bounds.npy + images/ + images_2/ of a synthetic forward-facing capture."""
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from gen_golden_datasets import synthetic_llff
    poses, bds, _ = synthetic_llff(seed, n=n, h=h, w=w)
    arr = np.concatenate([poses.transpose(2, 0, 1).reshape(n, 15), bds.T], 1)
    os.makedirs(root, exist_ok=True)
    np.save(os.path.join(root, "poses_bounds.npy"), arr)
    rng = np.random.Generator(np.random.PCG64(9))
    for sub, (hh, ww) in (("images", (h, w)), ("images_2", (h // 2, w // 2))):
        os.makedirs(os.path.join(root, sub))
        for k in range(n):
            Image.fromarray(rng.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8), "RGB").save(os.path.join(root, sub, f"{k:02d}.png"))


def test_llff_reader_on_a_synthetic_scene(tmp_path):
    D = _datasets()
    write_llff_scene(str(tmp_path))
    images, p, b, render_poses, i_test = D.load_llff_data(str(tmp_path), factor=2)
    assert images.shape == (4, 6, 8, 3) and p.shape == (4, 3, 5) and b.shape == (4, 2) and render_poses.shape == (120, 3, 5)
    assert p[0, 0, 4] == 6 and p[0, 1, 4] == 8 and abs(p[0, 2, 4] - 11.5 / 2) < 1e-6      # hwf follows the shrunk images
    # without a pre-shrunk directory the full-size images are reduced in memory
    images4, p4, _, _, _ = D.load_llff_data(str(tmp_path), factor=4)
    assert images4.shape == (4, 3, 4, 3) and p4[0, 0, 4] == 3
