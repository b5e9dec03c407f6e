"""Writes a tiny synthetic blender scene + config under gpurun_out/tiny and prints the config path
(python3 tools/rehearse_train.py); used to rehearse `python -m torch.distributed.run -m nerf_amd --config ...`."""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from test_datasets import write_blender_scene
base = os.path.abspath(os.path.join(root, "gpurun_out", "tiny"))
os.makedirs(base, exist_ok=True)
scene = os.path.join(base, "scene")
write_blender_scene(scene, n=(4, 1, 3), size=16)
cfg = os.path.join(base, "tiny.txt")
with open(cfg, "w") as fh:
    fh.write("expname = tiny\nbasedir = %s\ndatadir = %s\ndataset_type = blender\nno_batching = True\nuse_viewdirs = True\n"
             "white_bkgd = True\nlrate_decay = 500\nN_samples = 8\nN_importance = 8\nN_rand = 64\n" % (os.path.join(base, "logs"), scene))
print(cfg)
