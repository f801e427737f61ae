"""A small synthetic BlendedMVS tree in the on-disk formats the reference's datasets/bmvs.py and bmvs_finetune.py read (see
gens_amd/datasets/bmvs.py for the list).  Deterministic: the golden generator and the tests build the same tree from the same seed."""
import os

import numpy as np
from PIL import Image

from dtu_fixture import _look_at, write_pfm

RAW_HW = (72, 96)            # stands in for the 576 x 768 originals (the intrinsics in the cam files are for 768 x 576)
SCENE = "5a0271884e62597cdee0d0eb"


def make_bmvs_tree(root, seed=17, n_views=12):
    rng = np.random.default_rng(seed)
    for d in ("cams", "blended_images", "rendered_depth_maps"):
        os.makedirs(os.path.join(root, SCENE, d), exist_ok=True)
    centres = []
    for v in range(n_views):
        az, el = 2 * np.pi * v / n_views, 0.35 + 0.05 * rng.standard_normal()
        c = 3.0 * np.array([np.sin(az) * np.cos(el), np.sin(el), -np.cos(az) * np.cos(el)]) + rng.standard_normal(3) * 0.05
        centres.append(c)
        w2c = _look_at(c).astype(np.float32)
        k = np.array([[577.0 + rng.uniform(-3, 3), 0.0, 384.0 + rng.uniform(-2, 2)], [0.0, 578.0 + rng.uniform(-3, 3), 288.0], [0.0, 0.0, 1.0]])
        with open(os.path.join(root, SCENE, "cams", "{:0>8}_cam.txt".format(v)), "w") as f:
            f.write("extrinsic\n")
            for row in w2c:
                f.write(" ".join(repr(float(x)) for x in row) + " \n")
            f.write("\nintrinsic\n")
            for row in k:
                f.write(" ".join(repr(float(x)) for x in row) + " \n")
            f.write("\n1.9 0.0125 128 3.5 \n")
    centres = np.array(centres)
    d = np.linalg.norm(centres[:, None] - centres[None], axis=-1)
    d[np.eye(n_views) > 0] = 1e9
    with open(os.path.join(root, SCENE, "cams", "pair.txt"), "w") as f:
        f.write(f"{n_views}\n")
        for v in range(n_views):
            order = np.argsort(d[v])[:10]
            f.write(f"{v}\n10 " + " ".join(f"{int(s)} {1000.0 / (1 + d[v, s]):.4f}" for s in order) + " \n")
    h, w = RAW_HW
    yy, xx = np.mgrid[0:h, 0:w]
    for v in range(n_views):
        # smooth images: JPEG keeps them close to what was written, and the masked copy is exactly 0 outside the object
        img = np.stack([127 + 100 * np.sin(xx / (5.0 + c) + v) * np.cos(yy / (7.0 - c)) for c in range(3)], -1)
        blob = (((xx - w / 2 - 6 * np.sin(v)) / (0.36 * w)) ** 2 + ((yy - h / 2) / (0.42 * h)) ** 2) < 1.0
        Image.fromarray(img.clip(0, 255).astype(np.uint8)).save(os.path.join(root, SCENE, "blended_images", "{:0>8}.jpg".format(v)), quality=95)
        Image.fromarray((img * blob[..., None]).clip(0, 255).astype(np.uint8)).save(os.path.join(root, SCENE, "blended_images", "{:0>8}_masked.jpg".format(v)),
                                                                                   quality=95)
        depth = (2.6 + 0.4 * np.sin(xx / 13.0 + v) + 0.3 * np.cos(yy / 9.0)).astype(np.float32) * blob
        write_pfm(os.path.join(root, SCENE, "rendered_depth_maps", "{:0>8}.pfm".format(v)), depth)
    return root


def conf_values(root, mode):
    """The dataset section of confs/gens_bmvs.conf (data_dir, sizes) for this tree, as a plain dict."""
    c = {"dataset_name": "BMVSDataset", "data_dir": root, "num_src_view": 3, "interval_scale": 1, "num_interval": 192, "img_hw": [36, 48],
         "factor": 1.0, "scene": [SCENE], "ref_view": [2, 7]}
    if mode == "train":
        c["n_rays"] = 64
    else:
        c["val_res_level"] = 2
    return c


def finetune_conf_values(root):
    """The finetune_dataset section of confs/gens_bmvs_finetune.conf for this tree."""
    return {"dataset_name": "BMVSDatasetFinetune", "data_dir": root, "interval_scale": 1, "num_interval": 192, "img_hw": [36, 48], "n_rays": 48,
            "factor": 1.0, "num_views": 4, "scene": SCENE, "ref_view": 7, "val_res_level": 2}
