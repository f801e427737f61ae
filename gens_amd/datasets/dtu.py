"""DTU front-end without OpenCV: same constructor, item dictionary and random-draw order as the reference's DTUDataset
(/root/reference/datasets/dtu.py:74-439), so `runner.py` can iterate it unchanged.  The file formats it reads:

    Cameras/{view:08d}_cam.txt     MVSNet camera file: 4x4 world2cam, 3x3 intrinsics (1600x1200 pixels), depth_min / interval
    Cameras/pair.txt               per reference view, its ten best source views
    Rectified_raw/{scan}/rect_{view+1:03d}_{light}_r5000.png      image   (r7000 for view > 48)
    Depths_raw/{scan}/depth_visual_{view:04d}.png                 object mask (> 10)
    Depths_raw/{scan}/depth_map_{view:04d}.pfm                    depth
    pseudo_depths/{scan}/{view}_epoch0.npy                        pseudo depth of the reference view (train only)
"""
import os
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from . import camera as C


class DTUDataset(Dataset):
    def __init__(self, confs, mode):
        super().__init__()
        self.mode = mode
        self.data_dir = confs["data_dir"]
        self.num_src_view = confs.get_int("num_src_view")
        self.interval_scale = confs.get_float("interval_scale")
        self.num_interval = confs.get_int("num_interval")
        self.img_hw = confs["img_hw"]
        self.n_rays = confs.get_int("n_rays", 0)
        self.factor = confs.get_float("factor")
        self.total_views = 49
        self.split = confs.get_string("split", default=None)
        self.scene = confs.get_list("scene", default=None)
        self.light_idx = confs.get_list("light_idx", default=None)
        self.ref_view = confs.get_list("ref_view", default=None)
        if mode == "val":
            self.val_res_level = confs.get_int("val_res_level", default=1)
        self.intrs, self.w2cs, self.near_fars = self.read_cam_info()
        self.pairs = self.get_pairs()
        self.metas = self.build_list()
        self.pseudo_scale = 0.0037506045743823813                     # dtu.py:99

    # ------------------------------------------------------------------------------------------------ index (dtu.py:101-160)
    def get_pairs(self, num_select=10):
        pair_file = os.path.join(self.data_dir, "Cameras/pair.txt")
        if os.path.exists(pair_file):
            return C.read_pair_file(pair_file, num_select)
        return C.pairs_from_poses(self.w2cs, num_select)

    def build_list(self):
        if self.scene is not None:
            scans = self.scene
        elif self.split is not None:
            with open(self.split) as f:
                scans = [line.rstrip() for line in f.readlines()]
        else:
            raise ValueError("There are no scenes!")
        light_idxs = range(7) if self.light_idx is None else self.light_idx
        ref_views = list(range(self.total_views)) if self.ref_view is None else self.ref_view
        return [(scan, light, ref) for scan in scans for ref in ref_views for light in light_idxs]

    # ------------------------------------------------------------------------------------------------ cameras (dtu.py:162-191)
    def read_cam_info(self):
        intrs, w2cs, near_fars = [], [], []
        for vid in range(self.total_views):
            intr, w2c, near_far = C.read_cam_file(os.path.join(self.data_dir, "Cameras/{:0>8}_cam.txt".format(vid)), self.interval_scale,
                                                  self.num_interval)
            intr[0] *= self.img_hw[1] / 1600
            intr[1] *= self.img_hw[0] / 1200
            intrs.append(intr)
            w2cs.append(w2c)
            near_fars.append(near_far)
        return intrs, w2cs, near_fars

    def get_scale_mat(self, img_hw, intrs, w2cs, near_fars, factor=0.8):
        return C.get_scale_mat(img_hw, intrs, w2cs, near_fars, factor)

    # ------------------------------------------------------------------------------------------------ pixels (dtu.py:239-271)
    def read_img(self, filename):
        return C.resize_nearest(np.array(Image.open(filename), dtype=np.float32), self.img_hw)

    def read_numpy(self, filename):
        return C.resize_nearest(np.load(filename).astype(np.float32), self.img_hw)

    def read_depth(self, filename):
        return C.resize_nearest(np.array(C.read_pfm(filename)[0], dtype=np.float32), self.img_hw)

    # ------------------------------------------------------------------------------------------------ one item (dtu.py:273-436)
    def _files(self, scan, vid, light_idx):
        tag = "r7000" if vid > 48 else "r5000"
        return (os.path.join(self.data_dir, "Rectified_raw/{}/rect_{:0>3}_{}_{}.png".format(scan, vid + 1, light_idx, tag)),
                os.path.join(self.data_dir, "Depths_raw/{}/depth_visual_{:0>4}.png".format(scan, vid)),
                os.path.join(self.data_dir, "Depths_raw/{}/depth_map_{:0>4}.pfm".format(scan, vid)))

    def _object_mask(self, filename):
        m = (self.read_img(filename) > 10).astype(np.float32)
        return m if m.ndim == 2 else (np.mean(m, axis=-1) > 0).astype(np.float32)

    def __getitem__(self, idx):
        scan, light_idx, ref_view = self.metas[idx]
        train = self.mode == "train"
        # host RNG draws, in the reference's order: source views (python `random`), src_idx (numpy), then torch's generator below
        candidates = list(self.pairs[ref_view])
        keep = min(self.num_src_view, len(candidates))
        view_ids = [ref_view] + (random.sample(candidates[:6], keep) if train else candidates[:keep])
        src_idx = np.random.randint(1, len(view_ids))

        files = [self._files(scan, v, light_idx) for v in view_ids]
        imgs = torch.from_numpy(np.stack([self.read_img(f[0]) / 256.0 for f in files]).astype(np.float32))
        masks = torch.from_numpy(np.stack([self._object_mask(f[1]) for f in files]).astype(np.float32))
        views = C.NormalisedViews([self.intrs[v] for v in view_ids], [self.w2cs[v] for v in view_ids], [self.near_fars[v] for v in view_ids],
                                  self.img_hw, self.factor)
        depths = views.scaled([self.read_depth(f[2]) for f in files])
        if train:                                                          # pseudo depth of the reference view (else: its mask stands in)
            pseudo = self.read_numpy(os.path.join(self.data_dir, "pseudo_depths/{}/{}_epoch0.npy".format(scan, ref_view))) / self.pseudo_scale
        else:
            pseudo = masks[0].numpy()
        pseudo = views.scaled([pseudo])[0]

        h, w = self.img_hw
        item = {"imgs": imgs.permute(0, 3, 1, 2).contiguous(), "intrs": views.intrs, "c2ws": views.c2ws, "masks": masks,
                "scale_mat": views.scale_mat, "view_ids": torch.from_numpy(np.array(view_ids)).long()}
        if train:
            assert self.n_rays > 0, "No sampling rays!"
            px, py = C.sample_train_pixels(masks[0], self.n_rays)
        else:
            lvl = self.val_res_level
            px, py = C.lattice_pixels(h, w, lvl)
            item.update(bound_min=torch.tensor([-1, -1, -1], dtype=torch.float32), bound_max=torch.tensor([1, 1, 1], dtype=torch.float32),
                        scene=scan, file_name=scan + "_view" + str(ref_view) + "_light" + str(light_idx), hw=torch.Tensor([h // lvl, w // lvl]).int())
        rays_o, rays_d = C.rays_from_pixels(views.intrs[0], views.c2ws[0], px, py)
        near, far = views.near_fars[0].reshape(1, 2).split(split_size=1, dim=1)
        if train:
            pts = C.unproject_pseudo_points(pseudo, masks[0], views.intrs[0], views.c2ws[0])
            if pts is not None:
                item["pseudo_pts"] = pts
        at = (py.long(), px.long())
        item.update(rays_o=rays_o, rays_d=rays_d, near=near, far=far, color=imgs[0][at], depth=depths[0][at], pseudo_depth=pseudo[at],
                    depth_ref=depths[0], mask=masks[0][at], mask_ref=masks[0], pseudo_depth_ref=pseudo, src_idx=src_idx)
        return item

    def __len__(self):
        return len(self.metas)
