"""BlendedMVS front-end without OpenCV: same constructor, item dictionary and random-draw order as the reference's BMVSDataset
(/root/reference/datasets/bmvs.py:72-340).  Files per scene:

    {scene}/cams/pair.txt                        per reference view, its source views (all of them are kept, bmvs.py:117-121)
    {scene}/cams/{view:08d}_cam.txt              MVSNet camera file (intrinsics for 768 x 576 pixels)
    {scene}/blended_images/{view:08d}_masked.jpg image with the background masked out (the network input, bmvs.py:210)
    {scene}/rendered_depth_maps/{view:08d}.pfm   depth; the object mask is depth >= depth_min (bmvs.py:157-158)
"""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from . import camera as C


class BMVSDataset(Dataset):
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
        self.split = confs.get_string("split", default=None)
        self.scene = confs.get_list("scene", default=None)
        self.ref_view = confs.get_list("ref_view", default=None)
        self.src_views = confs.get_list("src_views", default=None)
        if mode == "val":
            self.val_res_level = confs.get_int("val_res_level", default=1)
        if self.scene is None:
            if self.split is None:
                raise ValueError("There are no scenes!")
            with open(self.split) as f:
                self.scene = [line.rstrip() for line in f.readlines()]
        self.metas = self.build_list()

    def build_list(self):
        """(scene, reference view, its source views) per item (bmvs.py:102-124)."""
        metas = []
        for scene in self.scene:
            with open(os.path.join(self.data_dir, scene, "cams", "pair.txt")) as f:
                lines = [line.rstrip() for line in f.readlines()]
            refs = list(range(int(lines[0]))) if self.ref_view is None else self.ref_view
            for ref_view in refs:
                if self.src_views is not None:
                    src_views = self.src_views
                else:
                    src_views = [int(x) for x in lines[2 * ref_view + 2].rstrip().split()[1::2]]
                metas.append((scene, ref_view, src_views))
        return metas

    def get_scale_mat(self, img_hw, intrs, w2cs, near_fars, factor=0.8):
        return C.get_scale_mat(img_hw, intrs, w2cs, near_fars, factor)

    def read_cam(self, filename):
        intr, w2c, near_far = C.read_cam_file(filename, self.interval_scale, self.num_interval)
        intr[0] *= self.img_hw[1] / 768                                  # bmvs.py:183-184
        intr[1] *= self.img_hw[0] / 576
        return intr, w2c, near_far

    def read_img(self, filename):
        return C.resize_nearest(np.array(Image.open(filename), dtype=np.float32), self.img_hw)

    def read_depth_and_mask(self, filename, depth_min):
        depth = np.array(C.read_pfm(filename)[0], dtype=np.float32)
        mask = np.array(depth >= depth_min, dtype=np.float32)
        return C.resize_nearest(depth, self.img_hw), C.resize_nearest(mask, self.img_hw)

    def __getitem__(self, idx):
        scan, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.num_src_view]
        scene_dir = os.path.join(self.data_dir, scan)
        cams = [self.read_cam(os.path.join(scene_dir, "cams", "%08d_cam.txt" % v)) for v in view_ids]               # (intr, w2c, near_far)
        imgs = torch.from_numpy(np.stack([self.read_img(os.path.join(scene_dir, "blended_images", "%08d_masked.jpg" % v)) / 256.0
                                          for v in view_ids]).astype(np.float32))
        depth_mask = [self.read_depth_and_mask(os.path.join(scene_dir, "rendered_depth_maps", "%08d.pfm" % v), cam[2][0])
                      for v, cam in zip(view_ids, cams)]
        views = C.NormalisedViews([c[0] for c in cams], [c[1] for c in cams], [c[2] for c in cams], self.img_hw, self.factor)
        depths = views.scaled([d for d, _ in depth_mask])
        masks = torch.from_numpy(np.stack([m for _, m in depth_mask]).astype(np.float32))

        h, w = self.img_hw
        item = {"imgs": imgs.permute(0, 3, 1, 2).contiguous(), "intrs": views.intrs, "c2ws": views.c2ws, "scale_mat": views.scale_mat,
                "view_ids": torch.from_numpy(np.array(view_ids)).long()}
        if self.mode == "train":
            assert self.n_rays > 0, "No sampling rays!"
            px, py = C.sample_train_pixels(masks[0], self.n_rays)
        else:
            lvl = self.val_res_level
            px, py = C.lattice_pixels(h, w, lvl)
            item.update(bound_min=torch.tensor([-1, -1, -1], dtype=torch.float32), bound_max=torch.tensor([1, 1, 1], dtype=torch.float32),
                        scene=scan, file_name=scan + "_view" + str(ref_view), hw=torch.Tensor([h // lvl, w // lvl]).int())
        rays_o, rays_d = C.rays_from_pixels(views.intrs[0], views.c2ws[0], px, py)
        near, far = views.near_fars[0].reshape(1, 2).split(split_size=1, dim=1)
        at = (py.long(), px.long())
        item.update(pixels_x=px, pixels_y=py, near_fars=views.near_fars, rays_o=rays_o, rays_d=rays_d, near=near, far=far, color=imgs[0][at],
                    depth=depths[0][at], mask=masks[0][at], masks=masks, depth_ref=depths[0], src_idx=1)
        return item

    def __len__(self):
        return len(self.metas)
