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
    def __getitem__(self, idx):
        scan, light_idx, ref_view = self.metas[idx]
        pairs = list(self.pairs[ref_view])
        if self.mode == "train":
            src_views = random.sample(pairs[:6], min(self.num_src_view, len(pairs)))
        else:
            src_views = pairs[:min(self.num_src_view, len(pairs))]
        view_ids = [ref_view] + src_views
        src_idx = np.random.randint(1, len(view_ids))
        h, w = self.img_hw
        w2c_ref_inv = np.linalg.inv(self.w2cs[ref_view])

        imgs, intrs, w2cs, near_fars, masks, depths = [], [], [], [], [], []
        for i, vid in enumerate(view_ids):
            tag = "r7000" if vid > 48 else "r5000"
            img_file = os.path.join(self.data_dir, "Rectified_raw/{}/rect_{:0>3}_{}_{}.png".format(scan, vid + 1, light_idx, tag))
            mask_file = os.path.join(self.data_dir, "Depths_raw/{}/depth_visual_{:0>4}.png".format(scan, vid))
            depth_file = os.path.join(self.data_dir, "Depths_raw/{}/depth_map_{:0>4}.pfm".format(scan, vid))
            mask = (self.read_img(mask_file) > 10).astype(np.float32)
            if mask.ndim > 2:
                mask = (np.mean(mask, axis=-1) > 0).astype(np.float32)
            imgs.append(self.read_img(img_file) / 256.0)
            intrs.append(self.intrs[vid])
            w2cs.append(self.w2cs[vid] @ w2c_ref_inv)                     # every pose relative to the reference camera
            near_fars.append(self.near_fars[vid])
            masks.append(mask)
            depths.append(self.read_depth(depth_file))
            if i == 0:
                if self.mode == "train":
                    pseudo_file = os.path.join(self.data_dir, "pseudo_depths/{}/{}_epoch0.npy".format(scan, vid))
                    ref_pseudo_depth = self.read_numpy(pseudo_file) / self.pseudo_scale
                else:
                    ref_pseudo_depth = masks[0]

        scale_mat, scale_factor = self.get_scale_mat(self.img_hw, intrs, w2cs, near_fars, factor=self.factor)
        c2ws, new_near_fars, new_intrs, new_depths = [], [], [], []
        for intr, w2c, depth in zip(intrs, w2cs, depths):                # cameras of the unit-sphere-normalised scene
            new_intr, c2w = C.load_K_Rt_from_P(None, (intr @ w2c @ scale_mat)[:3, :4])
            c2ws.append(c2w)
            new_intrs.append(new_intr)
            dist = np.sqrt(np.sum(c2w[:3, 3] ** 2)).astype(np.float32)
            new_near_fars.append([0.95 * (dist - 1), 1.05 * (dist + 1)])
            new_depths.append(scale_factor * depth)
        ref_pseudo_depth = torch.from_numpy((ref_pseudo_depth * scale_factor).astype(np.float32))

        imgs = torch.from_numpy(np.stack(imgs).astype(np.float32))
        intrs = torch.from_numpy(np.stack(new_intrs).astype(np.float32))
        c2ws = torch.from_numpy(np.stack(c2ws).astype(np.float32))
        near_fars = torch.from_numpy(np.stack(new_near_fars).astype(np.float32))
        masks = torch.from_numpy(np.stack(masks).astype(np.float32))
        depths = torch.from_numpy(np.stack(new_depths).astype(np.float32))
        outputs = {"imgs": imgs.permute(0, 3, 1, 2).contiguous(), "intrs": intrs, "c2ws": c2ws, "masks": masks,
                   "scale_mat": torch.from_numpy(w2c_ref_inv @ scale_mat), "view_ids": torch.from_numpy(np.array(view_ids)).long()}

        ys, xs = torch.meshgrid(torch.linspace(0, h - 1, h), torch.linspace(0, w - 1, w), indexing="ij")
        pixel_all = torch.stack([xs, ys], dim=-1)
        if self.mode == "train":
            assert self.n_rays > 0, "No sampling rays!"
            n = self.n_rays
            p_valid = pixel_all[masks[0] > 0.5]                          # three quarters of the rays inside the object mask
            pixels_x_i = torch.randint(low=0, high=w, size=[n // 4])
            pixels_y_i = torch.randint(low=0, high=h, size=[n // 4])
            p_select = p_valid[torch.randint(low=0, high=p_valid.shape[0], size=[n - n // 4])]
            pixels_x = torch.cat([p_select[:, 0], pixels_x_i], dim=0)
            pixels_y = torch.cat([p_select[:, 1], pixels_y_i], dim=0)
        else:
            lvl = self.val_res_level
            outputs.update({"bound_min": torch.tensor([-1, -1, -1], dtype=torch.float32), "bound_max": torch.tensor([1, 1, 1], dtype=torch.float32),
                            "scene": scan, "file_name": scan + "_view" + str(ref_view) + "_light" + str(light_idx),
                            "hw": torch.Tensor([h // lvl, w // lvl]).int()})
            pixels_y, pixels_x = torch.meshgrid(torch.linspace(0, h - 1, h // lvl), torch.linspace(0, w - 1, w // lvl), indexing="ij")
            pixels_x, pixels_y = pixels_x.reshape(-1), pixels_y.reshape(-1)

        at = (pixels_y.long(), pixels_x.long())
        p = torch.stack([pixels_x, pixels_y, torch.ones_like(pixels_y)], dim=-1).float()
        p = torch.matmul(intrs.inverse()[0, None, :3, :3], p[:, :, None]).squeeze()
        rays_d = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
        rays_d = torch.matmul(c2ws[0, None, :3, :3], rays_d[:, :, None]).squeeze()
        rays_o = c2ws[0, None, :3, 3].expand(rays_d.shape)
        near, far = near_fars[0].reshape(1, 2).split(split_size=1, dim=1)

        p_mask = (ref_pseudo_depth > 0) & (masks[0] > 0)
        if self.mode == "train" and p_mask.sum() > 100:                   # 2048 points un-projected from the pseudo depth (dtu.py:406-419)
            y, x = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
            x, y = x[p_mask].type_as(intrs), y[p_mask].type_as(intrs)
            p_depth = ref_pseudo_depth[p_mask]
            pick = torch.randint(low=0, high=x.shape[0], size=[2048])
            x, y, p_depth = x[pick], y[pick], p_depth[pick]
            xyz_ref = torch.matmul(intrs.inverse()[0, :3, :3], torch.stack((x, y, torch.ones_like(x)), dim=0) * p_depth.unsqueeze(0))
            xyz_world = torch.matmul(c2ws[0], torch.cat((xyz_ref, torch.ones_like(x).unsqueeze(0)), dim=0))[:3]
            outputs["pseudo_pts"] = xyz_world.permute(1, 0)

        outputs.update({"rays_o": rays_o, "rays_d": rays_d, "near": near, "far": far, "color": imgs[0][at], "depth": depths[0][at],
                        "pseudo_depth": ref_pseudo_depth[at], "depth_ref": depths[0], "mask": masks[0][at], "mask_ref": masks[0],
                        "pseudo_depth_ref": ref_pseudo_depth, "src_idx": src_idx})
        return outputs

    def __len__(self):
        return len(self.metas)
