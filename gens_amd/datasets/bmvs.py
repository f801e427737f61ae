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
        h, w = self.img_hw
        imgs, intrs, w2cs, near_fars, depths, masks = [], [], [], [], [], []
        for vid in view_ids:
            imgs.append(self.read_img(os.path.join(self.data_dir, scan, "blended_images", "%08d_masked.jpg" % vid)) / 256.0)
            intr, w2c, near_far = self.read_cam(os.path.join(self.data_dir, scan, "cams", "%08d_cam.txt" % vid))
            intrs.append(intr)
            w2cs.append(w2c)
            near_fars.append(near_far)
            depth, mask = self.read_depth_and_mask(os.path.join(self.data_dir, scan, "rendered_depth_maps", "%08d.pfm" % vid), near_far[0])
            depths.append(depth)
            masks.append(mask)
        w2c_ref_inv = np.linalg.inv(w2cs[0])
        w2cs = [w2c @ w2c_ref_inv for w2c in w2cs]                       # every pose relative to the reference camera
        scale_mat, scale_factor = self.get_scale_mat(self.img_hw, intrs, w2cs, near_fars, factor=self.factor)
        c2ws, new_near_fars, new_intrs, new_depths = [], [], [], []
        for intr, w2c, depth in zip(intrs, w2cs, depths):                # cameras of the unit-sphere-normalised scene
            new_intr, c2w = C.load_K_Rt_from_P(None, (intr @ w2c @ scale_mat)[:3, :4])
            c2ws.append(c2w)
            new_intrs.append(new_intr)
            dist = np.sqrt(np.sum(c2w[:3, 3] ** 2)).astype(np.float32)
            new_near_fars.append([0.95 * (dist - 1), 1.05 * (dist + 1)])
            new_depths.append(depth * scale_factor)
        depths = torch.from_numpy(np.stack(new_depths).astype(np.float32))
        masks = torch.from_numpy(np.stack(masks).astype(np.float32))
        imgs = torch.from_numpy(np.stack(imgs).astype(np.float32))
        intrs = torch.from_numpy(np.stack(new_intrs).astype(np.float32))
        c2ws = torch.from_numpy(np.stack(c2ws).astype(np.float32))
        near_fars = torch.from_numpy(np.stack(new_near_fars).astype(np.float32))
        outputs = {"imgs": imgs.permute(0, 3, 1, 2).contiguous(), "intrs": intrs, "c2ws": c2ws,
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
                            "scene": scan, "file_name": scan + "_view" + str(ref_view), "hw": torch.Tensor([h // lvl, w // lvl]).int(), "masks": masks})
            pixels_y, pixels_x = torch.meshgrid(torch.linspace(0, h - 1, h // lvl), torch.linspace(0, w - 1, w // lvl), indexing="ij")
            pixels_x, pixels_y = pixels_x.reshape(-1), pixels_y.reshape(-1)

        at = (pixels_y.long(), pixels_x.long())
        p = torch.stack([pixels_x, pixels_y, torch.ones_like(pixels_y)], dim=-1).float()
        p = torch.matmul(intrs.inverse()[0, None, :3, :3], p[:, :, None]).squeeze()
        rays_d = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
        rays_d = torch.matmul(c2ws[0, None, :3, :3], rays_d[:, :, None]).squeeze()
        rays_o = c2ws[0, None, :3, 3].expand(rays_d.shape)
        near, far = near_fars[0].reshape(1, 2).split(split_size=1, dim=1)
        outputs.update({"pixels_x": pixels_x, "pixels_y": pixels_y, "near_fars": near_fars, "rays_o": rays_o, "rays_d": rays_d, "near": near, "far": far,
                        "color": imgs[0][at], "depth": depths[0][at], "mask": masks[0][at], "masks": masks, "depth_ref": depths[0], "src_idx": 1})
        return outputs

    def __len__(self):
        return len(self.metas)
