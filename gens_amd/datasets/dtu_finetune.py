"""Per-scene fine-tune front-end (BASELINE config 5) without OpenCV: the reference's DTUDatasetFinetune
(/root/reference/datasets/dtu_finetune.py:74-345) -- all views of one scene resident, `get_all_images` for
`GenS.init_volumes` (runner.py:91), `get_random_rays(vid)` per step (runner.py:296) and `get_rays_at(vid)` for validation
(runner.py:346).  Same constructor keys, outputs and random-draw order."""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from . import camera as C


class DTUDatasetFinetune(Dataset):
    RAW_WH = (1600, 1200)                       # pixel size the intrinsics of the cam files refer to (dtu_finetune.py:171-172)
    HAS_PSEUDO_POINTS = True

    # where one scene's files live (overridden by the BlendedMVS variant)
    def pair_file(self):
        return os.path.join(self.data_dir, "Cameras/pair.txt")

    def cam_file(self, vid):
        return os.path.join(self.data_dir, "Cameras/{:0>8}_cam.txt".format(vid))

    def image_file(self, vid):
        return os.path.join(self.data_dir, "Rectified_raw/{}/rect_{:0>3}_3_r5000.png".format(self.scene, vid + 1))

    def mask_file(self, vid):
        return os.path.join(self.data_dir, "Depths_raw/{}/depth_visual_{:0>4}.png".format(self.scene, vid))

    def mask_from_pixels(self, m):
        return (m > 10).astype(np.float32)

    def __init__(self, confs, mode):
        super().__init__()
        self.mode = mode
        self.data_dir = confs["data_dir"]
        self.interval_scale = confs.get_float("interval_scale")
        self.num_interval = confs.get_int("num_interval")
        self.img_hw = confs["img_hw"]
        self.n_rays = confs.get_int("n_rays")
        self.factor = confs.get_float("factor")
        self.num_views = confs.get_int("num_views")
        self.scene = confs.get_string("scene")
        self.ref_view = confs.get_int("ref_view")
        self.val_res_level = confs.get_int("val_res_level", default=1)
        self.pairs = C.read_pair_file(self.pair_file())
        self.all_views = [self.ref_view] + list(self.pairs[self.ref_view])[:(self.num_views - 1)]
        intrs, c2ws, near_fars, self.scale_factor, self.trans_mat, scale_mat = self.read_cam_info()
        self.intrs = torch.from_numpy(np.stack(intrs).astype(np.float32))
        self.c2ws = torch.from_numpy(np.stack(c2ws).astype(np.float32))
        self.near_fars = torch.from_numpy(np.stack(near_fars).astype(np.float32))
        self.scale_mat = torch.from_numpy(self.trans_mat @ scale_mat)
        hw = self.img_hw
        self.images_lis = [self.image_file(v) for v in self.all_views]
        self.masks_lis = [self.mask_file(v) for v in self.all_views]
        images = [np.array(Image.open(f), dtype=np.float32) / 256.0 for f in self.images_lis]
        self.images = torch.from_numpy(np.stack([C.resize_nearest(im, hw) for im in images]).astype(np.float32))
        masks = [np.array(Image.open(f), dtype=np.float32) for f in self.masks_lis]
        self.masks = torch.from_numpy(np.stack([self.mask_from_pixels(C.resize_nearest(m, hw)) for m in masks]).astype(np.float32))
        if self.HAS_PSEUDO_POINTS:
            self._load_pseudo_points()

    def _load_pseudo_points(self):
        hw = self.img_hw
        self.pseudo_scale = 0.0037506045743823813                      # dtu_finetune.py:99
        self.pseudo_dense_lis = [os.path.join(self.data_dir, "pseudo_depths/{}/{}_epoch0.npy".format(self.scene, v)) for v in self.all_views]
        dense = np.stack([np.load(f).astype(np.float32) / self.pseudo_scale for f in self.pseudo_dense_lis])   # kept at file resolution
        self.dense_pseudo_depths = torch.from_numpy(dense.astype(np.float32)) * self.scale_factor
        pts = []
        for i in range(self.num_views):                                 # every view's pseudo depth un-projected once (:116-131)
            depth = self.dense_pseudo_depths[i]
            keep = depth > 0
            d_h, d_w = depth.shape
            y, x = torch.meshgrid(torch.arange(0, d_h), torch.arange(0, d_w), indexing="ij")
            x, y, p_depth = x[keep], y[keep], depth[keep]
            intr = self.intrs[i].clone()
            intr[0] *= d_w / hw[1]
            intr[1] *= d_h / hw[0]
            xyz_ref = torch.matmul(intr.inverse()[:3, :3], torch.stack((x, y, torch.ones_like(x)), dim=0) * p_depth.unsqueeze(0))
            xyz_world = torch.matmul(self.c2ws[i], torch.cat((xyz_ref, torch.ones_like(x).unsqueeze(0)), dim=0))[:3]
            pts.append(xyz_world.permute(1, 0))
        self.pseudo_ptses = torch.cat(pts, dim=0)

    def read_cam_info(self):
        """Cameras of `all_views` relative to the reference view, normalised to the unit sphere (:149-199)."""
        intrs, w2cs, near_fars = [], [], []
        for vid in self.all_views:
            intr, w2c, near_far = C.read_cam_file(self.cam_file(vid), self.interval_scale, self.num_interval)
            intr[0] *= self.img_hw[1] / self.RAW_WH[0]
            intr[1] *= self.img_hw[0] / self.RAW_WH[1]
            intrs.append(intr)
            w2cs.append(w2c)
            near_fars.append(near_far)
        w2c_ref_inv = np.linalg.inv(w2cs[0])
        new_w2cs = [w2c @ w2c_ref_inv for w2c in w2cs]
        scale_mat, scale_factor = C.get_scale_mat(self.img_hw, intrs, new_w2cs, near_fars, factor=self.factor)
        c2ws, new_near_fars, new_intrs = [], [], []
        for intr, w2c in zip(intrs, new_w2cs):
            new_intr, c2w = C.load_K_Rt_from_P(None, (intr @ w2c @ scale_mat)[:3, :4])
            c2ws.append(c2w)
            new_intrs.append(new_intr)
            dist = np.sqrt(np.sum(c2w[:3, 3] ** 2)).astype(np.float32)
            new_near_fars.append([0.95 * (dist - 1), 1.05 * (dist + 1)])
        return new_intrs, c2ws, new_near_fars, scale_factor, w2c_ref_inv, scale_mat

    def get_all_images(self):
        return {"imgs": self.images.permute(0, 3, 1, 2), "c2ws": self.c2ws, "intrs": self.intrs}

    def _rays(self, vid, pixels_x, pixels_y):
        p = torch.stack([pixels_x, pixels_y, torch.ones_like(pixels_y)], dim=-1).float()
        p = torch.matmul(self.intrs[vid].inverse()[None, :3, :3], p[:, :, None]).squeeze()
        rays_d = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
        rays_d = torch.matmul(self.c2ws[vid, None, :3, :3], rays_d[:, :, None]).squeeze()
        rays_o = self.c2ws[vid, None, :3, 3].expand(rays_d.shape)
        near, far = self.near_fars[vid].reshape(1, 2).split(split_size=1, dim=1)
        view_ids = [vid] + list(range(self.num_views))[:vid] + list(range(self.num_views))[vid + 1:]
        out = {"rays_o": rays_o, "rays_d": rays_d, "near": near, "far": far, "color": self.images[vid][(pixels_y.long(), pixels_x.long())],
               "intrs": self.intrs[view_ids], "c2ws": self.c2ws[view_ids], "view_ids": view_ids,
               "imgs": self.images[view_ids].permute(0, 3, 1, 2)}
        return out

    def get_random_rays(self, vid):
        vid = vid.item()
        pixels_x = torch.randint(low=0, high=self.img_hw[1], size=[self.n_rays])
        pixels_y = torch.randint(low=0, high=self.img_hw[0], size=[self.n_rays])
        out = self._rays(vid, pixels_x, pixels_y)
        if self.HAS_PSEUDO_POINTS:
            out["pseudo_pts"] = self.pseudo_ptses[torch.randint(low=0, high=self.pseudo_ptses.shape[0], size=[2048])]
        return out

    def get_rays_at(self, vid):
        lvl = self.val_res_level
        h, w = self.img_hw
        pixels_y, pixels_x = torch.meshgrid(torch.linspace(0, h - 1, h // lvl), torch.linspace(0, w - 1, w // lvl), indexing="ij")
        out = self._rays(vid, pixels_x.reshape(-1), pixels_y.reshape(-1))
        out.update({"scale_mat": self.scale_mat, "scene": self.scene, "bound_min": torch.tensor([-1, -1, -1], dtype=torch.float32),
                    "bound_max": torch.tensor([1, 1, 1], dtype=torch.float32), "hw": torch.Tensor([h // lvl, w // lvl]).int()})
        return out
