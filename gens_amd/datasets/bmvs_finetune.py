"""BlendedMVS per-scene fine-tune front-end without OpenCV: the reference's BMVSDatasetFinetune
(/root/reference/datasets/bmvs_finetune.py:74-309).  It is the DTU fine-tune dataset with the BlendedMVS file layout
({scene}/cams, {scene}/blended_images), 768 x 576 intrinsics, the object mask taken from the masked JPEG (any channel > 0),
no pseudo-depth points, and `masks` in the validation item."""
import os

import numpy as np

from .dtu_finetune import DTUDatasetFinetune


class BMVSDatasetFinetune(DTUDatasetFinetune):
    RAW_WH = (768, 576)                          # bmvs_finetune.py:147-148
    HAS_PSEUDO_POINTS = False

    def pair_file(self):
        return os.path.join(self.data_dir, f"{self.scene}/cams/pair.txt")

    def cam_file(self, vid):
        return os.path.join(self.data_dir, self.scene, "cams/{:0>8}_cam.txt".format(vid))

    def image_file(self, vid):
        return os.path.join(self.data_dir, self.scene, "blended_images/{:0>8}.jpg".format(vid))

    def mask_file(self, vid):
        return os.path.join(self.data_dir, self.scene, "blended_images/{:0>8}_masked.jpg".format(vid))

    def mask_from_pixels(self, m):
        return (np.mean(m, axis=-1) > 0).astype(np.float32)            # bmvs_finetune.py:106

    def get_rays_at(self, vid):
        out = super().get_rays_at(vid)
        out["masks"] = self.masks[out["view_ids"]]                      # bmvs_finetune.py:287,304
        return out
