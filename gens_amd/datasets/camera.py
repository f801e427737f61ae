"""Camera / file-format helpers of the dataset front-end, without OpenCV.

Each function cites the reference code it restates; the reference imports cv2 for exactly two things on this path --
`cv2.decomposeProjectionMatrix` (datasets/dtu.py:20) and `cv2.resize(..., INTER_NEAREST)` (dtu.py:243,255,266) -- both
are restated here in numpy.
"""
import re

import numpy as np
import torch


def decompose_projection_matrix(P):
    """P (3,4) = K [R | -R c] up to scale -> (K (3,3) upper triangular with positive diagonal, R (3,3) rotation, c4 (4,1)
    homogeneous camera centre), i.e. the first three outputs of cv2.decomposeProjectionMatrix (which factors P[:, :3] with
    RQDecomp3x3 -- positive diagonal, computed in double -- and takes the centre from the null space of P)."""
    P = np.asarray(P, dtype=np.float64)
    M = P[:, :3]
    # RQ through a QR of the row-reversed transpose:  E M = (E M^T E)^T ...  M = K R,  K upper triangular, R orthogonal
    E = np.eye(3)[::-1]
    q, r = np.linalg.qr((E @ M).T)
    K = E @ r.T @ E
    R = E @ q.T
    sign = np.sign(np.diag(K))
    sign[sign == 0] = 1.0
    K = K * sign[None, :]                      # K D D R with D = diag(sign), D^2 = I
    R = sign[:, None] * R
    if np.linalg.det(R) < 0:                   # P is only defined up to scale: a reflection means the sign of P was flipped
        R = -R
        K = K                                  # (K R) changes sign as a whole; K keeps its positive diagonal
    centre = -np.linalg.solve(M, P[:, 3])
    return K, R, np.concatenate([centre, [1.0]])[:, None]


def load_K_Rt_from_P(filename, P=None):
    """datasets/dtu.py:12-33: -> (intrinsics (4,4) float64 with K / K[2,2], pose = cam2world (4,4) float32)."""
    if P is None:
        lines = open(filename).read().splitlines()
        if len(lines) == 4:
            lines = lines[1:]
        lines = [[x[0], x[1], x[2], x[3]] for x in (x.split(" ") for x in lines)]
        P = np.asarray(lines).astype(np.float32).squeeze()
    dt = np.asarray(P).dtype if np.asarray(P).dtype in (np.float32, np.float64) else np.float64
    K, R, t = decompose_projection_matrix(P)
    K, R, t = K.astype(dt), R.astype(dt), t.astype(dt)          # OpenCV returns the input's type
    K = K / K[2, 2]
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.transpose()
    pose[:3, 3] = (t[:3] / t[3])[:, 0]
    return intrinsics, pose


def read_pfm(filename):
    """datasets/dtu.py:36-71: -> (data (H,W) or (H,W,3) float32 with row 0 at the top, scale)."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header == "PF":
            color = True
        elif header == "Pf":
            color = False
        else:
            raise Exception("Not a PFM file.")
        dim_match = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not dim_match:
            raise Exception("Malformed PFM header.")
        width, height = map(int, dim_match.groups())
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        scale = abs(scale)
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape)), scale


def write_pfm(filename, image, scale=1.0):
    """Inverse of read_pfm (little-endian), for fixtures and for saving depth maps."""
    image = np.asarray(image, dtype=np.float32)
    color = image.ndim == 3
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(f"{image.shape[1]} {image.shape[0]}\n".encode())
        f.write(f"{-abs(scale)}\n".encode())
        np.flipud(image).astype("<f4").tofile(f)


def resize_nearest(img, hw):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST): source index = min(floor(dst * src / dst_size), src - 1)."""
    img = np.asarray(img)
    h, w = int(hw[0]), int(hw[1])
    sh, sw = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(h) * (sh / h)).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(w) * (sw / w)).astype(np.int64), sw - 1)
    return img[ys][:, xs]


def read_cam_file(filename, interval_scale=1.0, num_interval=192):
    """MVSNet-style cam.txt (datasets/dtu.py:170-183): -> (intrinsics (4,4) float32, extrinsics = world2cam (4,4) float32,
    [depth_min, depth_max])."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    intrinsics_ = np.float32(np.diag([1, 1, 1, 1]))
    intrinsics_[:3, :3] = intrinsics
    depth_min = float(lines[11].split()[0])
    depth_interval = float(lines[11].split()[1]) * interval_scale
    return intrinsics_, extrinsics, [depth_min, depth_min + depth_interval * num_interval]


def read_pair_file(filename, num_select=10):
    """Cameras/pair.txt (datasets/dtu.py:106-116): -> (num_viewpoint, <=10) array of source views per reference view."""
    with open(filename) as f:
        num_viewpoint = int(f.readline())
        pairs = [[]] * num_viewpoint
        for _ in range(num_viewpoint):
            ref_view = int(f.readline().rstrip())
            src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
            pairs[ref_view] = np.array(src_views[:num_select])
    return np.array(pairs)


def pairs_from_poses(w2cs, num_select=10):
    """Fallback of datasets/dtu.py:118-125: the nearest camera centres."""
    c2ws = np.linalg.inv(np.stack(w2cs, axis=0))
    dists = np.linalg.norm(c2ws[:, None, :3, 3] - c2ws[None, :, :3, 3], axis=-1)
    dists[np.eye(dists.shape[0]) > 0] = 1e3
    return np.argsort(dists, axis=1)[:, :num_select]


def get_scale_mat(img_hw, intrs, w2cs, near_fars, factor=0.8):
    """datasets/dtu.py:193-229: bounding box of all view frusta -> (scale_mat (4,4) float32 that maps the unit sphere onto
    the scene, 1 / radius)."""
    bnds = np.zeros((3, 2))
    bnds[:, 0] = np.inf
    bnds[:, 1] = -np.inf
    im_h, im_w = img_hw
    for intr, w2c, near_far in zip(intrs, w2cs, near_fars):
        min_depth, max_depth = near_far
        depth = np.array([min_depth] * 4 + [max_depth] * 4)
        pts = np.stack([(np.array([0, 0, im_w, im_w, 0, 0, im_w, im_w]) - intr[0, 2]) * depth / intr[0, 0],
                        (np.array([0, im_h, 0, im_h, 0, im_h, 0, im_h]) - intr[1, 2]) * depth / intr[1, 1], depth])
        pts = pts.astype(np.float32)
        pts = np.linalg.inv(w2c) @ np.concatenate([pts, np.ones_like(pts[:1])], axis=0)
        pts = pts[:3]
        bnds[:, 0] = np.minimum(bnds[:, 0], pts.min(axis=1))
        bnds[:, 1] = np.maximum(bnds[:, 1], pts.max(axis=1))
    center = np.array(((bnds[0, 1] + bnds[0, 0]) / 2, (bnds[1, 1] + bnds[1, 0]) / 2, (bnds[2, 1] + bnds[2, 0]) / 2)).astype(np.float32)
    radius = (bnds[:, 1] - bnds[:, 0]).max(axis=0) / 2 * factor
    scale_mat = np.diag([radius, radius, radius, 1.0]).astype(np.float32)
    scale_mat[:3, 3] = center
    return scale_mat, 1.0 / radius


def near_far_from_sphere(rays_o, rays_d):
    """datasets/dtu.py:231-237."""
    a = torch.sqrt(torch.sum(rays_d ** 2, dim=-1, keepdim=True))
    b = torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = (-b) / a
    return mid - 1.0, mid + 1.0


# ----------------------------------------------------------------------------------------------------------------------
# the item a multi-view dataset hands to GenS.forward, assembled from parts (shared by the DTU / BlendedMVS train, val and fine-tune sets)
# ----------------------------------------------------------------------------------------------------------------------
class NormalisedViews:
    """Cameras of one item re-expressed in the frame the renderer assumes: reference camera at the origin of the world frame BEFORE the
    unit-sphere normalisation, then everything scaled so that the frusta's bounding box fits the unit sphere (datasets/dtu.py:330-356,
    bmvs.py:229-252).  Fields are float32 torch tensors stacked over the views; `scale_mat` maps normalised coordinates back to the
    dataset's world frame (what runner.py:231 multiplies the mesh by); `depth_scale` turns metric depths into normalised ones."""

    def __init__(self, intrs, w2cs, near_fars, img_hw, factor):
        ref_c2w = np.linalg.inv(w2cs[0])
        rel = [w2c @ ref_c2w for w2c in w2cs]                               # poses relative to the reference camera
        sphere, self.depth_scale = get_scale_mat(img_hw, intrs, rel, near_fars, factor=factor)
        k_new, poses, ranges = [], [], []
        for intr, w2c in zip(intrs, rel):
            k, c2w = load_K_Rt_from_P(None, (intr @ w2c @ sphere)[:3, :4])  # the projection of the normalised scene, split again
            centre_dist = np.sqrt(np.sum(c2w[:3, 3] ** 2)).astype(np.float32)
            k_new.append(k)
            poses.append(c2w)
            ranges.append([0.95 * (centre_dist - 1), 1.05 * (centre_dist + 1)])     # the unit sphere seen from this camera
        as_t = lambda seq: torch.from_numpy(np.stack(seq).astype(np.float32))  # noqa: E731
        self.intrs, self.c2ws, self.near_fars = as_t(k_new), as_t(poses), as_t(ranges)
        self.scale_mat = torch.from_numpy(ref_c2w @ sphere)

    def scaled(self, maps):
        """Depth-like maps (list of (H, W) arrays in dataset units) -> one float32 tensor in normalised units."""
        return torch.from_numpy(np.stack([m * self.depth_scale for m in maps]).astype(np.float32))


def lattice_pixels(h, w, level=1):
    """Every `level`-th pixel of an h x w image as flat float (x, y) vectors -- torch.linspace end points included, which is what the
    reference's validation rays use (dtu.py:389-393)."""
    ys, xs = torch.meshgrid(torch.linspace(0, h - 1, h // level), torch.linspace(0, w - 1, w // level), indexing="ij")
    return xs.reshape(-1), ys.reshape(-1)


def sample_train_pixels(mask, n_rays):
    """n_rays training pixels of an (h, w) object mask: the last quarter uniform over the image, the rest uniform over the pixels inside
    the mask -- drawn from torch's global generator in the reference's order: x of the uniform part, y of the uniform part, indices into
    the mask pixels (dtu.py:370-381).  -> float (x, y) vectors, mask part first."""
    h, w = mask.shape
    n_uniform = n_rays // 4
    ux = torch.randint(low=0, high=w, size=[n_uniform])
    uy = torch.randint(low=0, high=h, size=[n_uniform])
    gx, gy = lattice_pixels(h, w)
    inside = (mask > 0.5).reshape(-1)
    gx, gy = gx[inside], gy[inside]
    pick = torch.randint(low=0, high=gx.shape[0], size=[n_rays - n_uniform])
    return torch.cat([gx[pick], ux], dim=0), torch.cat([gy[pick], uy], dim=0)


def rays_from_pixels(intr, c2w, px, py):
    """Unit-length world-space rays of the camera (intr (4,4), c2w (4,4)) through pixel centres (px, py): -> rays_o (N,3) (the camera
    centre, expanded), rays_d (N,3)   (dtu.py:395-401)."""
    homog = torch.stack([px, py, torch.ones_like(py)], dim=-1).float()
    cam = torch.matmul(intr.inverse()[None, :3, :3], homog[:, :, None]).squeeze()
    cam = cam / torch.linalg.norm(cam, ord=2, dim=-1, keepdim=True)
    rays_d = torch.matmul(c2w[None, :3, :3], cam[:, :, None]).squeeze()
    return c2w[None, :3, 3].expand(rays_d.shape), rays_d


def unproject_pseudo_points(depth, mask, intr, c2w, count=2048, at_least=100):
    """`count` world points un-projected from a (pseudo) depth map at random pixels where both depth and mask are positive; None when
    fewer than `at_least` such pixels exist.  One draw from torch's global generator (dtu.py:406-419)."""
    usable = (depth > 0) & (mask > 0)
    if usable.sum() <= at_least:
        return None
    h, w = depth.shape
    rows, cols = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
    cols, rows, z = cols[usable].type_as(intr), rows[usable].type_as(intr), depth[usable]
    pick = torch.randint(low=0, high=cols.shape[0], size=[count])
    cols, rows, z = cols[pick], rows[pick], z[pick]
    cam = torch.matmul(intr.inverse()[:3, :3], torch.stack((cols, rows, torch.ones_like(cols)), dim=0) * z.unsqueeze(0))
    world = torch.matmul(c2w, torch.cat((cam, torch.ones_like(cols).unsqueeze(0)), dim=0))[:3]
    return world.permute(1, 0)
