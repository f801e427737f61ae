"""Writers for what `runner.py` stores per validated view, without trimesh / OpenCV:

    meshes/{scene}_epoch{e}.ply            the extracted surface, moved back to world coordinates by scale_mat   (runner.py:229-236)
    val_img/…png, val_normal/…png          rendered colour and normal images                                      (runner.py:243-244)
    val_render_depth/…png, val_sdf_depth/… depth maps through the magma colour map, fixed range [0, 2.5]          (runner.py:245-246, 379-392)

and the mask-based mesh cleaning of utils/clean_mesh.py:9-35 (drop faces with a vertex that fewer than two source masks see).
The ray-casting step of the reference's cleaning (clean_mesh_outside_frustum, utils/clean_mesh.py:38-99: pyembree through
trimesh) is not rebuilt; its tail, the removal of small connected components (:101-106), is `drop_small_components`."""
import os

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


# ------------------------------------------------------------------------------------------------------------------ meshes
def transform_vertices(vertices, matrix):
    """vertices (V,3), matrix (4,4) -> (V,3): what trimesh.Trimesh.apply_transform does to the vertices (runner.py:232)."""
    m = np.asarray(matrix, dtype=np.float64)
    v = np.asarray(vertices, dtype=np.float64)
    return v @ m[:3, :3].T + m[:3, 3]


def write_ply(path, vertices, triangles):
    """Binary little-endian PLY with float32 vertices and int32 triangles (the layout trimesh exports for a bare mesh)."""
    v = np.ascontiguousarray(vertices, dtype="<f4").reshape(-1, 3)
    t = np.ascontiguousarray(triangles, dtype="<i4").reshape(-1, 3)
    if t.size and (t.min() < 0 or t.max() >= len(v)):
        raise ValueError("write_ply: triangle index out of range")
    header = ("ply\nformat binary_little_endian 1.0\ncomment gens_amd\n"
              f"element vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
              f"element face {len(t)}\nproperty list uchar int vertex_indices\nend_header\n")
    faces = np.empty(len(t), dtype=[("n", "u1"), ("idx", "<i4", (3,))])
    faces["n"] = 3
    faces["idx"] = t
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())
        f.write(faces.tobytes())


def read_ply(path):
    """Inverse of write_ply (binary little-endian, float x/y/z vertices, uchar-counted int triangles) -> (vertices, triangles)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        nv = nf = None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            words = line.split()
            if words[:1] == [b"format"] and words[1] != b"binary_little_endian":
                raise ValueError(f"{path}: only binary_little_endian is read")
            if words[:2] == [b"element", b"vertex"]:
                nv = int(words[2])
            if words[:2] == [b"element", b"face"]:
                nf = int(words[2])
            if words[:1] == [b"end_header"]:
                break
        v = np.frombuffer(f.read(12 * nv), dtype="<f4").reshape(nv, 3)
        faces = np.frombuffer(f.read(13 * nf), dtype=[("n", "u1"), ("idx", "<i4", (3,))])
        if nf and not (faces["n"] == 3).all():
            raise ValueError(f"{path}: non-triangular face")
    return v.copy(), faces["idx"].copy()


@torch.no_grad()
def clean_mesh_by_mask(vertices, triangles, masks, intrs, c2ws, min_nb_visible=1):
    """utils/clean_mesh.py:9-35: keep the triangles whose three vertices project inside the (dilated) object mask of more than
    `min_nb_visible` views.  vertices (V,3) in the cameras' frame, masks (nv,H,W), intrs (nv,4,4), c2ws (nv,4,4) -> kept triangles."""
    points = torch.from_numpy(np.asarray(vertices)).float().permute(1, 0)
    masks, intrs, c2ws = masks.cpu(), intrs.cpu(), c2ws.cpu()
    nv, h, w = masks.shape
    pts_cam = torch.matmul(c2ws.inverse(), torch.cat([points, torch.ones_like(points[:1])], dim=0)[None])[:, :3]
    pts_img = torch.matmul(intrs[:, :3, :3], pts_cam)
    pts_xy = pts_img[:, :2] / torch.clamp(pts_img[:, 2:], 1e-8)
    pts_xy[:, 0] = 2 * pts_xy[:, 0] / (w - 1) - 1
    pts_xy[:, 1] = 2 * pts_xy[:, 1] / (h - 1) - 1
    in_mask = (pts_xy.abs() <= 1).all(dim=1) & (pts_img[:, -1] > 1e-8)
    grid = torch.clamp(pts_xy.permute(0, 2, 1).unsqueeze(1), -10, 10)
    warp_mask = F.grid_sample(masks.unsqueeze(1).float(), grid, align_corners=True).squeeze(1).squeeze(1)
    valid = ((warp_mask > 0) * in_mask).sum(dim=0) > min_nb_visible
    tri = torch.from_numpy(np.asarray(triangles).astype(np.int64))
    return np.asarray(triangles)[valid[tri].all(dim=-1).numpy()]


def drop_small_components(vertices, triangles, min_faces=500):
    """utils/clean_mesh.py:101-106 without trimesh: keep the connected components (faces sharing an edge) of at least `min_faces`
    faces and drop the vertices nothing references any more -> (vertices, triangles) re-indexed."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    tri = np.asarray(triangles, dtype=np.int64).reshape(-1, 3)
    v = np.asarray(vertices)
    if len(tri) == 0:
        return v[:0], tri
    # faces adjacent through a shared (undirected) edge: sort the three edges of every face, group equal edges
    edges = np.sort(np.stack([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], 1).reshape(-1, 2), axis=1)
    face_of = np.repeat(np.arange(len(tri)), 3)
    order = np.lexsort((edges[:, 1], edges[:, 0]))
    e, f = edges[order], face_of[order]
    same = (e[1:] == e[:-1]).all(axis=1)
    a, b = f[:-1][same], f[1:][same]
    n_comp, label = connected_components(coo_matrix((np.ones(len(a), dtype=np.int8), (a, b)), shape=(len(tri), len(tri))), directed=False)
    keep = np.bincount(label, minlength=n_comp)[label] >= min_faces
    tri = tri[keep]
    used = np.zeros(len(v), dtype=bool)
    used[tri.reshape(-1)] = True
    remap = np.cumsum(used) - 1
    return v[used], remap[tri].astype(np.asarray(triangles).dtype if len(tri) else np.int64)


def dilate_masks(masks, radius=11):
    """utils/clean_mesh.py:119-125: masks (nv,H,W[,3]) > 0.5, dilated by a disk of `radius` pixels (skimage.morphology.disk)."""
    from scipy import ndimage
    masks = masks.cpu()
    if masks.dim() > 3:
        masks = masks.mean(dim=-1)
    yy, xx = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    disk = (xx * xx + yy * yy) <= radius * radius
    return torch.stack([torch.from_numpy(ndimage.binary_dilation((m > 0.5).numpy(), structure=disk)) for m in torch.unbind(masks)])


# ------------------------------------------------------------------------------------------------------------------ images
def depth_to_rgb(depth, vmin=0.0, vmax=2.5):
    """runner.py:379-390: (H,W) depth -> (H,W,3) uint8 through matplotlib's magma map, linear in [vmin, vmax]."""
    import matplotlib as mpl
    import matplotlib.cm as cm
    mapper = cm.ScalarMappable(norm=mpl.colors.Normalize(vmin=vmin, vmax=vmax), cmap="magma")
    return (mapper.to_rgba(np.asarray(depth))[:, :, :3] * 255).astype(np.uint8)


def save_depth(depth, file_path):
    Image.fromarray(depth_to_rgb(depth)).save(file_path)


def save_validation_outputs(base_exp_dir, outputs, inputs, tag, image_tag=None, clean=False):
    """Store one validated view the way runner.py:215-246 (tag = "epoch{e}", image names from inputs["file_name"]) and
    runner.py:349-375 (tag = "step{s}", image_tag = the view index) do.  Returns the written paths."""
    scene = inputs["scene"]
    image_tag = inputs["file_name"] if image_tag is None else image_tag
    vertices, triangles = outputs["vertices"], outputs["triangles"]
    if clean:
        triangles = clean_mesh_by_mask(vertices, triangles, dilate_masks(inputs["masks"]), inputs["intrs"], inputs["c2ws"])
    vertices = transform_vertices(vertices, inputs["scale_mat"].detach().cpu().numpy())
    paths = {}
    for sub in ("meshes", "val_img", "val_normal", "val_sdf_depth", "val_render_depth"):
        os.makedirs(os.path.join(base_exp_dir, sub), exist_ok=True)
    paths["mesh"] = os.path.join(base_exp_dir, "meshes", f"{scene}_{tag}.ply")
    write_ply(paths["mesh"], vertices, triangles)
    paths["img"] = os.path.join(base_exp_dir, "val_img", f"{image_tag}_{tag}.png")
    Image.fromarray(outputs["img_fine"].astype(np.uint8)).save(paths["img"])
    paths["normal"] = os.path.join(base_exp_dir, "val_normal", f"{image_tag}_{tag}.png")
    Image.fromarray(outputs["normal_img"].astype(np.uint8)).save(paths["normal"])
    paths["render_depth"] = os.path.join(base_exp_dir, "val_render_depth", f"{image_tag}_{tag}.png")
    save_depth(outputs["render_depth"], paths["render_depth"])
    paths["sdf_depth"] = os.path.join(base_exp_dir, "val_sdf_depth", f"{image_tag}_{tag}.png")
    save_depth(outputs["sdf_depth"], paths["sdf_depth"])
    return paths
