"""Validation-output writers (gens_amd/io.py): PLY round trip, the reference's mask-based mesh cleaning (golden g14 generated from
utils/clean_mesh.py:9-35), depth colour map, and the directory layout runner.py:229-246 produces."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from gens_amd import io


def test_ply_round_trip_and_header(tmp_path):
    rng = np.random.default_rng(0)
    v = rng.standard_normal((37, 3))
    t = rng.integers(0, 37, size=(50, 3))
    path = str(tmp_path / "m.ply")
    io.write_ply(path, v, t)
    head = open(path, "rb").read(200).decode("ascii", "replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0\n") and "element vertex 37" in head and "element face 50" in head
    v2, t2 = io.read_ply(path)
    assert np.array_equal(v2, v.astype(np.float32)) and np.array_equal(t2, t)
    assert os.path.getsize(path) == head.index("end_header\n") + len("end_header\n") + 37 * 12 + 50 * 13
    io.write_ply(path, np.zeros((0, 3)), np.zeros((0, 3), dtype=int))               # an empty surface is a valid file
    v0, t0 = io.read_ply(path)
    assert v0.shape == (0, 3) and t0.shape == (0, 3)
    with pytest.raises(ValueError):
        io.write_ply(path, v, np.array([[0, 1, 37]]))


def test_transform_vertices_is_the_homogeneous_product():
    rng = np.random.default_rng(1)
    m = np.eye(4)
    m[:3, :4] = rng.standard_normal((3, 4))
    v = rng.standard_normal((20, 3))
    want = (m @ np.concatenate([v, np.ones((20, 1))], 1).T).T[:, :3]
    assert np.allclose(io.transform_vertices(v, m), want)


def test_clean_mesh_by_mask_matches_the_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g14_clean_mesh.npz"))
    masks, intrs, c2ws = (torch.from_numpy(g[k]) for k in ("masks", "intrs", "c2ws"))
    for nb in (1, 2):
        kept = io.clean_mesh_by_mask(g["vertices"], g["faces"], masks, intrs, c2ws, min_nb_visible=nb)
        assert np.array_equal(kept, g[f"kept{nb}"])
    assert 0 < len(g["kept2"]) < len(g["kept1"]) < len(g["faces"])


def test_dilate_masks_is_a_disk_dilation():
    m = torch.zeros(2, 40, 50)
    m[0, 20, 25] = 1.0
    m[1, 5:8, 5:8] = 0.6
    d = io.dilate_masks(m, radius=4)
    yy, xx = np.mgrid[0:40, 0:50]
    assert np.array_equal(d[0].numpy(), (yy - 20) ** 2 + (xx - 25) ** 2 <= 16)
    assert d[1].sum() > 9 and d.dtype == torch.bool
    assert io.dilate_masks(m[..., None].expand(2, 40, 50, 3), radius=4).equal(d)       # colour masks are averaged first


def test_depth_colour_map_and_validation_layout(tmp_path):
    depth = np.linspace(-0.5, 3.0, 12 * 16, dtype=np.float32).reshape(12, 16)
    rgb = io.depth_to_rgb(depth)
    import matplotlib
    magma = matplotlib.colormaps["magma"]
    want = (np.asarray(magma(np.clip(depth / 2.5, 0, 1)))[:, :, :3] * 255).astype(np.uint8)
    assert np.array_equal(rgb, want) and (rgb[0, 0] == rgb[0, 1]).all()                 # below the range: clipped to the first colour
    outputs = {"vertices": np.random.default_rng(2).standard_normal((9, 3)), "triangles": np.array([[0, 1, 2], [3, 4, 5]]),
               "img_fine": np.full((12, 16, 3), 300.7), "normal_img": np.full((12, 16, 3), 128.0), "sdf_depth": depth, "render_depth": depth + 0.1}
    scale = torch.eye(4)
    scale[:3, 3] = torch.tensor([1.0, 2.0, 3.0])
    inputs = {"scene": "scan24", "file_name": "scan24_view3_light3", "scale_mat": scale}
    paths = io.save_validation_outputs(str(tmp_path), outputs, inputs, "epoch7")
    assert paths["mesh"].endswith("meshes/scan24_epoch7.ply") and paths["img"].endswith("val_img/scan24_view3_light3_epoch7.png")
    assert sorted(os.listdir(tmp_path)) == ["meshes", "val_img", "val_normal", "val_render_depth", "val_sdf_depth"]
    v, t = io.read_ply(paths["mesh"])
    assert np.allclose(v, outputs["vertices"] + np.array([1.0, 2.0, 3.0]), atol=1e-6) and np.array_equal(t, outputs["triangles"])
    assert np.array(Image.open(paths["img"])).shape == (12, 16, 3)
    assert np.array_equal(np.array(Image.open(paths["sdf_depth"])), rgb)
    step = io.save_validation_outputs(str(tmp_path), outputs, inputs, "step99", image_tag=0)          # fine-tune naming (runner.py:373)
    assert step["normal"].endswith("val_normal/0_step99.png")


def test_drop_small_components_keeps_the_large_sheets():
    def sheet(n, offset):
        idx = np.arange(n * n).reshape(n, n) + offset
        return np.concatenate([np.stack([idx[:-1, :-1], idx[1:, :-1], idx[:-1, 1:]], -1).reshape(-1, 3),
                               np.stack([idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]], -1).reshape(-1, 3)])
    big, small = sheet(20, 0), sheet(4, 400)                                     # 722 and 18 faces, no shared vertices
    rng = np.random.default_rng(3)
    verts = rng.standard_normal((400 + 16 + 5, 3))                               # + five vertices nothing references
    tris = np.concatenate([small, big])
    v, t = io.drop_small_components(verts, tris, min_faces=500)
    assert len(t) == len(big) and len(v) == 400 and t.max() == 399
    assert np.allclose(v[t], verts[big])                                         # same triangles, re-indexed
    v2, t2 = io.drop_small_components(verts, tris, min_faces=10)
    assert len(t2) == len(tris) and len(v2) == 416
    v3, t3 = io.drop_small_components(verts, tris[:0], min_faces=10)
    assert len(t3) == 0 and len(v3) == 0
    # two triangles that only touch in ONE vertex are different components (trimesh's face_adjacency is edge adjacency)
    bow = np.array([[0, 1, 2], [2, 3, 4]])
    assert len(io.drop_small_components(verts, bow, min_faces=2)[1]) == 0
