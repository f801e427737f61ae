"""Dataset front-end: gens_amd.datasets.{DTUDataset, DTUDatasetFinetune, BMVSDataset, BMVSDatasetFinetune}
against the reference's classes (/root/reference/datasets/*.py) run on the same synthetic trees (tests/dtu_fixture.py,
tests/bmvs_fixture.py) with the same RNG seeds -- goldens g12 / g13 (tests/golden/make_golden.py; cv2 stubbed there, see
_install_cv2_stub)."""
import os
import random

import numpy as np
import pytest
import torch

from gens_amd.config import Conf
from gens_amd.datasets import DTUDataset, camera
import sys
sys.path.insert(0, os.path.dirname(__file__))          # the fixtures import each other by bare name (the golden generator does too)
import bmvs_fixture  # noqa: E402
import dtu_fixture  # noqa: E402


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    return dtu_fixture.make_dtu_tree(str(tmp_path_factory.mktemp("dtu")))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "g12_dtu_dataset.npz"))


@pytest.mark.parametrize("mode,idx", [("val", 1), ("train", 0)])
def test_dtu_item_matches_the_reference(tree, golden, mode, idx):
    ds = DTUDataset(Conf(dtu_fixture.conf_values(tree, mode)), mode)
    assert len(ds) == int(golden[f"{mode}_len"])
    random.seed(5)
    np.random.seed(6)
    torch.manual_seed(7)
    item = ds[idx]
    want_keys = {k.split(".", 1)[1] for k in golden.files if k.startswith(mode + ".")}
    assert set(item) == want_keys
    for k in sorted(want_keys):
        want, got = golden[f"{mode}.{k}"], item[k]
        if isinstance(got, str):
            assert got == str(want), k
        elif isinstance(got, (int, np.integer)):
            assert int(got) == int(want), k
        else:
            got = got.numpy()
            assert got.shape == want.shape and str(got.dtype) == str(want.dtype), (k, got.shape, got.dtype, want.shape, want.dtype)
            if got.dtype.kind in "iu":
                assert np.array_equal(got, want), k
            else:      # the camera decomposition runs through a different factorisation (QR vs scipy RQ / SVD): float32 round-off
                scale = max(1.0, float(np.abs(want).max()))
                assert np.abs(got - want).max() <= 2e-5 * scale, (k, float(np.abs(got - want).max()))


def test_projection_matrix_round_trip():
    rng = np.random.default_rng(1)
    for _ in range(50):
        k = np.array([[rng.uniform(300, 3000), rng.uniform(-2, 2), rng.uniform(100, 900)], [0, rng.uniform(300, 3000), rng.uniform(100, 700)], [0, 0, 1.0]])
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        c = rng.standard_normal(3) * 2
        p = rng.uniform(0.2, 30) * k @ np.concatenate([q, (-q @ c)[:, None]], 1)
        intr, pose = camera.load_K_Rt_from_P(None, p)
        assert np.allclose(intr[:3, :3], k, rtol=1e-9, atol=1e-7) and intr.dtype == np.float64 and pose.dtype == np.float32
        assert np.allclose(pose[:3, :3], q.T, atol=1e-6) and np.allclose(pose[:3, 3], c, atol=1e-6)


def test_pfm_round_trip_and_nearest_resize(tmp_path):
    rng = np.random.default_rng(2)
    for shape in ((7, 5), (6, 9, 3)):
        img = rng.standard_normal(shape).astype(np.float32)
        f = str(tmp_path / f"x{len(shape)}.pfm")
        camera.write_pfm(f, img, scale=2.5)
        back, scale = camera.read_pfm(f)
        assert np.array_equal(back, img) and scale == 2.5
    a = np.arange(12 * 16).reshape(12, 16)
    assert np.array_equal(camera.resize_nearest(a, (6, 8)), a[::2, ::2])                   # exact 2x decimation picks the even samples
    assert np.array_equal(camera.resize_nearest(a, (12, 16)), a)
    up = camera.resize_nearest(a, (24, 16))
    assert np.array_equal(up[::2], a) and np.array_equal(up[1::2], a)
    with pytest.raises(Exception):
        (tmp_path / "bad.pfm").write_bytes(b"P6\n1 1\n-1\n\0\0\0\0")
        camera.read_pfm(str(tmp_path / "bad.pfm"))


def test_pairs_from_poses_when_no_pair_file(tree, tmp_path):
    ds = DTUDataset(Conf(dtu_fixture.conf_values(tree, "val")), "val")
    from_file = ds.pairs
    nearest = camera.pairs_from_poses(ds.w2cs, 10)
    assert nearest.shape == from_file.shape == (49, 10)
    assert np.array_equal(nearest[:, 0], from_file[:, 0])          # the fixture's pair.txt ranks by camera distance too


def test_get_loader_runs_a_val_pass(tree):
    from gens_amd.datasets import get_loader
    conf = Conf(dtu_fixture.conf_values(tree, "val"))
    loader, sampler, ds = get_loader(conf, "val", distributed=False)
    loader = torch.utils.data.DataLoader(ds, 1, sampler=sampler, num_workers=0, collate_fn=lambda d: d[0])
    items = list(loader)
    assert len(items) == 2 and items[0]["rays_o"].shape == (30 * 40, 3) and tuple(items[0]["hw"].tolist()) == (30, 40)
    with pytest.raises(NotImplementedError):
        get_loader(Conf({"dataset_name": "NoSuchDataset"}), "val", False)


def test_finetune_dataset_matches_the_reference(tree, golden):
    """DTUDatasetFinetune: resident views, get_all_images / get_random_rays / get_rays_at (runner.py:91,296,346)."""
    from gens_amd.datasets import DTUDatasetFinetune
    torch.manual_seed(11)
    ft = DTUDatasetFinetune(Conf(dtu_fixture.finetune_conf_values(tree)), "finetune")
    items = {"all": ft.get_all_images(), "rand": ft.get_random_rays(torch.tensor(1)), "at": ft.get_rays_at(2)}

    def same(got, want, what):
        if isinstance(got, str):
            assert got == str(want), what
        elif isinstance(got, list):
            assert got == [int(x) for x in want], what
        else:
            got = got.numpy()
            assert got.shape == want.shape, (what, got.shape, want.shape)
            scale = max(1.0, float(np.abs(want).max()))
            assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 2e-5 * scale, what

    same(ft.pseudo_ptses, golden["ft.pseudo_ptses"], "pseudo_ptses")
    same(ft.scale_mat, golden["ft.scale_mat"], "scale_mat")
    for name, item in items.items():
        want_keys = {k.split(".", 2)[2] for k in golden.files if k.startswith(f"ft.{name}.")}
        assert set(item) == want_keys, (name, set(item) ^ want_keys)
        for k in want_keys:
            same(item[k], golden[f"ft.{name}.{k}"], f"{name}.{k}")


# ------------------------------------------------------------------------------------------------------- BlendedMVS
@pytest.fixture(scope="module")
def bmvs_tree(tmp_path_factory):
    return bmvs_fixture.make_bmvs_tree(str(tmp_path_factory.mktemp("bmvs")))


@pytest.fixture(scope="module")
def bmvs_golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "g13_bmvs_dataset.npz"))


def _same(got, want, what):
    if isinstance(got, str):
        assert got == str(want), what
    elif isinstance(got, (int, np.integer)):
        assert int(got) == int(want), what
    elif isinstance(got, list):
        assert got == [int(x) for x in want], what
    else:
        got = got.numpy()
        assert got.shape == want.shape and str(got.dtype) == str(want.dtype), (what, got.shape, got.dtype, want.shape, want.dtype)
        if got.dtype.kind in "iu":
            assert np.array_equal(got, want), what
        else:          # the camera decomposition runs through a different factorisation (QR vs scipy RQ / SVD): float32 round-off
            scale = max(1.0, float(np.abs(want).max()))
            assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 2e-5 * scale, what


@pytest.mark.parametrize("mode,idx", [("val", 1), ("train", 0)])
def test_bmvs_item_matches_the_reference(bmvs_tree, bmvs_golden, mode, idx):
    from gens_amd.datasets import BMVSDataset
    ds = BMVSDataset(Conf(bmvs_fixture.conf_values(bmvs_tree, mode)), mode)
    assert len(ds) == int(bmvs_golden[f"{mode}_len"])
    random.seed(5)
    np.random.seed(6)
    torch.manual_seed(7)
    item = ds[idx]
    want_keys = {k.split(".", 1)[1] for k in bmvs_golden.files if k.startswith(mode + ".")}
    assert set(item) == want_keys
    for k in sorted(want_keys):
        _same(item[k], bmvs_golden[f"{mode}.{k}"], k)
    assert float(item["masks"].mean()) > 0.2 and float(item["masks"].mean()) < 0.9       # the object mask is neither empty nor full


def test_bmvs_finetune_dataset_matches_the_reference(bmvs_tree, bmvs_golden):
    from gens_amd.datasets import BMVSDatasetFinetune, get_loader
    torch.manual_seed(11)
    ft = get_loader(Conf(bmvs_fixture.finetune_conf_values(bmvs_tree)), "finetune", False)
    assert isinstance(ft, BMVSDatasetFinetune) and not hasattr(ft, "pseudo_ptses")
    items = {"all": ft.get_all_images(), "rand": ft.get_random_rays(torch.tensor(1)), "at": ft.get_rays_at(2)}
    _same(ft.scale_mat, bmvs_golden["ft.scale_mat"], "scale_mat")
    _same(ft.masks, bmvs_golden["ft.masks"], "masks")
    for name, item in items.items():
        want_keys = {k.split(".", 2)[2] for k in bmvs_golden.files if k.startswith(f"ft.{name}.")}
        assert set(item) == want_keys, (name, set(item) ^ want_keys)
        for k in want_keys:
            _same(item[k], bmvs_golden[f"ft.{name}.{k}"], f"{name}.{k}")
