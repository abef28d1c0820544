"""On-disk formats (sugar_amd/io.py): the 3DGS PLY and cameras.json.  The camera matrices are pinned against golden values
produced by the reference's own loader (tests/golden/make_cameras_golden.py: camera_to_JSON -> load_gs_cameras -> GSCamera);
the PLY layout is restated from gaussian_model.py:177-256 (the reference needs `plyfile`, absent here: PARITY UNPINNED for
the byte layout beyond the header/ordering rules asserted below)."""
import json
import math
import os

import numpy as np
import torch

from sugar_amd import io as sio

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "cameras_golden.npz"))


def test_cameras_json_matches_the_reference_loader(tmp_path):
    p = tmp_path / "cameras.json"
    p.write_bytes(GOLD["json"].tobytes())
    cams, names = sio.cameras_from_json(str(p))
    assert names == list(GOLD["names"])  # sorted by image name
    assert len(cams) == 6
    for i, c in enumerate(cams):
        assert (c.image_height, c.image_width) == tuple(GOLD["sizes"][i])
        assert abs(c.tanfovx - math.tan(GOLD["fov"][i, 0] / 2)) < 1e-12 and abs(c.tanfovy - math.tan(GOLD["fov"][i, 1] / 2)) < 1e-12
        assert torch.equal(c.viewmatrix, torch.tensor(GOLD["world_view"][i]))
        assert torch.equal(c.projmatrix, torch.tensor(GOLD["full_proj"][i]))
        assert torch.equal(c.campos, torch.tensor(GOLD["center"][i]))
    # images larger than max_img_size were scaled down; the others were not
    assert max(cams[0].image_width, cams[1].image_width) <= 1920 and {c.image_width for c in cams} == {1600, 1920}


def test_camera_json_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    r, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                  [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                  [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
    T = rng.standard_normal(3)
    e = sio.camera_to_json(3, "a", R, T, 1.0, 0.7, 640, 480)
    p = tmp_path / "c.json"
    p.write_text(json.dumps([e, dict(e, id=4), dict(e, img_name="0")]))
    cams, names = sio.cameras_from_json(str(p))
    assert names == ["0", "a"]  # the duplicate name keeps one camera
    direct = sio.camera_from_RT(R, T, 1.0, 0.7, 640, 480)
    assert torch.allclose(cams[1].viewmatrix, direct.viewmatrix, atol=1e-6) and torch.allclose(cams[1].projmatrix, direct.projmatrix, atol=1e-5)
    # row-vector convention: a world point maps with p @ viewmatrix, and the camera centre maps to the origin
    assert torch.allclose(torch.cat([direct.campos, torch.ones(1)]) @ direct.viewmatrix, torch.tensor([0.0, 0, 0, 1]), atol=1e-5)


def test_gaussian_ply_round_trip_and_layout(tmp_path):
    g = torch.Generator().manual_seed(0)
    P, M = 257, 16
    t = dict(xyz=torch.randn(P, 3, generator=g), features=torch.randn(P, M, 3, generator=g), opacity=torch.randn(P, 1, generator=g),
             scaling=torch.randn(P, 3, generator=g), rotation=torch.randn(P, 4, generator=g))
    path = str(tmp_path / "sub" / "point_cloud.ply")
    sio.save_gaussian_ply(path, **t)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {P}"]
    props = [l.split()[2] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    assert props == sio.gaussian_ply_attributes(M) and len(props) == 62     # 3+3+3+45+1+3+4 (gaussian_model.py:177-189)
    rows = np.frombuffer(body, dtype="<f4").reshape(P, 62)
    assert np.array_equal(rows[:, 0:3], t["xyz"].numpy()) and not rows[:, 3:6].any()       # normals are zeros
    # f_rest is channel-major: f_rest_k = coefficient 1 + k % 15 of channel k // 15 (transpose(1, 2).flatten(1))
    assert np.array_equal(rows[:, 9 + 0], t["features"][:, 1, 0].numpy())
    assert np.array_equal(rows[:, 9 + 14], t["features"][:, 15, 0].numpy())
    assert np.array_equal(rows[:, 9 + 15], t["features"][:, 1, 1].numpy())
    assert np.array_equal(rows[:, 6:9], t["features"][:, 0, :].numpy())
    back = sio.load_gaussian_ply(path)
    for k in t:
        assert back[k].dtype == torch.float32 and torch.equal(back[k], t[k]), k
    # lower SH degree and reordered / extra properties are handled by name
    sio.save_gaussian_ply(str(tmp_path / "d1.ply"), t["xyz"], t["features"][:, :4], t["opacity"], t["scaling"], t["rotation"])
    assert sio.load_gaussian_ply(str(tmp_path / "d1.ply"))["features"].shape == (P, 4, 3)


def test_ply_reader_accepts_reordered_properties_and_comments(tmp_path):
    """properties are found by name (load_ply does the same through plyfile), header comments are skipped"""
    g = torch.Generator().manual_seed(3)
    P, M = 11, 4
    feats = torch.randn(P, M, 3, generator=g)
    t = dict(xyz=torch.randn(P, 3, generator=g), opacity=torch.randn(P, 1, generator=g), scaling=torch.randn(P, 3, generator=g),
             rotation=torch.randn(P, 4, generator=g))
    names = sio.gaussian_ply_attributes(M)
    cols = {"x": t["xyz"][:, 0], "y": t["xyz"][:, 1], "z": t["xyz"][:, 2], "nx": torch.zeros(P), "ny": torch.zeros(P), "nz": torch.zeros(P),
            "opacity": t["opacity"][:, 0]}
    for c in range(3):
        cols[f"f_dc_{c}"] = feats[:, 0, c]
        for k in range(M - 1):
            cols[f"f_rest_{c * (M - 1) + k}"] = feats[:, 1 + k, c]
    for i in range(3):
        cols[f"scale_{i}"] = t["scaling"][:, i]
    for i in range(4):
        cols[f"rot_{i}"] = t["rotation"][:, i]
    order = list(reversed(names))  # any order
    header = "ply\nformat binary_little_endian 1.0\ncomment written by a test\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in order) + "end_header\n"
    body = np.stack([cols[n].numpy().astype("<f4") for n in order], axis=1).tobytes()
    path = tmp_path / "reordered.ply"
    path.write_bytes(header.encode() + body)
    back = sio.load_gaussian_ply(str(path))
    assert torch.equal(back["features"], feats)
    for k in t:
        assert torch.equal(back[k], t[k]), k
    import pytest
    (tmp_path / "ascii.ply").write_text("ply\nformat ascii 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(ValueError):
        sio.load_gaussian_ply(str(tmp_path / "ascii.ply"))


def test_reference_save_ply_is_read_back_by_load_gaussian_ply_and_the_other_way(tmp_path, monkeypatch):
    """The reference's own GaussianModel.save_ply (gaussian_model.py:191-208) writing through the `plyfile` stand-in, read by
    sugar_amd.io -- and sugar_amd.io's writer read through the stand-in by the statements of GaussianModel.load_ply
    (:215-247, its device="cuda" tensor construction aside).  With the real `plyfile` installed the same test pins the layout
    for real; with the stand-in it pins the column order / reshape rules of the two reference functions against each other."""
    import pytest
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("no reference tree")
    from sugar_amd import shims
    for name in ("plyfile",):
        monkeypatch.delitem(__import__("sys").modules, name, raising=False)
    shims.install()
    _, GaussianModel, _, _ = ref_env.import_gaussian_splatting()
    import plyfile
    g = GaussianModel(3)
    P, rng = 37, torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=rng)
    g._xyz, g._features_dc, g._features_rest = rnd(P, 3), rnd(P, 1, 3), rnd(P, 15, 3)
    g._opacity, g._scaling, g._rotation = rnd(P, 1), rnd(P, 3), rnd(P, 4)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    g.save_ply(path)
    d = sio.load_gaussian_ply(path)
    assert torch.equal(d["xyz"], g._xyz) and torch.equal(d["opacity"], g._opacity)
    assert torch.equal(d["features"], torch.cat([g._features_dc, g._features_rest], dim=1))
    assert torch.equal(d["scaling"], g._scaling) and torch.equal(d["rotation"], g._rotation)
    # the other direction: our writer, the reference's reading statements
    path2 = str(tmp_path / "ours.ply")
    sio.save_gaussian_ply(path2, d["xyz"], d["features"], d["opacity"], d["scaling"], d["rotation"])
    assert open(path2, "rb").read() == open(path, "rb").read()  # byte-identical files
    v = plyfile.PlyData.read(path2).elements[0]
    rest = sorted([p.name for p in v.properties if p.name.startswith("f_rest_")], key=lambda x: int(x.split("_")[-1]))
    extra = np.stack([np.asarray(v[n]) for n in rest], axis=1).reshape(P, 3, 15)                  # :229-235
    assert np.array_equal(extra.transpose(0, 2, 1), g._features_rest.numpy())                        # :251 transpose(1, 2)


def test_sfm_point_cloud_round_trip_through_the_plyfile_stand_in(tmp_path):
    """storePly / fetchPly (gaussian_splatting/scene/dataset_readers.py:107-130): float positions and normals, uchar colours"""
    import pytest
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("no reference tree")
    from sugar_amd import shims
    shims.install()
    ref_env.import_gaussian_splatting()
    from scene.dataset_readers import fetchPly, storePly
    rng = np.random.default_rng(2)
    xyz = rng.standard_normal((50, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, (50, 3)).astype(np.uint8)
    p = str(tmp_path / "points3D.ply")
    storePly(p, xyz, rgb)
    pcd = fetchPly(p)
    assert np.array_equal(pcd.points, xyz) and np.allclose(pcd.colors, rgb / 255.0) and np.array_equal(pcd.normals, np.zeros_like(xyz))


def test_colmap_pose_quaternions_of_the_dataset_writer_invert_the_reference_conversion():
    """oracle/reference_trainer.py::_rotmat_to_qvec (used to write images.txt for the trainer harness) against the reference's
    qvec2rotmat (gaussian_splatting/scene/colmap_loader.py:43-53), all trace branches"""
    import pytest
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("no reference tree")
    from sugar_amd import shims
    shims.install()
    ref_env.import_gaussian_splatting()
    from scene.colmap_loader import qvec2rotmat
    from oracle.reference_trainer import _rotmat_to_qvec
    rng = np.random.default_rng(0)
    mats = [np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0]), np.diag([-1.0, -1.0, 1.0]), np.eye(3)]
    for _ in range(300):
        q, _r = np.linalg.qr(rng.standard_normal((3, 3)))
        mats.append(q * np.sign(np.linalg.det(q)))
    for R in mats:
        qv = _rotmat_to_qvec(R)
        assert abs(np.linalg.norm(qv) - 1.0) < 1e-12 and np.allclose(qvec2rotmat(qv), R, atol=1e-12)
