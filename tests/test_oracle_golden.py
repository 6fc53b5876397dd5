"""Pin the oracle (oracle/fusion_ref.py) against fixtures produced by running the
unmodified reference (tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest
from PIL import Image

from oracle import fusion_ref as O


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def kat_depth(h, w):
    j, i = np.mgrid[0:h, 0:w]
    return ((7 * j + 3 * i + 1) % 256).astype(np.uint8)


def test_manifest_intact(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))["files"]
    assert len(man) >= 22
    for rel, digest in man.items():
        assert _sha(os.path.join(golden_dir, rel)) == digest, rel


def test_kat1_unproject_bytes(golden_dir, tmp_path):
    out = tmp_path / "kat.txt"
    O.gentxtcord_loop(str(out), kat_depth(4, 6))
    want = open(os.path.join(golden_dir, "kat_unproject_4x6.txt")).read()
    assert out.read_text() == want
    # SURVEY 8(c) KAT-1 literal values
    lines = want.splitlines()
    assert lines[0] == "-0.5329860041206481,-0.3999473402668649,1"
    assert lines[-1] == "-19.41234961883173,-14.613075945000576,37"
    # vectorised flavour is bit-identical to the text (repr round-trips fp64)
    vec = O.unproject(kat_depth(4, 6))
    np.testing.assert_array_equal(vec, O.read_xyz_txt(os.path.join(golden_dir, "kat_unproject_4x6.txt")))


def test_quat_to_rinv_matches_scipy_transfer(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "poses.json")))
    for q, want in zip(g["quats_xyzw"], g["scipy_transfer"]):
        got = O.quat_to_rinv(q)
        np.testing.assert_allclose(got, np.array(want), rtol=0, atol=2e-15)
    for q, want in zip(g["get_r_wxyz_input"], g["get_r"]):
        np.testing.assert_allclose(O.get_r_wxyz(q), np.array(want), rtol=0, atol=2e-15)
    # KAT-2 literal
    r = O.quat_to_rinv([0.1, 0.2, 0.3, 0.9])
    np.testing.assert_allclose(r[0], [0.726315789473684, 0.6105263157894736, -0.31578947368421045], atol=2e-15)


def test_point_camera(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "poses.json")))
    for case in g["point_camera"]:
        got = O.se3_apply(np.array([case["p"]]), O.quat_to_rinv(case["q"]), case["t"])[0]
        np.testing.assert_allclose(got, case["p_world"], rtol=0, atol=1e-13)


def _load_scene(golden_dir):
    scene = os.path.join(golden_dir, "scene3")
    names, quats, ts = O.parse_pose_file(os.path.join(scene, "camera_pose", "image_colmap_simi_2.txt"))
    depths = np.stack([np.array(Image.open(os.path.join(scene, "depth", n)).convert("L")) for n in names])
    return scene, names, quats, ts, depths


def test_scene3_full_pipeline(golden_dir, tmp_path):
    scene, names, quats, ts, depths = _load_scene(golden_dir)
    assert names == ["000.png", "007.png", "frame_b.png"]
    # camera txt per frame: byte exact with the loop flavour
    for k, n in enumerate(names):
        p = tmp_path / "cam.txt"
        O.gentxtcord_loop(str(p), depths[k])
        assert p.read_text() == open(os.path.join(scene, "point", n[:-4] + ".txt")).read()
    # world txt holds the LAST frame only; loop flavour reproduces the bytes
    xs, ys, zs = [], [], []
    for k, n in enumerate(names):
        O.get_pointdata_loop(os.path.join(scene, "point", n[:-4] + ".txt"), quats[k], ts[k], xs, ys, zs,
                             str(tmp_path / "world.txt"))
    assert (tmp_path / "world.txt").read_text() == \
        open(os.path.join(scene, "point_world", "small_worldpoint_5_23_5.txt")).read()
    # fused PLY: byte exact from the loop flavour, and from the vectorised flavour
    want_ply = open(os.path.join(scene, "ply", "small_035_p8.ply")).read()
    assert O.format_ply(np.stack([xs, ys, zs], 1)) == want_ply
    fused = O.fuse_frames(depths, quats, ts)
    assert fused.shape == (3 * 24 * 32, 3)
    np.testing.assert_allclose(fused, np.stack([xs, ys, zs], 1), rtol=0, atol=2e-12)
    np.testing.assert_allclose(fused[-768:], O.read_xyz_txt(os.path.join(scene, "point_world",
                               "small_worldpoint_5_23_5.txt")), rtol=0, atol=2e-12)
    got_ply = O.format_ply(fused)
    if got_ply != want_ply:  # a 1e-13 difference may flip a %.4f digit; must be rare and tiny
        a = O.read_ply_vertices(os.path.join(scene, "ply", "small_035_p8.ply"))
        b = np.array([[float(v) for v in s.split()] for s in got_ply.split("end_header\n")[1].strip().split("\n")])
        assert np.abs(a - b).max() <= 1.0001e-4
        assert (a != b).sum() <= 2
    assert O.read_ply_vertices(os.path.join(scene, "ply", "small_035_p8.ply")).shape == (2304, 3)


def test_icp_apply_merge(golden_dir, tmp_path):
    d = os.path.join(golden_dir, "icp_apply")
    T = O.parse_T_file(os.path.join(d, "T_data.txt"))
    np.testing.assert_array_equal(T, np.array(json.load(open(os.path.join(d, "T_parsed.json")))))
    xs, ys, zs = [], [], []
    with open(tmp_path / "w.txt", "w") as f:
        O.local_world_loop(os.path.join(d, "point", "0.txt"), f, T, xs, ys, zs, False)
        O.local_world_loop(os.path.join(d, "point", "24.txt"), f, T, xs, ys, zs, True)
    assert (tmp_path / "w.txt").read_text() == open(os.path.join(d, "point_world", "03_testT.txt")).read()
    assert O.format_ply(np.stack([xs, ys, zs], 1)) == open(os.path.join(d, "ply", "icp", "024.ply")).read()
    a = O.read_xyz_txt(os.path.join(d, "point", "0.txt"))
    b = O.apply_T(O.read_xyz_txt(os.path.join(d, "point", "24.txt")), T)
    merged = np.concatenate([a, b])
    np.testing.assert_allclose(merged, O.read_xyz_txt(os.path.join(d, "point_world", "03_testT.txt")),
                               rtol=0, atol=1e-12)


def test_p2c_480x640_digest(golden_dir, tmp_path):
    g = json.load(open(os.path.join(golden_dir, "p2c_480x640.json")))
    depth = np.random.default_rng(g["seed"]).integers(1, 256, tuple(g["shape"]), dtype=np.uint8)
    out = tmp_path / "p.txt"
    O.gentxtcord_loop(str(out), depth)
    assert _sha(str(out)) == g["sha256_txt"]
    vec = O.unproject(depth)
    for k, (x, y, z) in g["ret_samples"].items():
        assert tuple(vec[int(k)]) == (x, y, float(z))
    assert g["short_raster_error"] == "IndexError"  # p2c hard-codes 480x640 (p2c:34-35)
    assert O.format_ply(vec[:7]) == open(os.path.join(golden_dir, "p2c_first7.ply")).read()
    # the coloured writer (genply_noRGB, p2c:55-91) on the reference's own image
    from PIL import Image
    rgb = np.array(Image.open(os.path.join(golden_dir, "p2c_rgb_2x3.png")).convert("RGB")).reshape(-1, 3)
    assert O.format_ply_rgb(vec[:6], rgb) == open(os.path.join(golden_dir, "p2c_first6_rgb.ply")).read()


def test_c1_192x640_digest(golden_dir, tmp_path):
    g = json.load(open(os.path.join(golden_dir, "c1_192x640.json")))
    depth = np.random.default_rng(g["seed"]).integers(1, 256, tuple(g["shape"]), dtype=np.uint8)
    fused = O.fuse_frames(depth[None], [g["q_xyzw"]], [g["t"]])
    assert fused.shape[0] == g["n"] == 122880
    for k, xyz in g["world_xyz"].items():
        np.testing.assert_allclose(fused[int(k)], xyz, rtol=0, atol=1e-12)
    np.testing.assert_allclose(fused.sum(0), g["world_sum_xyz"], rtol=1e-12)
    # loop-faithful flavour reproduces both text files byte for byte
    got = O.fuse_frames_loop(depth[None], [g["q_xyzw"]], [g["t"]], str(tmp_path))
    assert _sha(str(tmp_path / "cam_0.txt")) == g["sha256_cam_txt"]
    assert _sha(str(tmp_path / "world.txt")) == g["sha256_world_txt"]
    np.testing.assert_allclose(got, fused, rtol=0, atol=1e-12)


def test_quat_to_rinv_is_scipys_on_quaternions_of_every_scale():
    """c2w:53-55 through SciPy itself (importable here and on the GPU box): `np.matrix(R.from_quat(q).as_matrix()).I` for 20,000
    random quaternions scaled over four decades (SciPy normalises; the order in which the four squares are added shows in
    the last bit) against the oracle's restatement AND the product's (poses.scipy_transfer), bit for bit.  The golden
    poses.json holds three quaternions: not enough to notice that an np.dot-based norm is an ulp off for one in eight."""
    Rotation = pytest.importorskip("scipy.spatial.transform").Rotation
    import importlib
    from helpers import PKG
    poses = importlib.import_module(PKG + ".poses")
    rng = np.random.default_rng(31337)
    for _ in range(20000):
        q = rng.normal(size=4) * 10 ** rng.uniform(-2, 2)
        want = np.asarray(np.matrix(Rotation.from_quat(q).as_matrix()).I)
        assert np.array_equal(O.quat_to_rinv(q), want), q
        assert np.array_equal(np.asarray(poses.scipy_transfer(q)), want), q
