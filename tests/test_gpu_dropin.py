"""GPU: the drop-in scripts, run exactly as the reference's are (python <script>.py from a working
directory with ./depth ./camera_pose ./point ./point_world ./ply), against the files the unmodified
reference wrote for the same inputs (tests/golden/)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from helpers import PKG, ROOT, r3d as _r3d
from oracle import fusion_ref as O

pytestmark = pytest.mark.gpu

SCRIPTS = os.path.join(ROOT, PKG)


def run_script(rel, cwd, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(SCRIPTS, rel)] + list(args), cwd=cwd, env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_camera_to_world_script_reproduces_reference_files(tmp_path, golden_dir):
    scene = os.path.join(golden_dir, "scene3")
    for d in ("depth", "camera_pose"):
        shutil.copytree(os.path.join(scene, d), tmp_path / d)
    for d in ("point", "point_world", "ply"):
        os.makedirs(tmp_path / d)
    out = run_script("transfer/camera_to_world.py", str(tmp_path))
    assert "Write into .ply file Done." in out
    # camera txt per frame: byte identical (fp64 product u*Z is the reference's own operation)
    for name in ("000", "007", "frame_b"):
        assert (tmp_path / "point" / (name + ".txt")).read_bytes() == \
            open(os.path.join(scene, "point", name + ".txt"), "rb").read()
    # fused PLY: byte identical at 4 decimals
    assert (tmp_path / "ply" / "small_035_p8.ply").read_bytes() == \
        open(os.path.join(scene, "ply", "small_035_p8.ply"), "rb").read()
    # world txt (last frame only, like the reference): fp64 values equal up to summation order
    got = O.read_xyz_txt(str(tmp_path / "point_world" / "small_worldpoint_5_23_5.txt"))
    want = O.read_xyz_txt(os.path.join(scene, "point_world", "small_worldpoint_5_23_5.txt"))
    assert got.shape == want.shape == (768, 3)
    assert (np.abs(got - want) / (1 + np.linalg.norm(want, axis=1, keepdims=True))).max() <= 1e-12


def test_camera_to_world_device_text_equals_host_text(tmp_path):
    """The script's default way (text formatted on the GPU, csrc/r3d_textfmt.hip) and the host formatter's way
    (R3D_HOST_TEXT=1, rounds 1-4) write the same bytes into every file: 7 frames of 120 x 200, 16-bit depth PNGs too."""
    from PIL import Image
    rng = np.random.default_rng(77)
    F, H, W = 7, 120, 200
    for sub, bits in (("a8", 8), ("b8", 8), ("a16", 16), ("b16", 16)):
        for d in ("depth", "point", "point_world", "ply", "camera_pose"):
            os.makedirs(tmp_path / sub / d)
    for bits in (8, 16):
        rs = np.random.default_rng(bits)
        with open(tmp_path / ("a%d" % bits) / "camera_pose" / "image_colmap_simi_2.txt", "w") as f:
            f.write("id,tx,ty,tz,qx,qy,qz,qw,name,extra\n")
            for k in range(F):
                if bits == 8:
                    Image.fromarray(rs.integers(0, 256, (H, W), dtype=np.uint8)).save(tmp_path / "a8" / "depth" / ("%03d.png" % k))
                else:
                    Image.fromarray(rs.integers(0, 65536, (H, W), dtype=np.uint16)).save(tmp_path / "a16" / "depth" / ("%03d.png" % k))
                q, t = rs.normal(size=4), rs.normal(size=3) * 10
                f.write(",".join([str(k)] + [repr(float(x)) for x in t] + [repr(float(x)) for x in q] + ["%03d.png" % k, "x"]) + "\n")
        a, b = tmp_path / ("a%d" % bits), tmp_path / ("b%d" % bits)
        shutil.copytree(a / "depth", b / "depth", dirs_exist_ok=True)
        shutil.copy(a / "camera_pose" / "image_colmap_simi_2.txt", b / "camera_pose")
        out_a = run_script("transfer/camera_to_world.py", str(a))
        out_b = run_script("transfer/camera_to_world.py", str(b), env={"R3D_HOST_TEXT": "1"})
        assert out_a.count("Write into .ply file Done.") == out_b.count("Write into .ply file Done.") == 1
        for rel in ["ply/small_035_p8.ply", "point_world/small_worldpoint_5_23_5.txt"] + ["point/%03d.txt" % k for k in range(F)]:
            assert (a / rel).read_bytes() == (b / rel).read_bytes(), (bits, rel)
        assert os.path.getsize(a / "ply" / "small_035_p8.ply") > F * H * W * 20
    # f1's optional binary flag: the same cloud as float32, in a standard binary PLY; every other file as before
    a = tmp_path / "a8"
    text_ply = O.read_ply_vertices(str(a / "ply" / "small_035_p8.ply"))
    run_script("transfer/camera_to_world.py", str(a), env={"R3D_PLY_BINARY": "1"})
    R = _r3d()
    got = R.cloud_io.read_ply(str(a / "ply" / "small_035_p8.ply"))
    assert got.shape == (F * H * W, 3) and os.path.getsize(a / "ply" / "small_035_p8.ply") == len(R.device_text.ply_header_binary(F * H * W)) + 12 * F * H * W
    assert np.abs(got - text_ply).max() <= 0.5001e-4 + 1e-6 * np.abs(got).max()          # the text holds 4 decimals, the floats ~7 digits
    assert (a / "point" / "000.txt").read_bytes() == (tmp_path / "b8" / "point" / "000.txt").read_bytes()


def test_camera_to_world_functions_keep_reference_semantics(tmp_path, golden_dir, monkeypatch):
    import importlib
    c2w = importlib.import_module(PKG + ".transfer.camera_to_world")
    scene = os.path.join(golden_dir, "scene3")
    monkeypatch.chdir(tmp_path)
    os.makedirs("point_world")
    names, quats, ts = O.parse_pose_file(os.path.join(scene, "camera_pose", "image_colmap_simi_2.txt"))
    xs, ys, zs = [], [], []
    for k, n in enumerate(names):      # get_pointdata mutates the caller's lists and rewrites the world txt
        c2w.get_pointdata(os.path.join(scene, "point", n[:-4] + ".txt"), quats[k], ts[k], xs, ys, zs)
    assert len(xs) == len(ys) == len(zs) == 2304 and isinstance(xs[0], float)
    want = O.read_ply_vertices(os.path.join(scene, "ply", "small_035_p8.ply"))
    assert np.abs(np.stack([xs, ys, zs], 1) - want).max() <= 0.5001e-4
    c2w.genply([xs, ys, zs], "out.ply", len(xs))
    assert open("out.ply", "rb").read() == open(os.path.join(scene, "ply", "small_035_p8.ply"), "rb").read()
    assert c2w.gentxtcord("cam.txt", np.zeros((2, 4), np.uint8)) is None
    p = c2w.point_camera(np.array([1.0, 0.0, 2.0]), c2w.scipy_transfer([0.1, 0.2, 0.3, 0.9]), np.array([1.0, 2, 3]))
    np.testing.assert_allclose(p.ravel(), [-0.9052631578947368, -1.8947368421052633, -0.768421052631579], atol=1e-15)


def test_pixel_to_camera_script_config_c1(tmp_path, golden_dir):
    """BASELINE config 1: a single 640x192 depth PNG -> camera-frame txt + PLY (the reference script
    itself raises IndexError on this size and TypeError before writing any PLY)."""
    from PIL import Image
    for d in ("depth", "point", "ply"):
        os.makedirs(tmp_path / d)
    depth = np.random.default_rng(0).integers(1, 256, (192, 640), dtype=np.uint8)
    Image.fromarray(np.stack([depth // 2, depth, depth // 3], 2).astype(np.uint8), "RGB").save(tmp_path / "depth" / "24.png")
    run_script("transfer/pixel_to_camera.py", str(tmp_path))
    import hashlib, json
    g = json.load(open(os.path.join(golden_dir, "c1_192x640.json")))
    assert hashlib.sha256((tmp_path / "point" / "24.txt").read_bytes()).hexdigest() == g["sha256_cam_txt"]
    cam = O.unproject(depth)
    assert (tmp_path / "ply" / "24.ply").read_bytes() == O.format_ply(cam).encode()


def test_pixel_to_camera_functions(tmp_path, golden_dir):
    import importlib, json
    p2c = importlib.import_module(PKG + ".transfer.pixel_to_camera")
    g = json.load(open(os.path.join(golden_dir, "p2c_480x640.json")))
    depth = np.random.default_rng(g["seed"]).integers(1, 256, tuple(g["shape"]), dtype=np.uint8)
    xs, ys, zs = p2c.gentxtcord(str(tmp_path / "p.txt"), depth)
    import hashlib
    assert hashlib.sha256((tmp_path / "p.txt").read_bytes()).hexdigest() == g["sha256_txt"]
    for k, (x, y, z) in g["ret_samples"].items():
        assert (xs[int(k)], ys[int(k)], zs[int(k)]) == (x, y, z)
    assert isinstance(zs[0], int)
    p2c.genply_RGB([xs[:7], ys[:7], zs[:7]], str(tmp_path / "f.ply"))
    assert (tmp_path / "f.ply").read_bytes() == open(os.path.join(golden_dir, "p2c_first7.ply"), "rb").read()
    from PIL import Image
    rgb = np.random.default_rng(1).integers(0, 256, (2, 3, 3), dtype=np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "c.png")
    p2c.genply_noRGB([xs[:6], ys[:6], zs[:6]], str(tmp_path / "c.png"), str(tmp_path / "c.ply"))
    lines = (tmp_path / "c.ply").read_text().split("\n")
    assert lines[2].strip() == "element vertex 6" and lines[9].strip() == "property uchar alpha"
    assert lines[11].split() == ["%.4f" % xs[0], "%.4f" % ys[0], "%.4f" % zs[0]] + [str(v) for v in rgb[0, 0]] + ["0"]
    # and the reference's own bytes for its own image
    p2c.genply_noRGB([xs[:6], ys[:6], zs[:6]], os.path.join(golden_dir, "p2c_rgb_2x3.png"), str(tmp_path / "g.ply"))
    assert (tmp_path / "g.ply").read_bytes() == open(os.path.join(golden_dir, "p2c_first6_rgb.ply"), "rb").read()


def test_transfer_T_icp_script_reproduces_reference_files(tmp_path, golden_dir):
    d = os.path.join(golden_dir, "icp_apply")
    shutil.copytree(os.path.join(d, "point"), tmp_path / "point")
    shutil.copy(os.path.join(d, "T_data.txt"), tmp_path / "T_data.txt")
    os.makedirs(tmp_path / "point_world")
    os.makedirs(tmp_path / "ply" / "icp")
    run_script("other_tools/transfer_T_icp.py", str(tmp_path))
    assert (tmp_path / "ply" / "icp" / "024.ply").read_bytes() == open(os.path.join(d, "ply", "icp", "024.ply"), "rb").read()
    got = O.read_xyz_txt(str(tmp_path / "point_world" / "03_testT.txt"))
    want = O.read_xyz_txt(os.path.join(d, "point_world", "03_testT.txt"))
    assert got.shape == want.shape == (100, 3)
    np.testing.assert_array_equal(got[:50], want[:50])            # pass-through cloud: exact
    assert np.abs(got[50:] - want[50:]).max() <= 1e-12 * 500


def test_transfer_T_icp_estimate_recovers_transform(tmp_path):
    from oracle import icp_ref as OI
    R = _r3d()
    # SURVEY's C3 recipe (s=1.7, 10 degrees, |t|=0.5; target uniform in a 20 m cube + N(0, 0.01)) from TXT FILES, no guess
    src, tgt, T_true, _ = OI.synthetic_pair(n_tgt=200000, n_src=200000, s=1.7, angle_deg=10.0, t_norm=0.5, seed=7)
    tgt = (tgt.astype(np.float64) + np.random.default_rng(8).normal(size=tgt.shape) * 0.01).astype(np.float32)
    for dd in ("point", "point_world", os.path.join("ply", "icp")):
        os.makedirs(tmp_path / dd)
    R.cloud_io.write_xyz_txt(str(tmp_path / "point" / "0.txt"), tgt.astype(np.float64))
    R.cloud_io.write_xyz_txt(str(tmp_path / "point" / "24.txt"), src.astype(np.float64))
    out = run_script("other_tools/transfer_T_icp.py", str(tmp_path), "--estimate")
    assert "ICP:" in out
    T = R.get_T(str(tmp_path / "T_data.txt"))
    np.testing.assert_allclose(T, T_true, rtol=0, atol=1e-3)
    merged = R.cloud_io.read_ply(str(tmp_path / "ply" / "icp" / "024.ply"))
    assert merged.shape == (400000, 3)
    # the moved cloud lands on the target: every merged point of the second half has a target point within the noise
    from scipy.spatial import cKDTree
    d, _ = cKDTree(tgt.astype(np.float64)).query(merged[200000::997])
    assert d.max() < 0.1


def test_camera_to_world_script_c1_scene_digests(tmp_path, golden_dir):
    """245,760 points (2 frames of 640x192): the drop-in's PLY and camera txt files hash to what the reference wrote."""
    import hashlib
    import json
    from PIL import Image
    g = json.load(open(os.path.join(golden_dir, "c1_scene_2x192x640.json")))
    for d in ("depth", "camera_pose", "point", "point_world", "ply"):
        os.makedirs(tmp_path / d)
    rng = np.random.default_rng(g["seed"])
    lines = ["id,tx,ty,tz,qx,qy,qz,qw,name,tail\n"]
    for k, fr in enumerate(g["frames"]):
        depth = rng.integers(0, 256, size=tuple(g["shape"]), dtype=np.uint8)
        Image.fromarray(depth, mode="L").save(tmp_path / "depth" / fr["name"])
        q, t = rng.normal(size=4), rng.normal(size=3) * 10
        assert [float(v) for v in q] == fr["q"] and [float(v) for v in t] == fr["t"]
        lines.append("%d,%r,%r,%r,%r,%r,%r,%r,%s,x\n" % ((k,) + tuple(float(v) for v in t) + tuple(float(v) for v in q)
                                                        + (fr["name"],)))
    (tmp_path / "camera_pose" / "image_colmap_simi_2.txt").write_text("".join(lines))
    run_script("transfer/camera_to_world.py", str(tmp_path))
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
    for fr in g["frames"]:
        assert sha(tmp_path / "point" / (fr["name"][:-4] + ".txt")) == g["sha256_point"][fr["name"]]
    assert sha(tmp_path / "ply" / "small_035_p8.ply") == g["sha256_ply"]
    world = O.read_xyz_txt(str(tmp_path / "point_world" / "small_worldpoint_5_23_5.txt"))
    for k, row in g["world_last_frame_rows"].items():
        np.testing.assert_allclose(world[int(k)], row, rtol=0, atol=1e-11)
    np.testing.assert_allclose(world.sum(0), g["world_last_frame_sum"], rtol=1e-11)


@pytest.mark.parametrize("world,host_text", [(2, "0"), (3, "0"), (4, "0"), (2, "1"), (4, "1")])
def test_camera_to_world_script_frame_sharded_over_ranks(tmp_path, golden_dir, mock_rccl, world, host_text):
    """BASELINE config 4's shape through the drop-in: `torch.distributed.run --nproc-per-node N camera_to_world.py`.
    scene3 has 3 frames: 2 ranks = ragged blocks (2 + 1), 3 ranks = one each, 4 ranks = an EMPTY last block.  The ranks share
    the box's one GPU against the stand-in transport (R3D_RCCL_PATH; RCCL refuses two ranks per device); the worker
    processes never import torch.  Every file must equal what the reference wrote for the same inputs -- with every rank
    formatting its own files' text on the GPU (the default) and through host memory and the host formatter (R3D_HOST_TEXT=1)."""
    scene = os.path.join(golden_dir, "scene3")
    for d in ("depth", "camera_pose"):
        shutil.copytree(os.path.join(scene, d), tmp_path / d)
    for d in ("point", "point_world", "ply"):
        os.makedirs(tmp_path / d)
    port = 29650 + (os.getpid() + 11 * world) % 120
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(SCRIPTS, "transfer", "camera_to_world.py")]
    r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", R3D_RCCL_PATH=mock_rccl,
                                R3D_SHARE_GPU="1", R3D_HOST_TEXT=host_text))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("Write into .ply file Done.") == 1                  # rank 0 only
    for name in ("000", "007", "frame_b"):
        assert (tmp_path / "point" / (name + ".txt")).read_bytes() == \
            open(os.path.join(scene, "point", name + ".txt"), "rb").read()
    assert (tmp_path / "ply" / "small_035_p8.ply").read_bytes() == \
        open(os.path.join(scene, "ply", "small_035_p8.ply"), "rb").read()
    got = O.read_xyz_txt(str(tmp_path / "point_world" / "small_worldpoint_5_23_5.txt"))
    want = O.read_xyz_txt(os.path.join(scene, "point_world", "small_worldpoint_5_23_5.txt"))
    assert got.shape == want.shape and (np.abs(got - want) / (1 + np.linalg.norm(want, axis=1, keepdims=True))).max() <= 1e-12


@pytest.mark.parametrize("fault", ["missing", "other_size"])
def test_sharded_dropin_fails_on_every_rank_instead_of_hanging(tmp_path, golden_dir, mock_rccl, fault):
    """A rank whose depth frames cannot be read, or have another raster size than its peers', must not leave the others
    waiting in a collective that has no timeout: the ranks agree on (ok, H, W, bytes per value) first and ALL raise.  Three
    ranks, the fault in the LAST rank's block; the job has to come back non-zero well inside the deadline."""
    import time
    from PIL import Image
    scene = os.path.join(golden_dir, "scene3")
    for d in ("depth", "camera_pose"):
        shutil.copytree(os.path.join(scene, d), tmp_path / d)
    for d in ("point", "point_world", "ply"):
        os.makedirs(tmp_path / d)
    names, _q, _t = _r3d().read_pose_file(str(tmp_path / "camera_pose" / "image_colmap_simi_2.txt"))
    victim = tmp_path / "depth" / names[-1]
    if fault == "missing":
        os.remove(victim)
    else:
        Image.fromarray(np.zeros((10, 12), np.uint8), mode="L").save(victim)
    port = 29780 + (os.getpid() + (7 if fault == "missing" else 13)) % 100
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(SCRIPTS, "transfer", "camera_to_world.py")]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", R3D_RCCL_PATH=mock_rccl,
                                R3D_SHARE_GPU="1"))
    assert r.returncode != 0 and time.time() - t0 < 120, (r.returncode, time.time() - t0)
    text = r.stdout + r.stderr
    assert ("could not read their depth frames" in text) if fault == "missing" else ("depth rasters differ" in text), text[-3000:]
    assert not (tmp_path / "ply" / "small_035_p8.ply").exists()


@pytest.mark.parametrize("colour_format", ["png", "jpeg"])
def test_pixel_to_camera_script_writes_the_coloured_ply_when_the_image_exists(tmp_path, colour_format):
    """p2c:136 calls its writer with (points, ./img/24.png, ply path) -- the coloured writer's signature.  With ./img/24.png
    present the drop-in's main() writes `x y z R G B 0` rows under the uchar header of p2c:71-87; the oracle's formatter
    (pinned by the reference-generated coloured fixture in tests/test_oracle_golden.py) gives the expected bytes."""
    from PIL import Image
    for d in ("depth", "point", "ply", "img"):
        os.makedirs(tmp_path / d)
    rng = np.random.default_rng(4)
    depth = rng.integers(1, 256, (48, 64), dtype=np.uint8)
    rgb = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    Image.fromarray(np.stack([depth // 2, depth, depth // 3], 2).astype(np.uint8), "RGB").save(tmp_path / "depth" / "24.png")
    if colour_format == "png":
        Image.fromarray(rgb, "RGB").save(tmp_path / "img" / "24.png")
    else:       # AirSim's scene images are JPG: JPEG bytes under the name the script opens; the colours are what PIL decodes
        Image.fromarray(rgb, "RGB").save(tmp_path / "img" / "24.png", format="JPEG", quality=90)
        rgb = np.asarray(Image.open(tmp_path / "img" / "24.png").convert("RGB"))
    out = run_script("transfer/pixel_to_camera.py", str(tmp_path))
    assert "Write into .ply file Done." in out
    assert (tmp_path / "ply" / "24.ply").read_bytes() == O.format_ply_rgb(O.unproject(depth), rgb.reshape(-1, 3)).encode()


def test_camera_to_world_script_on_jpeg_depth_files(tmp_path):
    """Config 5's input flavour: AirSim writes its depth images as 3-channel JPG (airsim/main.cpp:1369-1392) and
    camera_to_world.py:160 reads depth with IMREAD_GRAYSCALE, which for a JPEG is libjpeg's grey output (the luma channel).
    The drop-in decodes them natively; every file it writes must equal the oracle's loops on the rasters libjpeg itself
    (through PIL's draft mode) gives for the same files -- camera txts and PLY byte for byte."""
    from PIL import Image
    rng = np.random.default_rng(41)
    F, H, W = 4, 40, 72
    for d in ("depth", "camera_pose", "point", "point_world", "ply"):
        os.makedirs(tmp_path / d)
    yy, xx = np.mgrid[0:H, 0:W]
    rasters = []
    with open(tmp_path / "camera_pose" / "image_colmap_simi_2.txt", "w") as f:
        f.write("id,tx,ty,tz,qx,qy,qz,qw,name,extra\n")
        for k in range(F):
            img = (np.stack([90 + 60 * np.sin(xx / 9.0 + k), 120 + 50 * np.cos(yy / 7.0), 100 + (xx + yy) % 80], 2)
                   + rng.normal(0, 6, (H, W, 3))).clip(0, 255).astype(np.uint8)
            p = tmp_path / "depth" / ("%03d.jpg" % k)
            Image.fromarray(img, "RGB").save(p, quality=(95, 80, 60, 90)[k], subsampling=(0, 2, 1, 2)[k])
            im = Image.open(p)
            im.draft("L", im.size)
            rasters.append(np.array(im))
            q, t = rng.normal(size=4), rng.normal(size=3) * 10
            f.write(",".join([str(k)] + [repr(float(x)) for x in t] + [repr(float(x)) for x in q] + ["%03d.jpg" % k, "x"]) + "\n")
    out = run_script("transfer/camera_to_world.py", str(tmp_path))
    assert "Write into .ply file Done." in out
    names, quats, ts = O.parse_pose_file(str(tmp_path / "camera_pose" / "image_colmap_simi_2.txt"))
    ref = tmp_path / "ref"
    os.makedirs(ref)
    xs, ys, zs = [], [], []
    for k in range(F):
        cam = str(ref / ("%03d.txt" % k))
        O.gentxtcord_loop(cam, rasters[k])
        O.get_pointdata_loop(cam, quats[k], ts[k], xs, ys, zs, str(ref / "world.txt"))
        assert (tmp_path / "point" / ("%03d.txt" % k)).read_bytes() == open(cam, "rb").read(), k
    O.genply_loop([xs, ys, zs], str(ref / "fused.ply"))
    assert (tmp_path / "ply" / "small_035_p8.ply").read_bytes() == open(ref / "fused.ply", "rb").read()
