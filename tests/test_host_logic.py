"""CPU: host-side logic of the package (pose maths, file parsers, native text writers) against
the oracle and the golden fixtures.  Nothing here needs a GPU."""
import importlib
import json
import os

import numpy as np
import pytest

from helpers import PKG, ROOT
from oracle import fusion_ref as O


@pytest.fixture(scope="module")
def R():
    return importlib.import_module(PKG)


def test_scipy_transfer_and_get_r_match_golden(R, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "poses.json")))
    for q, want in zip(g["quats_xyzw"], g["scipy_transfer"]):
        got = R.scipy_transfer(q)
        assert isinstance(got, np.matrix)
        np.testing.assert_array_equal(np.asarray(got), np.array(want))      # SciPy's arithmetic restated + same inverse: bit equal
        np.testing.assert_allclose(np.asarray(got), O.quat_to_rinv(q), atol=2e-15)
    for q, want in zip(g["get_r_wxyz_input"], g["get_r"]):
        np.testing.assert_allclose(np.asarray(R.get_r(q)), np.array(want), atol=2e-15)
    with pytest.raises(ValueError):
        R.scipy_transfer([0, 0, 0, 0])


def test_rotation_restatement_is_bitwise_scipy_and_the_package_does_not_import_scipy(R):
    """poses._rotation_matrix_xyzw restates SciPy's from_quat().as_matrix() operation for operation, so the drop-ins do
    not pay for importing scipy.spatial; when SciPy is importable here the two must agree bit for bit."""
    import subprocess, sys
    code = "import importlib, sys; importlib.import_module(%r); sys.exit(1 if 'scipy' in sys.modules else 0)" % PKG
    assert subprocess.run([sys.executable, "-c", code], cwd=ROOT).returncode == 0
    Rotation = pytest.importorskip("scipy.spatial.transform").Rotation
    P = importlib.import_module(PKG + ".poses")
    rng = np.random.default_rng(0)
    for _ in range(5000):
        q = rng.normal(size=4) * 10.0 ** rng.uniform(-3, 3)
        np.testing.assert_array_equal(P._rotation_matrix_xyzw(q), Rotation.from_quat(q).as_matrix())


def test_pose_table_layout(R):
    q = [[0.1, 0.2, 0.3, 0.9], [0, 0, 0, 1]]
    t = [[1, 2, 3], [4, 5, 6]]
    tab = R.pose_table(q, t)
    assert tab.shape == (2, 12) and tab.dtype == np.float64
    np.testing.assert_array_equal(tab[0, :9].reshape(3, 3), np.asarray(R.scipy_transfer(q[0])))
    np.testing.assert_array_equal(tab[:, 9:], np.array(t, dtype=np.float64))
    np.testing.assert_array_equal(tab[1, :9].reshape(3, 3), np.eye(3))
    with pytest.raises(ValueError):
        R.pose_table(q, t[:1])
    T = R.pose_to_T(tab[0, :9].reshape(3, 3), tab[0, 9:])
    p = np.array([1.0, 0.0, 2.0])
    np.testing.assert_allclose(T[:3, :3] @ p + T[:3, 3], O.se3_apply(p[None], tab[0, :9].reshape(3, 3), t[0])[0],
                               atol=1e-14)


def test_read_pose_file_matches_oracle_and_rejects_trailing_name(R, golden_dir, tmp_path):
    path = os.path.join(golden_dir, "scene3", "camera_pose", "image_colmap_simi_2.txt")
    names, quats, ts = R.read_pose_file(path)
    on, oq, ot = O.parse_pose_file(path)
    assert names == on
    np.testing.assert_array_equal(quats, oq)
    np.testing.assert_array_equal(ts, ot)
    bad = tmp_path / "bad.txt"
    bad.write_text("h\n1,0,0,0,0,0,0,1,a.png\n")
    with pytest.raises(ValueError):
        R.read_pose_file(str(bad))
    short = tmp_path / "short.txt"
    short.write_text("h\n1,0,0\n")
    with pytest.raises(ValueError):
        R.read_pose_file(str(short))
    empty = tmp_path / "empty.txt"
    empty.write_text("header only\n")
    n, q, t = R.read_pose_file(str(empty))
    assert n == [] and q.shape == (0, 4) and t.shape == (0, 3)


def test_get_T_and_write_T_round_trip(R, golden_dir, tmp_path):
    d = os.path.join(golden_dir, "icp_apply")
    T = R.get_T(os.path.join(d, "T_data.txt"))
    np.testing.assert_array_equal(T, np.array(json.load(open(os.path.join(d, "T_parsed.json")))))
    rng = np.random.default_rng(0)
    M = rng.normal(size=(4, 4)) * 10.0 ** rng.integers(-6, 6, size=(4, 4))
    R.write_T(str(tmp_path / "T.txt"), M)
    np.testing.assert_array_equal(R.get_T(str(tmp_path / "T.txt")), M)


def test_native_writers_match_reference_bytes(R, golden_dir, tmp_path):
    scene = os.path.join(golden_dir, "scene3")
    names, quats, ts = O.parse_pose_file(os.path.join(scene, "camera_pose", "image_colmap_simi_2.txt"))
    from PIL import Image
    depths = np.stack([np.array(Image.open(os.path.join(scene, "depth", n)).convert("L")) for n in names])
    # camera txt: X,Y from the oracle's fp64 (bit-equal to the reference), Z printed as the raw integer
    for k, n in enumerate(names):
        cam = O.unproject(depths[k])
        want = open(os.path.join(scene, "point", n[:-4] + ".txt"), "rb").read()
        assert R.cloud_io.format_xyz_txt(cam, z_raw=depths[k]) == want
        R.cloud_io.write_xyz_txt(str(tmp_path / "c.txt"), cam, z_raw=depths[k])
        assert (tmp_path / "c.txt").read_bytes() == want
    # world txt + PLY from the values the reference itself wrote (repr round-trips fp64)
    world = R.cloud_io.read_xyz_txt(os.path.join(scene, "point_world", "small_worldpoint_5_23_5.txt"))
    assert R.cloud_io.format_xyz_txt(world) == open(os.path.join(scene, "point_world",
                                                                 "small_worldpoint_5_23_5.txt"), "rb").read()
    np.testing.assert_array_equal(world, O.read_xyz_txt(os.path.join(scene, "point_world",
                                                                     "small_worldpoint_5_23_5.txt")))
    fused = O.fuse_frames(depths, quats, ts)
    assert R.cloud_io.format_ply(fused) == O.format_ply(fused).encode()
    R.cloud_io.write_ply(str(tmp_path / "p.ply"), fused)
    assert (tmp_path / "p.ply").read_bytes() == O.format_ply(fused).encode()
    np.testing.assert_array_equal(R.cloud_io.read_ply(os.path.join(scene, "ply", "small_035_p8.ply")),
                                  O.read_ply_vertices(os.path.join(scene, "ply", "small_035_p8.ply")))
    # append mode and the empty cloud
    R.cloud_io.write_xyz_txt(str(tmp_path / "a.txt"), world[:3])
    R.cloud_io.write_xyz_txt(str(tmp_path / "a.txt"), world[3:5], append=True)
    assert (tmp_path / "a.txt").read_bytes() == R.cloud_io.format_xyz_txt(world[:5])
    assert R.cloud_io.format_ply(np.zeros((0, 3))) == O.format_ply(np.zeros((0, 3))).encode()
    assert R.cloud_io.format_xyz_txt(np.zeros((0, 3), np.float32)) == b""


def test_native_formatters_fuzz_against_python(R):
    rng = np.random.default_rng(1)
    x = rng.normal(size=(20001, 3)) * 10.0 ** rng.integers(-9, 13, size=(20001, 3))
    x[0] = [0.0, -0.0, 0.00005]
    x[1] = [0.00015, 0.00025, -0.00005]
    x[2] = [1e15, 1e16, 123456789012345678.0]
    x[3] = [1e-4, 1e-5, 9.999999e-5]
    x[4] = [2.0 ** 40, 2.0 ** 41 + 0.5, 1e300]
    x[5] = [np.inf, -np.inf, 0.12345]
    for dt in (np.float64, np.float32):
        with np.errstate(over="ignore"):
            a = x.astype(dt)
        assert R.cloud_io.format_ply(a) == O.format_ply(a.astype(np.float64)).encode()
        want = "".join("%r,%r,%r\n" % (float(p[0]), float(p[1]), float(p[2])) for p in a).encode()
        assert R.cloud_io.format_xyz_txt(a) == want


def test_depth_readers(R, golden_dir, tmp_path):
    from PIL import Image
    p = os.path.join(golden_dir, "scene3", "depth", "000.png")
    np.testing.assert_array_equal(R.cloud_io.read_depth_gray(p), np.array(Image.open(p)))
    rgb = np.random.default_rng(0).integers(0, 256, (5, 7, 3), dtype=np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "c.png")
    bgr = R.cloud_io.read_depth_unchanged(str(tmp_path / "c.png"))
    np.testing.assert_array_equal(bgr[:, :, 1], rgb[:, :, 1])
    np.testing.assert_array_equal(bgr[:, :, 0], rgb[:, :, 2])
    with pytest.raises(FileNotFoundError):
        R.cloud_io.read_depth_gray(str(tmp_path / "missing.png"))


def test_type_checks_happen_before_any_device_work(R):
    with pytest.raises(TypeError):
        R.unproject(np.zeros((4, 4), np.int64))
    with pytest.raises(ValueError):
        R.unproject(np.zeros((4,), np.uint8))
    with pytest.raises(ValueError):
        R.cloud_io.write_xyz_txt("/tmp/x.txt", np.zeros((3, 2)))


def test_coloured_ply_matches_reference_bytes(R, golden_dir, tmp_path):
    """genply_noRGB's layout (p2c:55-91), pinned by running the reference with PIL's Image put into its namespace."""
    import json
    from PIL import Image
    g = json.load(open(os.path.join(golden_dir, "p2c_480x640.json")))
    depth = np.random.default_rng(g["seed"]).integers(1, 256, tuple(g["shape"]), dtype=np.uint8)
    cam = O.unproject(depth)[:6]
    rgb = np.array(Image.open(os.path.join(golden_dir, "p2c_rgb_2x3.png")).convert("RGB")).reshape(-1, 3)
    want = open(os.path.join(golden_dir, "p2c_first6_rgb.ply"), "rb").read()
    R.cloud_io.write_ply_rgb(str(tmp_path / "c.ply"), cam, rgb)
    assert (tmp_path / "c.ply").read_bytes() == want
    R.cloud_io.write_ply_rgb(str(tmp_path / "c32.ply"), cam.astype(np.float32), rgb)
    assert (tmp_path / "c32.ply").read_bytes() == want          # these values survive the f32 round trip at 4 decimals
    with pytest.raises(ValueError):
        R.cloud_io.write_ply_rgb(str(tmp_path / "bad.ply"), cam, rgb[:5])


def test_sfm2npy_helper(tmp_path, monkeypatch):
    """c2w:32-38 surface completeness: PFM (bottom-up float rows) -> npy of what cv.imread(path) gives with DEFAULT flags:
    uint8 H x W x 3 (no IMREAD_ANYDEPTH => saturate_cast<uchar>, round half to even; grey replicated; colour in BGR order;
    samples divided by |scale|).  The float samples stay reachable through read_pfm(..., unchanged=True)."""
    c2w = importlib.import_module(PKG + ".transfer.camera_to_world")
    monkeypatch.chdir(tmp_path)
    os.makedirs("pfm")
    os.makedirs("npy")
    img = np.array([[0.5, 1.5, 2.5, 3.49], [-3.0, 254.5, 255.5, 300.0], [7.0, 8.25, 9.75, 100.0]], np.float32)
    with open("pfm/a.pfm", "wb") as f:
        f.write(b"Pf\n4 3\n-1.0\n")
        f.write(img[::-1].astype("<f4").tobytes())
    assert c2w.sfm2npy("a") == "./npy/a.npy"
    got = np.load("npy/a.npy")
    want = np.array([[0, 2, 2, 3], [0, 254, 255, 255], [7, 8, 10, 100]], np.uint8)
    assert got.dtype == np.uint8 and got.shape == (3, 4, 3)
    for c in range(3):
        np.testing.assert_array_equal(got[:, :, c], want)
    np.testing.assert_array_equal(c2w.read_pfm("pfm/a.pfm", unchanged=True), img)
    # big-endian colour file with a scale factor: samples / |scale|, channels reversed to BGR
    rgb = np.stack([img, img + 10, img + 20], 2)
    with open("pfm/c.pfm", "wb") as f:
        f.write(b"PF\n4 3\n2.0\n")
        f.write((rgb[::-1] * 2).astype(">f4").tobytes())
    np.testing.assert_array_equal(c2w.read_pfm("pfm/c.pfm", unchanged=True), rgb[:, :, ::-1])
    got = c2w.read_pfm("pfm/c.pfm")
    np.testing.assert_array_equal(got[:, :, 2], want)                  # R is the last channel
    np.testing.assert_array_equal(got[:, :, 0], np.clip(np.rint(img + 20), 0, 255).astype(np.uint8))


def test_read_xyz_txt_takes_first_three_fields_and_names_bad_lines(R, tmp_path):
    """The reference reads data_p[0:3] of every line (c2w:97-98): extra fields are ignored, a missing final newline is
    fine, a malformed line is an error that names the line."""
    p = tmp_path / "a.txt"
    p.write_text("1.5,2,3\n4,5,6.25\n")
    np.testing.assert_array_equal(R.cloud_io.read_xyz_txt(str(p)), [[1.5, 2, 3], [4, 5, 6.25]])
    p.write_text("1.5,2,3\n4,5,6.25")                                  # no trailing newline
    np.testing.assert_array_equal(R.cloud_io.read_xyz_txt(str(p)), [[1.5, 2, 3], [4, 5, 6.25]])
    p.write_text("1,2,3,255,0,7\n4,5,6,1,2,3\n")                         # x,y,z,r,g,b rows: NOT two points per line
    np.testing.assert_array_equal(R.cloud_io.read_xyz_txt(str(p)), [[1, 2, 3], [4, 5, 6]])
    p.write_text("1,2,3\n\n4,5,6\n")                                     # blank line skipped
    np.testing.assert_array_equal(R.cloud_io.read_xyz_txt(str(p)), [[1, 2, 3], [4, 5, 6]])
    p.write_text("")
    assert R.cloud_io.read_xyz_txt(str(p)).shape == (0, 3)
    p.write_text("1,2,3\n4,5\n")
    with pytest.raises(ValueError, match="line 2"):
        R.cloud_io.read_xyz_txt(str(p))
    p.write_text("1,2,3\n4,x,6\n")
    with pytest.raises(ValueError, match="line 2"):
        R.cloud_io.read_xyz_txt(str(p))


def test_colmap_depth_prep_drop_in(tmp_path):
    """other_tools/data_transfer.py:5-16 (parity unpinned: OpenCV is absent): 640x480 grey uint8 .npy.  The reference's
    `cv2.resize(img, (640, 480), cv2.INTER_NEAREST)` passes the constant as `dst`, so OpenCV resizes BILINEARLY: that is the
    default here, in OpenCV's fixed point (checked against a float bilinear to within one grey level, against the exact
    box-filter case, against the identity); interpolation="nearest" is the index rule its comment intends.  Q14 grey weights."""
    from PIL import Image
    dt = importlib.import_module(PKG + ".other_tools.data_transfer")
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "a.png")
    out = dt.get_data(str(tmp_path / "a.png"), str(tmp_path / "a.npy"), interpolation="nearest")
    saved = np.load(tmp_path / "a.npy")
    assert saved.shape == (480, 640) and saved.dtype == np.uint8
    np.testing.assert_array_equal(out, saved)
    for (y, x) in ((0, 0), (479, 639), (100, 333), (240, 320)):
        sy, sx = min(int(np.floor(y * 37 / 480)), 36), min(int(np.floor(x * 53 / 640)), 52)
        r, g, b = (int(v) for v in rgb[sy, sx])
        assert saved[y, x] == (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14
    grey = np.full((5, 5, 3), 200, np.uint8)
    assert (dt.bgr_to_gray(grey) == 200).all()                   # the weights sum to 2^14
    big = rng.integers(0, 256, (960, 1280, 3), dtype=np.uint8)    # exact 2x downscale picks every other pixel
    np.testing.assert_array_equal(dt.resize_nearest(big, 640, 480), big[::2, ::2])
    # the default: what the reference's call does under OpenCV
    out = dt.get_data(str(tmp_path / "a.png"), str(tmp_path / "b.npy"))
    np.testing.assert_array_equal(out, dt.bgr_to_gray(dt.resize_linear_u8(rgb[..., ::-1], 640, 480)))
    assert not np.array_equal(out, saved)
    for (h, w) in ((37, 53), (1080, 1920), (100, 2000), (479, 641)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = dt.resize_linear_u8(img, 640, 480).astype(np.float64)
        fx, fy = (np.arange(640) + 0.5) * (w / 640) - 0.5, (np.arange(480) + 0.5) * (h / 480) - 0.5
        x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
        ax, ay = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
        xa, xb, ya, yb = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1), np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
        f = img.astype(np.float64)
        ref = (f[ya][:, xa] * (1 - ax) + f[ya][:, xb] * ax) * (1 - ay) + (f[yb][:, xa] * (1 - ax) + f[yb][:, xb] * ax) * ay
        assert np.abs(got - ref).max() < 1.0
    b4 = big.astype(np.int64)
    np.testing.assert_array_equal(dt.resize_linear_u8(big, 640, 480),
                                  ((b4[0::2, 0::2] + b4[0::2, 1::2] + b4[1::2, 0::2] + b4[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    same = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    np.testing.assert_array_equal(dt.resize_linear_u8(same, 640, 480), same)
    flat = np.full((33, 71), 137, np.uint8)
    assert (dt.resize_linear_u8(flat, 640, 480) == 137).all()     # the fixed-point weights of a pixel pair sum to 2048
    with pytest.raises(ValueError):
        dt.get_data(str(tmp_path / "a.png"), str(tmp_path / "c.npy"), interpolation="cubic")


def test_native_rgb_png_batch_matches_pil(R, tmp_path):
    """f3 for the RGBD path: 8-bit RGB / RGBA / grey PNGs -> [F,H,W,3] R,G,B bytes, byte-identical to PIL's decode;
    palette and 16-bit files go through the fallback; mismatched sizes are an error."""
    from PIL import Image
    rng = np.random.default_rng(12)
    H, W = 37, 53
    imgs, paths = [], []
    for k in range(5):
        a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        a[k::7] = a[k::7] // 2 * 2          # some smooth rows so every PNG filter type gets used
        p = tmp_path / ("c%d.png" % k)
        Image.fromarray(a, "RGB").save(p, compress_level=1 + k)
        imgs.append(a)
        paths.append(str(p))
    got = R.cloud_io.read_rgb_batch(paths)
    assert got.shape == (5, H, W, 3) and got.dtype == np.uint8
    np.testing.assert_array_equal(got, np.stack(imgs))
    rgba = np.dstack([imgs[0], rng.integers(0, 256, (H, W), dtype=np.uint8)])
    Image.fromarray(rgba, "RGBA").save(tmp_path / "a.png")
    grey = rng.integers(0, 256, (H, W), dtype=np.uint8)
    Image.fromarray(grey, "L").save(tmp_path / "g.png")
    got2 = R.cloud_io.read_rgb_batch([str(tmp_path / "a.png"), str(tmp_path / "g.png")])
    np.testing.assert_array_equal(got2[0], imgs[0])                   # alpha dropped
    np.testing.assert_array_equal(got2[1], np.repeat(grey[..., None], 3, axis=2))
    Image.fromarray(imgs[1], "RGB").convert("P").save(tmp_path / "p.png")   # palette: PIL fallback, still [.,.,3]
    got3 = R.cloud_io.read_rgb_batch([str(tmp_path / "p.png")])
    np.testing.assert_array_equal(got3[0], np.array(Image.open(tmp_path / "p.png").convert("RGB")))
    Image.fromarray(imgs[0][:10], "RGB").save(tmp_path / "small.png")
    with pytest.raises(Exception):
        R.cloud_io.read_rgb_batch([paths[0], str(tmp_path / "small.png")])
    with pytest.raises(FileNotFoundError):
        R.cloud_io.read_rgb_batch([str(tmp_path / "missing.png")])
    out = np.empty((5, H, W, 3), np.uint8)
    assert R.cloud_io.read_rgb_batch(paths, out=out) is out


def test_16_bit_grey_depth_png_keeps_its_high_byte_like_opencv(R, tmp_path):
    """cv.imread(path, IMREAD_GRAYSCALE) (c2w:160) reads a 16-bit PNG through libpng's strip_16: v >> 8.  PIL's 'L'
    conversion would clip at 255 instead -- a different depth map.  (Parity unpinned: OpenCV is absent here; the rule is
    libpng's documented transformation.)"""
    from PIL import Image
    try:
        import cv2  # noqa: F401
        pytest.skip("OpenCV present: its own decode is used")
    except ImportError:
        pass
    rng = np.random.default_rng(3)
    v = rng.integers(0, 65536, (21, 34), dtype=np.uint16)
    paths = []
    for k in range(3):
        p = tmp_path / ("d%d.png" % k)
        Image.fromarray((v + k).astype(np.uint16)).save(p)
        paths.append(str(p))
    np.testing.assert_array_equal(R.cloud_io.read_depth_gray(paths[0]), (v >> 8).astype(np.uint8))
    got = R.cloud_io.read_depth_batch(paths)
    assert got.dtype == np.uint8 and got.shape == (3, 21, 34)
    for k in range(3):
        np.testing.assert_array_equal(got[k], ((v + k).astype(np.uint16) >> 8).astype(np.uint8))


def test_native_txt_and_ply_parsers_agree_with_python_float(R, tmp_path):
    """f1's 'fast X,Y,Z parser' (r3d_parse_xyz_text): correctly rounded like float() on every spelling repr() and '%.4f'
    produce plus the ones people type (exponents, '+', blanks, CRLF, nan/inf, overflow, subnormals); anything outside the
    subset it shares with float() (underscores ...) is handed to the Python parser, so results never differ."""
    rng = np.random.default_rng(11)
    vals = np.concatenate([rng.normal(size=3000) * 10.0 ** rng.integers(-30, 30, 3000),
                           rng.integers(-10**6, 10**6, 300).astype(np.float64),
                           np.array([0.0, -0.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf, -np.inf, np.nan,
                                     0.1, 1 / 3, 1e22, 1e23, 123456789012345678.0, 2 ** 53 + 2.0, 9007199254740993.0])])
    vals = np.resize(vals, (vals.size // 3) * 3).reshape(-1, 3)
    p = tmp_path / "v.txt"
    R.cloud_io.write_xyz_txt(str(p), vals)                              # repr() spellings
    got = R.cloud_io.read_xyz_txt(str(p))
    np.testing.assert_array_equal(got.view(np.uint64), vals.view(np.uint64))        # bit for bit, NaN and -0.0 included
    want = [[float(v) for v in line.split(',')[:3]] for line in p.read_text().split('\n') if line.strip()]
    np.testing.assert_array_equal(got.view(np.uint64), np.array(want).view(np.uint64))
    odd = ("1e3,+2.5E-2, 3 \r\n  -4,5.,.5,9,9\n\n 1e400,-1e400,1e-400\nNaN,iNf,-infinity\n"
           "0.1000000000000000055511151231257827,0.30000000000000004,9007199254740993\n7,8,9")
    p.write_text(odd)
    want = np.array([[float(v) for v in line.split(',')[:3]] for line in odd.split('\n') if line.strip()])
    np.testing.assert_array_equal(R.cloud_io.read_xyz_txt(str(p)).view(np.uint64), want.view(np.uint64))
    p.write_text("1_000,2,3\n4,5,6\n")                                  # float() takes underscores, from_chars does not
    np.testing.assert_array_equal(R.cloud_io.read_xyz_txt(str(p)), [[1000, 2, 3], [4, 5, 6]])
    for bad in ("1,2\n", "1,,3\n", "1,2,x\n", "1,2,3\n4;5;6\n", "+-1,2,3\n"):
        p.write_text(bad)
        with pytest.raises(ValueError):
            R.cloud_io.read_xyz_txt(str(p))
    # big file: several parser threads, line boundaries in odd places
    big = rng.normal(size=(200001, 3)) * 100
    R.cloud_io.write_xyz_txt(str(p), big)
    np.testing.assert_array_equal(R.cloud_io.read_xyz_txt(str(p)), big)
    q = tmp_path / "v.ply"
    R.cloud_io.write_ply(str(q), big)
    back = R.cloud_io.read_ply(str(q))
    assert back.shape == big.shape and np.abs(back - big).max() <= 0.5001e-4
    np.testing.assert_array_equal(back[:100], np.array([[float("%.4f" % v) for v in row] for row in big[:100]]))
    rgb = rng.integers(0, 256, (1000, 3), dtype=np.uint8)
    R.cloud_io.write_ply_rgb(str(q), big[:1000], rgb)                   # colour columns are ignored
    np.testing.assert_array_equal(R.cloud_io.read_ply(str(q)), back[:1000])
    q.write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nelement face 1\nend_header\n1 2 3\n4 5 6\n3 0 1 1\n")
    np.testing.assert_array_equal(R.cloud_io.read_ply(str(q)), [[1, 2, 3], [4, 5, 6]])      # rows after the vertices are not vertices


def test_native_parser_fast_path_is_correctly_rounded_on_a_million_numbers(R, tmp_path):
    """The parser's fast path (64-bit digits x exact power of ten in extended precision, midpoints sent to the exact path)
    against float() on the spellings that take it: repr() of ordinary magnitudes, '%.4f', '%.17g', '%.19g' (19 digits),
    short decimals, and the classic double-rounding traps."""
    rng = np.random.default_rng(99)
    x = rng.normal(size=250000) * 10.0 ** rng.integers(-8, 9, 250000)
    toks = [repr(float(v)) for v in x[:90000]] + ["%.4f" % v for v in x[90000:150000]] + ["%.17g" % v for v in x[150000:200000]] \
        + ["%.19g" % v for v in x[200000:230000]] + ["%.3e" % v for v in x[230000:250000]]
    toks += ["9007199254740993", "9007199254740992.5", "9007199254740993.0000001", "0.500000000000000166533453693773481063544750213623046875",
             "1.00000000000000011102230246251565404236316680908203125", "1.00000000000000011102230246251565404236316680908203124",
             "4503599627370496.5", "4503599627370497.5", "0.1", "123456789012345678", "1234567890123456789", "12345678901234567890",
             "1e27", "1e-27", "9999999999999999999e27", "1e28", "0.000", "-0.0", "000012.500", "1.e2", ".5e1", "5E-1"]
    toks += ["%d.%04d5" % (a, b) for a, b in zip(rng.integers(0, 5000, 30000), rng.integers(0, 10000, 30000))]     # ...5 ties at 1e-5
    while len(toks) % 3:
        toks.append("1")
    p = tmp_path / "m.txt"
    p.write_text("".join("%s,%s,%s\n" % tuple(toks[i:i + 3]) for i in range(0, len(toks), 3)))
    got = R.cloud_io.read_xyz_txt(str(p)).reshape(-1)
    want = np.array([float(t) for t in toks])
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    assert bad.size == 0, [(toks[k], got[k], want[k]) for k in bad[:5]]


def test_depth_maps_saved_in_colour_decode_natively_when_their_channels_agree(R, tmp_path):
    """A depth map written as an RGB(A) PNG with R = G = B (how p2c's input is shaped) is its own grey value under every
    colour -> grey rule, so the native batch decoder takes it (no per-file PIL / OpenCV round trip); one pixel with differing
    channels and the file goes to the fallback, whose conversion rule is its own."""
    from PIL import Image
    rng = np.random.default_rng(21)
    H, W = 40, 56
    greys, paths = [], []
    for k, mode in enumerate(("RGB", "RGBA", "LA", "L", "RGB")):
        g = rng.integers(0, 256, (H, W), dtype=np.uint8)
        if mode == "RGB":
            img = Image.fromarray(np.stack([g, g, g], 2), "RGB")
        elif mode == "RGBA":
            img = Image.fromarray(np.stack([g, g, g, rng.integers(0, 256, (H, W), dtype=np.uint8)], 2), "RGBA")
        elif mode == "LA":
            img = Image.fromarray(np.stack([g, rng.integers(0, 256, (H, W), dtype=np.uint8)], 2), "LA")
        else:
            img = Image.fromarray(g, "L")
        p = tmp_path / ("d%d.png" % k)
        img.save(p)
        greys.append(g)
        paths.append(str(p))
    out = np.empty((5, H, W), np.uint8)
    lib = R.load_library()
    import ctypes as C
    arr = (C.c_char_p * 5)(*[os.fsencode(p) for p in paths])
    assert lib.r3d_png_gray_decode_batch(arr, 5, out.ctypes.data, H, W, 8) == 0          # all five natively
    np.testing.assert_array_equal(out, np.stack(greys))
    np.testing.assert_array_equal(R.cloud_io.read_depth_batch(paths), np.stack(greys))
    real = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)                                  # real colour: not ours to convert
    Image.fromarray(real, "RGB").save(tmp_path / "c.png")
    arr1 = (C.c_char_p * 1)(os.fsencode(str(tmp_path / "c.png")))
    assert lib.r3d_png_gray_decode_batch(arr1, 1, out.ctypes.data, H, W, 8) != 0
    got = R.cloud_io.read_depth_batch([paths[0], str(tmp_path / "c.png")])                  # falls back per file, still works
    np.testing.assert_array_equal(got[0], greys[0])
    assert got.shape == (2, H, W)


def _gray_opencv_png(r, g, b):
    """libpng's 8-bit rgb_to_gray as OpenCV's PNG reader requests it (weights 0.299, 0.587 -> 15-bit integers, truncated twice)."""
    rc, gc = 29900 * 32768 // 100000, 58700 * 32768 // 100000
    r, g, b = (np.asarray(c).astype(np.int64) for c in (r, g, b))
    v = (rc * r + gc * g + (32768 - rc - gc) * b) >> 15
    return np.where((r == g) & (g == b), r, v).astype(np.uint8)


def _gray_cvtcolor(r, g, b):
    r, g, b = (np.asarray(c).astype(np.int64) for c in (r, g, b))
    return ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.uint8)


def test_colour_to_grey_rules_known_answers_and_sweep(R):
    """f3: the two integer colour -> grey rules OpenCV has (include/r3d.h R3D_GRAY_*), cv2 being absent: known answers worked
    out by hand from the two libraries' formulas, the rounding edges, and 200k random triples against an independent NumPy
    restatement.  The three candidates (libpng-as-OpenCV-calls-it, cvtColor, PIL 'L') really are three different maps."""
    kat = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [1, 1, 1], [254, 255, 255], [12, 200, 77],
                    [0, 1, 0], [3, 0, 0], [0, 0, 9], [128, 127, 128]], np.uint8)
    # (9797 R + 19234 G + 3737 B) >> 15, equal channels kept           (4899 R + 9617 G + 1868 B + 8192) >> 14
    want_png = [76, 149, 29, 255, 0, 1, 254, 129, 0, 0, 1, 127]       # 2498235>>15, 4904670>>15, 952935>>15, ...
    want_cvt = [76, 150, 29, 255, 0, 1, 255, 130, 1, 1, 1, 127]       # 1257437>>14, 2460527>>14, 484532>>14, ...
    np.testing.assert_array_equal(R.cloud_io.rgb_to_gray(kat, "opencv_png"), want_png)
    np.testing.assert_array_equal(R.cloud_io.rgb_to_gray(kat, "cvtcolor"), want_cvt)
    rng = np.random.default_rng(77)
    px = rng.integers(0, 256, (200000, 3), dtype=np.uint8)
    px[:5000, 1] = px[:5000, 0]                                        # many near-grey pixels: the equal-channel branch and its edge
    px[:2500, 2] = px[:2500, 0]
    r, g, b = px[:, 0], px[:, 1], px[:, 2]
    got_png, got_cvt = R.cloud_io.rgb_to_gray(px, "opencv_png"), R.cloud_io.rgb_to_gray(px, "cvtcolor")
    np.testing.assert_array_equal(got_png, _gray_opencv_png(r, g, b))
    np.testing.assert_array_equal(got_cvt, _gray_cvtcolor(r, g, b))
    r, g, b = (c.astype(np.int64) for c in (r, g, b))
    pil_l = ((19595 * r + 38470 * g + 7471 * b + 32768) >> 16).astype(np.uint8)
    assert (got_png != got_cvt).mean() > 0.3 and (got_png != pil_l).mean() > 0.3 and (got_cvt != pil_l).mean() > 0.001
    assert np.abs(got_png.astype(int) - got_cvt).max() == 1
    # every exact tie of the cvtColor rounding: (4899 R + 9617 G + 1868 B) mod 16384 == 8192 rounds UP
    rr, gg, bb = np.meshgrid(np.arange(256), np.arange(256), np.arange(0, 256, 5), indexing="ij")
    s = 4899 * rr + 9617 * gg + 1868 * bb
    tie = (s % 16384) == 8192
    if tie.any():
        t = np.stack([rr[tie], gg[tie], bb[tie]], 1).astype(np.uint8)
        np.testing.assert_array_equal(R.cloud_io.rgb_to_gray(t, "cvtcolor"), (s[tie] // 16384 + 1).astype(np.uint8))
    rgba = np.concatenate([px[:1000], rng.integers(0, 256, (1000, 1), dtype=np.uint8)], 1)      # alpha is ignored
    np.testing.assert_array_equal(R.cloud_io.rgb_to_gray(rgba, "cvtcolor"), got_cvt[:1000])
    L = importlib.import_module(R.__name__ + "._lib")
    with pytest.raises(R.R3DError):
        L.check(R.load_library().r3d_rgb_to_gray_u8(px.ctypes.data, 10, 3, 7, px.ctypes.data))


def test_colour_depth_files_take_the_opencv_rules_not_pils(R, tmp_path, monkeypatch):
    """cv.imread(path, IMREAD_GRAYSCALE) (c2w:160) on files that really are colour: PNG -> libpng's rule, natively, in the
    batch decoder too (8- and 16-bit, with and without alpha); BMP -> the cvtColor rule; JPEG -> refused with the fallback
    named, accepted on request."""
    from PIL import Image
    try:
        import cv2  # noqa: F401
        pytest.skip("OpenCV present: its own decode is used")
    except ImportError:
        pass
    L = importlib.import_module(R.__name__ + "._lib")
    rng = np.random.default_rng(5)
    H, W = 33, 47
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    Image.fromarray(rgb, "RGB").save(tmp_path / "c.png")
    Image.fromarray(np.dstack([rgb, rng.integers(0, 256, (H, W), dtype=np.uint8)]), "RGBA").save(tmp_path / "ca.png")
    want = _gray_opencv_png(r, g, b)
    for name in ("c.png", "ca.png"):
        np.testing.assert_array_equal(R.cloud_io.read_depth_gray(str(tmp_path / name)), want)
        np.testing.assert_array_equal(R.cloud_io.read_depth_gray(str(tmp_path / name), rule="cvtcolor"), _gray_cvtcolor(r, g, b))
    got = R.cloud_io.read_depth_batch([str(tmp_path / "c.png"), str(tmp_path / "ca.png")])
    np.testing.assert_array_equal(got, np.stack([want, want]))
    monkeypatch.setenv("R3D_GRAY_RULE", "cvtcolor")
    np.testing.assert_array_equal(R.cloud_io.read_depth_batch([str(tmp_path / "c.png")])[0], _gray_cvtcolor(r, g, b))
    monkeypatch.delenv("R3D_GRAY_RULE")
    assert (want != np.array(Image.open(tmp_path / "c.png").convert("L"))).mean() > 0.3       # PIL's 'L' is another map
    # 16-bit RGB PNG (hand-made: PIL does not write them): libpng converts at 16 bits, rounds, THEN keeps the high byte
    import struct
    import zlib
    v16 = rng.integers(0, 65536, (H, W, 3), dtype=np.uint16)
    v16[:4] = v16[:4, :, :1]                                                                   # some equal-channel rows
    raw = b"".join(b"\x00" + v16[y].astype(">u2").tobytes() for y in range(H))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 16, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    (tmp_path / "c16.png").write_bytes(png)
    r6, g6, b6 = (v16[..., k].astype(np.int64) for k in range(3))
    w16 = np.where((r6 == g6) & (g6 == b6), r6, (9797 * r6 + 19234 * g6 + 3737 * b6 + 16384) >> 15)
    np.testing.assert_array_equal(R.cloud_io.read_depth_gray(str(tmp_path / "c16.png")), (w16 >> 8).astype(np.uint8))
    # BMP: decoded in colour, converted by cvtColor's rule
    Image.fromarray(rgb, "RGB").save(tmp_path / "c.bmp")
    np.testing.assert_array_equal(R.cloud_io.read_depth_gray(str(tmp_path / "c.bmp")), _gray_cvtcolor(r, g, b))
    # JPEG: sequential files decode natively (libjpeg's grey output = the luma channel; tests below pin the bytes); a
    # progressive one is refused by default, with the way out in the message
    Image.fromarray(rgb, "RGB").save(tmp_path / "c.jpg", quality=95)
    j = R.cloud_io.read_depth_gray(str(tmp_path / "c.jpg"))
    assert j.shape == (H, W) and j.dtype == np.uint8 and np.abs(j.astype(int) - _gray_cvtcolor(r, g, b)).mean() < 12
    np.testing.assert_array_equal(R.cloud_io.read_depth_batch([str(tmp_path / "c.jpg")])[0], j)
    Image.fromarray(rgb, "RGB").save(tmp_path / "p.jpg", quality=95, progressive=True)
    with pytest.raises(R.cloud_io.UnsupportedDepthFormat) as e:
        R.cloud_io.read_depth_gray(str(tmp_path / "p.jpg"))
    assert e.value.code == L.ERR_UNSUPPORTED and "allow_pil_jpeg" in str(e.value) and "R3D_ALLOW_PIL_JPEG" in str(e.value) and "progressive" in str(e.value)
    with pytest.raises(R.cloud_io.UnsupportedDepthFormat):
        R.cloud_io.read_depth_batch([str(tmp_path / "p.jpg")])
    jp = R.cloud_io.read_depth_gray(str(tmp_path / "p.jpg"), allow_pil_jpeg=True)
    assert jp.shape == (H, W) and np.abs(jp.astype(int) - j).max() <= 2          # the same image, another coding
    monkeypatch.setenv("R3D_ALLOW_PIL_JPEG", "1")
    np.testing.assert_array_equal(R.cloud_io.read_depth_batch([str(tmp_path / "p.jpg")])[0], jp)
    # a palette PNG is not guessed at either
    Image.fromarray(rgb, "RGB").quantize(16).save(tmp_path / "p.png")
    with pytest.raises(R.cloud_io.UnsupportedDepthFormat):
        R.cloud_io.read_depth_gray(str(tmp_path / "p.png"))


def test_gamma_tagged_colour_png_is_refused_not_approximated(R, tmp_path):
    """libpng converts colour to grey in LINEAR light when the file says what its gamma is (gAMA / sRGB / iCCP): that path is
    not restated, so such a file with real colour in it is refused under the OpenCV-PNG rule (and still converts under the
    cvtColor rule, which knows no gamma); the same file with R = G = B everywhere decodes, as it does in libpng."""
    from PIL import Image
    from PIL.PngImagePlugin import PngInfo
    try:
        import cv2  # noqa: F401
        pytest.skip("OpenCV present: its own decode is used")
    except ImportError:
        pass
    rng = np.random.default_rng(9)
    rgb = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    g = rng.integers(0, 256, (20, 30), dtype=np.uint8)
    import struct
    tag = PngInfo()
    tag.add(b"gAMA", struct.pack(">I", 45455))                                           # file gamma 1 / 2.2
    Image.fromarray(rgb, "RGB").save(tmp_path / "tagged.png", pnginfo=tag)
    Image.fromarray(np.dstack([g, g, g]), "RGB").save(tmp_path / "tagged_grey.png", pnginfo=tag)
    assert b"gAMA" in (tmp_path / "tagged.png").read_bytes()[:200]
    with pytest.raises(R.cloud_io.UnsupportedDepthFormat) as e:
        R.cloud_io.read_depth_gray(str(tmp_path / "tagged.png"))
    assert "gAMA" in str(e.value) and "linear" in str(e.value)
    with pytest.raises(R.cloud_io.UnsupportedDepthFormat):
        R.cloud_io.read_depth_batch([str(tmp_path / "tagged.png")])
    np.testing.assert_array_equal(R.cloud_io.read_depth_gray(str(tmp_path / "tagged.png"), rule="cvtcolor"),
                                  _gray_cvtcolor(rgb[..., 0], rgb[..., 1], rgb[..., 2]))
    np.testing.assert_array_equal(R.cloud_io.read_depth_gray(str(tmp_path / "tagged_grey.png")), g)


def _pil_luma(path):
    """libjpeg(-turbo) asked for greyscale output through PIL's draft mode: what OpenCV's JPEG reader requests for IMREAD_GRAYSCALE."""
    from PIL import Image
    im = Image.open(path)
    im.draft("L", im.size)
    return np.array(im)


@pytest.mark.parametrize("hw", [(1, 1), (8, 8), (16, 16), (37, 53), (100, 300), (384, 1280), (1080, 1920)])
def test_native_jpeg_grey_equals_libjpeg_byte_for_byte(R, tmp_path, hw):
    """f3 / config 5's depth files: csrc/r3d_jpeg.cpp restates libjpeg's grey output (luma alone, "islow" integer IDCT) -- the
    request OpenCV's JPEG reader makes for IMREAD_GRAYSCALE (c2w:160 on AirSim's JPG depth).  Pinned against libjpeg-turbo itself
    (PIL's draft('L') decode): every raster byte for byte, over sizes that are not multiples of the MCU, qualities 5..100, the
    three usual chroma subsamplings and grey files, optimised Huffman tables, restart intervals; batches too."""
    from PIL import Image
    H, W = hw
    rng = np.random.default_rng(H * 7 + W)
    yy, xx = np.mgrid[0:H, 0:W]
    img = (np.stack([128 + 100 * np.sin(xx / 17.0 + yy / 9.0), 128 + 90 * np.cos(xx / 5.0), 100 + yy % 97], 2)
           + rng.normal(0, 15, (H, W, 3))).clip(0, 255).astype(np.uint8)
    noise = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    big = H * W > 500_000
    variants = [dict(quality=q, subsampling=sub) for q in ((75, 95) if big else (5, 30, 75, 95, 100)) for sub in (0, 1, 2)]
    variants += [dict(quality=90, optimize=True), dict(quality=85, restart_marker_blocks=7), dict(quality=85, restart_marker_rows=1, subsampling=2)]
    paths = []
    for k, kw in enumerate(variants):
        p = str(tmp_path / ("v%d.jpg" % k))
        try:
            Image.fromarray(noise if k % 5 == 4 else img, "RGB").save(p, **kw)
        except (TypeError, ValueError, OSError):          # an older Pillow without the restart options
            continue
        paths.append(p)
    for k, q in enumerate((20, 90)):
        p = str(tmp_path / ("g%d.jpg" % k))
        Image.fromarray(img[..., 0], "L").save(p, quality=q)
        paths.append(p)
    assert len(paths) >= 9
    for p in paths:
        np.testing.assert_array_equal(R.cloud_io.read_depth_gray(p), _pil_luma(p), err_msg=p)
    for backend in ("native", "pil", None):          # the library's decoder, libjpeg-turbo on a thread pool, the default choice
        got = R.cloud_io.read_depth_batch(paths, jpeg_backend=backend)
        assert got.shape == (len(paths), H, W)
        for k, p in enumerate(paths):
            np.testing.assert_array_equal(got[k], _pil_luma(p), err_msg="%s %s" % (backend, p))


def test_native_jpeg_refuses_what_it_does_not_restate(R, tmp_path):
    from PIL import Image
    import ctypes as C
    L = importlib.import_module(R.__name__ + "._lib")
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    lib = R.load_library()
    h, w = C.c_int(), C.c_int()
    Image.fromarray(img, "RGB").save(tmp_path / "p.jpg", progressive=True)
    Image.fromarray(img, "RGB").convert("CMYK").save(tmp_path / "c.jpg")
    (tmp_path / "t.jpg").write_bytes(b"\xff\xd8\xff\xe0\x00\x10JFIF")
    (tmp_path / "n.jpg").write_bytes(b"not a jpeg at all")
    assert lib.r3d_jpeg_gray_info(os.fsencode(str(tmp_path / "p.jpg")), C.byref(h), C.byref(w)) == L.ERR_UNSUPPORTED
    assert (h.value, w.value) == (40, 56)                                  # the size is still reported
    assert lib.r3d_jpeg_gray_info(os.fsencode(str(tmp_path / "c.jpg")), C.byref(h), C.byref(w)) == L.ERR_UNSUPPORTED
    assert lib.r3d_jpeg_gray_info(os.fsencode(str(tmp_path / "t.jpg")), C.byref(h), C.byref(w)) == L.ERR_INVALID
    assert lib.r3d_jpeg_gray_info(os.fsencode(str(tmp_path / "n.jpg")), C.byref(h), C.byref(w)) == L.ERR_INVALID
    assert lib.r3d_jpeg_gray_info(os.fsencode(str(tmp_path / "missing.jpg")), C.byref(h), C.byref(w)) == L.ERR_INVALID
    # truncated entropy data: zeros are fed behind the end like libjpeg does -- no crash, a raster comes back
    Image.fromarray(img, "RGB").save(tmp_path / "ok.jpg", quality=90)
    data = (tmp_path / "ok.jpg").read_bytes()
    (tmp_path / "cut.jpg").write_bytes(data[: len(data) * 2 // 3])
    out = np.empty((40, 56), np.uint8)
    arr = (C.c_char_p * 1)(os.fsencode(str(tmp_path / "cut.jpg")))
    assert lib.r3d_jpeg_gray_decode_batch(arr, 1, out.ctypes.data, 40, 56) in (L.OK, L.ERR_INVALID)
    arr = (C.c_char_p * 1)(os.fsencode(str(tmp_path / "ok.jpg")))
    assert lib.r3d_jpeg_gray_decode_batch(arr, 1, out.ctypes.data, 41, 56) == L.ERR_INVALID      # another size than the batch's


@pytest.mark.parametrize("hw", [(1, 1), (2, 3), (4, 6), (8, 8), (17, 33), (61, 83), (144, 256), (480, 640), (1080, 1920)])
def test_native_jpeg_colour_equals_pil_byte_for_byte(R, tmp_path, hw):
    """The colour planes of the RGBD path: the reference opens them with PIL (p2c:58-60), AirSim writes them as JPG.  The native
    decoder (islow IDCT on every component, libjpeg's fancy chroma upsampling, jdcolor's fixed-point YCbCr->RGB) gives PIL's
    bytes: sizes that are not multiples of the MCU (and ones whose chroma rows hold one or two samples: replication instead of the
    triangle filter), qualities, 4:4:4 / 4:2:2 / 4:2:0, grey files, restart intervals; batches through read_rgb_batch."""
    from PIL import Image
    H, W = hw
    rng = np.random.default_rng(H * 11 + W)
    yy, xx = np.mgrid[0:H, 0:W]
    img = (np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 5.0), 128 + 90 * np.cos(xx / 3.0), 100 + yy % 97], 2)
           + rng.normal(0, 25, (H, W, 3))).clip(0, 255).astype(np.uint8)
    noise = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    big = H * W > 500_000
    variants = [dict(quality=q, subsampling=sub) for q in ((85,) if big else (10, 85, 100)) for sub in (0, 1, 2)]
    variants += [dict(quality=90, optimize=True, subsampling=2), dict(quality=85, restart_marker_blocks=5, subsampling=1),
                 dict(quality=85, restart_marker_rows=1, subsampling=2)]
    paths = []
    for k, kw in enumerate(variants):
        p = str(tmp_path / ("v%d.jpg" % k))
        try:
            Image.fromarray(noise if k % 4 == 3 else img, "RGB").save(p, **kw)
        except (TypeError, ValueError, OSError):          # an older Pillow without the restart options / tiny optimised files
            continue
        paths.append(p)
    p = str(tmp_path / "g.jpg")
    Image.fromarray(img[..., 0], "L").save(p, quality=80)
    paths.append(p)
    assert len(paths) >= 4
    for backend in ("native", "pil", None):
        got = R.cloud_io.read_rgb_batch(paths, jpeg_backend=backend)
        assert got.shape == (len(paths), H, W, 3) and got.dtype == np.uint8
        for k, p in enumerate(paths):
            np.testing.assert_array_equal(got[k], np.asarray(Image.open(p).convert("RGB")), err_msg="%s %s" % (backend, p))


def test_native_jpeg_colour_refusals_fall_back_to_pil(R, tmp_path):
    """What the native colour decoder does not restate goes to PIL -- for colour that IS the reference's reader, so the bytes
    are the reference's either way; a corrupt or missing file is an error, not a fallback."""
    from PIL import Image
    import ctypes as C
    L = importlib.import_module(R.__name__ + "._lib")
    lib = R.load_library()
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    Image.fromarray(img, "RGB").save(tmp_path / "p.jpg", progressive=True)
    h, w, c = C.c_int(), C.c_int(), C.c_int()
    assert lib.r3d_jpeg_rgb_info(os.fsencode(str(tmp_path / "p.jpg")), C.byref(h), C.byref(w), C.byref(c)) == L.ERR_UNSUPPORTED
    assert (h.value, w.value, c.value) == (24, 40, 3)
    got = R.cloud_io.read_rgb_batch([str(tmp_path / "p.jpg")])
    np.testing.assert_array_equal(got[0], np.asarray(Image.open(tmp_path / "p.jpg").convert("RGB")))
    # a sequential first file and a progressive second one: the batch decoder refuses, PIL takes the list
    Image.fromarray(img, "RGB").save(tmp_path / "s.jpg")
    got = R.cloud_io.read_rgb_batch([str(tmp_path / "s.jpg"), str(tmp_path / "p.jpg")])
    for k, n in enumerate(("s.jpg", "p.jpg")):
        np.testing.assert_array_equal(got[k], np.asarray(Image.open(tmp_path / n).convert("RGB")))
    # formats that are neither PNG nor JPEG go to PIL, alone or mixed with others (the reference opens any colour image with
    # Image.open, p2c:58-60)
    Image.fromarray(img, "RGB").save(tmp_path / "c.bmp")
    Image.fromarray(img, "RGB").save(tmp_path / "c.tiff")
    Image.fromarray(img, "RGB").save(tmp_path / "c.png")
    for names in (["c.bmp"], ["c.tiff"], ["c.bmp", "c.tiff"], ["c.png", "c.bmp"]):
        got = R.cloud_io.read_rgb_batch([str(tmp_path / n) for n in names])
        for k in range(len(names)):
            np.testing.assert_array_equal(got[k], img)
    (tmp_path / "t.jpg").write_bytes(b"\xff\xd8\xff\xe0\x00\x10JFIF")
    assert lib.r3d_jpeg_rgb_info(os.fsencode(str(tmp_path / "t.jpg")), C.byref(h), C.byref(w), C.byref(c)) == L.ERR_INVALID
    with pytest.raises(R.R3DError):
        R.cloud_io.read_rgb_batch([str(tmp_path / "t.jpg")])
    out = np.empty((1, 25, 40, 3), np.uint8)
    arr = (C.c_char_p * 1)(os.fsencode(str(tmp_path / "s.jpg")))
    assert lib.r3d_jpeg_rgb_decode_batch(arr, 1, out.ctypes.data, 25, 40) == L.ERR_INVALID       # another size than the batch's
    assert lib.r3d_jpeg_rgb_decode_batch(None, 1, out.ctypes.data, 25, 40) == L.ERR_INVALID


@pytest.mark.parametrize("n_files", [1, 3, 40])
def test_txt_batch_writer_equals_one_call_per_file(R, tmp_path, n_files):
    """The per-frame camera txts of the frame loop (c2w:163-165) in one native call: same bytes as one write_xyz_txt per
    file -- with the integer third column of the camera txt and without, f32 and f64, few files (threads inside each file)
    and many (one file per thread at a time), values that take every branch of the repr() formatter."""
    rng = np.random.default_rng(n_files)
    per = 70_001 if n_files <= 3 else 1_013                # more than one 65536-point block per file / many small files
    xyz = rng.normal(0, 50, (n_files * per, 3))
    xyz[::97] *= 1e-7
    xyz[5::89] *= 1e18
    xyz[3::211, 0] = 0.0
    xyz[7::223, 1] = -0.0
    xyz[11::227, 2] = np.inf
    xyz[13::229, 0] = np.nan
    z = rng.integers(0, 65536, n_files * per).astype(np.uint16)
    for dtype in (np.float64, np.float32):
        for z_raw in (None, z, (z & 255).astype(np.uint8)):
            a = xyz.astype(dtype)
            paths = [str(tmp_path / ("b%d.txt" % k)) for k in range(n_files)]
            R.cloud_io.write_xyz_txt_batch(paths, a, z_raw=z_raw)
            for k, p in enumerate(paths):
                want = R.cloud_io.format_xyz_txt(a[k * per:(k + 1) * per], z_raw=None if z_raw is None else z_raw[k * per:(k + 1) * per])
                assert open(p, "rb").read() == want, (dtype, k)
    R.cloud_io.write_xyz_txt_batch([], np.zeros((0, 3)))
    with pytest.raises(ValueError):
        R.cloud_io.write_xyz_txt_batch([str(tmp_path / "x.txt")] * 2, np.zeros((3, 3)))
    with pytest.raises(R.R3DError) as e:
        R.cloud_io.write_xyz_txt_batch([str(tmp_path / "no_such_dir" / ("x%d.txt" % k)) for k in range(40)], np.zeros((40, 3)))
    assert "cannot open" in str(e.value)


def test_pow10_table_of_the_shortest_digits_algorithm_matches_its_definition():
    """csrc/r3d_pow10_table.h (generated by tools/gen_pow10_table.py) entry by entry: g(k) = ceil(10^k / 2^r) with
    r = floor(log2 10^k) - 127, recomputed here with exact integers -- a wrong last bit would misprint one double in 2^64,
    which no random test finds."""
    import re
    from fractions import Fraction
    text = open(os.path.join(ROOT, PKG, "csrc", "r3d_pow10_table.h")).read()
    rows = re.findall(r"\{0x([0-9A-F]{16})ull, 0x([0-9A-F]{16})ull\},\s*// (-?\d+)", text)
    assert [int(k) for _, _, k in rows] == list(range(-292, 325))
    for hi, lo, k in rows:
        k = int(k)
        g = (int(hi, 16) << 64) | int(lo, 16)
        assert 1 << 127 <= g < 1 << 128
        # (g - 1) 2^r < 10^k <= g 2^r with r = floor(log2 10^k) - 127, in integers
        x = Fraction(10) ** k
        r = x.numerator.bit_length() - x.denominator.bit_length()
        if Fraction(2) ** r > x:
            r -= 1
        r -= 127
        assert Fraction(g - 1) * Fraction(2) ** r < x <= Fraction(g) * Fraction(2) ** r, k


def _repr_lines(R, values):
    xyz = np.zeros((len(values), 3), np.float64)
    xyz[:, 0] = values
    return [ln.split(b",")[0].decode() for ln in R.cloud_io.format_xyz_txt(xyz).split(b"\n")[:-1]]


def test_native_repr_is_pythons_for_every_exponent_and_at_every_kind_of_boundary(R):
    """The txt formatter's shortest-digits routine against repr(): every binary exponent (powers of two -- the doubles whose
    lower neighbour is closer --, their neighbours on both sides, the largest and a middle significand in each binade),
    subnormals down to 5e-324, the largest double, integers around 2^53, powers of ten and their neighbours (where the digit
    count changes), the switch points of repr's notation (1e16, 1e-4) and values with many trailing zeros."""
    vals = []
    for e in range(-1074, 1024):
        p = float(2.0 ** e) if e >= -1022 else float(np.ldexp(1.0, e))
        vals += [p, np.nextafter(p, np.inf), np.nextafter(p, 0.0)]
        if -1022 <= e < 1023:
            vals += [float(np.ldexp(1.5, e)), float(np.ldexp(1.0 + 2.0 ** -52 * 0x5555555555555, e)), float(np.ldexp(2.0 - 2.0 ** -52, e))]
    for k in range(-323, 309):
        t = float("1e%d" % k)
        vals += [t, np.nextafter(t, np.inf), np.nextafter(t, 0.0), float("9.5e%d" % k), float("1.2345678901234567e%d" % k), float("5e%d" % k)]
    vals += [5e-324, 1.7976931348623157e308, 2.2250738585072014e-308, 2.225073858507201e-308, 9007199254740991.0, 9007199254740992.0,
             9007199254740994.0, 1e16, 9999999999999998.0, 1e-4, 0.0001, 0.00009999999999999999, 123456789012345680.0, 1e22, 1e23,
             100.0, 1000000.0, 120000.0, 0.1, 0.2, 0.30000000000000004, 1 / 3, 2 / 3, 1e15, 123.0, 0.5, 4.35, 4.350000000000001]
    vals = np.array(vals, np.float64)
    vals = np.concatenate([vals, -vals[::7]])
    got = _repr_lines(R, vals)
    for v, g in zip(vals.tolist(), got):
        assert g == repr(v), (v.hex(), g, repr(v))


def test_native_repr_is_pythons_on_random_bit_patterns(R):
    """2 M doubles drawn as random 64-bit patterns (every exponent equally likely) and 1 M of the camera txt's kind
    (((i - cx) / fx) * Z) against repr()."""
    rng = np.random.default_rng(20261004)
    bits = rng.integers(0, 1 << 64, 2_000_000, dtype=np.uint64)
    vals = bits.view(np.float64)
    vals = vals[np.isfinite(vals)]
    fx, fy, cx, cy = R.REF_INTRINSICS
    cam = ((rng.integers(0, 1280, 1_000_000) - cx) / fx) * rng.integers(0, 256, 1_000_000).astype(np.float64)
    for block in (vals, cam):
        got = _repr_lines(R, block)
        want = [repr(v) for v in block.tolist()]
        if got != want:
            bad = next(k for k in range(len(want)) if got[k] != want[k])
            raise AssertionError((float(block[bad]).hex(), got[bad], want[bad]))


def test_native_percent_4f_is_pythons_on_ties_and_random_values(R):
    """"%.4f" rounds the EXACT binary value half-to-even.  The formatter's quick way (a double product, trusted only away from
    the .5 boundary) must hand every doubtful case to its exact integer way: real ties (odd multiples of 1/32 scaled by
    powers of two -- the only doubles whose fifth decimal is an exact 5), values one ulp either side of them, values whose
    product with 10^4 lands within an ulp of .5, large values without fraction bits, tiny ones, and 2 M random ones."""
    rng = np.random.default_rng(7)
    ties = (2 * rng.integers(0, 1 << 20, 200_000) + 1) / 32.0 * rng.choice([1.0, 0.5, 0.25, 0.125, 2.0, 1024.0], 200_000)
    ties = np.concatenate([ties, np.arange(1, 20001, 2) / 32.0 / 625.0 * 625.0])
    near = np.concatenate([np.nextafter(ties, np.inf), np.nextafter(ties, -np.inf)])
    k = rng.integers(0, 10 ** 9, 400_000)
    almost = (k + 0.5) / 1e4                                   # decimal ties: not representable, the double decides
    big = rng.uniform(2.0 ** 30, 2.0 ** 40, 100_000)
    big = np.concatenate([big, np.floor(big) + 0.5, [2.0 ** 40 - 2.0 ** -13, 2.0 ** 40, 2.0 ** 41 + 0.5, 1e15, 123456789012345680.0]])
    tiny = np.concatenate([rng.uniform(0, 1e-3, 100_000), [5e-5, 4.9999999999999996e-5, 5.000000000000001e-5, 1.5e-4, 2.5e-4, 5e-324, 0.0, -0.0]])
    rand = rng.normal(0, 1, 2_000_000) * 10.0 ** rng.integers(-6, 9, 2_000_000)
    vals = np.concatenate([ties, near, almost, big, tiny, rand])
    vals = np.concatenate([vals, -vals[::5]])
    vals = vals[: len(vals) // 3 * 3].reshape(-1, 3)
    body = R.cloud_io.format_ply(vals).split(b"end_header\n    ", 1)[1]
    got = body.decode().split()
    want = ["%.4f" % v for v in vals.reshape(-1).tolist()]
    assert len(got) == len(want)
    if got != want:
        bad = next(i for i in range(len(want)) if got[i] != want[i])
        raise AssertionError((float(vals.reshape(-1)[bad]).hex(), got[bad], want[bad]))


def test_batches_of_mixed_formats_are_read_file_by_file(R, tmp_path):
    """A pose file may name PNG and JPEG frames side by side: the native batch decoders take one format per call, a mixed
    list goes file by file with the same per-format rules; a missing file is named."""
    from PIL import Image
    rng = np.random.default_rng(5)
    g = rng.integers(0, 256, (24, 40), dtype=np.uint8)
    c = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    Image.fromarray(g, "L").save(tmp_path / "a.png")
    Image.fromarray(c, "RGB").save(tmp_path / "b.jpg", quality=90)
    Image.fromarray(c, "RGB").save(tmp_path / "c.png")
    paths = [str(tmp_path / n) for n in ("a.png", "b.jpg", "c.png")]
    got = R.cloud_io.read_depth_batch(paths)
    for k, p in enumerate(paths):
        np.testing.assert_array_equal(got[k], R.cloud_io.read_depth_gray(p))
    np.testing.assert_array_equal(got[0], g)
    np.testing.assert_array_equal(got[1], _pil_luma(paths[1]))
    rgb = R.cloud_io.read_rgb_batch(paths)
    for k, p in enumerate(paths):
        np.testing.assert_array_equal(rgb[k], np.asarray(Image.open(p).convert("RGB")))
    with pytest.raises(FileNotFoundError) as e:
        R.cloud_io.read_depth_batch(paths + [str(tmp_path / "gone.png")])
    assert "gone.png" in str(e.value)
    with pytest.raises(FileNotFoundError) as e:
        R.cloud_io.read_rgb_batch(paths + [str(tmp_path / "gone.jpg")])
    assert "gone.jpg" in str(e.value)


def _png_with_filter(path, img, bit_depth, filter_type):
    """A PNG whose every scanline uses ONE filter type (encoders choose per line; this pins each of the five)."""
    import struct
    import zlib
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    colour_type = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    rows = img.astype(">u2" if bit_depth == 16 else np.uint8).reshape(h, -1).view(np.uint8).astype(np.int64)   # bytes of each line
    bpp = ch * bit_depth // 8
    out = bytearray()
    prev = np.zeros(rows.shape[1], np.int64)
    for y in range(h):
        cur = rows[y]
        left = np.concatenate([np.zeros(bpp, np.int64), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int64), prev[:-bpp]])
        if filter_type == 0:
            pred = np.zeros_like(cur)
        elif filter_type == 1:
            pred = left
        elif filter_type == 2:
            pred = prev
        elif filter_type == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        out.append(filter_type)
        out += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, colour_type, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(bytes(out), 6)) + chunk(b"IEND", b""))


@pytest.mark.parametrize("filter_type", [0, 1, 2, 3, 4])
def test_png_unfilter_every_type_and_pixel_size(R, tmp_path, filter_type):
    """The five PNG line filters x bytes per pixel 1 (grey 8), 2 (grey 16 / grey+alpha), 3 (RGB), 4 (RGBA), 6 (RGB 16),
    8 (RGBA 16): files written with one filter type on every line, read back through the native decoders and through PIL."""
    from PIL import Image
    rng = np.random.default_rng(filter_type)
    yy, xx = np.mgrid[0:23, 0:37]
    smooth = ((np.sin(xx / 5.0) + np.cos(yy / 3.0) + 2) * 60).astype(np.int64)
    cases = [(smooth.astype(np.uint8), 8), ((smooth * 250 + rng.integers(0, 9, smooth.shape)).astype(np.uint16), 16),
             (np.stack([smooth, 255 - smooth], 2).astype(np.uint8), 8),
             (np.stack([smooth, smooth // 2, rng.integers(0, 256, smooth.shape)], 2).astype(np.uint8), 8),
             (rng.integers(0, 256, smooth.shape + (4,), dtype=np.uint8), 8),
             (rng.integers(0, 65536, smooth.shape + (3,)).astype(np.uint16), 16),
             (rng.integers(0, 65536, smooth.shape + (4,)).astype(np.uint16), 16)]
    for k, (img, depth) in enumerate(cases):
        p = str(tmp_path / ("f%d_%d.png" % (filter_type, k)))
        _png_with_filter(p, img, depth, filter_type)
        pil = np.asarray(Image.open(p))
        assert pil.shape[:2] == img.shape[:2]
        got = R.cloud_io.read_depth_gray(p)                          # IMREAD_GRAYSCALE semantics through the native decoder
        if img.ndim == 2 and depth == 8:
            np.testing.assert_array_equal(got, img)
        elif img.ndim == 2:
            np.testing.assert_array_equal(got, (img >> 8).astype(np.uint8))
        elif depth == 8 and img.shape[2] in (3, 4):
            np.testing.assert_array_equal(R.cloud_io.read_rgb_batch([p])[0], img[..., :3])
            np.testing.assert_array_equal(got, R.cloud_io.rgb_to_gray(img[..., :3], "opencv_png"))
        elif depth == 8:                                             # grey + alpha: alpha dropped
            np.testing.assert_array_equal(got, img[..., 0])
        else:                                                        # 16-bit colour: libpng's 16-bit rule (rounds), then the high byte
            r, g, b = (img[..., c].astype(np.int64) for c in range(3))
            g16 = np.where((r == g) & (g == b), r, (9797 * r + 19234 * g + 3737 * b + 16384) >> 15)
            np.testing.assert_array_equal(got, (g16 >> 8).astype(np.uint8))


def test_file_writers_across_slab_boundaries(R, tmp_path):
    """The file writers format slab k + 1 while a helper thread writes slab k (8 M points per slab for the PLY, 4 M for the
    coloured PLY and the txt): clouds a little over one and over two slabs must give the bytes of the in-memory formatters,
    whose output is pinned to the reference's by the golden tests."""
    rng = np.random.default_rng(9)
    n = (8 << 20) + 12_345
    xyz = (rng.normal(0, 30, (n, 3))).astype(np.float32)
    p = tmp_path / "big.ply"
    R.cloud_io.write_ply(str(p), xyz)
    head = ("ply\n    format ascii 1.0\n    element vertex %d\n    property float x\n    property float y\n"
            "    property float z\n    end_header\n    " % n).encode()
    data = p.read_bytes()
    assert data.startswith(head) and data.endswith(b"\n    ")
    body = data[len(head):-5]
    pos = 0
    for lo in range(0, n, 1_000_003):                       # compare block by block: a block's text does not depend on its neighbours
        want = R.cloud_io.format_ply(xyz[lo:lo + 1_000_003]).split(b"end_header\n    ", 1)[1][:-5]
        assert body[pos:pos + len(want)] == want, lo
        pos += len(want)
    assert pos == len(body)
    p.unlink()
    m = 2 * (4 << 20) + 777                                  # two full slabs and a bit
    t = tmp_path / "big.txt"
    R.cloud_io.write_xyz_txt(str(t), xyz[:m])
    data = t.read_bytes()
    pos = 0
    for lo in range(0, m, 1_000_003):
        want = R.cloud_io.format_xyz_txt(xyz[lo:min(m, lo + 1_000_003)])
        assert data[pos:pos + len(want)] == want, lo
        pos += len(want)
    assert pos == len(data)
    t.unlink()
    k = (4 << 20) + 4_321
    rgb = rng.integers(0, 256, (k, 3), dtype=np.uint8)
    c = tmp_path / "big_rgb.ply"
    R.cloud_io.write_ply_rgb(str(c), xyz[:k], rgb)
    lines = c.read_bytes().split(b"end_header\n    ", 1)[1]
    # rows: the plain PLY's "x y z " followed by "R G B 0"
    plain = R.cloud_io.format_ply(xyz[:k]).split(b"end_header\n    ", 1)[1][:-5].split(b" \n")
    got_rows = lines[:-5].split(b"\n")
    assert len(got_rows) == k + 0 or got_rows[-1] == b""
    for i in (0, 1, (4 << 20) - 1, 4 << 20, (4 << 20) + 1, k - 1):
        assert got_rows[i] == plain[i] + b" %d %d %d 0" % tuple(int(v) for v in rgb[i]), i
    import hashlib
    want = hashlib.sha256()
    for i0 in range(0, k, 500_000):
        blk = b"".join(plain[i] + b" %d %d %d 0\n" % (rgb[i, 0], rgb[i, 1], rgb[i, 2]) for i in range(i0, min(k, i0 + 500_000)))
        want.update(blk)
    assert hashlib.sha256(lines[:-5]).hexdigest() == want.hexdigest()


def test_device_text_header_and_bench_modules_on_the_cpu(R):
    """What the GPU text path shares with the host: the PLY header / trailer device_text puts around the GPU's rows are the
    host formatter's own bytes (genply's template, c2w:122-132; genply_noRGB's, p2c:62-75); and bench.py's workload modules
    (tools/bench_*.py) import without a GPU -- the driver's line must not die on an import."""
    import subprocess
    import sys
    T = R.device_text
    xyz = np.array([[1.0, 2.0, 3.0], [-4.5, 0.25, 1e-5]])
    whole = R.cloud_io.format_ply(xyz)
    head = T.ply_header(2)
    assert whole.startswith(head) and whole.endswith(T.PLY_TRAILER)
    assert whole[len(head):-len(T.PLY_TRAILER)] == b"1.0000 2.0000 3.0000 \n-4.5000 0.2500 0.0000 \n"
    assert T.ply_header(0) == R.cloud_io.format_ply(np.zeros((0, 3)))[:-len(T.PLY_TRAILER)]
    colour = T.ply_header(7, colour=True).decode()
    assert "element vertex 7\n" in colour and colour.count("property uchar") == 4 and colour.endswith("end_header\n    ")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); "
            "import bench_common, bench_regimes, bench_e2e, bench_secondary, bench_assemble; "
            "assert bench_assemble.step_text('inputs_overlap').endswith('fused]'); print('ok')"
            % (ROOT, os.path.join(ROOT, "tools")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--inputs" in r.stdout


def test_binary_ply_flag_round_trips(R, tmp_path):
    """f1's optional binary flag: a standard binary_little_endian PLY of float32 x, y, z -- readable by
    read_ply, header without the reference template's indents, 12 bytes per vertex, f64 clouds rounded once."""
    xyz = np.random.default_rng(0).normal(0, 50, (70_001, 3))
    xyz[5] = (np.inf, -0.0, 1e-40)
    for dt in (np.float64, np.float32):
        R.cloud_io.write_ply_binary(str(tmp_path / "b.ply"), xyz.astype(dt))
        raw = (tmp_path / "b.ply").read_bytes()
        head = R.device_text.ply_header_binary(len(xyz))
        assert raw.startswith(head) and len(raw) == len(head) + 12 * len(xyz)
        assert head == b"ply\nformat binary_little_endian 1.0\nelement vertex 70001\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
        assert raw[len(head):] == xyz.astype(np.float32).astype("<f4").tobytes()
        np.testing.assert_array_equal(R.cloud_io.read_ply(str(tmp_path / "b.ply")).astype(np.float32), xyz.astype(np.float32))
    R.cloud_io.write_ply_binary(str(tmp_path / "e.ply"), np.zeros((0, 3)))
    assert R.cloud_io.read_ply(str(tmp_path / "e.ply")).shape == (0, 3)
    with pytest.raises(R.R3DError):
        R.cloud_io.write_ply_binary(str(tmp_path / "no_such_dir" / "x.ply"), xyz)
