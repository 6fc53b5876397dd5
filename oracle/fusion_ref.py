"""Oracle (test infrastructure): fp64 restatement of the reference's fusion arithmetic.

PARITY PINNED by tests/golden/ (see oracle/__init__.py).  Citations are to files under
the reference checkout (`p2c` = transfer/pixel_to_camera.py, `c2w` =
transfer/camera_to_world.py, `icp` = other_tools/transfer_T_icp.py).

Two flavours of every stage:
  * vectorised NumPy fp64 (fast; the checker for GPU parity tests), and
  * `*_loop` functions that keep the reference's per-point Python loops and its
    text-file round trip (byte-exact files; the `cpu_baseline` that bench.py times).
"""
import numpy as np

# Reference intrinsics, hard-coded at p2c:25-28 and c2w:68-71.
REF_FX, REF_FY, REF_CX, REF_CY = 600.391, 600.079, 320, 240


# --------------------------------------------------------------------------- unprojection
def pixel_rays(h, w, fx=REF_FX, fy=REF_FY, cx=REF_CX, cy=REF_CY):
    """u[i] = (i-cx)/fx, v[j] = (j-cy)/fy in fp64 -- the Z-independent factor of
    `X = (i - cx)/fx*Z`, `Y = (j - cy)/fy*Z` (c2w:78-79, p2c:37-38; Python evaluates
    the division first, then the product with Z)."""
    u = (np.arange(w, dtype=np.float64) - cx) / fx
    v = (np.arange(h, dtype=np.float64) - cy) / fy
    return u, v


def unproject(depth, fx=REF_FX, fy=REF_FY, cx=REF_CX, cy=REF_CY):
    """Camera-frame points of a depth raster, row-major (j outer, i inner), every pixel
    kept, no masking (c2w:74-83).  Returns float64 [H*W, 3]."""
    depth = np.asarray(depth)
    h, w = depth.shape
    u, v = pixel_rays(h, w, fx, fy, cx, cy)
    z = depth.astype(np.float64)
    out = np.empty((h, w, 3), dtype=np.float64)
    out[:, :, 0] = u[None, :] * z
    out[:, :, 1] = v[:, None] * z
    out[:, :, 2] = z
    return out.reshape(-1, 3)


# --------------------------------------------------------------------------- pose
def quat_to_rinv(q_xyzw):
    """c2w:53-55 `np.matrix(R.from_quat(q).as_matrix()).I`: scalar-LAST quaternion,
    normalised (SciPy does), rotation matrix, then a GENERAL matrix inverse (not a
    transpose).  Returns float64 [3,3]."""
    q = np.asarray(q_xyzw, dtype=np.float64)
    # SciPy normalises with scipy.linalg.norm(quat, axis=1) = sqrt(add.reduce(q * q)): the four squares added one after
    # the other.  (np.dot(q, q) -- BLAS ddot -- sums them in another order and is an ulp off for one quaternion in eight:
    # found in round 4 by tools/stress_dropin.py; tests/test_oracle_golden.py pins this function against SciPy itself now.)
    n = np.sqrt(np.add.reduce(q * q))
    if n == 0.0:
        raise ValueError("zero-norm quaternion")
    x, y, z, w = q / n
    r = np.array([
        [x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w]])
    return np.linalg.inv(r)


def get_r_wxyz(q_wxyz):
    """c2w:40-52 / icp:13-25 `get_r`: scalar-FIRST quaternion, NOT normalised, then a
    general inverse.  Unused by the reference's own main paths but part of its surface."""
    q = np.asarray(q_wxyz, dtype=np.float64)
    r = np.zeros((3, 3))
    r[0, 0] = 1 - 2 * q[2] * q[2] - 2 * q[3] * q[3]
    r[1, 1] = 1 - 2 * q[1] * q[1] - 2 * q[3] * q[3]
    r[2, 2] = 1 - 2 * q[1] * q[1] - 2 * q[2] * q[2]
    r[0, 1] = 2 * q[1] * q[2] - 2 * q[0] * q[3]
    r[0, 2] = 2 * q[1] * q[3] + 2 * q[0] * q[2]
    r[1, 0] = 2 * q[1] * q[2] + 2 * q[0] * q[3]
    r[1, 2] = 2 * q[2] * q[3] - 2 * q[0] * q[1]
    r[2, 0] = 2 * q[1] * q[3] - 2 * q[0] * q[2]
    r[2, 1] = 2 * q[2] * q[3] + 2 * q[0] * q[1]
    return np.linalg.inv(r)


def se3_apply(p_cam, rinv, t):
    """c2w:57-59 `p_world = R^-1 . (p_cam - t)` for an [N,3] block, fp64."""
    p = np.asarray(p_cam, dtype=np.float64) - np.asarray(t, dtype=np.float64)[None, :]
    return p @ np.asarray(rinv, dtype=np.float64).T


def fuse_frames(depths, quats_xyzw, ts, fx=REF_FX, fy=REF_FY, cx=REF_CX, cy=REF_CY):
    """c2w:149-174: per pose line unproject then SE(3), concatenated in pose-file order.
    depths: [F,H,W]; returns float64 [F*H*W, 3]."""
    depths = np.asarray(depths)
    f, h, w = depths.shape
    out = np.empty((f, h * w, 3), dtype=np.float64)
    for k in range(f):
        out[k] = se3_apply(unproject(depths[k], fx, fy, cx, cy), quat_to_rinv(quats_xyzw[k]), ts[k])
    return out.reshape(-1, 3)


def apply_T(p, T):
    """icp:10-12,82-87: `(T . [x,y,z,1]^T)[0:3]` -- a general 4x4 (scale lives in T)."""
    p = np.asarray(p, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    ph = np.concatenate([p, np.ones((p.shape[0], 1))], axis=1)
    return (ph @ T.T)[:, :3]


# --------------------------------------------------------------------------- text formats
def parse_pose_file(path):
    """c2w:138-158: first line is a header; per line fields[1:4]=t, [4:8]=q (xyzw),
    [8]=depth file name.  Returns (names, quats [F,4], ts [F,3])."""
    names, quats, ts = [], [], []
    with open(path) as f:
        f.readline()
        for line in f:
            fields = line.split(',')
            ts.append([float(s) for s in fields[1:4]])
            quats.append([float(s) for s in fields[4:8]])
            names.append(fields[8])
    return names, np.array(quats).reshape(-1, 4), np.array(ts).reshape(-1, 3)


def parse_T_file(path):
    """icp:33-43: four whitespace-separated rows -> 4x4 fp64."""
    T = np.zeros((4, 4))
    with open(path) as f:
        for i in range(4):
            T[i] = [float(s) for s in f.readline().split()[:4]]
    return T


def read_xyz_txt(path):
    """Inverse of the `X,Y,Z\\n` text lines (c2w:97-98 reads them back the same way)."""
    rows = []
    with open(path) as f:
        for line in f:
            rows.append([float(s) for s in line.split(',')[0:3]])
    return np.array(rows, dtype=np.float64).reshape(-1, 3)


PLY_HEAD = ("ply\n    format ascii 1.0\n    element vertex %d\n    property float x\n"
            "    property float y\n    property float z\n    end_header\n    ")
PLY_TAIL = "\n    "


def format_ply(xyz):
    """c2w:112-134 (== icp:46-68 == p2c:98-124): the exact byte layout, including the
    4-space indentation the triple-quoted template carries and the trailing space of
    every vertex row."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    body = "".join("%.4f %.4f %.4f \n" % (p[0], p[1], p[2]) for p in xyz)
    return PLY_HEAD % xyz.shape[0] + body + PLY_TAIL


PLY_HEAD_RGB = ("ply\n    format ascii 1.0\n    element vertex %d\n    property float x\n    property float y\n"
                "    property float z\n    property uchar red\n    property uchar green\n    property uchar blue\n"
                "    property uchar alpha\n    end_header\n    ")


def format_ply_rgb(xyz, rgb):
    """p2c:55-91 (genply_noRGB, the coloured writer): rows `x y z R G B 0`, no trailing space, same indentation."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    rgb = np.asarray(rgb).reshape(-1, 3)
    body = "".join("%.4f %.4f %.4f %d %d %d 0\n" % (p[0], p[1], p[2], int(c[0]), int(c[1]), int(c[2])) for p, c in zip(xyz, rgb))
    return PLY_HEAD_RGB % xyz.shape[0] + body + PLY_TAIL


def read_ply_vertices(path):
    """Parse a PLY written in the reference layout back to float64 [N,3]."""
    with open(path) as f:
        lines = f.read().split('\n')
    k = next(i for i, s in enumerate(lines) if s.strip() == 'end_header')
    n = next(int(s.split()[-1]) for s in lines if s.strip().startswith('element vertex'))
    rows = [[float(v) for v in s.split()[:3]] for s in lines[k + 1:k + 1 + n]]
    return np.array(rows, dtype=np.float64).reshape(-1, 3)


# --------------------------------------------------------------------------- loop-faithful
def gentxtcord_loop(filename, depth, fx=REF_FX, fy=REF_FY, cx=REF_CX, cy=REF_CY):
    """c2w:67-83 with its loop structure: one Python iteration, three str() and one
    write per pixel; Z keeps the raster's scalar type so `str(Z)` prints an integer
    for uint8 rasters."""
    with open(filename, 'w') as f:
        for j, row in enumerate(depth):
            for i, Z in enumerate(row):
                X = (i - cx) / fx * Z
                Y = (j - cy) / fy * Z
                f.write(str(X) + ',' + str(Y) + ',' + str(Z) + '\n')


def get_pointdata_loop(p_path, q, t, xs, ys, zs, world_path):
    """c2w:86-105 with its loop structure: re-read the camera txt, one np.dot on an
    np.matrix per point, append to the caller's lists, rewrite the world txt ('w' mode,
    so it holds the last frame only)."""
    r = np.matrix(quat_to_rinv(q))
    t = np.asarray(t, dtype=np.float64)
    with open(p_path) as fin, open(world_path, 'w') as fout:
        for line in fin:
            p = np.array([float(s) for s in line.split(',')[0:3]])
            pw = np.array(np.dot(r, (p - t).T).T)
            xs.append(pw[0, 0])
            ys.append(pw[1, 0])
            zs.append(pw[2, 0])
            fout.write(str(pw[0, 0]) + ',' + str(pw[1, 0]) + ',' + str(pw[2, 0]) + '\n')


def local_world_loop(path_local, fout, T, xs, ys, zs, flag):
    """icp:71-97: pass-through (flag False) or homogeneous 4x4 apply (flag True)."""
    T = np.asarray(T, dtype=np.float64)
    with open(path_local) as fin:
        for line in fin:
            p = np.ones(4)
            p[0:3] = [float(s) for s in line.split(',')[0:3]]
            pw = np.array(np.dot(T, p.T).T) if flag else p
            xs.append(pw[0])
            ys.append(pw[1])
            zs.append(pw[2])
            fout.write(str(pw[0]) + ',' + str(pw[1]) + ',' + str(pw[2]) + '\n')


def fuse_frames_loop(depths, quats_xyzw, ts, workdir):
    """The whole per-frame path of c2w:149-172 (unproject -> txt -> re-read -> SE(3) ->
    world txt) with the reference's loop structure.  Returns float64 [N,3]."""
    import os
    xs, ys, zs = [], [], []
    for k in range(len(depths)):
        cam = os.path.join(workdir, "cam_%d.txt" % k)
        gentxtcord_loop(cam, depths[k])
        get_pointdata_loop(cam, quats_xyzw[k], ts[k], xs, ys, zs, os.path.join(workdir, "world.txt"))
    return np.stack([np.array(xs), np.array(ys), np.array(zs)], axis=1)


def genply_loop(coords, pc_file):
    """c2w:112-134 with its loop structure: the three coordinate lists go into a (3, N) float64 array, every column is
    formatted with three "%.4f" calls and one str.format into a list of rows, and the file is ONE write of the header
    template with the joined rows inside.  Same bytes as format_ply (tests/test_oracle_golden.py); here for its cost."""
    n = len(coords[0])
    table = np.zeros((3, n))
    for axis in range(3):
        table[axis] = coords[axis]
    fmt4 = lambda value: "%.4f" % value
    rows = []
    for col in table.T:
        rows.append("{} {} {} \n".format(fmt4(col[0]), fmt4(col[1]), fmt4(col[2])))
    with open(pc_file, "w") as f:
        f.write(PLY_HEAD % len(rows) + "".join(rows) + PLY_TAIL)


# --------------------------------------------------------------------------- tolerance
def parity_errors(got, ref):
    """SURVEY 8(d) tolerance metrics: (max point-wise ||d||/max(||ref||,1e-6),
    max per-component |d| / max(|ref|, 1e-3*||ref||, 1e-9))."""
    got = np.asarray(got, dtype=np.float64).reshape(-1, 3)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1, 3)
    d = got - ref
    nr = np.linalg.norm(ref, axis=1)
    e_norm = np.linalg.norm(d, axis=1) / np.maximum(nr, 1e-6)
    e_comp = np.abs(d) / np.maximum(np.maximum(np.abs(ref), 1e-3 * nr[:, None]), 1e-9)
    if got.shape[0] == 0:
        return 0.0, 0.0
    return float(e_norm.max()), float(e_comp.max())
