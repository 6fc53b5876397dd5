"""Host-side (fp64) pose handling: the per-frame quantities are 12 numbers, so they stay in
Python exactly as the reference computes them; only the per-point work goes to the GPU.

Reference: transfer/camera_to_world.py:40-55 (get_r, scipy_transfer), :138-158 (pose file),
other_tools/transfer_T_icp.py:33-43 (get_T).
"""
import numpy as np

import math


def _rotation_matrix_xyzw(q):
    """Rotation matrix of a scalar-last quaternion, normalised first: SciPy's Rotation.from_quat(q).as_matrix() restated
    operation for operation (sequential 4-term dot, one sqrt, four divisions, then the ten products) -- bit-identical to
    SciPy 1.15 on 50,000 random quaternions (tests/test_host_logic.py) and to the reference-generated fixture, without
    paying ~0.2-0.4 s to import scipy.spatial in every drop-in run."""
    x, y, z, w = [float(v) for v in np.asarray(q, dtype=np.float64).reshape(4)]
    n = math.sqrt(x * x + y * y + z * z + w * w)
    if not n > 0.0:
        raise ValueError("quaternion has zero norm")
    x /= n
    y /= n
    z /= n
    w /= n
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    return np.array([
        [x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw)],
        [2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw)],
        [2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2]])


def scipy_transfer(quat):
    """Inverse rotation of a scalar-LAST quaternion (normalised first), as an np.matrix
    like the reference returns (c2w:53-55: `np.matrix(R.from_quat(q).as_matrix()).I`)."""
    return np.matrix(np.linalg.inv(_rotation_matrix_xyzw(quat)))


def get_r(q):
    """Inverse rotation of a scalar-FIRST quaternion WITHOUT normalisation (c2w:40-52)."""
    w, x, y, z = [float(v) for v in q]
    r = np.array([
        [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y],
        [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
        [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])
    return np.matrix(np.linalg.inv(r))


def pose_table(quats_xyzw, ts):
    """[F,12] float64 rows = (Rinv row-major, t) -- the layout r3d_fuse_frames reads."""
    quats_xyzw = np.asarray(quats_xyzw, dtype=np.float64).reshape(-1, 4)
    ts = np.asarray(ts, dtype=np.float64).reshape(-1, 3)
    if len(quats_xyzw) != len(ts):
        raise ValueError("need one translation per quaternion")
    table = np.empty((len(ts), 12), dtype=np.float64)
    for k in range(len(ts)):
        table[k, :9] = np.asarray(scipy_transfer(quats_xyzw[k])).reshape(9)
        table[k, 9:] = ts[k]
    return table


def pose_to_T(rinv, t):
    """4x4 that maps p_cam -> Rinv (p_cam - t); lets r3d_apply_T do a standalone SE(3) apply."""
    T = np.eye(4)
    T[:3, :3] = np.asarray(rinv, dtype=np.float64)
    T[:3, 3] = -np.asarray(rinv, dtype=np.float64) @ np.asarray(t, dtype=np.float64)
    return T


def str_tofloat(data):
    """c2w:28-30 (np.float is gone from numpy >= 1.24; float() is what it aliased)."""
    return np.array([float(s) for s in data])


def read_pose_file(path):
    """Pose CSV of c2w:138-158: one header line, then `id,tx,ty,tz,qx,qy,qz,qw,name,...`.
    Like the reference, the file name must not be the LAST field (a trailing newline would
    stick to it); unlike the reference that case raises a clear error instead of a TypeError
    from a failed imread."""
    names, quats, ts = [], [], []
    with open(path, 'r') as f:
        f.readline()
        for lineno, line in enumerate(f, start=2):
            if not line.strip():
                continue
            fields = line.split(',')
            if len(fields) < 9:
                raise ValueError("%s:%d: expected at least 9 comma-separated fields" % (path, lineno))
            name = fields[8]
            if name.endswith('\n'):
                raise ValueError("%s:%d: the depth file name is the last field and carries the newline; "
                                 "the reference format has at least one field after it" % (path, lineno))
            ts.append(str_tofloat(fields[1:4]))
            quats.append(str_tofloat(fields[4:8]))
            names.append(name)
    return names, np.array(quats, dtype=np.float64).reshape(-1, 4), np.array(ts, dtype=np.float64).reshape(-1, 3)


def get_T(path_txt):
    """icp:33-43: 4 whitespace-separated rows -> 4x4 float64."""
    T = np.zeros((4, 4))
    with open(path_txt, 'r') as f:
        for i in range(4):
            row = str_tofloat(f.readline().split())
            T[i, 0:4] = row[0:4]
    return T


def write_T(path_txt, T):
    """Write a 4x4 in the format get_T() parses (repr() keeps every fp64 bit)."""
    T = np.asarray(T, dtype=np.float64).reshape(4, 4)
    with open(path_txt, 'w') as f:
        for row in T:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
