"""Oracle (test infrastructure): rigid point-to-plane ICP restated on the CPU.  PARITY UNPINNED.

The reference holds no ICP code (other_tools/transfer_T_icp.py:99-108 only reads a T_data.txt that its authors produced by
hand in CloudCompare, readme.md:25,54); what it does fix is the USE: the relative pose of two single-view camera clouds,
./point/0.txt and ./point/24.txt (icp:107-108).  This file restates the build's definition of that step exactly as
include/r3d.h specifies it, in NumPy fp64:
  * normals of an organised cloud: central differences of the four raster neighbours, fp64 cross product, zero vector at
    raster borders / missing depth / depth jumps (r3d_normals_organized);
  * admissible pairs, the residual r = n . (p - q), the 24 direction classes of the target normals, and the per-class gate
    gate_scale x ("lower" order statistic of (float)(r r)) (r3d_icp_plane_residuals, r3d_select_quantile_f32);
  * the 29 sums, the 6x6 solve and the exponential map (r3d_icp_plane_accumulate, r3d_plane_step_from_sums);
  * the loop (r3d_icp_iterate_plane), with icp_ref's fp32 nearest-neighbour definition.
Anchors: synthetic two-view scenes with a known relative pose (3d_reconstruction_system_amd/synthetic.two_views), and in the
tests an independent SciPy implementation (cKDTree neighbours, lstsq solve).
"""
import numpy as np

from . import icp_ref

PLANE_SUMS = 29


def organized_normals(xyz, height, width, max_jump=0.05, viewpoint=None):
    """[F*H*W, 3] float32 normals of an organised cloud given as [F*H*W, 3] (or [F,H,W,3]) float32."""
    P = np.asarray(xyz, dtype=np.float32).reshape(-1, height, width, 3).astype(np.float64)
    F = P.shape[0]
    vp = np.zeros(3) if viewpoint is None else np.asarray(viewpoint, dtype=np.float64)
    out = np.zeros((F, height, width, 3), dtype=np.float32)
    if height < 3 or width < 3:
        return out.reshape(-1, 3)
    c = P[:, 1:-1, 1:-1]
    l, r, u, d = P[:, 1:-1, :-2], P[:, 1:-1, 2:], P[:, :-2, 1:-1], P[:, 2:, 1:-1]

    def rng_of(x):
        e = x - vp
        return np.sqrt(e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1] + e[..., 2] * e[..., 2])

    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        rc = rng_of(c)
        ok = np.isfinite(rc) & (rc > 0)
        mj = float(np.float32(max_jump))
        for nb in (l, r, u, d):
            rn = rng_of(nb)
            ok &= np.isfinite(rn) & (rn > 0) & (np.abs(rn - rc) <= mj * rc)
        a, b = r - l, d - u
        n = np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                      a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                      a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)
        ln = np.sqrt(n[..., 0] * n[..., 0] + n[..., 1] * n[..., 1] + n[..., 2] * n[..., 2])
        ok &= np.isfinite(ln) & (ln > 0)
        n = n / ln[..., None]
        e = c - vp
        flip = (n[..., 0] * e[..., 0] + n[..., 1] * e[..., 1] + n[..., 2] * e[..., 2]) > 0
        n = np.where(flip[..., None], -n, n)
        n = np.where(ok[..., None], n, 0.0)
    out[:, 1:-1, 1:-1] = n.astype(np.float32)
    return out.reshape(-1, 3)


def quantile_lower(values, q):
    """(value, count): element of rank floor(q (m - 1)) of the m finite values; (+inf, 0) when there are none."""
    v = np.asarray(values, dtype=np.float32)
    v = np.sort(v[np.isfinite(v)])
    if v.size == 0:
        return np.float32(np.inf), 0
    return v[int(np.floor(q * (v.size - 1)))], int(v.size)


def plane_pairs(src, tgt, nrm, idx, d2=None, max_d2=-1.0):
    """(admissible mask, r fp64, p, n) of the pairs (src[k], tgt[idx[k]], nrm[idx[k]])."""
    p = np.asarray(src, dtype=np.float32).astype(np.float64)
    idx = np.asarray(idx).astype(np.int64)
    inside = idx < tgt.shape[0]
    j = np.where(inside, idx, 0)
    q = np.asarray(tgt, dtype=np.float32).astype(np.float64)[j]
    n32 = np.asarray(nrm, dtype=np.float32)[j]
    n = n32.astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        r = n[:, 0] * (p[:, 0] - q[:, 0]) + n[:, 1] * (p[:, 1] - q[:, 1]) + n[:, 2] * (p[:, 2] - q[:, 2])
        ok = inside & ~np.all(n32 == 0, axis=1) & np.isfinite(r) & np.isfinite((p[:, 0] + p[:, 1]) + p[:, 2])
        if max_d2 >= 0:
            ok &= np.asarray(d2, dtype=np.float32) <= np.float32(max_d2)
    return ok, r, p, n


def normal_classes(n32):
    """Direction class 0..23 of stored f32 normals: 8 major + 4 [n_major < 0] + 2 [n_(major+1) < 0] + [n_(major+2) < 0], major =
    axis of the largest |component| (lowest axis on ties)."""
    n32 = np.asarray(n32, dtype=np.float32)
    a = np.abs(n32)
    major = np.zeros(n32.shape[0], dtype=np.int64)
    major[(a[:, 1] > a[:, 0]) & (a[:, 1] >= a[:, 2])] = 1
    major[(a[:, 2] > a[:, 0]) & (a[:, 2] > a[:, 1])] = 2
    rows = np.arange(n32.shape[0])
    return (major * 8 + (n32[rows, major] < 0) * 4 + (n32[rows, (major + 1) % 3] < 0) * 2
            + (n32[rows, (major + 2) % 3] < 0)).astype(np.int64)


def plane_residuals(src, tgt, nrm, idx, d2=None, max_d2=-1.0):
    """(float32 (r r) per source row, +inf where the pair is not admissible; direction class per row, 255 there)."""
    ok, r, _p, _n = plane_pairs(src, tgt, nrm, idx, d2, max_d2)
    j = np.where(np.asarray(idx).astype(np.int64) < tgt.shape[0], np.asarray(idx).astype(np.int64), 0)
    cls = np.where(ok, normal_classes(np.asarray(nrm, dtype=np.float32)[j]), 255).astype(np.uint8)
    with np.errstate(invalid="ignore", over="ignore"):
        return np.where(ok, (r * r).astype(np.float32), np.float32(np.inf)), cls


def kept_pairs(src, tgt, nrm, idx, d2=None, max_d2=-1.0, trim_q=0.0, gate_scale=1.0):
    """(mask of the pairs that take part, r, p, n)."""
    ok, r, p, n = plane_pairs(src, tgt, nrm, idx, d2, max_d2)
    if 0.0 < trim_q < 1.0:
        r2, cls = plane_residuals(src, tgt, nrm, idx, d2, max_d2)
        keep = np.zeros_like(ok)
        for c in range(24):
            m = ok & (cls == c)
            if not m.any():
                continue
            gate, _m = quantile_lower(r2[m], float(np.float32(trim_q)))
            keep |= m & (r2 <= np.float32(gate) * np.float32(gate_scale))
        ok = keep
    return ok, r, p, n


def plane_sums(src, tgt, nrm, idx, d2=None, max_d2=-1.0, trim_q=0.0, gate_scale=1.0):
    """The 29 fp64 sums over the pairs that take part."""
    ok, r, p, n = kept_pairs(src, tgt, nrm, idx, d2, max_d2, trim_q, gate_scale)
    p, n, r = p[ok], n[ok], r[ok]
    J = np.concatenate([np.cross(p, n), n], axis=1)
    s = np.zeros(PLANE_SUMS)
    s[0] = p.shape[0]
    s[1] = (r * r).sum()
    s[2:8] = (J * r[:, None]).sum(0)
    A = J.T @ J
    s[8:] = A[np.triu_indices(6)]
    return s


def step_from_sums(s):
    """(T 4x4, rms): the rigid step of r3d_plane_step_from_sums; raises ValueError when it is undefined."""
    s = np.asarray(s, dtype=np.float64)
    n = s[0]
    rms = float(np.sqrt(max(s[1], 0.0) / n)) if n > 0 else 0.0
    if not n >= 6:
        raise ValueError("fewer than 6 pairs")
    A = np.zeros((6, 6))
    A[np.triu_indices(6)] = s[8:]
    A = A + np.triu(A, 1).T
    tr_rot, tr_tra = np.trace(A[:3, :3]), np.trace(A[3:, 3:])
    if not (tr_rot > 0 and tr_tra > 0):
        raise ValueError("no pairs with a lever arm")
    length = np.sqrt(tr_rot / tr_tra)
    sc = np.array([1 / length] * 3 + [1.0] * 3)
    As = A * np.outer(sc, sc)
    w = np.linalg.eigvalsh(As)
    if not w[0] > 1e-10 * np.trace(As):
        raise ValueError("normal equations singular: a freedom is unconstrained")
    x = np.linalg.solve(As, -s[2:8] * sc) * sc
    om, v = x[:3], x[3:]
    th = np.linalg.norm(om)
    K = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-6:
        a, b = 1 - th * th / 6, 0.5 - th * th / 24
    else:
        a, b = np.sin(th) / th, (1 - np.cos(th)) / (th * th)
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + a * K + b * (K @ K)
    T[:3, 3] = v
    return T, rms


def icp_point_to_plane(src, tgt, tgt_normals, T0=None, max_iter=30, trim_q=0.5, gate_scale=20.0, max_d2=-1.0, tol=1e-9,
                       nn=None):
    """The loop of r3d_icp_iterate_plane (with d_src_orig) on the CPU: returns (T mapping the ORIGINAL src onto tgt, rms
    history).  nn(cur, tgt) -> (idx, d2): defaults to icp_ref's brute-force fp32 definition."""
    nn = nn or icp_ref.nearest_neighbours
    T_total = np.eye(4) if T0 is None else np.array(T0, dtype=np.float64)
    src = np.asarray(src, dtype=np.float32)
    start = icp_ref.apply_T32(src, T_total) if T0 is not None else src       # the cloud "when the state was reset"
    T_run = np.eye(4)
    cur = start
    hist = []
    for _ in range(max_iter):
        idx, d2 = nn(cur, tgt)
        s = plane_sums(cur, tgt, tgt_normals, idx, d2, max_d2, trim_q, gate_scale)
        T, rms = step_from_sums(s)
        hist.append(rms)
        T_run = T @ T_run
        cur = icp_ref.apply_T32(start, T_run)
        if np.abs(T - np.eye(4)).max() <= tol:
            break
    return T_run @ T_total, hist


def relative_pose(pose_a, pose_b):
    """T_ab (camera b -> camera a) from two pose-file rows (q xyzw, t) in the convention p_cam = R p_world + t."""
    def mat(q, t):
        x, y, z, w = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        return T
    return mat(*pose_a) @ np.linalg.inv(mat(*pose_b))
