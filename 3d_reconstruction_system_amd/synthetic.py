"""Synthetic scenes (data generators, no compute path): depth rasters + poses of a pinhole camera inside a box room.

The reference ships no sample data (its depth images and poses live behind a cloud-drive link, data/*.md), so the
benchmarks, smoke test and parity tests of the registration path feed on these.  Poses follow the pose file's convention
(camera_to_world.py:57-59, 155-158): p_cam = R p_world + t, quaternion scalar-last.
"""
import numpy as np

ROOM_LO = np.array([-4.0, -1.5, -3.0])
ROOM_HI = np.array([4.0, 1.5, 3.0])


def room_view(h, w, yaw, centre, fx=None, fy=None, pitch=0.0, lo=ROOM_LO, hi=ROOM_HI):
    """z-depth raster (float64 [h,w]) of the inside of the box [lo, hi] seen from `centre` by a camera turned by `yaw` about
    the vertical (y) axis and then by `pitch` about its own x axis; returns (depth, q_xyzw, t, (fx, fy, cx, cy))."""
    fx = 0.8 * w if fx is None else fx
    fy = fx if fy is None else fy
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    u = (np.arange(w) - cx) / fx
    v = (np.arange(h) - cy) / fy
    rays = np.stack([np.broadcast_to(u[None, :], (h, w)), np.broadcast_to(v[:, None], (h, w)), np.ones((h, w))], -1)
    cyaw, syaw, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cyaw, 0, syaw], [0, 1, 0], [-syaw, 0, cyaw]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rwc = Ry @ Rx                                              # camera -> world
    c = np.asarray(centre, dtype=np.float64)
    d = rays @ Rwc.T                                            # ray directions in the world, per unit camera z
    with np.errstate(divide="ignore", invalid="ignore"):
        t_hi = np.where(d > 0, (hi - c) / d, np.inf)
        t_lo = np.where(d < 0, (lo - c) / d, np.inf)
    z = np.minimum(t_hi, t_lo).min(-1)                          # camera-frame z at the first wall hit
    R = Rwc.T                                                   # world -> camera
    t = -R @ c
    # quaternion of R = Rx(-pitch) Ry(-yaw), scalar-last
    qy = np.array([0.0, np.sin(-yaw / 2), 0.0, np.cos(-yaw / 2)])
    qx = np.array([np.sin(-pitch / 2), 0.0, 0.0, np.cos(-pitch / 2)])
    q = quat_mul(qx, qy)
    return z, q, t, (fx, fy, cx, cy)


def quat_mul(a, b):
    """Hamilton product a * b of scalar-last quaternions (rotation b first, then a)."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def room_views(n_frames, h, w, seed=0):
    """n_frames views turning once about the vertical axis near the room's centre: (depth f32 [F,h,w], q [F,4], t [F,3], K)."""
    rng = np.random.default_rng(seed)
    depths, quats, ts, K = [], [], [], None
    for f in range(n_frames):
        c = rng.uniform(-0.8, 0.8, 3) * np.array([1.0, 0.3, 1.0])
        z, q, t, K = room_view(h, w, 2 * np.pi * f / n_frames + 0.1, c)
        depths.append(z.astype(np.float32))
        quats.append(q)
        ts.append(t)
    return np.stack(depths), np.array(quats), np.array(ts), K


def pose_matrix(q_xyzw, t):
    """4x4 world -> camera matrix [R t; 0 1] of a pose-file row (unit quaternion assumed)."""
    x, y, z, w = np.asarray(q_xyzw, dtype=np.float64) / np.linalg.norm(q_xyzw)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


def two_views(h=480, w=640, yaw_deg=15.0, baseline=(0.35, 0.05, -0.2), depth_noise=0.0, seed=0, pitch_deg=3.0,
              yaw_a=0.9, pitch_a=0.2):
    """Two partially overlapping single views of the room, the registration problem of readme.md:25:
    returns dict(depth_a, depth_b [h,w] f32, pose_a, pose_b (q, t), K, T_ab) where T_ab maps camera-b coordinates onto
    camera-a coordinates (the T_data.txt that merges ./point/b.txt into ./point/a.txt's frame, icp:99-108).
    View a looks into a corner of the room, tilted so that two walls AND the floor are in sight (one wall alone leaves three
    freedoms of a rigid motion open); view b is turned by yaw_deg / pitch_deg from there and moved by `baseline`."""
    rng = np.random.default_rng(seed)
    ca = np.array([0.3, -0.1, 0.4])
    cb = ca + np.asarray(baseline, dtype=np.float64)
    za, qa, ta, K = room_view(h, w, yaw_a, ca, pitch=pitch_a)
    zb, qb, tb, _ = room_view(h, w, yaw_a + np.deg2rad(yaw_deg), cb, pitch=pitch_a + np.deg2rad(pitch_deg))
    if depth_noise > 0:
        za = za * (1.0 + rng.normal(size=za.shape) * depth_noise)
        zb = zb * (1.0 + rng.normal(size=zb.shape) * depth_noise)
    Ta, Tb = pose_matrix(qa, ta), pose_matrix(qb, tb)
    return {"depth_a": za.astype(np.float32), "depth_b": zb.astype(np.float32), "pose_a": (qa, ta), "pose_b": (qb, tb), "K": K,
            "T_ab": Ta @ np.linalg.inv(Tb)}
