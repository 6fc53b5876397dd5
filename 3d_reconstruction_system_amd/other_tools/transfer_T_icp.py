#!/usr/bin/env python3
"""MI355X drop-in for the reference's other_tools/transfer_T_icp.py: merge two camera-frame
clouds, the second one moved by the 4x4 in T_data.txt.

The reference file is a module-level script (it runs on import) that only APPLIES a T obtained
from an external ICP tool.  Here the same work sits in main() behind a __main__ guard, the apply
runs on the GPU, and `estimate_T()` adds the missing step: it computes T_data.txt on the GPU
(icp.icp_similarity: moments init -> symmetric dead-zone ICP -> plain ICP with exact culled NN, fused
cross-covariance sums and a device-side Umeyama solve; no initial guess needed, closes the
monocular-depth vs COLMAP scale gap of readme.md:25,104) so the external tool is no longer needed.
Run `python transfer_T_icp.py` for the reference behaviour, `python transfer_T_icp.py --estimate`
to (re)compute T_data.txt from ./point/24.txt -> ./point/0.txt first.

`--estimate-rigid` is the reference's own case (readme.md:25: "match the point clouds corresponding to two images"):
./point/0.txt and ./point/24.txt are two PARTIALLY OVERLAPPING single-view camera clouds in the same (depth-map) unit, so
the T between them is rigid.  It runs point-to-plane ICP on the GPU (icp.icp_point_to_plane: normals from the target's
pixel raster, per-direction-class rank gate, 6x6 solve on the device), from `--init FILE` (a 4x4 in get_T's format) or from
the relative COLMAP pose of the two images (`--colmap POSEFILE NAME_A NAME_B [--colmap-scale S]`, S = guess of the unit
ratio, default 1), and writes T_data.txt.  With --colmap it also prints and writes ./scale.txt = |t_icp| / |t_colmap|, the
factor that brings COLMAP's translations to the depth maps' unit (readme.md:25).
"""
import os
import sys

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "transfer"))
    import _common  # type: ignore
else:
    from ..transfer import _common

r3d = _common.package()

str_tofloat = r3d.str_tofloat
get_r = r3d.get_r
scipy_transfer = r3d.scipy_transfer
get_T = r3d.get_T


def point_camera(p1, r_inverse):
    """(T . p1) for one homogeneous point, on the host as icp:10-12."""
    p_world = np.dot(r_inverse, (p1).T)
    return np.array(p_world.T)


def genply(gtxyz, pc_file, lenth_point):
    xyz = np.empty((lenth_point, 3), dtype=np.float64)
    xyz[:, 0] = gtxyz[0]
    xyz[:, 1] = gtxyz[1]
    xyz[:, 2] = gtxyz[2]
    r3d.cloud_io.write_ply(pc_file, xyz)
    print("Write into .ply file Done.")


def local_world(path_local, file_write, T, xcord, ycord, zcord, flag):
    """Read a camera txt; flag=True moves it by T on the GPU, flag=False passes it through;
    append to the caller's lists and to the open file `file_write` (icp:71-97)."""
    print('start transfer')
    pts = r3d.cloud_io.read_xyz_txt(path_local)
    if flag:
        pts = r3d.apply_T(pts, T, ctx=_common.context())
    xcord.extend(pts[:, 0].tolist())
    ycord.extend(pts[:, 1].tolist())
    zcord.extend(pts[:, 2].tolist())
    text = r3d.cloud_io.format_xyz_txt(pts)
    file_write.write(text if 'b' in getattr(file_write, 'mode', 'w') else text.decode('ascii'))


def estimate_T(path_source='./point/24.txt', path_target='./point/0.txt', path_T='T_data.txt', **kw):
    """Similarity ICP source -> target on the GPU; writes T in the format get_T() reads."""
    src = r3d.cloud_io.read_xyz_txt(path_source)
    tgt = r3d.cloud_io.read_xyz_txt(path_target)
    icp = __import__(r3d.__name__ + ".icp", fromlist=["icp_similarity"])
    T, info = icp.icp_similarity(src, tgt, ctx=_common.context(), **kw)
    r3d.write_T(path_T, T)
    return T, info


def infer_raster_shape(pts):
    """(H, W) of a camera txt written by gentxtcord (p2c:34-44: row-major, X = (i - cx)/fx * Z): X/Z climbs along a row and
    falls back at every row start.  Z = 0 pixels are skipped; raises ValueError when the rows cannot be told apart."""
    pts = np.asarray(pts, dtype=np.float64)
    n = pts.shape[0]
    valid = np.flatnonzero(np.isfinite(pts).all(axis=1) & (pts[:, 2] != 0))
    if valid.size < 4:
        raise ValueError("too few pixels with depth to find the raster width")
    u = pts[valid, 0] / pts[valid, 2]
    span = u.max() - u.min()
    drops = valid[1:][np.diff(u) < -0.5 * span]
    if drops.size == 0:
        raise ValueError("cloud is not organised in raster rows (X/Z never falls back)")
    steps = np.diff(np.concatenate([[0], drops]))
    w = int(np.bincount(steps).argmax())
    if w < 2 or n % w != 0 or np.mean(steps == w) < 0.8:
        raise ValueError("no regular raster rows found (best width %d for %d points): pass the shape explicitly" % (w, n))
    return n // w, w


def _pose_of(pose_file, name):
    for png, q, t in zip(*r3d.read_pose_file(pose_file)):
        if png == name or os.path.splitext(png)[0] == os.path.splitext(name)[0]:
            return q, t
    raise ValueError("no pose line for image '%s' in %s" % (name, pose_file))


def estimate_T_rigid(path_source='./point/24.txt', path_target='./point/0.txt', path_T='T_data.txt', shape=None, init=None,
                     colmap=None, colmap_scale=1.0, path_scale='./scale.txt', **kw):
    """Rigid point-to-plane ICP source -> target on the GPU (two single-view clouds); writes T in get_T()'s format.
    colmap = (pose_file, name_target, name_source): start from COLMAP's relative pose (translation x colmap_scale) and
    report / write the scale |t_icp| / |t_colmap|.  Returns (T, info)."""
    src = r3d.cloud_io.read_xyz_txt(path_source)
    tgt = r3d.cloud_io.read_xyz_txt(path_target)
    icp = __import__(r3d.__name__ + ".icp", fromlist=["icp_point_to_plane"])
    if shape is None:
        shape = infer_raster_shape(tgt)
    poses = None
    if colmap is not None:
        pose_file, name_a, name_b = colmap
        poses = (_pose_of(pose_file, name_a), _pose_of(pose_file, name_b))
        T_rel = icp.scale_from_baselines(np.eye(4), *poses)[1]
        if init is None:
            init = T_rel.copy()
            init[:3, 3] *= float(colmap_scale)
    T, info = icp.icp_point_to_plane(src, tgt, tgt_shape=shape, init=init, ctx=_common.context(), **kw)
    r3d.write_T(path_T, T)
    info["raster_shape"] = tuple(int(v) for v in shape)
    if poses is not None:
        info["scale"] = icp.scale_from_baselines(T, *poses)[0]
        with open(path_scale, 'w') as f:
            f.write(repr(float(info["scale"])) + "\n")
    return T, info


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    path_T = 'T_data.txt'
    path_world = './point_world/03_testT.txt'
    path_ply = './ply/icp/024.ply'
    if '--estimate-rigid' in argv:
        kw = {}
        if '--shape' in argv:
            k = argv.index('--shape')
            kw["shape"] = (int(argv[k + 1]), int(argv[k + 2]))
        if '--init' in argv:
            kw["init"] = get_T(argv[argv.index('--init') + 1])
        if '--colmap' in argv:
            k = argv.index('--colmap')
            kw["colmap"] = (argv[k + 1], argv[k + 2], argv[k + 3])
        if '--colmap-scale' in argv:
            kw["colmap_scale"] = float(argv[argv.index('--colmap-scale') + 1])
        T, info = estimate_T_rigid(path_T=path_T, **kw)
        print('point-to-plane ICP on a %dx%d raster: %d iterations, %d pairs, rms %.6g'
              % (info["raster_shape"] + (info["iterations"], int(info["pairs"]), info["rms_history"][-1])))
        if "scale" in info:
            print('scale |t_icp| / |t_colmap| = %r' % info["scale"])
    elif '--estimate' in argv:
        T, info = estimate_T(path_T=path_T)
        print('ICP: %d coarse + %d fine iterations, rms %.6g' % (info["coarse_iterations"], info["iterations"],
                                                                  info["rms_history"][-1]))
    T = get_T(path_T)
    xcord, ycord, zcord = [], [], []
    with open(path_world, 'w') as file_w:
        local_world('./point/0.txt', file_w, T, xcord, ycord, zcord, False)
        local_world('./point/24.txt', file_w, T, xcord, ycord, zcord, True)
    genply([xcord, ycord, zcord], path_ply, len(xcord))


if __name__ == '__main__':
    main()
