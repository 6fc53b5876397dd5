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


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    path_T = 'T_data.txt'
    path_world = './point_world/03_testT.txt'
    path_ply = './ply/icp/024.ply'
    if '--estimate' in argv:
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
