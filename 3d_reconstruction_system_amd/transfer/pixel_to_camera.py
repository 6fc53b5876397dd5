#!/usr/bin/env python3
# -*- coding:utf-8 -*-
"""MI355X drop-in for the reference's transfer/pixel_to_camera.py: one depth PNG -> camera-frame
points (txt) -> PLY.

Same entry point and defaults (`./depth/24.png` -> `./point/24.txt` -> `./ply/24.ply`).  As
shipped, the reference script crashes after writing the txt (it calls genply_RGB with three
arguments, p2c:136 vs :98), its coloured writer needs an un-imported PIL (p2c:58) and
gentxtcord is hard-wired to 480x640 (p2c:34-35).  Here the raster size comes from the image, both
writers work, and main() does what its three-argument call (points, ./img/24.png, ply path) spells out: the COLOURED PLY
when that colour image exists, the plain PLY otherwise.
"""
import os
import sys
import time

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _common  # type: ignore
else:
    from . import _common

r3d = _common.package()


def gentxtcord(filename, depth):
    """Camera-frame points of `depth` written as `X,Y,Z` lines; returns [xcord, ycord, zcord]
    (three Python lists, Z as the raster's own integers) like p2c:24-44."""
    depth = np.ascontiguousarray(depth)
    cam = r3d.unproject(depth, intrinsics=_common.intrinsics(), out_dtype=np.float64, ctx=_common.context())
    integral = depth.dtype in (np.uint8, np.uint16)
    r3d.cloud_io.write_xyz_txt(filename, cam, z_raw=depth if integral else None)
    zs = depth.reshape(-1).tolist() if integral else cam[:, 2].tolist()
    return [cam[:, 0].tolist(), cam[:, 1].tolist(), zs]


def _as_xyz(gtxyz):
    n = len(gtxyz[0])
    xyz = np.empty((n, 3), dtype=np.float64)
    xyz[:, 0] = gtxyz[0]
    xyz[:, 1] = gtxyz[1]
    xyz[:, 2] = gtxyz[2]
    return xyz


def genply_noRGB(gtxyz, imgpath, pc_file):
    """The COLOURED writer (the reference's names are swapped): x y z R G B 0 rows with the
    uchar red/green/blue/alpha header of p2c:55-91."""
    t1 = time.time()
    # the bytes PIL's Image.open(imgpath) gives (p2c:58): PNG and JPEG decoded natively, anything else by PIL itself
    img = r3d.cloud_io.read_rgb_batch([imgpath])[0]
    xyz = _as_xyz(gtxyz)
    if img.shape[0] * img.shape[1] != xyz.shape[0]:
        raise ValueError("colour image has %d pixels, cloud has %d points" % (img.shape[0] * img.shape[1], xyz.shape[0]))
    r3d.cloud_io.write_ply_rgb(pc_file, xyz, img.reshape(-1, 3))
    print("Write into .ply file Done.", time.time() - t1)


def genply_RGB(gtxyz, pc_file):
    """The PLAIN writer (p2c:98-124), native formatter, reference byte layout."""
    r3d.cloud_io.write_ply(pc_file, _as_xyz(gtxyz))
    print("Write into .ply file Done.")


def main():
    num = 24
    depth_path = './depth/' + str(num) + '.png'
    point_path = './point/' + str(num) + '.txt'
    imgpath = './img/' + str(num) + '.png'
    pc_file = './ply/' + str(num) + '.ply'
    gt = r3d.cloud_io.read_depth_unchanged(depth_path)
    gray_img = gt[:, :, 1] if gt.ndim == 3 else gt   # p2c:134 takes channel 1 of a 3-channel PNG
    gt_cord = gentxtcord(point_path, gray_img)
    if os.path.exists(imgpath):
        genply_noRGB(gt_cord, imgpath, pc_file)      # p2c:136 passes (gt_cord, imgpath, pc_file): the coloured writer's signature
    else:
        genply_RGB(gt_cord, pc_file)


if __name__ == '__main__':
    main()
