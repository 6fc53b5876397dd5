#!/usr/bin/env python3
"""MI355X drop-in for the reference's octomap/txt_transfer_octomap.py: `X,Y,Z` world txt -> OctoMap .bt.

The reference is a module-level script with absolute paths of its author's disk; here the same steps sit in
main(input_txt, output_bt) (defaults from argv) and `txt_read(file_path, tree)` keeps its signature.  The
per-point `tree.updateNode(data, True)` loop becomes one bulk insert on the GPU.
"""
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "transfer"))
    import _common  # type: ignore
else:
    from ..transfer import _common

r3d = _common.package()
voxelmap = __import__(r3d.__name__ + ".voxelmap", fromlist=["OcTree"])
str_tofloat = r3d.str_tofloat


def txt_read(file_path, tree):
    """Every `X,Y,Z` line of file_path becomes a hit in `tree` (octomap/txt_transfer_octomap.py:16-28)."""
    pts = r3d.cloud_io.read_xyz_txt(file_path)
    print('the generation: ', 0)
    tree.insertPointCloud(pts)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    file_txt = argv[0] if len(argv) > 0 else './point_world/small_worldpoint_5_23_5.txt'
    file_bt = argv[1] if len(argv) > 1 else './bt/worldpoint.bt'
    tree = voxelmap.OcTree(0.1)
    txt_read(file_txt, tree)
    tree.updateInnerOccupancy()
    tree.writeBinary(bytes(file_bt, encoding='utf-8'))


if __name__ == '__main__':
    main()
