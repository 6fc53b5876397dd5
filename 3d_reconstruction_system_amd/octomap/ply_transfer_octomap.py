#!/usr/bin/env python3
"""MI355X drop-in for the reference's octomap/ply_transfer_octomap.py (== other_tools/ply_transfer_octomap.py):
ASCII PLY in the reference layout -> OctoMap .bt.

Reference quirks kept on purpose (octomap/ply_transfer_octomap.py:19-37): it discards EIGHT lines although the
header has seven, so the first vertex is dropped, and it stops after 5,400,001 vertices.
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
voxelmap = __import__(r3d.__name__ + ".voxelmap", fromlist=["OcTree"])
str_tofloat = r3d.str_tofloat

SKIPPED_LINES = 8
MAX_POINTS = 5400001


def txt_read(file_path, tree):
    with open(file_path, 'r') as f:
        lines = f.read().split('\n')
    rows = [s.split()[:3] for s in lines[SKIPPED_LINES:] if s.split()]
    rows = [r for r in rows if len(r) == 3][:MAX_POINTS]
    print('the generation: ', 0)
    tree.insertPointCloud(np.array(rows, dtype=np.float64).reshape(-1, 3))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    file_txt = argv[0] if len(argv) > 0 else './point/26_31_R-T.ply'
    file_bt = argv[1] if len(argv) > 1 else './bt/airsim_26_31_R-T.bt'
    tree = voxelmap.OcTree(0.1)
    txt_read(file_txt, tree)
    tree.updateInnerOccupancy()
    tree.writeBinary(bytes(file_bt, encoding='utf-8'))


if __name__ == '__main__':
    main()
