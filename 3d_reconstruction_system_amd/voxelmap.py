"""Occupied-voxel set on the MI355X + OctoMap ".bt" export: what the reference does point by point
through the third-party python-octomap binding (octomap/txt_transfer_octomap.py:16-36):

    tree = octomap.OcTree(0.1); tree.updateNode(xyz, True) ...; tree.updateInnerOccupancy(); tree.writeBinary(path)

`OcTree` below offers that same small surface; points are buffered on the host and inserted in bulk by the
HIP hash-set kernel, the pruned octree is serialised by the library's host code.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from .device import default_context


def format_bt(codes_sorted, resolution=0.1):
    """(.bt bytes, node count) for ascending unique 48-bit Morton codes.  Host only (no GPU needed)."""
    codes = np.ascontiguousarray(codes_sorted, dtype=np.uint64)
    lib = L.load()
    n, nodes = C.c_size_t(), C.c_int64()
    L.check(lib.r3d_octree_format_bt(codes.ctypes.data, codes.shape[0], float(resolution), None, 0, C.byref(n),
                                     C.byref(nodes)))
    buf = C.create_string_buffer(max(n.value, 1))
    L.check(lib.r3d_octree_format_bt(codes.ctypes.data, codes.shape[0], float(resolution), buf, n.value, C.byref(n),
                                     C.byref(nodes)))
    return buf.raw[:n.value], nodes.value


class VoxelSet:
    """HBM-resident hash set of occupied voxels (r3d_voxelset)."""

    def __init__(self, resolution=0.1, capacity=1 << 20, ctx=None):
        self.ctx = ctx or default_context()
        self.resolution = float(resolution)
        h = C.c_void_p()
        L.check(self.ctx.lib.r3d_voxelset_create(self.ctx.handle, self.resolution, int(capacity), C.byref(h)))
        self.handle = h.value
        self.ctx.adopt(self)

    def close(self):
        if self.handle:
            self.ctx.lib.r3d_voxelset_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        L.check(self.ctx.lib.r3d_voxelset_clear(self.handle))

    def insert(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise ValueError("cloud must be [N,3]")
        L.check(self.ctx.lib.r3d_voxelset_insert_host(self.handle, xyz.ctypes.data, xyz.shape[0]))

    def insert_device(self, d_xyz, n_points):
        L.check(self.ctx.lib.r3d_voxelset_insert(self.handle, d_xyz, int(n_points)))

    def insert_codes_device(self, d_codes, n_codes):
        L.check(self.ctx.lib.r3d_voxelset_insert_codes(self.handle, d_codes, int(n_codes)))

    def union_across(self, comm):
        """Collective over a comm.Comm: afterwards this set holds the occupied voxels of EVERY rank's set (config 5: frames
        sharded, one map).  Only distinct codes cross the fabric, through the C ABI's all-gather of unequal shards."""
        L.check(self.ctx.lib.r3d_voxelset_union(self.handle, comm.handle))

    def stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(self.ctx.lib.r3d_voxelset_stats(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return {"voxels": a.value, "ignored_points": b.value, "overflow": c.value}

    def codes(self):
        """Ascending unique Morton codes (uint64) of the occupied voxels."""
        n = C.c_int64()
        L.check(self.ctx.lib.r3d_voxelset_codes(self.handle, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint64)
        L.check(self.ctx.lib.r3d_voxelset_codes(self.handle, out.ctypes.data, out.shape[0], C.byref(n)))
        return out


def voxelize(xyz, resolution=0.1, ctx=None):
    """Ascending Morton codes of the voxels hit by an [N,3] cloud; the table is sized from N and regrown on overflow."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    cap = max(1 << 16, 2 * xyz.shape[0])
    while True:
        vs = VoxelSet(resolution, cap, ctx)
        try:
            vs.insert(xyz)
            st = vs.stats()
            if st["overflow"] == 0:
                return vs.codes(), st
        finally:
            vs.close()
        cap *= 4


class OcTree:
    """The slice of python-octomap's OcTree the reference scripts use."""

    def __init__(self, resolution):
        self.resolution = float(resolution)
        self._pending = []
        self._blocks = []
        self._codes = None

    def updateNode(self, point, occupied=True):
        if not occupied:
            raise NotImplementedError("the reference only inserts hits (updateNode(point, True))")
        self._pending.append((float(point[0]), float(point[1]), float(point[2])))
        self._codes = None

    def insertPointCloud(self, xyz):
        self._blocks.append(np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3))
        self._codes = None

    def _flush(self):
        if self._codes is None:
            blocks = list(self._blocks)
            if self._pending:
                blocks.append(np.array(self._pending, dtype=np.float64).astype(np.float32))
            pts = np.concatenate(blocks) if blocks else np.zeros((0, 3), np.float32)
            if pts.shape[0]:
                self._codes, self._stats = voxelize(pts, self.resolution)
            else:
                self._codes, self._stats = np.zeros(0, np.uint64), {"voxels": 0, "ignored_points": 0, "overflow": 0}
        return self._codes

    def updateInnerOccupancy(self):
        self._flush()

    def size(self):
        return format_bt(self._flush(), self.resolution)[1]

    def writeBinary(self, filename):
        if isinstance(filename, bytes):
            filename = filename.decode("utf-8")
        codes = self._flush()
        nodes = C.c_int64()
        L.check(L.load().r3d_octree_write_bt(os.fsencode(filename), codes.ctypes.data, codes.shape[0], self.resolution,
                                             C.byref(nodes)))
        return True
