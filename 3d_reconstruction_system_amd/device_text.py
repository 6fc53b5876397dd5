"""f1 on the device: the reference's txt lines and PLY rows of HBM-resident clouds, formatted by the GPU
(csrc/r3d_textfmt.hip) and copied into files by host threads -- no fp64 cloud crosses PCIe, the text does.

    w = TextWriter(ctx)
    w.add_ply('./ply/small_035_p8.ply', d_world, np.float64, n)                       # genply        c2w:112-134
    w.add_xyz_txt(paths, d_cam, np.float64, n, d_z_raw=d_depth, z_dtype=np.uint8)     # gentxtcord    c2w:73-83
    w.write()                                                                         # every file, side by side

Byte-identical to cloud_io.write_ply / write_xyz_txt (the host formatter, which checks this one in tests/test_gpu_textfmt.py).
A cloud the device formatter refuses (a "%.4f" coordinate of magnitude >= 2^40) is downloaded and written by the host
formatter instead: same bytes, no other difference.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from .device import depth_code, xyz_code

TEXT_XYZ_TXT, TEXT_PLY_ROWS, TEXT_PLY_ROWS_RGB = 0, 1, 2


class _TextFile(C.Structure):
    _fields_ = [("path", C.c_char_p), ("head", C.c_char_p), ("head_bytes", C.c_size_t), ("d_bytes", C.c_void_p),
                ("n_bytes", C.c_size_t), ("tail", C.c_char_p), ("tail_bytes", C.c_size_t)]


def ply_header_binary(n_points):
    """Header of cloud_io.write_ply_binary's file (a standard PLY header: no indents)."""
    return ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
            "end_header\n" % n_points).encode()


def ply_header(n_points, colour=False):
    """The header genply writes (c2w:122-129; genply_noRGB p2c:62-75 with colour): the template's 4-space indents included,
    up to and including the indent in front of the first vertex row."""
    extra = "    property uchar red\n    property uchar green\n    property uchar blue\n    property uchar alpha\n" if colour else ""
    return ("ply\n    format ascii 1.0\n    element vertex %d\n    property float x\n    property float y\n    property float z\n"
            "%s    end_header\n    " % (n_points, extra)).encode()


PLY_TRAILER = b"\n    "


def format_text(ctx, kind, d_xyz, dtype, n_points, d_aux=None, aux_code=0, segment_points=0, d_text=None, text_cap=0):
    """r3d_format_text_device: (n_bytes, segment offsets [n_segments + 1]).  d_text None: sizes only."""
    n_seg = 0 if n_points == 0 else (1 if segment_points <= 0 else -(-n_points // segment_points))
    offs = np.zeros(n_seg + 1, np.int64)
    total = C.c_int64(0)
    L.check(ctx.lib.r3d_format_text_device(ctx.handle, int(kind), d_xyz, xyz_code(dtype), int(n_points), d_aux, int(aux_code),
                                           int(segment_points), d_text, int(text_cap), offs.ctypes.data, C.byref(total)))
    return total.value, offs


class TextWriter:
    """Collects the text files of device-resident clouds, then formats and writes them all at once: one device text buffer,
    every file a range of it, the files written side by side by the library's host threads."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._jobs = []       # (kind, d_xyz, dtype, n, d_aux, aux_code, segment_points, n_bytes, offsets, files)
        self._raw = []        # (path, head, device pointer, bytes, tail): device bytes that go into a file as they are
        self._host_jobs = []  # callables: what the device formatter refused

    def _add(self, kind, d_xyz, dtype, n, d_aux, aux_code, seg, files_of):
        try:
            n_bytes, offs = format_text(self.ctx, kind, d_xyz, dtype, n, d_aux, aux_code, seg)
        except L.R3DError as e:
            if e.code != L.ERR_UNSUPPORTED:
                raise
            return False
        self._jobs.append((kind, d_xyz, dtype, n, d_aux, aux_code, seg, n_bytes, offs, files_of(offs)))
        return True

    def add_ply(self, path, d_xyz, dtype, n_points, d_rgb=None, rgb_stride=3):
        """genply's file (c2w:112-134), or genply_noRGB's (p2c:55-91) when d_rgb is given ([n][3] bytes, or rgba words with
        rgb_stride=4)."""
        path = os.fspath(path)
        kind = TEXT_PLY_ROWS if d_rgb is None else TEXT_PLY_ROWS_RGB
        head = ply_header(n_points, d_rgb is not None)
        if n_points == 0 or not self._add(kind, d_xyz, dtype, n_points, d_rgb, rgb_stride if d_rgb is not None else 0, 0,
                                          lambda offs: [(path, head, 0, int(offs[-1]), PLY_TRAILER)]):
            self._host_jobs.append(lambda: self._host_ply(path, d_xyz, dtype, n_points, d_rgb, rgb_stride))

    def add_ply_binary(self, path, d_xyz_f32, n_points):
        """cloud_io.write_ply_binary's file from a device-resident FLOAT32 cloud: header + the cloud's own bytes."""
        self._raw.append((os.fspath(path), ply_header_binary(n_points), d_xyz_f32 if n_points else None, n_points * 12, b""))

    def add_xyz_txt(self, paths, d_xyz, dtype, n_points, d_z_raw=None, z_dtype=None):
        """`X,Y,Z\\n` files with repr() floats: file k takes rows [k*P, (k+1)*P), P = n_points / len(paths) -- the
        ./point/<stem>.txt of the frame loop (c2w:163-165) with the raster's integers as third column, or a world txt."""
        paths = [os.fspath(p) for p in paths]
        if not paths:
            if n_points:
                raise ValueError("points but no files to put them in")
            return
        if n_points % len(paths):
            raise ValueError("%d points do not divide into %d files" % (n_points, len(paths)))
        per = n_points // len(paths)
        aux_code = depth_code(z_dtype) if d_z_raw is not None else 0
        if d_z_raw is not None and np.dtype(z_dtype) not in (np.dtype(np.uint8), np.dtype(np.uint16)):
            raise TypeError("the integer third column is uint8 or uint16")
        if n_points == 0:
            self._host_jobs.append(lambda: [open(p, "wb").close() for p in paths])
            return
        ok = self._add(TEXT_XYZ_TXT, d_xyz, dtype, n_points, d_z_raw, aux_code, per,
                       lambda offs: [(p, b"", int(offs[k]), int(offs[k + 1] - offs[k]), b"") for k, p in enumerate(paths)])
        assert ok            # repr() rows are never refused

    def _host_ply(self, path, d_xyz, dtype, n, d_rgb, stride):
        from . import cloud_io
        xyz = np.empty((n, 3), dtype)
        if n:
            L.check(self.ctx.lib.r3d_download(self.ctx.handle, xyz.ctypes.data, d_xyz, xyz.nbytes))
        if d_rgb is None:
            return cloud_io.write_ply(path, xyz)
        col = np.empty((n, stride), np.uint8)
        if n:
            L.check(self.ctx.lib.r3d_download(self.ctx.handle, col.ctypes.data, d_rgb, col.nbytes))
        cloud_io.write_ply_rgb(path, xyz, col[:, :3])

    def write(self):
        """Formats every collected text into one device buffer and writes all files; returns the bytes of text written."""
        ctx = self.ctx
        from .transfer import _common
        total = sum(j[7] for j in self._jobs)
        raw_total = sum(r[3] for r in self._raw)
        files, keep = [], []
        buf = ctx.alloc(max(total, 16)) if self._jobs else None
        try:
            _common.stamp("text buffer allocated (%d bytes)" % total)
            at = 0
            for kind, d_xyz, dtype, n, d_aux, aux_code, seg, n_bytes, offs, jf in self._jobs:
                got, _ = format_text(ctx, kind, d_xyz, dtype, n, d_aux, aux_code, seg, buf.ptr + at, n_bytes)
                if got != n_bytes:
                    raise RuntimeError("device text changed size between the two passes (%d -> %d bytes)" % (n_bytes, got))
                for path, head, off, nb, tail in jf:
                    keep += [os.fsencode(path), head, tail]
                    files.append(_TextFile(keep[-3], head or None, len(head), buf.ptr + at + off, nb, tail or None, len(tail)))
                at += n_bytes
            for path, head, ptr, nb, tail in self._raw:
                keep += [os.fsencode(path), head, tail]
                files.append(_TextFile(keep[-3], head or None, len(head), ptr, nb, tail or None, len(tail)))
            ctx.sync()
            _common.stamp("text formatted (pass A + scan + pass B per cloud)")
            if files:
                arr = (_TextFile * len(files))(*files)
                L.check(ctx.lib.r3d_write_device_text_files(ctx.handle, arr, len(files)))
            for job in self._host_jobs:
                job()
        finally:
            if buf is not None:
                buf.free()
            self._jobs, self._raw, self._host_jobs = [], [], []
        return total + raw_total
