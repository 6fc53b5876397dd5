#!/usr/bin/env python3
# -*- coding:utf-8 -*-
"""MI355X drop-in for the reference's transfer/camera_to_world.py.

Same entry point (`python camera_to_world.py` from a directory holding ./camera_pose, ./depth,
./point, ./point_world, ./ply), same functions, same files written.  What changed is where the
per-point arithmetic runs: all frames of the pose file are unprojected and moved to the world
frame by ONE fused HIP launch (fp64 registers, reference evaluation order) instead of two Python
loops per frame with a text file between them.  The txt/PLY text is produced ON THE GPU from its
fp64 result (csrc/r3d_textfmt.hip), byte-compatible with Python's repr()/"%.4f"; the host copies it into files.

Several GPUs: start it under a one-process-per-GPU launcher, e.g.
    python -m torch.distributed.run --nproc-per-node 8 camera_to_world.py
and the pose file's frames are cut into contiguous blocks, one per rank; each rank decodes and fuses its block, one
RCCL all-gather (through the library's C ABI, no torch in this process) assembles the world cloud, every rank writes
the ./point/<stem>.txt files of ITS frames, rank 0 writes the world txt and the PLY.  Same bytes as one GPU.

Deliberate differences from the reference (all documented in DESIGN.md):
  * str_tofloat uses float() (np.float no longer exists);
  * a missing depth image raises FileNotFoundError naming the file (reference: TypeError on None);
  * nothing else: Z=0 pixels are kept, the world txt holds the last frame only, the PLY holds all.
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

# ---- little tools (c2w:28-59) ----------------------------------------------------------------
str_tofloat = r3d.str_tofloat
get_r = r3d.get_r
scipy_transfer = r3d.scipy_transfer


def point_camera(p1, r_inverse, t):
    """One point (or an [N,3] block) camera -> world: Rinv . (p1 - t).  Kept on the host for
    single points, exactly as c2w:57-59; blocks go to the GPU."""
    p1 = np.asarray(p1, dtype=np.float64)
    if p1.ndim == 2 and p1.shape[0] > 64:
        return r3d.se3_apply(p1, np.asarray(r_inverse), np.asarray(t, dtype=np.float64), ctx=_common.context())
    p_world = np.dot(r_inverse, (p1 - t).T)
    return np.array(p_world.T)


def read_pfm(path, unchanged=False):
    """Portable Float Map as `cv.imread(path)` returns it with DEFAULT flags (c2w:35): uint8 H x W x 3, BGR -- the float
    samples divided by |scale|, rows flipped to top-down, grey replicated to three channels, then saturate_cast<uchar>
    (round half to even, clamp to 0..255).  Restated from OpenCV's PFM reader (modules/imgcodecs/src/grfmt_pfm.cpp) and
    imread's "no IMREAD_ANYDEPTH => 8 bit" rule; cv2 is not in this image, so it is pinned by formula only.
    unchanged=True gives the float32 samples instead (IMREAD_UNCHANGED: H x W, or H x W x 3 in BGR order)."""
    with open(path, 'rb') as f:
        kind = f.readline().strip()
        if kind not in (b'Pf', b'PF'):
            raise ValueError("%s is not a PFM file" % path)
        w, h = [int(v) for v in f.readline().split()]
        scale = float(f.readline().strip())
        if scale == 0.0:
            raise ValueError("%s: PFM scale factor 0" % path)
        ch = 3 if kind == b'PF' else 1
        data = np.frombuffer(f.read(w * h * ch * 4), dtype='<f4' if scale < 0 else '>f4').astype(np.float32)
    img = data.reshape(h, w, ch)[::-1] * np.float32(1.0 / abs(scale))
    img = img[:, :, ::-1] if ch == 3 else img[:, :, 0]                              # colour: BGR like OpenCV
    if unchanged:
        return np.ascontiguousarray(img, dtype=np.float32)
    if ch == 1:
        img = np.repeat(img[:, :, None], 3, axis=2)
    with np.errstate(invalid='ignore'):
        u8 = np.clip(np.rint(np.nan_to_num(img, nan=0.0)), 0, 255).astype(np.uint8)  # np.rint rounds half to even, as cvRound
    return np.ascontiguousarray(u8)


def sfm2npy(transfer_name):
    """./pfm/<name>.pfm -> ./npy/<name>.npy (c2w:32-38; a COLMAP/MVS depth map converter, not on the fusion path)."""
    pfm_path = './pfm/' + transfer_name + '.pfm'
    npy_path = './npy/' + transfer_name + '.npy'
    mat = read_pfm(pfm_path)
    print(type(mat))
    np.save(npy_path, mat)
    return npy_path


# ---- conversion functions (c2w:67-105) -------------------------------------------------------
def gentxtcord(filename, depth):
    """Camera-frame `X,Y,Z` text of one raster (c2w:67-83).  Returns None like the reference."""
    depth = np.ascontiguousarray(depth)
    cam = r3d.unproject(depth, intrinsics=_common.intrinsics(), out_dtype=np.float64, ctx=_common.context())
    z_raw = depth if depth.dtype in (np.uint8, np.uint16) else None
    r3d.cloud_io.write_xyz_txt(filename, cam, z_raw=z_raw)


def get_pointdata(p_path, q, t, xcord, ycord, zcord):
    """Read a camera txt, move it to the world frame with pose (q, t), append to the caller's
    three lists and rewrite ./point_world/small_worldpoint_5_23_5.txt (c2w:86-105)."""
    point_world_path = './point_world/small_worldpoint_5_23_5.txt'
    cam = r3d.cloud_io.read_xyz_txt(p_path)
    world = r3d.se3_apply(cam, np.asarray(scipy_transfer(q)), np.asarray(t, dtype=np.float64),
                          ctx=_common.context())
    xcord.extend(world[:, 0].tolist())
    ycord.extend(world[:, 1].tolist())
    zcord.extend(world[:, 2].tolist())
    r3d.cloud_io.write_xyz_txt(point_world_path, world)


# ---- visualisation (c2w:112-134) -------------------------------------------------------------
def genply(gtxyz, pc_file, lenth_point):
    """ASCII PLY in the reference's exact layout.  gtxyz = [xs, ys, zs] (or an [N,3] array)."""
    if isinstance(gtxyz, np.ndarray) and gtxyz.ndim == 2 and gtxyz.shape[1] == 3:
        xyz = gtxyz[:lenth_point]
    else:
        xyz = np.empty((lenth_point, 3), dtype=np.float64)
        xyz[:, 0] = gtxyz[0]
        xyz[:, 1] = gtxyz[1]
        xyz[:, 2] = gtxyz[2]
    r3d.cloud_io.write_ply(pc_file, xyz)
    print("Write into .ply file Done.")


# ---- flow (c2w:138-174) ----------------------------------------------------------------------
def fuse_pose_file(qt_path, depth_dir='./depth/', out_dtype=np.float64):
    """Parse the pose file, load every depth raster and fuse all frames in one launch.
    Returns (names, depths [F,H,W], world [F*H*W,3])."""
    _common.stamp(None)
    names, quats, ts = r3d.read_pose_file(qt_path)
    if not names:
        return names, np.empty((0, 0, 0), np.uint8), np.empty((0, 3), out_dtype)
    depths = r3d.cloud_io.read_depth_batch([os.path.join(depth_dir, n) for n in names])
    _common.stamp("pose file + %d depth files decoded" % len(names))
    ts = ts * _common.pose_scale()               # 1 unless R3D_POSE_SCALE says otherwise (COLMAP unit -> depth unit)
    ctx = _common.context()
    _common.stamp("GPU context")
    world = r3d.fuse_frames(depths, quats, ts, intrinsics=_common.intrinsics(), out_dtype=out_dtype, ctx=ctx)
    _common.stamp("fused launch, world cloud in host memory")
    return names, depths, world


def _write_camera_txts(names, cam, depths, per):
    """./point/<stem>.txt for every frame, one native call: the files are spread over the host threads (one whole file per
    thread at a time)."""
    integral = depths.dtype in (np.uint8, np.uint16)
    r3d.cloud_io.write_xyz_txt_batch(['./point/' + n[0:-4] + '.txt' for n in names], cam[:len(names) * per],
                                     z_raw=depths[:len(names)] if integral else None)


def _get_file_name_sharded(qt_path):
    """get_file_name when a launcher started one process per GPU (WORLD_SIZE > 1): BASELINE config 4.  Every rank fuses its
    block of frames into its slot of the world cloud, one all-gather (C ABI, RCCL) assembles the cloud in HBM on every rank,
    and the text is formatted where the data is: every rank writes the ./point/<stem>.txt of ITS frames from its GPU, rank 0
    the world txt and the fused PLY from the gathered cloud -- same bytes as one GPU (R3D_HOST_TEXT=1: through host memory
    and the host formatter, rounds 2-4's way)."""
    D = _common.module("dist")
    ctx, comm = _common.sharded_context()
    if comm.rank == 0:
        print('data start transfer')
    t1 = time.time()
    host_text = os.environ.get("R3D_HOST_TEXT", "0") not in ("", "0")
    names, lo, hi, depths, world = D.fuse_pose_file_sharded(qt_path, './depth/', _common.intrinsics(), np.float64, ctx, comm,
                                                            pose_scale=_common.pose_scale(), keep_on_device=not host_text)
    n_frames = len(names)
    if host_text:
        return _sharded_files_host_text(names, lo, hi, depths, world, ctx, comm, t1)
    bufs = [b for b in (world["d_full"], world["d_depth"]) if b is not None]
    try:        # a rank that fails while writing its files must still meet the others at the barrier, then raise
        per = world["per"]
        text = r3d.device_text.TextWriter(ctx)
        if comm.rank == 0:
            if n_frames:
                text.add_ply('./ply/small_035_p8.ply', world["d_full"].ptr, np.float64, n_frames * per)
            else:
                r3d.cloud_io.write_ply('./ply/small_035_p8.ply', np.empty((0, 3)))
        if n_frames and not _common.skip_intermediate():
            if hi > lo:
                cam = ctx.camera(depths.shape[1], depths.shape[2], *_common.intrinsics())
                d_cam = ctx.alloc((hi - lo) * per * 24)
                bufs.append(d_cam)
                r3d.unproject_device(ctx, cam, world["d_depth"].ptr, depths.dtype, hi - lo, d_cam.ptr, np.float64)
                integral = depths.dtype in (np.uint8, np.uint16)
                text.add_xyz_txt(['./point/' + nm[0:-4] + '.txt' for nm in names[lo:hi]], d_cam.ptr, np.float64, (hi - lo) * per,
                                 d_z_raw=world["d_depth"].ptr if integral else None, z_dtype=depths.dtype if integral else None)
            if comm.rank == 0:
                text.add_xyz_txt(['./point_world/small_worldpoint_5_23_5.txt'], world["d_full"].ptr + (n_frames - 1) * per * 24, np.float64, per)
        text.write()
        t2 = time.time()
        if comm.rank == 0:
            print('##################')
            print("%d frames cost ." % n_frames, t2 - t1)
            print("Write into .ply file Done.")
    finally:
        for b in bufs:
            try:
                b.free()
            except Exception:
                pass
        comm.barrier()                         # nobody leaves (and tears RCCL down) while rank 0 still needs its peers
        comm.close()


def _sharded_files_host_text(names, lo, hi, depths, world, ctx, comm, t1):
    n_frames = len(names)
    ply_done = None
    try:        # a rank that fails while writing its files must still meet the others at the barrier, then raise
        if comm.rank == 0:
            ply_done = _ply_in_background(world, './ply/small_035_p8.ply')
        if n_frames and not _common.skip_intermediate():
            if hi > lo:
                per = depths.shape[1] * depths.shape[2]
                cam = r3d.unproject(depths, intrinsics=_common.intrinsics(), out_dtype=np.float64, ctx=ctx)
                _write_camera_txts(names[lo:hi], cam, depths, per)          # this rank's frames only
            if comm.rank == 0:
                per = world.shape[0] // n_frames
                r3d.cloud_io.write_xyz_txt('./point_world/small_worldpoint_5_23_5.txt', world[(n_frames - 1) * per:])
        t2 = time.time()
        if comm.rank == 0:
            print('##################')
            print("%d frames cost ." % n_frames, t2 - t1)
    finally:
        try:
            if ply_done is not None:
                _finish_ply(ply_done)
        finally:
            comm.barrier()                         # nobody leaves (and tears RCCL down) while rank 0 still needs its peers
            comm.close()


def _finish_ply(ply_done):
    """Wait for the background PLY from a `finally`: its error is raised only when nothing else is already propagating (the
    first failure is the one the caller should see)."""
    import sys
    if sys.exc_info()[0] is None:
        return ply_done()
    try:
        ply_done()
    except Exception:
        pass


def _ply_in_background(world, pc_file):
    """genply's file, written by a helper thread while the caller goes on (the native writer releases the GIL): the fused
    PLY is one file and its writer one thread -- 1.3 GB in 0.3 s on the MI355X box's host, with the formatter threads half
    idle -- so it runs beside the per-frame txt files instead of after them.  Returns a function that waits for it, re-raises
    what it raised and prints genply's line."""
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1)
    job = pool.submit(r3d.cloud_io.write_ply, pc_file, world)

    def wait():
        try:
            job.result()
        finally:
            pool.shutdown()
        print("Write into .ply file Done.")
    return wait


def get_file_name(qt_path):
    """The frame loop (c2w:138-174) with every frame in flight at once: the rasters go up (1 B/point), ONE fused launch
    makes the world cloud and one more the camera clouds, both fp64 and both STAYING in HBM; the GPU then formats the
    reference's text from them (csrc/r3d_textfmt.hip: repr() lines for ./point/<stem>.txt and the world txt, "%.4f" rows for
    the PLY) and host threads copy that text into the files -- the PLY beside the per-frame txts.  Same bytes as the host
    formatter (R3D_HOST_TEXT=1 takes that way: clouds to host memory, csrc/r3d_format.cpp)."""
    if _common.world_size() > 1:
        return _get_file_name_sharded(qt_path)
    if os.environ.get("R3D_HOST_TEXT", "0") not in ("", "0"):
        return _get_file_name_host_text(qt_path)
    print('data start transfer')
    t1 = time.time()
    _common.stamp(None)
    names, quats, ts = r3d.read_pose_file(qt_path)
    n_frames = len(names)
    if not n_frames:
        return _get_file_name_host_text(qt_path, announce=False)     # a header-only PLY: nothing for the GPU to do
    # the HIP context comes up (70 ms, no GIL held) on a helper thread while the library's host threads decode the rasters
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as pool:
        ctx_job = pool.submit(_common.context)
        try:
            depths = r3d.cloud_io.read_depth_batch([os.path.join('./depth/', n) for n in names])
        finally:
            ctx_error = ctx_job.exception()
        if ctx_error is not None:
            raise ctx_error
        ctx = ctx_job.result()
    _common.stamp("pose file + %d depth files decoded, GPU context up" % n_frames)
    per = depths.shape[1] * depths.shape[2]
    n = n_frames * per
    cam = ctx.camera(depths.shape[1], depths.shape[2], *_common.intrinsics())
    table = r3d.pose_table(quats, ts * _common.pose_scale())       # 1 unless R3D_POSE_SCALE says otherwise
    bufs = []
    try:
        d_depth, d_pose, d_world = ctx.alloc(depths.nbytes).upload(depths), ctx.alloc(table.nbytes).upload(table), ctx.alloc(n * 24)
        bufs += [d_depth, d_pose, d_world]
        r3d.fuse_frames_device(ctx, cam, d_depth.ptr, depths.dtype, n_frames, d_pose.ptr, d_world.ptr, np.float64)
        text = r3d.device_text.TextWriter(ctx)
        if _common.ply_binary():      # f1's optional flag (R3D_PLY_BINARY=1): a standard binary PLY of the f32 cloud, NOT the reference's text
            d_world32 = ctx.alloc(n * 12)
            bufs.append(d_world32)
            r3d.fuse_frames_device(ctx, cam, d_depth.ptr, depths.dtype, n_frames, d_pose.ptr, d_world32.ptr, np.float32)
            text.add_ply_binary('./ply/small_035_p8.ply', d_world32.ptr, n)
        else:
            text.add_ply('./ply/small_035_p8.ply', d_world.ptr, np.float64, n)     # the same bytes genply(world, ...) writes
        if not _common.skip_intermediate():
            d_cam = ctx.alloc(n * 24)
            bufs.append(d_cam)
            r3d.unproject_device(ctx, cam, d_depth.ptr, depths.dtype, n_frames, d_cam.ptr, np.float64)
            integral = depths.dtype in (np.uint8, np.uint16)
            # the per-frame camera txt the reference leaves in ./point/ (third column: the raster's own integers)
            text.add_xyz_txt(['./point/' + nm[0:-4] + '.txt' for nm in names], d_cam.ptr, np.float64, n,
                             d_z_raw=d_depth.ptr if integral else None, z_dtype=depths.dtype if integral else None)
            # the reference reopens this file with 'w' for every frame: it ends up holding the last one
            text.add_xyz_txt(['./point_world/small_worldpoint_5_23_5.txt'], d_world.ptr + (n_frames - 1) * per * 24, np.float64, per)
        _common.stamp("fused launches + text sizes (device)")
        written = text.write()
        _common.stamp("%d bytes of text formatted on the GPU and written" % written)
    finally:
        for b in bufs:
            b.free()
    t2 = time.time()
    print('##################')
    print("%d frames cost ." % n_frames, t2 - t1)
    print("Write into .ply file Done.")


def _get_file_name_host_text(qt_path, announce=True):
    """The same files through the HOST formatter (rounds 1-4's way; R3D_HOST_TEXT=1): both fp64 clouds come to host memory."""
    if announce:
        print('data start transfer')
    t1 = time.time()
    names, depths, world = fuse_pose_file(qt_path)
    n_frames = len(names)
    ply_done = _ply_in_background(world, './ply/small_035_p8.ply')      # the same bytes genply(world, ...) writes
    try:
        if n_frames and not _common.skip_intermediate():
            per = depths.shape[1] * depths.shape[2]
            cam = r3d.unproject(depths, intrinsics=_common.intrinsics(), out_dtype=np.float64, ctx=_common.context())
            _common.stamp("camera clouds in host memory")
            _write_camera_txts(names, cam, depths, per)   # the per-frame camera txt the reference leaves in ./point/
            _common.stamp("%d camera txt files (PLY being written beside them)" % n_frames)
            # the reference reopens this file with 'w' for every frame: it ends up holding the last one
            r3d.cloud_io.write_xyz_txt('./point_world/small_worldpoint_5_23_5.txt', world[(n_frames - 1) * per:])
            _common.stamp("world txt")
        t2 = time.time()
        print('##################')
        print("%d frames cost ." % n_frames, t2 - t1)
    finally:
        _finish_ply(ply_done)
        _common.stamp("rest of the PLY")


def main():
    qt_path = './camera_pose/image_colmap_simi_2.txt'
    get_file_name(qt_path)


if __name__ == '__main__':
    main()
