"""Depth raster -> camera frame -> world frame on the MI355X, through the C ABI.

Host-array API (NumPy in, NumPy out; H2D + kernel + D2H inside the library) and raw
device-pointer API (asynchronous on the context's stream; what bench.py and dist.py use with
HBM-resident batches).  No CPU implementation lives here: without libr3d_hip.so and a gfx950
device these functions raise.

Reference arithmetic: transfer/pixel_to_camera.py:24-44, transfer/camera_to_world.py:53-105,
other_tools/transfer_T_icp.py:10-12,71-97.
"""
import numpy as np

from . import _lib as L
from .device import default_context, depth_code, xyz_code
from .poses import pose_table

# the reference's hard-coded intrinsics (p2c:25-28, c2w:68-71)
REF_INTRINSICS = (600.391, 600.079, 320, 240)


def _as_batch(depths):
    depths = np.asarray(depths)
    if depths.ndim == 2:
        depths = depths[None]
    if depths.ndim != 3:
        raise ValueError("depth must be [H,W] or [F,H,W] (got shape %s)" % (depths.shape,))
    depth_code(depths.dtype)
    return np.ascontiguousarray(depths)


def unproject(depth, intrinsics=REF_INTRINSICS, out_dtype=np.float32, depth_scale=1.0, ctx=None):
    """Camera-frame points of one raster [H,W] or a batch [F,H,W]: float [F*H*W,3], row-major,
    every pixel emitted (Z=0 pixels included, like the reference)."""
    d = _as_batch(depth)
    f, h, w = d.shape
    out = np.empty((f * h * w, 3), dtype=out_dtype)
    xyz_code(out.dtype)
    if f * h * w == 0:
        return out
    ctx = ctx or default_context()
    cam = ctx.camera(h, w, *intrinsics)
    L.check(ctx.lib.r3d_unproject_host(ctx.handle, cam.handle, d.ctypes.data, depth_code(d.dtype), f,
                                       float(depth_scale), out.ctypes.data, xyz_code(out.dtype)))
    return out


def _out_array(out, n, out_dtype):
    if out is None:
        return np.empty((n, 3), dtype=out_dtype)
    if not (isinstance(out, np.ndarray) and out.shape == (n, 3) and out.flags.c_contiguous):
        raise ValueError("out must be a C-contiguous [%d,3] array" % n)
    return out


def fuse_frames(depths, quats_xyzw, ts, intrinsics=REF_INTRINSICS, out_dtype=np.float32, depth_scale=1.0,
                ctx=None, out=None):
    """World-frame cloud of F frames, concatenated in frame order: p_w = Rinv_f (p_cam - t_f).
    out: optional preallocated [F*H*W,3] array (e.g. ctx.pinned_empty(...) for full-rate PCIe)."""
    d = _as_batch(depths)
    f, h, w = d.shape
    table = pose_table(quats_xyzw, ts)
    if table.shape[0] != f:
        raise ValueError("%d frames but %d poses" % (f, table.shape[0]))
    out = _out_array(out, f * h * w, out_dtype)
    xyz_code(out.dtype)
    if f * h * w == 0:
        return out
    ctx = ctx or default_context()
    cam = ctx.camera(h, w, *intrinsics)
    L.check(ctx.lib.r3d_fuse_frames_host(ctx.handle, cam.handle, d.ctypes.data, depth_code(d.dtype), f,
                                         float(depth_scale), table.ctypes.data, out.ctypes.data,
                                         xyz_code(out.dtype)))
    return out


def fuse_frames_rgb(depths, rgb, quats_xyzw=None, ts=None, intrinsics=REF_INTRINSICS, out_dtype=np.float32, depth_scale=1.0,
                    ctx=None):
    """RGBD fusion (BASELINE config 5; colour attach of genply_noRGB, p2c:55-91): depths [F,H,W] + rgb [F,H,W,3] uint8
    -> (xyz [F*H*W,3], rgba [F*H*W] uint32 with bytes R,G,B,0).  quats/ts None: camera-frame points."""
    d = _as_batch(depths)
    f, h, w = d.shape
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    if rgb.shape != (f, h, w, 3) and not (f == 1 and rgb.shape == (h, w, 3)):
        raise ValueError("rgb must be [F,H,W,3] uint8 matching the depth batch (got %s)" % (rgb.shape,))
    table = None
    if quats_xyzw is not None:
        table = pose_table(quats_xyzw, ts)
        if table.shape[0] != f:
            raise ValueError("%d frames but %d poses" % (f, table.shape[0]))
    n = f * h * w
    out = np.empty((n, 3), dtype=out_dtype)
    rgba = np.empty(n, dtype=np.uint32)
    code = xyz_code(out.dtype)
    if n == 0:
        return out, rgba
    ctx = ctx or default_context()
    cam = ctx.camera(h, w, *intrinsics)
    L.check(ctx.lib.r3d_fuse_frames_rgb_host(ctx.handle, cam.handle, d.ctypes.data, depth_code(d.dtype), f, float(depth_scale),
                                             table.ctypes.data if table is not None else None, rgb.ctypes.data,
                                             out.ctypes.data, code, rgba.ctypes.data))
    return out, rgba


def se3_apply(xyz, rinv, t, out_dtype=None, ctx=None):
    """Rinv . (p - t) for every point of an [N,3] cloud -- point_camera() (c2w:57-59) in bulk."""
    xyz = np.ascontiguousarray(xyz)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise ValueError("cloud must be [N,3]")
    out = np.empty(xyz.shape, dtype=out_dtype or xyz.dtype)
    xyz_code(xyz.dtype), xyz_code(out.dtype)
    if xyz.shape[0] == 0:
        return out
    ctx = ctx or default_context()
    pose = np.concatenate([np.asarray(rinv, dtype=np.float64).reshape(9), np.asarray(t, dtype=np.float64).reshape(3)])
    L.check(ctx.lib.r3d_se3_apply_host(ctx.handle, xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0],
                                       pose.ctypes.data, out.ctypes.data, xyz_code(out.dtype)))
    return out


def apply_T(xyz, T, out_dtype=None, ctx=None):
    """(T . [x,y,z,1])[0:3] for every point of an [N,3] float32/float64 cloud."""
    xyz = np.ascontiguousarray(xyz)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise ValueError("cloud must be [N,3]")
    out = np.empty(xyz.shape, dtype=out_dtype or xyz.dtype)
    xyz_code(xyz.dtype), xyz_code(out.dtype)
    if xyz.shape[0] == 0:
        return out
    ctx = ctx or default_context()
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4))
    L.check(ctx.lib.r3d_apply_T_host(ctx.handle, xyz.ctypes.data, xyz_code(xyz.dtype), xyz.shape[0],
                                     T.ctypes.data, out.ctypes.data, xyz_code(out.dtype)))
    return out


# ---- raw device-pointer entry points (async on ctx's stream) -------------------------------
def unproject_device(ctx, cam, d_depth, depth_dtype, n_frames, d_out, out_dtype, depth_scale=1.0):
    L.check(ctx.lib.r3d_unproject(ctx.handle, cam.handle, d_depth, depth_code(depth_dtype), int(n_frames),
                                  float(depth_scale), d_out, xyz_code(out_dtype)))


def fuse_frames_device(ctx, cam, d_depth, depth_dtype, n_frames, d_pose, d_out, out_dtype, depth_scale=1.0):
    L.check(ctx.lib.r3d_fuse_frames(ctx.handle, cam.handle, d_depth, depth_code(depth_dtype), int(n_frames),
                                    float(depth_scale), d_pose, d_out, xyz_code(out_dtype)))


def fuse_frames_rgb_device(ctx, cam, d_depth, depth_dtype, n_frames, d_pose, d_rgb, d_out, out_dtype, d_rgba,
                           depth_scale=1.0):
    L.check(ctx.lib.r3d_fuse_frames_rgb(ctx.handle, cam.handle, d_depth, depth_code(depth_dtype), int(n_frames),
                                        float(depth_scale), d_pose, d_rgb, d_out, xyz_code(out_dtype), d_rgba))


def fuse_frames_voxel_device(ctx, cam, d_depth, depth_dtype, n_frames, d_pose, d_rgb, d_out, d_rgba, voxel_set, depth_scale=1.0):
    """The f32 cloud (d_rgb / d_rgba None = no colour; d_pose None = camera frame) AND its occupied voxels into `voxel_set`
    (voxelmap.VoxelSet) in one launch: r3d_fuse_frames_rgb + r3d_voxelset_insert without reading the cloud back."""
    L.check(ctx.lib.r3d_fuse_frames_voxel(ctx.handle, cam.handle, d_depth, depth_code(depth_dtype), int(n_frames),
                                          float(depth_scale), d_pose, d_rgb, d_out, d_rgba, voxel_set.handle))


def apply_T_device(ctx, d_in, in_dtype, n_points, T, d_out, out_dtype):
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4))
    L.check(ctx.lib.r3d_apply_T(ctx.handle, d_in, xyz_code(in_dtype), int(n_points), T.ctypes.data, d_out,
                                xyz_code(out_dtype)))
