"""ctypes binding of libr3d_hip.so (C ABI: include/r3d.h).

The HIP library IS the product: there is no CPU fallback.  If the shared object is
missing, or it cannot open a gfx950 device, every compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libr3d_hip.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_NODEVICE, ERR_UNSUPPORTED = -1, -2, -3, -4, -5
DEPTH_U8, DEPTH_U16, DEPTH_F32 = 0, 1, 2
F32, F64 = 0, 1
CTX_EXTERNAL_STREAM = 1

_ERR_NAMES = {ERR_INVALID: "R3D_ERR_INVALID", ERR_HIP: "R3D_ERR_HIP", ERR_NOMEM: "R3D_ERR_NOMEM",
              ERR_NODEVICE: "R3D_ERR_NODEVICE", ERR_UNSUPPORTED: "R3D_ERR_UNSUPPORTED"}


class R3DError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (_ERR_NAMES.get(code, "R3D_ERR"), code, message))
        self.code = code


class R3DLibraryMissing(ImportError):
    pass


_vp, _i, _i64, _d, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float, C.c_size_t
_pvp, _pi, _pd, _pf, _psz = C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_double), \
    C.POINTER(C.c_float), C.POINTER(C.c_size_t)

# name -> (restype, argtypes); mirrors include/r3d.h declaration by declaration
SIGNATURES = {
    "r3d_version": (_i, []),
    "r3d_last_error": (C.c_char_p, []),
    "r3d_device_count": (_i, [_pi]),
    "r3d_ctx_create": (_i, [_i, _vp, _i, _pvp]),
    "r3d_ctx_destroy": (_i, [_vp]),
    "r3d_ctx_sync": (_i, [_vp]),
    "r3d_ctx_stream": (_i, [_vp, _pvp]),
    "r3d_ctx_set_tuning": (_i, [_vp, C.c_char_p, _i]),
    "r3d_ctx_get_tuning": (_i, [_vp, C.c_char_p, _pi]),
    "r3d_dev_alloc": (_i, [_vp, _sz, _pvp]),
    "r3d_dev_free": (_i, [_vp, _vp]),
    "r3d_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "r3d_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "r3d_memcpy_d2d": (_i, [_vp, _vp, _vp, _sz]),
    "r3d_download": (_i, [_vp, _vp, _vp, _sz]),
    "r3d_memset": (_i, [_vp, _vp, _i, _sz]),
    "r3d_cache_prefetch": (_i, [_vp, _vp, _sz]),
    "r3d_host_alloc": (_i, [_vp, _sz, _pvp]),
    "r3d_host_free": (_i, [_vp, _vp]),
    "r3d_timer_start": (_i, [_vp]),
    "r3d_timer_stop": (_i, [_vp, _pf]),
    "r3d_camera_create": (_i, [_vp, _i, _i, _d, _d, _d, _d, _pvp]),
    "r3d_camera_destroy": (_i, [_vp]),
    "r3d_selftest_magic_div": (_i, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "r3d_unproject": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _i]),
    "r3d_unproject_host": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _i]),
    "r3d_fuse_frames": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _i]),
    "r3d_fuse_frames_host": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _i]),
    "r3d_fuse_frames_rgb": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _i, _vp]),
    "r3d_fuse_frames_rgb_host": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _i, _vp]),
    "r3d_backproject_depth_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "r3d_backproject_depth_grad_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "r3d_parse_xyz_text": (_i, [_vp, _sz, _i, _vp, _i64, _vp, _vp]),
    "r3d_format_text_device": (_i, [_vp, _i, _vp, _i, _i64, _vp, _i, _i64, _vp, _sz, _vp, _vp]),
    "r3d_write_device_text_files": (_i, [_vp, _vp, _i]),
    "r3d_write_ply_binary": (_i, [C.c_char_p, _vp, _i, _i64]),
    "r3d_project3d_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "r3d_project3d_grad_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "r3d_se3_apply": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _i]),
    "r3d_se3_apply_host": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _i]),
    "r3d_apply_T": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _i]),
    "r3d_apply_T_host": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _i]),
    "r3d_apply_T_dev": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _i]),
    "r3d_apply_T_many": (_i, [_vp, _vp, _i, _i64, _vp, _i, _vp, _i]),
    "r3d_icp_nn": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "r3d_icp_nn_host": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "r3d_nn_index_create": (_i, [_vp, _vp, _i64, _pvp]),
    "r3d_nn_index_destroy": (_i, [_vp]),
    "r3d_nn_index_rebuild": (_i, [_vp, _vp, _i64]),
    "r3d_gather_rows_strided": (_i, [_vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "r3d_gather_rows": (_i, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "r3d_permutation_invert": (_i, [_vp, _vp, _i64, _vp]),
    "r3d_remap_u32": (_i, [_vp, _vp, _i64, _vp, _i64]),
    "r3d_nn_index_query": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _vp]),
    "r3d_nn_index_sort_cloud": (_i, [_vp, _vp, _i64, _vp]),
    "r3d_nn_index_sort_cloud_valid": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "r3d_cloud_zero_rows_to_nan": (_i, [_vp, _vp, _i64]),
    "r3d_icp_accumulate": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _f, _vp]),
    "r3d_icp_accumulate_dev": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _f, _f, _vp]),
    "r3d_nn_index_query_sums": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _f, _f, _vp]),
    "r3d_umeyama_from_sums": (_i, [_vp, _i, _vp, _vp]),
    "r3d_icp_state_reset": (_i, [_vp, _vp]),
    "r3d_icp_solve_dev": (_i, [_vp, _vp, _i, _vp]),
    "r3d_icp_iterate": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i, _i, _f, _vp]),
    "r3d_normals_organized": (_i, [_vp, _vp, _i64, _i, _i, _f, _vp, _vp]),
    "r3d_select_quantile_f32": (_i, [_vp, _vp, _i64, _d, _vp, _vp]),
    "r3d_select_quantile_f32_dev": (_i, [_vp, _vp, _i64, _d, _vp]),
    "r3d_trimmed_means_f32": (_i, [_vp, _vp, _i, _i64, _d, _vp]),
    "r3d_icp_plane_residuals": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _f, _vp, _vp]),
    "r3d_icp_plane_accumulate": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _f, _f, _f, _vp]),
    "r3d_plane_step_from_sums": (_i, [_vp, _vp, _vp]),
    "r3d_icp_iterate_plane": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _f, _f, _f, _vp]),
    "r3d_comm_unique_id": (_i, [_vp]),
    "r3d_comm_create": (_i, [_vp, _vp, _i, _i, _pvp]),
    "r3d_comm_destroy": (_i, [_vp]),
    "r3d_comm_info": (_i, [_vp, _pi, _pi, C.POINTER(C.c_char_p)]),
    "r3d_comm_rccl_report": (_i, [_vp, _pi, _pi, _pi, _pi]),
    "r3d_comm_allgather": (_i, [_vp, _vp, _vp, _vp, _i]),
    "r3d_allgather_xyz": (_i, [_vp, _vp, _vp, _i, _vp, _i]),
    "r3d_allgather_inputs": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i]),
    "r3d_comm_allreduce_sum_f64": (_i, [_vp, _vp, _i64]),
    "r3d_format_ply": (_i, [_vp, _i, _i64, _vp, _sz, _psz]),
    "r3d_write_ply": (_i, [C.c_char_p, _vp, _i, _i64]),
    "r3d_write_ply_rgb": (_i, [C.c_char_p, _vp, _i, _vp, _i64]),
    "r3d_write_ply_rgba": (_i, [C.c_char_p, _vp, _i, _vp, _i64]),
    "r3d_write_xyz_txt": (_i, [C.c_char_p, _vp, _i, _i64, _vp, _i, _i]),
    "r3d_write_xyz_txt_batch": (_i, [_vp, _i, _vp, _i, _i64, _vp, _i]),
    "r3d_format_xyz_txt": (_i, [_vp, _i, _i64, _vp, _i, _vp, _sz, _psz]),
    "r3d_png_gray_info": (_i, [C.c_char_p, _pi, _pi, _pi]),
    "r3d_png_gray_decode_batch": (_i, [_vp, _i, _vp, _i, _i, _i]),
    "r3d_png_gray8_info": (_i, [C.c_char_p, _pi, _pi]),
    "r3d_png_gray8_decode_batch": (_i, [_vp, _i, _vp, _i, _i, _i]),
    "r3d_rgb_to_gray_u8": (_i, [_vp, C.c_int64, _i, _i, _vp]),
    "r3d_jpeg_gray_info": (_i, [C.c_char_p, _pi, _pi]),
    "r3d_jpeg_gray_decode_batch": (_i, [_vp, _i, _vp, _i, _i]),
    "r3d_jpeg_rgb_info": (_i, [C.c_char_p, _pi, _pi, _pi]),
    "r3d_jpeg_rgb_decode_batch": (_i, [_vp, _i, _vp, _i, _i]),
    "r3d_png_rgb_info": (_i, [C.c_char_p, _pi, _pi, _pi]),
    "r3d_png_rgb_decode_batch": (_i, [_vp, _i, _vp, _i, _i]),
    "r3d_voxelset_create": (_i, [_vp, _d, _i64, _pvp]),
    "r3d_voxelset_destroy": (_i, [_vp]),
    "r3d_voxelset_clear": (_i, [_vp]),
    "r3d_voxelset_insert": (_i, [_vp, _vp, _i64]),
    "r3d_voxelset_insert_host": (_i, [_vp, _vp, _i64]),
    "r3d_fuse_frames_voxel": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "r3d_voxelset_insert_codes": (_i, [_vp, _vp, _i64]),
    "r3d_voxelset_union": (_i, [_vp, _vp]),
    "r3d_voxelset_stats": (_i, [_vp, _vp, _vp, _vp]),
    "r3d_voxelset_codes": (_i, [_vp, _vp, _i64, _vp]),
    "r3d_sort_u64": (_i, [_vp, _vp, _i64, _i]),
    "r3d_octree_format_bt": (_i, [_vp, _i64, _d, _vp, _sz, _psz, _vp]),
    "r3d_octree_write_bt": (_i, [C.c_char_p, _vp, _i64, _d, _vp]),
}

_lib = None


def load():
    """Load libr3d_hip.so once; raises R3DLibraryMissing with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise R3DLibraryMissing(
            "%s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C 3d_reconstruction_system_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def hip_runtimes_loaded():
    """Paths of every libamdhip64 mapped into this process.  PyTorch-ROCm wheels bundle their own copy and ask for it
    by the unversioned name, so a process that loads libr3d_hip.so BEFORE importing torch ends up with two HIP runtimes:
    device addresses still work across them, but a hipStream_t made by one is garbage to the other."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


def require_single_hip_runtime(what):
    libs = hip_runtimes_loaded()
    if len(libs) > 1:
        raise R3DError(ERR_INVALID, "%s needs ONE HIP runtime in the process, found %s: import torch before the first use "
                                    "of this package (torch's bundled runtime is then shared), or keep to the default "
                                    "stream" % (what, ", ".join(libs)))


def last_error():
    msg = load().r3d_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    if rc != OK:
        raise R3DError(rc, last_error())
    return rc
