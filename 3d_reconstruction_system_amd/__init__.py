"""MI355X-native depth -> world point-cloud fusion (drop-in for the hot path of
rainfall1998/3D_reconstruction_system: transfer/pixel_to_camera.py, transfer/camera_to_world.py,
other_tools/transfer_T_icp.py).

The directory name is not a Python identifier; import it with
    r3d = importlib.import_module("3d_reconstruction_system_amd")

Every batch of points is computed in libr3d_hip.so (hand-written gfx950 HIP kernels behind the C ABI of
include/r3d.h).  There is no CPU path for batches: without the library or without an MI355X the compute
entry points raise.  What stays on the host is what SURVEY 8(b) puts there -- quaternion -> Rinv, 4x4 parsing,
text formatting / parsing and PNG decoding (native C++ in the same library) -- plus the reference's one-point
helper `point_camera(p1, r_inverse, t)` for up to 64 points (camera_to_world.py:57-59 is a single np.dot).
"""
from ._lib import (R3DError, R3DLibraryMissing, LIB_PATH, load as load_library,  # noqa: F401
                   DEPTH_U8, DEPTH_U16, DEPTH_F32, F32, F64)
from .device import Context, Camera, DeviceBuffer, default_context  # noqa: F401
from .fusion import (REF_INTRINSICS, unproject, fuse_frames, fuse_frames_rgb, se3_apply, apply_T,  # noqa: F401
                     unproject_device, fuse_frames_device, fuse_frames_rgb_device, fuse_frames_voxel_device,
                     apply_T_device)
from .poses import (scipy_transfer, get_r, pose_table, pose_to_T, read_pose_file, get_T, write_T,  # noqa: F401
                    str_tofloat)
from . import cloud_io, device_text  # noqa: F401

__version__ = "0.2.0"
