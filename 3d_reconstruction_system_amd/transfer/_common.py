"""Shared plumbing of the drop-in scripts: locate the package whether the script is imported as
part of it or run as `python camera_to_world.py` from a data directory, and read the optional
environment overrides.

Environment (all optional; the defaults are the reference's hard-coded values):
  R3D_FX R3D_FY R3D_CX R3D_CY   pinhole intrinsics        (p2c:25-28, c2w:68-71)
  R3D_DEVICE                    GPU index                 (default 0)
  R3D_SKIP_INTERMEDIATE=1       do not write ./point/<stem>.txt and ./point_world/*.txt
  R3D_POSE_SCALE                a number, or the path of a file holding one (e.g. the ./scale.txt that
                                `transfer_T_icp.py --estimate-rigid --colmap ...` writes): every pose translation is multiplied
                                by it -- COLMAP's unit brought to the depth maps' unit (readme.md:25).  Default 1 = the reference.
  WORLD_SIZE RANK LOCAL_RANK    set by a one-process-per-GPU launcher (torch.distributed.run ...): frames are sharded
  R3D_HOST_TEXT=1               camera_to_world.py: format the txt / PLY text on the host (csrc/r3d_format.cpp) instead of on the GPU
                                (csrc/r3d_textfmt.hip); same bytes, for A/B timing
  R3D_PLY_BINARY=1              camera_to_world.py: write ./ply/small_035_p8.ply as a standard binary PLY (float32 x, y, z) instead of the
                                reference's ASCII layout -- f1's optional flag; 12 B/vertex instead of ~26
  R3D_TIMING=1                  stage times on stderr (stamp() below)
"""
import importlib
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ROOT = os.path.dirname(_PKG_DIR)


_t_last = None


def stamp(label):
    """R3D_TIMING=1: milliseconds since the previous stamp, on stderr."""
    global _t_last
    if os.environ.get("R3D_TIMING", "0") in ("", "0"):
        return
    import time
    now = time.perf_counter()
    if _t_last is not None:
        sys.stderr.write("[r3d timing] %8.1f ms  %s\n" % ((now - _t_last) * 1e3, label))
    _t_last = now


def package():
    name = os.path.basename(_PKG_DIR)
    if name in sys.modules:
        return sys.modules[name]
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    return importlib.import_module(name)


def intrinsics():
    r3d = package()
    fx, fy, cx, cy = r3d.REF_INTRINSICS

    def num(key, default):
        v = os.environ.get(key)
        if v is None:
            return default
        f = float(v)
        return int(f) if f.is_integer() and isinstance(default, int) else f

    return (num("R3D_FX", fx), num("R3D_FY", fy), num("R3D_CX", cx), num("R3D_CY", cy))


def context():
    return package().default_context(int(os.environ.get("R3D_DEVICE", "0")))


def pose_scale():
    v = os.environ.get("R3D_POSE_SCALE")
    if not v:
        return 1.0
    try:
        return float(v)
    except ValueError:
        with open(v) as f:
            return float(f.read().split()[0])


def ply_binary():
    return os.environ.get("R3D_PLY_BINARY", "0") not in ("", "0")


def skip_intermediate():
    return os.environ.get("R3D_SKIP_INTERMEDIATE", "0") not in ("", "0")


def module(name):
    return importlib.import_module(package().__name__ + "." + name)


def world_size():
    return int(os.environ.get("WORLD_SIZE", "1"))


_sharded = None


def sharded_context():
    """(Context on this rank's GPU, Comm over all ranks) for a launcher-started job; created once."""
    global _sharded
    if _sharded is None:
        CM = module("comm")
        ctx = package().default_context(CM.env_local_device())
        _sharded = (ctx, CM.Comm.from_env(ctx))
    return _sharded
