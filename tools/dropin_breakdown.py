"""Where the drop-in's wall time goes: the stages of transfer/camera_to_world.py on bench.py's 100-frame scene, timed one by
one in ONE process (interpreter start and imports included as their own lines).  usage: python tools/dropin_breakdown.py [frames]"""
import os
import sys
import time

T0 = time.perf_counter()
import numpy as np  # noqa: E402

T_NUMPY = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import importlib
    import shutil
    import tempfile
    from PIL import Image
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    H, W = 384, 1280
    td = tempfile.mkdtemp(prefix="r3d_bd_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    cwd = os.getcwd()
    try:
        for d in ("depth", "camera_pose", "point", "point_world", "ply"):
            os.makedirs(os.path.join(td, d))
        rng = np.random.default_rng(1234)
        base = 40 + 30 * np.sin(np.add.outer(np.arange(H), np.arange(W + 7 * frames)) / 37.0)
        lines = ["id,tx,ty,tz,qx,qy,qz,qw,name,tail\n"]
        for k in range(frames):
            depth = np.clip(base[:, 7 * k:7 * k + W] + rng.integers(0, 6, (H, W)), 1, 255).astype(np.uint8)
            Image.fromarray(depth, "L").save(os.path.join(td, "depth", "%04d.png" % k), compress_level=1)
            q, t = rng.normal(size=4), rng.normal(size=3) * 10
            lines.append("%d,%r,%r,%r,%r,%r,%r,%r,%04d.png,x\n" % ((k,) + tuple(map(float, t)) + tuple(map(float, q)) + (k,)))
        with open(os.path.join(td, "camera_pose", "image_colmap_simi_2.txt"), "w") as f:
            f.writelines(lines)
        os.chdir(td)
        marks = [("python start -> numpy imported (this process)", T_NUMPY - T0)]
        t = time.perf_counter()

        def mark(what):
            nonlocal t
            now = time.perf_counter()
            marks.append((what, now - t))
            t = now
        c2w = importlib.import_module("3d_reconstruction_system_amd.transfer.camera_to_world")
        r3d = c2w.r3d
        mark("import package + drop-in module (loads the .so)")
        ctx = c2w._common.context()
        mark("context (HIP init, stream)")
        names, quats, ts = r3d.read_pose_file('./camera_pose/image_colmap_simi_2.txt')
        mark("pose file")
        depths = r3d.cloud_io.read_depth_batch([os.path.join('./depth/', n) for n in names])
        mark("decode %d PNGs" % frames)
        world = r3d.fuse_frames(depths, quats, ts, intrinsics=c2w._common.intrinsics(), out_dtype=np.float64, ctx=ctx)
        mark("fuse_frames -> f64 world cloud in host memory (%.2f GB)" % (world.nbytes / 1e9))
        cam = r3d.unproject(depths, intrinsics=c2w._common.intrinsics(), out_dtype=np.float64, ctx=ctx)
        mark("unproject -> f64 camera clouds in host memory")
        per = H * W
        c2w._write_camera_txts(names, cam, depths, per)
        mark("%d camera txt files" % frames)
        r3d.cloud_io.write_xyz_txt('./point_world/small_worldpoint_5_23_5.txt', world[(frames - 1) * per:])
        mark("world txt (last frame)")
        c2w.genply(world, './ply/small_035_p8.ply', world.shape[0])
        mark("fused ASCII PLY")
        total = sum(v for _, v in marks)
        for what, v in marks:
            print("%8.1f ms  %s" % (v * 1e3, what))
        print("%8.1f ms  total" % (total * 1e3))
        written = sum(os.path.getsize(os.path.join(d, n)) for d in ("point", "point_world", "ply") for n in os.listdir(d))
        print("%.3f GB written" % (written / 1e9))
    finally:
        os.chdir(cwd)
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
