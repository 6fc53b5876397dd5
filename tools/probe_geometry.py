import sys, importlib, time
sys.path.insert(0, '.')
import numpy as np
r3d = importlib.import_module("3d_reconstruction_system_amd")
ctx = r3d.Context(0)
rng = np.random.default_rng(0)
def timed(fn, iters):
    t0 = time.perf_counter(); k = 0
    while k < 5 or time.perf_counter() - t0 < 0.08:
        fn(); k += 1
        if k % 5 == 0: ctx.sync()
    ctx.sync()
    g = []
    for _ in range(5):
        ctx.timer_start()
        for _ in range(iters): fn()
        g.append(ctx.timer_stop() / iters)
    return sorted(g)[2]
for (F, H, W) in ((50, 1080, 1920), (100, 384, 1280)):
    n = F * H * W
    cam = ctx.camera(H, W, 960.0, 960.0, W / 2, H / 2)
    tab = r3d.pose_table(rng.normal(size=(F, 4)), rng.normal(size=(F, 3)) * 10)
    d_pose = ctx.alloc(tab.nbytes).upload(tab)
    rgb = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    d_rgb = ctx.alloc(rgb.nbytes).upload(rgb); del rgb
    d_xyz, d_rgba, d_xyz64 = ctx.alloc(n * 12), ctx.alloc(n * 4), ctx.alloc(n * 24)
    for dt, bpp in ((np.uint8, 13), (np.float32, 16)):
        depth = (rng.random((F, H, W)) * 200 + 1).astype(dt)
        d_depth = ctx.alloc(depth.nbytes).upload(depth); del depth
        for blocks in (0, 256 * 16, 10 ** 9):
            ctx.set_tuning("fuse_blocks", blocks)
            ms = timed(lambda: r3d.fuse_frames_device(ctx, cam, d_depth.ptr, dt, F, d_pose.ptr, d_xyz.ptr, np.float32), 20)
            ms3 = timed(lambda: r3d.fuse_frames_device(ctx, cam, d_depth.ptr, dt, F, d_pose.ptr, d_xyz64.ptr, np.float64), 20)
            ms2 = timed(lambda: r3d.fuse_frames_rgb_device(ctx, cam, d_depth.ptr, dt, F, d_pose.ptr, d_rgb.ptr, d_xyz.ptr, np.float32, d_rgba.ptr), 20)
            print("%dx%dx%d %-7s blocks=%-10d plain %.4f ms %.2f TB/s | rgb %.4f ms %.2f TB/s" % (F, H, W, np.dtype(dt).name, blocks, ms, n * bpp / ms / 1e9, ms2, n * (bpp + 7) / ms2 / 1e9), "| f64 %.4f ms %.2f TB/s" % (ms3, n * (bpp + 12) / ms3 / 1e9), flush=True)
        ctx.set_tuning("fuse_blocks", 0)
        d_depth.free()
    for b in (d_pose, d_rgb, d_xyz, d_rgba, d_xyz64): b.free()
