#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run only in the build container (needs /root/reference); the GPU box never sees
the reference, only the data files this script wrote.  Usage:

    python tests/golden/make_golden.py

What is imported from the reference (read-only, never copied):
    /root/reference/transfer/pixel_to_camera.py      gentxtcord, genply_RGB
    /root/reference/transfer/camera_to_world.py      gentxtcord, scipy_transfer, get_r,
                                                     point_camera, get_pointdata, genply, main
    /root/reference/other_tools/transfer_T_icp.py    executed with runpy (module-level script)

Environment shim (this file's own code, not reference code): the reference was
written against numpy 1.18 / OpenCV 4.2.  Here OpenCV is absent and numpy is 2.x,
so we register a tiny `cv2` stand-in whose imread() is PIL-backed and restore the
removed alias np.float = float.  Nothing else is altered.

Every fixture is DATA: inputs + the outputs the reference produced for them.
"""
import hashlib
import io
import json
import os
import runpy
import shutil
import sys
import types
import contextlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def install_shim():
    np.float = float  # removed in numpy 1.24; reference calls map(np.float, ...)
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_UNCHANGED = -1
    cv2.IMREAD_GRAYSCALE = 0

    def imread(path, flag=1):
        try:
            im = Image.open(path)
        except Exception:
            return None
        if flag == 0:
            return np.array(im.convert("L"))
        a = np.array(im)
        if a.ndim == 3 and a.shape[2] >= 3:
            a = a[:, :, [2, 1, 0] + list(range(3, a.shape[2]))]  # RGB(A) -> BGR(A)
        return a

    cv2.imread = imread
    sys.modules["cv2"] = cv2


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


@contextlib.contextmanager
def chdir(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


def kat_depth(h, w):
    j, i = np.mgrid[0:h, 0:w]
    return ((7 * j + 3 * i + 1) % 256).astype(np.uint8)


def main():
    install_shim()
    sys.path.insert(0, os.path.join(REF, "transfer"))
    with contextlib.redirect_stdout(io.StringIO()):
        import pixel_to_camera as p2c
        import camera_to_world as c2w

    # ---- G1: c2w.gentxtcord on the 4x6 known-answer raster (SURVEY KAT-1) ----
    out = os.path.join(HERE, "kat_unproject_4x6.txt")
    c2w.gentxtcord(out, kat_depth(4, 6))

    # ---- G2: quaternion -> R^-1 (scipy_transfer, scalar-last, normalising) and get_r (wxyz, raw) ----
    rng = np.random.default_rng(42)
    quats = [[0.1, 0.2, 0.3, 0.9], [0.0, 0.0, 0.0, 1.0], [1.0, 0.0, 0.0, 0.0],
             [0.5, -0.5, 0.5, -0.5], [2.0, -1.0, 0.5, 3.0]]
    quats += rng.normal(size=(6, 4)).tolist()
    poses = {"quats_xyzw": quats, "scipy_transfer": [], "get_r_wxyz_input": [], "get_r": [],
             "point_camera": []}
    for q in quats:
        rinv = c2w.scipy_transfer(np.array(q))
        poses["scipy_transfer"].append(np.asarray(rinv).tolist())
        qn = np.array(q) / np.linalg.norm(q)
        wxyz = [qn[3], qn[0], qn[1], qn[2]]
        poses["get_r_wxyz_input"].append(wxyz)
        poses["get_r"].append(np.asarray(c2w.get_r(wxyz)).tolist())
    # point_camera on a few points with the first pose (KAT-2)
    rinv = c2w.scipy_transfer(np.array(quats[0]))
    t = np.array([1.0, 2.0, 3.0])
    for p in [[1.0, 0.0, 2.0], [0.0, 0.0, 0.0], [-3.5, 2.25, 130.0]]:
        pw = c2w.point_camera(np.array(p), rinv, t)
        poses["point_camera"].append({"p": p, "t": t.tolist(), "q": quats[0],
                                      "p_world": [pw[0, 0], pw[1, 0], pw[2, 0]]})
    with open(os.path.join(HERE, "poses.json"), "w") as f:
        json.dump(poses, f, indent=1)

    # ---- G3: the whole camera_to_world.main() on a 3-frame 24x32 scene ----
    scene = os.path.join(HERE, "scene3")
    shutil.rmtree(scene, ignore_errors=True)
    for d in ("depth", "camera_pose", "point", "point_world", "ply"):
        os.makedirs(os.path.join(scene, d))
    rng = np.random.default_rng(3)
    names = ["000.png", "007.png", "frame_b.png"]
    lines = ["id,tx,ty,tz,qx,qy,qz,qw,name,tail\n"]
    for k, name in enumerate(names):
        depth = rng.integers(0, 256, size=(24, 32), dtype=np.uint8)  # includes Z=0 pixels
        Image.fromarray(depth, mode="L").save(os.path.join(scene, "depth", name))
        q = rng.normal(size=4)
        if k == 1:
            q = q * 3.7  # un-normalised on purpose: scipy normalises
        t = rng.normal(size=3) * 10
        lines.append("%d,%r,%r,%r,%r,%r,%r,%r,%s,x\n" % ((k + 1,) + tuple(float(v) for v in t)
                                                     + tuple(float(v) for v in q) + (name,)))
    with open(os.path.join(scene, "camera_pose", "image_colmap_simi_2.txt"), "w") as f:
        f.writelines(lines)
    with chdir(scene), contextlib.redirect_stdout(io.StringIO()):
        c2w.main()

    # ---- G4: transfer_T_icp.py (apply T to cloud B, pass cloud A through, merge) ----
    icpd = os.path.join(HERE, "icp_apply")
    shutil.rmtree(icpd, ignore_errors=True)
    for d in ("point", "point_world", os.path.join("ply", "icp")):
        os.makedirs(os.path.join(icpd, d))
    rng = np.random.default_rng(11)
    c2w.gentxtcord(os.path.join(icpd, "point", "0.txt"), rng.integers(1, 256, (5, 10), dtype=np.uint8))
    c2w.gentxtcord(os.path.join(icpd, "point", "24.txt"), rng.integers(1, 256, (5, 10), dtype=np.uint8))
    s, th = 1.7, np.deg2rad(90.0)
    T = np.eye(4)
    T[:3, :3] = s * np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    T[:3, 3] = [1.0, 2.0, 3.0]
    with open(os.path.join(icpd, "T_data.txt"), "w") as f:
        for row in T:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
    with chdir(icpd), contextlib.redirect_stdout(io.StringIO()):
        ns = runpy.run_path(os.path.join(REF, "other_tools", "transfer_T_icp.py"))
        ns["file_w"].close()  # the reference never closes it; flush so the bytes land
        Tread = ns["get_T"]("T_data.txt")
    with open(os.path.join(icpd, "T_parsed.json"), "w") as f:
        json.dump(np.asarray(Tread).tolist(), f)

    # ---- G5: p2c.gentxtcord on its hard-coded 480x640 raster: digest + samples only ----
    rng = np.random.default_rng(0)
    d480 = rng.integers(1, 256, (480, 640), dtype=np.uint8)
    tmp = os.path.join("/tmp", "r3d_golden_p2c.txt")
    xs, ys, zs = p2c.gentxtcord(tmp, d480)
    with open(tmp) as f:
        all_lines = f.readlines()
    picks = [0, 1, 639, 640, 100 * 640 + 37, 240 * 640 + 320, 307199]
    g5 = {"seed": 0, "shape": [480, 640], "gen": "default_rng(0).integers(1,256,(480,640),uint8)",
          "sha256_txt": sha256_file(tmp), "n_lines": len(all_lines),
          "sample_lines": {str(k): all_lines[k] for k in picks},
          "ret_samples": {str(k): [float(xs[k]), float(ys[k]), int(zs[k])] for k in picks},
          "ret_z_type": type(zs[0]).__name__}
    # p2c.genply_RGB (the plain writer, despite its name) on the first 7 points
    plytmp = os.path.join(HERE, "p2c_first7.ply")
    with contextlib.redirect_stdout(io.StringIO()):
        p2c.genply_RGB([xs[:7], ys[:7], zs[:7]], plytmp)
    # p2c.genply_noRGB (the COLOURED writer) references an `Image` it never imports (p2c:58); with PIL's Image
    # placed in its namespace (environment shim, like cv2) it runs: 6 points + a 2x3 RGB image
    p2c.Image = Image
    rgb = np.random.default_rng(1).integers(0, 256, (2, 3, 3), dtype=np.uint8)
    Image.fromarray(rgb, "RGB").save(os.path.join(HERE, "p2c_rgb_2x3.png"))
    with contextlib.redirect_stdout(io.StringIO()):
        p2c.genply_noRGB([xs[:6], ys[:6], zs[:6]], os.path.join(HERE, "p2c_rgb_2x3.png"), os.path.join(HERE, "p2c_first6_rgb.ply"))
    try:
        p2c.gentxtcord(tmp, d480[:192])
        g5["short_raster_error"] = None
    except Exception as e:  # reference behaviour on a 192-row raster
        g5["short_raster_error"] = type(e).__name__
    with open(os.path.join(HERE, "p2c_480x640.json"), "w") as f:
        json.dump(g5, f, indent=1)
    os.remove(tmp)

    # ---- G6: config C1 (640x192) through c2w.gentxtcord + get_pointdata: digests + samples ----
    c1 = os.path.join("/tmp", "r3d_golden_c1")
    shutil.rmtree(c1, ignore_errors=True)
    os.makedirs(os.path.join(c1, "point_world"))
    rng = np.random.default_rng(0)
    d192 = rng.integers(1, 256, (192, 640), dtype=np.uint8)
    q = np.array([0.1, 0.2, 0.3, 0.9])
    t = np.array([1.0, 2.0, 3.0])
    with chdir(c1):
        c2w.gentxtcord("cam.txt", d192)
        xs, ys, zs = [], [], []
        c2w.get_pointdata("cam.txt", q, t, xs, ys, zs)
        with open("cam.txt") as f:
            cam_lines = f.readlines()
        with open("point_world/small_worldpoint_5_23_5.txt") as f:
            world_lines = f.readlines()
        picks = [0, 1, 639, 640, 100 * 640 + 37, 122879]
        g6 = {"seed": 0, "shape": [192, 640], "q_xyzw": q.tolist(), "t": t.tolist(),
              "gen": "default_rng(0).integers(1,256,(192,640),uint8)",
              "sha256_cam_txt": sha256_file("cam.txt"),
              "sha256_world_txt": sha256_file("point_world/small_worldpoint_5_23_5.txt"),
              "n": len(xs),
              "cam_lines": {str(k): cam_lines[k] for k in picks},
              "world_lines": {str(k): world_lines[k] for k in picks},
              "world_xyz": {str(k): [float(xs[k]), float(ys[k]), float(zs[k])] for k in picks},
              "world_sum_xyz": [float(np.sum(xs)), float(np.sum(ys)), float(np.sum(zs))]}
    with open(os.path.join(HERE, "c1_192x640.json"), "w") as f:
        json.dump(g6, f, indent=1)
    shutil.rmtree(c1)

    # ---- G7: camera_to_world.main() on 2 frames of 192x640 (config C1 size): digests of every file it writes ----
    big = os.path.join("/tmp", "r3d_golden_c1scene")
    shutil.rmtree(big, ignore_errors=True)
    for d in ("depth", "camera_pose", "point", "point_world", "ply"):
        os.makedirs(os.path.join(big, d))
    rng = np.random.default_rng(77)
    lines = ["id,tx,ty,tz,qx,qy,qz,qw,name,tail\n"]
    g7 = {"seed": 77, "shape": [192, 640], "frames": []}
    for k in range(2):
        depth = rng.integers(0, 256, size=(192, 640), dtype=np.uint8)
        name = "c1_%d.png" % k
        Image.fromarray(depth, mode="L").save(os.path.join(big, "depth", name))
        q = rng.normal(size=4)
        t = rng.normal(size=3) * 10
        g7["frames"].append({"name": name, "q": [float(v) for v in q], "t": [float(v) for v in t]})
        lines.append("%d,%r,%r,%r,%r,%r,%r,%r,%s,x\n" % ((k,) + tuple(float(v) for v in t) + tuple(float(v) for v in q) + (name,)))
    with open(os.path.join(big, "camera_pose", "image_colmap_simi_2.txt"), "w") as f:
        f.writelines(lines)
    with chdir(big), contextlib.redirect_stdout(io.StringIO()):
        c2w.main()
    g7["gen"] = "per frame: default_rng(77) stream: integers(0,256,(192,640),uint8), normal(4), normal(3)*10"
    g7["sha256_ply"] = sha256_file(os.path.join(big, "ply", "small_035_p8.ply"))
    g7["sha256_point"] = {fr["name"]: sha256_file(os.path.join(big, "point", fr["name"][:-4] + ".txt")) for fr in g7["frames"]}
    world = np.loadtxt(os.path.join(big, "point_world", "small_worldpoint_5_23_5.txt"), delimiter=",")
    g7["world_last_frame_sum"] = [float(v) for v in world.sum(0)]
    g7["world_last_frame_rows"] = {str(k): [float(v) for v in world[k]] for k in (0, 1, 61439, 122879)}
    with open(os.path.join(HERE, "c1_scene_2x192x640.json"), "w") as f:
        json.dump(g7, f, indent=1)
    shutil.rmtree(big)

    # ---- manifest ----
    manifest = {}
    for root, _, files in os.walk(HERE):
        for fn in sorted(files):
            if fn in ("MANIFEST.json", "make_golden.py") or fn.endswith(".pyc"):
                continue
            p = os.path.join(root, fn)
            manifest[os.path.relpath(p, HERE)] = sha256_file(p)
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "reference": REF,
                   "numpy": np.__version__, "files": manifest}, f, indent=1, sort_keys=True)
    print("wrote", len(manifest), "fixture files under", HERE)


if __name__ == "__main__":
    main()
