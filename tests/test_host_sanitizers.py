"""CPU: the host-only translation units (text formatters, PNG and JPEG decoders) under AddressSanitizer + UBSan with hostile
inputs (random bit patterns as doubles, corrupted PNG and JPEG files).  GPU sanitizers are unavailable on the pool."""
import os
import subprocess

import pytest

from helpers import PKG, ROOT


def test_host_code_is_clean_under_asan_ubsan(tmp_path, golden_dir):
    exe = str(tmp_path / "host_fuzz")
    src = os.path.join(ROOT, PKG, "csrc")
    build = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                            "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "c", "host_fuzz.cpp"), os.path.join(src, "r3d_format.cpp"),
                            os.path.join(src, "r3d_png.cpp"), os.path.join(src, "r3d_jpeg.cpp"), "-I", src, "-lz", "-lpthread", "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-3000:]
    from PIL import Image
    import numpy as np
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, (33, 47, 3), dtype=np.uint8), "RGB").save(tmp_path / "c.png")
    Image.fromarray(rng.integers(0, 256, (33, 47, 4), dtype=np.uint8), "RGBA").save(tmp_path / "a.png")
    yy, xx = np.mgrid[0:61, 0:83]
    smooth = (np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 5.0), 128 + 90 * np.cos(xx / 3.0), 100 + yy % 97], 2)
              + rng.normal(0, 10, (61, 83, 3))).clip(0, 255).astype(np.uint8)
    Image.fromarray(smooth, "RGB").save(tmp_path / "s420.jpg", quality=85, subsampling=2)
    Image.fromarray(smooth, "RGB").save(tmp_path / "s444.jpg", quality=95, subsampling=0, optimize=True)
    Image.fromarray(smooth[..., 0], "L").save(tmp_path / "grey.jpg", quality=60)
    Image.fromarray(smooth, "RGB").save(tmp_path / "s422.jpg", quality=70, subsampling=1)
    Image.fromarray(smooth[:3, :4], "RGB").save(tmp_path / "tiny.jpg", quality=70, subsampling=2)
    run = subprocess.run([exe, os.path.join(golden_dir, "scene3", "depth", "000.png"), str(tmp_path / "c.png"),
                          str(tmp_path / "a.png"), str(tmp_path / "s420.jpg"), str(tmp_path / "s444.jpg"), str(tmp_path / "grey.jpg"),
                          str(tmp_path / "s422.jpg"), str(tmp_path / "tiny.jpg")],
                         capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "host fuzz OK" in run.stdout and "corrupted JPEG trials" in run.stdout


def test_host_pools_are_race_free_under_tsan(tmp_path, golden_dir):
    """The same hostile-input driver and the copy crew of the host pipeline under ThreadSanitizer: the pools (decoders, text
    formatters with their writer thread, the per-call crew) share scratch, counters and buffers across threads."""
    src = os.path.join(ROOT, PKG, "csrc")
    common = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=thread", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"), "-I", src]
    crew = str(tmp_path / "crew_tsan")
    build = subprocess.run(common + [os.path.join(ROOT, "tests", "c", "crew_tsan.cpp"), "-lpthread", "-o", crew], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([crew], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "crew ok" in run.stdout and "ThreadSanitizer" not in run.stderr, run.stdout[-1000:] + run.stderr[-3000:]
    exe = str(tmp_path / "host_fuzz_tsan")
    build = subprocess.run(common + [os.path.join(ROOT, "tests", "c", "host_fuzz.cpp"), os.path.join(src, "r3d_format.cpp"),
                                     os.path.join(src, "r3d_png.cpp"), os.path.join(src, "r3d_jpeg.cpp"), "-lz", "-lpthread", "-o", exe],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    from PIL import Image
    import numpy as np
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:61, 0:83]
    smooth = (np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 5.0), 128 + 90 * np.cos(xx / 3.0), 100 + yy % 97], 2)
              + rng.normal(0, 10, (61, 83, 3))).clip(0, 255).astype(np.uint8)
    Image.fromarray(smooth, "RGB").save(tmp_path / "s420.jpg", quality=85, subsampling=2)
    run = subprocess.run([exe, os.path.join(golden_dir, "scene3", "depth", "000.png"), str(tmp_path / "s420.jpg")],
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0 and "host fuzz OK" in run.stdout and "ThreadSanitizer" not in run.stderr, run.stdout[-1000:] + run.stderr[-3000:]
