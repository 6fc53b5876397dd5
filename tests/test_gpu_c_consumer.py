"""GPU: a plain C program (gcc, C99) links libr3d_hip.so through include/r3d.h and runs the fused path."""
import os
import subprocess

import pytest

from helpers import PKG, ROOT

pytestmark = pytest.mark.gpu


def test_c_consumer(tmp_path):
    exe = str(tmp_path / "cabi_smoke")
    libdir = os.path.join(ROOT, PKG)
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "c", "cabi_smoke.c"), "-o", exe, "-L", libdir, "-lr3d_hip", "-lm",
                            "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "C ABI smoke OK" in run.stdout
