import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# PyTorch-ROCm bundles its own libamdhip64 and asks for it by the unversioned name: whoever loads first decides whether
# the process ends up with one HIP runtime or two (3d_reconstruction_system_amd/_lib.py: hip_runtimes_loaded).  Tests mix
# torch tensors / streams with the library, so torch goes first.
try:
    import torch  # noqa: F401,E402
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the reference's surface returns np.matrix (c2w:53-55); numpy nags about that class
    config.addinivalue_line("filterwarnings", "ignore:the matrix subclass:PendingDeprecationWarning")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def mock_rccl(tmp_path_factory):
    """tests/c/mock_rccl.cpp: the ten nccl* symbols r3d_comm.hip binds, moving bytes between processes through /dev/shm
    (RCCL refuses two ranks on one device; the test box has one GPU).  Test infrastructure, bound via R3D_RCCL_PATH."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path_factory.mktemp("mock") / "libmockrccl.so")
    build = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-fPIC", "-shared",
                            os.path.join(root, "tests", "c", "mock_rccl.cpp"), "-o", so], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    return so
