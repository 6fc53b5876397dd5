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


# The driver runs `pytest -m gpu -x`: the first failure ends the run.  Parity tests therefore go FIRST, in the order of the
# hot path (fusion -> drop-ins -> C consumer -> ICP -> voxel map -> torch op), the exchange step after them, and everything
# that starts bench.py or needs more than one GPU LAST -- a wobble in a measurement must never stand between the run and the
# correctness evidence (round 3 lost 450 parity tests to a timing assertion in the file that sorted first).
GPU_ORDER = ["test_gpu_fusion", "test_gpu_textfmt", "test_gpu_dropin", "test_gpu_c_consumer", "test_gpu_icp", "test_gpu_plane_icp", "test_gpu_voxel",
             "test_gpu_torch_ops", "test_gpu_config5_full", "test_gpu_comm", "test_gpu_dist", "test_gpu_eight_ranks", "test_gpu_bench_contract",
             "test_gpu_multi_device"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in GPU_ORDER:
            return 1 + GPU_ORDER.index(name)
        return 0 if not name.startswith("test_gpu_") else len(GPU_ORDER) - 1.5      # unknown GPU files: before the bench files
    items.sort(key=rank)                                                            # stable: file order is kept within a file


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


@pytest.fixture(scope="session")
def real_rccl():
    """Tests that create a communicator on the REAL RCCL (one rank) ask for this first.  RCCL's bootstrap probes the box's
    network interfaces and host name; on a box where that stalls (seen once: minutes per ncclCommInitRank on an otherwise
    healthy GPU box) an in-process call could hang the whole run.  So the bring-up is tried ONCE in a child process with a
    deadline; if it does not come up the dependent tests are skipped with that reason -- the N>1 logic is still covered
    through the stand-in transport, which needs no network at all."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib, sys; sys.path.insert(0, %r); r3d = importlib.import_module('3d_reconstruction_system_amd'); "
            "CM = importlib.import_module('3d_reconstruction_system_amd.comm'); ctx = r3d.Context(0); "
            "c = CM.Comm(ctx, CM.Comm.unique_id(), 0, 1); c.barrier(); c.close(); ctx.close(); print('rccl up')" % root)
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=90)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL did not bring up a one-rank communicator within 90 s on this box (bootstrap stall)")
    if r.returncode != 0 or "rccl up" not in r.stdout:
        pytest.fail("one-rank RCCL communicator failed: " + r.stdout[-500:] + r.stderr[-1500:])
    return True
