"""GPU: the sharded path end to end with the real HIP compute hook -- two ranks, one shared GPU, gloo."""
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_frames", [6, 5])
def test_two_ranks_hip_fuse_and_gather(tmp_path, n_frames):
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29700 + (os.getpid() + n_frames) % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), out,
           str(n_frames)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for rank in range(2):
        line = open("%s.rank%d" % (out, rank)).read()
        assert "ok=1" in line, line
