"""GPU: the rehearsal of the multi-rank exchange at BASELINE config 4's sizes on the ONE-GPU test box, over the stand-in
transport (tests/c/mock_rccl.cpp bound through R3D_RCCL_PATH; RCCL itself refuses two ranks on one device).

Why not eight ranks: the GPU boxes this suite runs on allow at most SIX processes on the card at once (the run is killed
beyond that: a five-rank version of this file was, with the test runner and the launcher counted in) -- so the rehearsal uses
up to FOUR ranks, and what needs eight (the shard arithmetic of a 1001-frame job over 8 ranks, the byte offsets of config 4
and 5 over 8 ranks) runs on the CPU in tests/test_dist_gloo.py.  Eight real ranks on eight GPUs are the driver's scaling run.

What is covered here that the small multi-rank tests (test_gpu_comm.py, test_gpu_dropin.py) do not reach:
  * byte offsets beyond 2^32 inside ONE exchange (config 4's cloud is 5.9 GB): `recv + off[from]`, both assemblies;
  * a ragged 1001-frame job over four ranks, all three all-gather algorithms where they apply, the map union, the all-reduce;
  * bench.py --gpus 4 exactly as the driver launches it, every assembly strategy on the line;
  * the sharded drop-in at four ranks on a scene that leaves the last block empty.
Ordered after every single-process parity test (tests/conftest.py)."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from helpers import PKG, ROOT
from oracle import fusion_ref as O

pytestmark = pytest.mark.gpu


_launches = [0]


def launch(world, args, mock, timeout=900, cwd=ROOT, env=None):
    _launches[0] += 1                       # a port of its own for every launch of this file (no reuse inside TIME_WAIT)
    port = 29700 + (os.getpid() * 7 + _launches[0] * 17) % 250
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=str(cwd),
                          env=dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", R3D_RCCL_PATH=mock, R3D_SHARE_GPU="1",
                                   **(env or {})))


def shm_free_bytes():
    try:
        st = os.statvfs("/dev/shm")
        return st.f_bavail * st.f_frsize
    except OSError:
        return 0


def test_config4_bytes_offsets_beyond_2_32_in_one_exchange(tmp_path, mock_rccl):
    """760 + 3 frames of 1280x384 over two ranks: 4.50 GB of xyz, rank 1's slot starts at byte 4 482 662 400 > 2^32."""
    if shm_free_bytes() < 6 << 30:
        pytest.skip("the stand-in transport needs ~4.5 GB of /dev/shm for this message (%d MB free)" % (shm_free_bytes() >> 20))
    out = str(tmp_path / "big")
    r = launch(2, [os.path.join(ROOT, "tests", "_dist_big_worker.py"), out, "760", "3"], mock_rccl)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for rank in range(2):
        line = open("%s.rank%d" % (out, rank)).read()
        assert "ok=1" in line and "R3D_RCCL_PATH" in line and "total_bytes=%d" % (763 * 384 * 1280 * 12) in line, line
    assert "slot_offset_bytes=%d" % (760 * 384 * 1280 * 12) in open(out + ".rank1").read()
    assert 760 * 384 * 1280 * 12 > 1 << 32


def test_four_ranks_ragged_1001_frames(tmp_path, mock_rccl):
    """1001 frames over four ranks (3 x 251 + 248): both assemblies, auto / direct all-gather, in-place slots, the voxel-set union
    and the all-reduce -- every rank's result against the single-GPU cloud (tests/_dist_mock_worker.py)."""
    out = str(tmp_path / "res")
    r = launch(4, [os.path.join(ROOT, "tests", "_dist_mock_worker.py"), out, "1001"], mock_rccl)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for rank in range(4):
        line = open("%s.rank%d" % (out, rank)).read()
        assert "ok=1" in line and "R3D_RCCL_PATH" in line, line
    assert "lo=0 hi=251" in open(out + ".rank0").read() and "lo=753 hi=1001" in open(out + ".rank3").read()


def test_config4_whole_1000_frames_over_four_ranks_bit_for_bit(tmp_path, mock_rccl):
    """BASELINE config 4 at its full size: 1000 frames of 1280x384 = 491.52 M points, 5.9 GB of xyz, through dist.ShardedFusion
    over four ranks -- both assemblies, every rank comparing EVERY bit of the assembled cloud with the cloud of one launch over
    all 1000 frames (tests/_dist_c4_worker.py)."""
    if shm_free_bytes() < 24 << 30:
        pytest.skip("the stand-in transport needs ~18 GB of /dev/shm for this exchange (%d MB free)" % (shm_free_bytes() >> 20))
    out = str(tmp_path / "c4")
    r = launch(4, [os.path.join(ROOT, "tests", "_dist_c4_worker.py"), out, "1000"], mock_rccl)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for rank in range(4):
        line = open("%s.rank%d" % (out, rank)).read()
        assert "ok=1" in line and "R3D_RCCL_PATH" in line and "points=491520000" in line, line
        assert "outputs: True" in line and "inputs: True" in line, line
    assert "lo=0 hi=250" in open(out + ".rank0").read() and "lo=750 hi=1000" in open(out + ".rank3").read()


def test_four_ranks_equal_shards_take_the_nccl_allgather_too(tmp_path, mock_rccl):
    out = str(tmp_path / "res")
    r = launch(4, [os.path.join(ROOT, "tests", "_dist_mock_worker.py"), out, "1000"], mock_rccl)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for rank in range(4):
        assert "ok=1" in open("%s.rank%d" % (out, rank)).read()


def test_bench_four_ranks_as_the_driver_launches_it(mock_rccl):
    """bench.py --gpus 4 under torch.distributed.run (torch's group on gloo, the exchange through r3d_comm_* on the stand-in
    transport): all six assembly strategies measured and reported, the pipelined one checked against the plain one by the
    bench itself, ONE line, weak scaling bookkeeping right."""
    r = launch(4, [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--frames", "16"], mock_rccl,
               env={"R3D_DIST_BACKEND": "gloo", "R3D_BENCH_TRANSPORT": "r3d"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and "R3D_RCCL_PATH" in d["transport"] and "watchdog" not in d
    for m in ("none", "outputs", "inputs", "outputs_direct", "inputs_direct", "inputs_overlap"):
        assert "ms_per_step" in d["assemble"][m], (m, d["assemble"][m])
        assert d["assemble"][m]["same_bits_as_single_launch"] is True, (m, d["assemble"][m])      # EVERY strategy, bit for bit
    # first-contact instrumentation: the communicator's own world / rank, the transport, RCCL's log object
    assert d["comm"]["launcher_world"] == 4 and d["comm"]["rccl"]["world"] == 4 and d["comm"]["rccl"]["rank"] == 0
    assert "R3D_RCCL_PATH" in d["comm"]["rccl"]["origin"] and "rccl_log" in d["comm"]
    assert d["config"]["points_per_step"] == 4 * 16 * 384 * 1280 and d["config"]["assemble"] != "none"
    assert d["assemble"]["outputs"]["fabric_bytes_in_per_gpu"] == 3 * 16 * 384 * 1280 * 12
    assert d["assemble"]["inputs"]["fabric_bytes_in_per_gpu"] == 3 * 16 * (384 * 1280 + 96)
    assert abs(d["value"] - 4 * 16 * 384 * 1280 / d["ms_per_step"] / 1e3) / d["value"] < 1e-3


def test_sharded_dropin_over_four_ranks_with_an_empty_block(tmp_path, golden_dir, mock_rccl):
    """`torch.distributed.run --nproc-per-node 4 camera_to_world.py` on a 9-frame scene (blocks of 3, 3, 3 and an EMPTY
    fourth): every file equals the single-process run's, byte for byte."""
    from PIL import Image
    rng = np.random.default_rng(31)
    F, H, W = 9, 40, 56
    for sub in ("a", "b"):
        for d in ("depth", "point", "point_world", "ply", "camera_pose"):
            os.makedirs(tmp_path / sub / d)
    with open(tmp_path / "a" / "camera_pose" / "image_colmap_simi_2.txt", "w") as f:
        f.write("id,tx,ty,tz,qx,qy,qz,qw,name,extra\n")
        for k in range(F):
            Image.fromarray(rng.integers(0, 256, (H, W), dtype=np.uint8)).save(tmp_path / "a" / "depth" / ("%03d.png" % k))
            q, t = rng.normal(size=4), rng.normal(size=3) * 10
            f.write(",".join([str(k)] + [repr(float(x)) for x in t] + [repr(float(x)) for x in q] + ["%03d.png" % k, "x"]) + "\n")
    shutil.copytree(tmp_path / "a" / "depth", tmp_path / "b" / "depth", dirs_exist_ok=True)
    shutil.copy(tmp_path / "a" / "camera_pose" / "image_colmap_simi_2.txt", tmp_path / "b" / "camera_pose")
    script = os.path.join(ROOT, PKG, "transfer", "camera_to_world.py")
    one = subprocess.run([sys.executable, script], cwd=tmp_path / "a", capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    four = launch(4, [script], mock_rccl, cwd=tmp_path / "b")
    assert four.returncode == 0, four.stdout[-2000:] + four.stderr[-3000:]
    assert four.stdout.count("Write into .ply file Done.") == 1
    for rel in ["ply/small_035_p8.ply", "point_world/small_worldpoint_5_23_5.txt"] + ["point/%03d.txt" % k for k in range(F)]:
        assert (tmp_path / "a" / rel).read_bytes() == (tmp_path / "b" / rel).read_bytes(), rel
    # ... and the cloud itself against the oracle (the files above are the product's on both sides)
    got = O.read_ply_vertices(str(tmp_path / "b" / "ply" / "small_035_p8.ply"))
    assert got.shape == (F * H * W, 3)
