"""CPU: the static ISA check that guards the workgroup barriers (tools/isa_barrier_check.py) -- its dataflow on hand-written
assembly, and the compiled voxel kernels themselves (hipcc cross-compiles gfx950 here)."""
import importlib.util
import os
import shutil

import pytest

from helpers import ROOT

spec = importlib.util.spec_from_file_location("isa_barrier_check", os.path.join(ROOT, "tools", "isa_barrier_check.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def body(text):
    return [l for l in text.strip("\n").split("\n")]


def test_straight_line_code():
    ok = body("""
	ds_write_b32 v1, v2
	s_waitcnt lgkmcnt(0)
	s_barrier
	ds_read_b32 v3, v1
	s_waitcnt lgkmcnt(0)
	s_endpgm
""")
    assert chk.check_function(ok) == []
    bad = body("""
	ds_add_u32 v1, v2
	v_mov_b32_e32 v3, 0
	s_barrier
	s_endpgm
""")
    assert [i for i, _ in chk.check_function(bad)] == [2]


def test_cross_lane_moves_are_not_memory_and_other_counters_do_not_count():
    assert chk.check_function(body("""
	ds_bpermute_b32 v1, v2, v3
	s_barrier
	s_endpgm
""")) == []
    assert len(chk.check_function(body("""
	ds_write_b64 v1, v[2:3]
	s_waitcnt vmcnt(0)
	s_waitcnt lgkmcnt(1)
	s_barrier
	s_endpgm
"""))) == 1
    assert chk.check_function(body("""
	ds_write_b64 v1, v[2:3]
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_barrier
	s_endpgm
""")) == []


def test_the_round_3_pattern_a_result_less_atomic_on_the_back_edge():
    """the loop header's barrier is clean from the preheader and dirty along the back edge"""
    loop = body("""
	s_waitcnt lgkmcnt(0)
	s_branch .LBB0_2
.LBB0_1:
	s_add_i32 s4, s4, 1
	s_cmp_ge_u32 s4, s5
	s_cbranch_scc1 .LBB0_4
.LBB0_2:
	s_barrier
	ds_read_b32 v2, v1
	s_waitcnt lgkmcnt(0)
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_1
	ds_add_u32 v1, v3
	s_branch .LBB0_1
.LBB0_4:
	s_waitcnt lgkmcnt(0)
	s_barrier
	s_endpgm
""")
    found = chk.check_function(loop)
    assert len(found) == 1 and found[0][1].startswith("s_cbranch_scc1")
    fixed = [l if "s_add_i32" not in l else "\ts_waitcnt lgkmcnt(0)\n" + l for l in loop]
    assert chk.check_function("\n".join(fixed).split("\n")) == []


@pytest.mark.skipif(shutil.which(chk.HIPCC) is None and not os.path.exists(chk.HIPCC), reason="hipcc not available")
def test_no_compiled_kernel_has_a_barrier_with_lds_traffic_in_flight(capsys):
    """Every translation unit of the library (the bug was first seen in r3d_fuse.hip's fuse_voxel_kernel; r3d_sort.hip and the
    selection / compaction kernels of r3d_plane.hip have barrier loops with LDS atomics too) -- what the tool's own __main__ does."""
    import glob
    files = sorted(glob.glob(os.path.join(chk.CSRC, "r3d_*.hip")))
    assert len(files) >= 10
    assert chk.main(files) == 0
    assert "barriers checked" in capsys.readouterr().out
