#!/usr/bin/env python3
"""Static check of the gfx950 ISA the compiler made from csrc/*.hip: does any path reach an s_barrier with an LDS memory
operation still in flight (no `s_waitcnt lgkmcnt(0)` behind it)?

Why: round 3 found voxel_insert_kernel / fuse_voxel_kernel's per-tile `ds_add_u32` (an atomicAdd whose result is unused) on
the loop's back edge followed by the loop header's s_barrier with NO wait in between -- hipcc (ROCm 7.2) dropped the wait of
__syncthreads()' release fence there.  Another wave's ds_read behind the barrier could then overtake the add in the LDS
queue, the waves disagreed about a workgroup-uniform counter, took a different number of barriers and read LDS data of the
wrong tile (fused kernel: 0.02 % wrong colour words, a few lost voxel codes per million).  The kernels now wait explicitly;
this script finds any other instance.

  python tools/isa_barrier_check.py [file.hip ...]      # default: every csrc/r3d_*.hip; exit code 1 if anything is found
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d_reconstruction_system_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-I" + os.path.join(ROOT, "include"),
         "--cuda-device-only", "-S"]
# DS instructions that do not touch LDS memory (cross-lane moves): they count in lgkmcnt but order nothing in memory
NOT_MEMORY = ("ds_bpermute", "ds_permute", "ds_swizzle", "ds_nop")


def functions(text):
    """{name: [lines]} of every function body in an assembly file"""
    out, name, body = {}, None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L") and not line.startswith("\t"):
            if name:
                out[name] = body
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if line.strip().startswith(".end_amdhsa_kernel") or line.strip().startswith(".Lfunc_end"):
                out[name] = body
                name, body = None, []
    return out


def check_function(lines):
    """[(line_no, text)] of barriers reachable with an LDS memory operation pending"""
    ins = []          # (text, label or None)
    labels = {}
    for l in lines:
        t = l.split(";")[0].strip()
        if not t:
            continue
        m = re.match(r"^(\.LBB\w+):$", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith("."):
            continue
        ins.append(t)
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, t in enumerate(ins):
        op = t.split()[0]
        tgt = t.split()[-1] if op.startswith("s_branch") or op.startswith("s_cbranch") else None
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            if tgt in labels and labels[tgt] < n:
                succ[i].append(labels[tgt])
            continue
        if op.startswith("s_cbranch") and tgt in labels and labels[tgt] < n:
            succ[i].append(labels[tgt])
        if i + 1 < n:
            succ[i].append(i + 1)
    pending_in = [False] * n
    work = [0] if n else []
    seen = [False] * n
    while work:
        i = work.pop()
        p = pending_in[i]
        t = ins[i]
        op = t.split()[0]
        if op.startswith("ds_") and not op.startswith(NOT_MEMORY):
            p = True
        elif op == "s_waitcnt" and "lgkmcnt(0)" in t:
            p = False
        for s in succ[i]:
            if not seen[s] or (p and not pending_in[s]):
                seen[s] = True
                pending_in[s] = pending_in[s] or p
                work.append(s)
    return [(i, ins[i - 1] if i else "") for i in range(n) if ins[i].split()[0] == "s_barrier" and pending_in[i]]


def main(files):
    bad = 0
    for f in files:
        with tempfile.TemporaryDirectory() as tmp_dir:
            out = os.path.join(tmp_dir, "kernels.s")
            r = subprocess.run([HIPCC] + FLAGS + [f, "-o", out], capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit("%s does not compile:\n%s" % (f, r.stderr[-2000:]))
            text = open(out).read()
        n_barriers = 0
        for name, body in functions(text).items():
            n_barriers += sum(1 for l in body if l.strip() == "s_barrier")
            for (i, prev) in check_function(body):
                bad += 1
                print("%s: %s: s_barrier (instruction %d, after `%s`) reachable with an LDS operation in flight"
                      % (os.path.basename(f), name[:90], i, prev))
        print("%s: %d barriers checked" % (os.path.basename(f), n_barriers), flush=True)
    return bad


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "r3d_*.hip")))
    sys.exit(1 if main(files) else 0)
