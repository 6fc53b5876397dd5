"""Shared by bench.py and the tools/bench_*.py modules it imports: the workload's constants and two helpers."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BENCH = os.path.join(ROOT, "bench.py")

H, W, FRAMES_PER_GPU = 384, 1280, 100
BYTES_PER_POINT = 13          # 1 B u8 depth read + 12 B f32 xyz written
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0         # ... and the float4 copy it measures (read + write mixed): the practical ceiling of a 1:1 stream
XGMI_LINK_GBS = 153.0         # one xGMI link, per direction (7 links per GPU, full mesh of 8)
RASTER_COPIES = 16            # rotating rasters of the headline step: 16 x 49 MB = 786 MB, three Infinity Caches' worth


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_child(workload, timeout=300):
    """`bench.py --workload <workload>` as a child process; its last JSON line as a dict, or {"failed": ...}.  The N=1
    headline starts its side measurements this way BEFORE it touches the GPU itself: the same kernel symbol at other regimes
    must not mix into the rocprofv3 statistics of the parent's launches."""
    try:
        r = subprocess.run([sys.executable, BENCH, "--workload", workload], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"failed": "child exited with %d: %s" % (r.returncode, (r.stderr or r.stdout)[-200:])}
        return json.loads(lines[-1])
    except Exception as e:  # pragma: no cover
        return {"failed": "%s: %s" % (type(e).__name__, str(e)[:160])}
