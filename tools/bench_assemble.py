"""The N > 1 half of bench.py: how the fused world cloud is assembled on every rank (BASELINE config 4's exchange step).

Strategies (all leave the same bits in `full`, checked here against a single-launch sample for EVERY strategy):
  none            shards stay resident (no exchange): the quantity that scales with N
  outputs         fuse own frames, all-gather the xyz shards (12 B/point over xGMI; north_star's wording)
  inputs          all-gather rasters + poses (1 B/point), fuse every rank's frames locally
  *_direct        the same exchange as one grouped send/recv per peer instead of ncclAllGather (r3d_comm only)
  inputs_overlap  'inputs' as a pipeline: slices travel on a side stream while the landed ones are fused

The exchange code has only ever met RCCL with one rank (no multi-GPU node was available to the build), so the survey runs
under a watchdog: bench.py measures the shards-stay-resident job FIRST by the full contract, and if a strategy then stalls
for WATCHDOG_S seconds rank 0 prints THAT line, flagged, and every rank exits with code 4.
"""
import glob
import importlib
import json
import os
import re
import tempfile
import threading
import time

import numpy as np

from bench_common import H, W, XGMI_LINK_GBS

WATCHDOG_S = int(os.environ.get("R3D_BENCH_WATCHDOG_S", "240"))
OVERLAP_CHUNKS = 4            # slices of the pipelined 'inputs' assembly
STEP_TEXT = {"none": "1 fused launch over this rank's frames (shards stay resident)",
             "outputs": "1 fused launch + all-gather of xyz shards (12 B/point over xGMI)",
             "inputs": "all-gather of depth+poses (1 B/point over xGMI) + 1 fused launch over all ranks' frames on every rank "
                       "(replicated compute: each GPU writes the whole cloud into its own HBM)"}


def step_text(mode):
    return (STEP_TEXT[mode.replace("_direct", "").replace("_overlap", "")]
            + (" [grouped send/recv per peer]" if mode.endswith("_direct") else "")
            + (" [pipelined: %d slices gathered on a side stream while the landed ones are fused]" % OVERLAP_CHUNKS
               if mode.endswith("_overlap") else ""))


def rccl_debug_env(world):
    """First contact with RCCL at N > 1 should answer SURVEY 5's ring-vs-direct question without a second lease: unless the
    caller set NCCL_DEBUG (or R3D_BENCH_NCCL_DEBUG=0), RCCL logs its INIT and TUNING lines into one file per process, which
    rank 0 parses after the survey (`comm.rccl_log`).  Must run before anything loads RCCL."""
    if world <= 1 and os.environ.get("R3D_BENCH_FORCE_COLLECTIVES", "0") in ("", "0"):
        return None
    if os.environ.get("R3D_BENCH_NCCL_DEBUG", "1") in ("", "0"):
        return None
    if "NCCL_DEBUG_FILE" in os.environ:          # the caller routes RCCL's log already: leave it alone
        return None
    d = tempfile.mkdtemp(prefix="r3d_nccl_")
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,TUNING", NCCL_DEBUG_FILE=os.path.join(d, "rccl.%p.log"))
    return d


def parse_rccl_log(log_dir):
    """What RCCL said it chose: one entry per distinct (collective, bytes) TUNING line -- `AllGather: 49152000 Bytes -> Algo
    RING proto SIMPLE channel{Lo..Hi}={0..15}` -- and the INIT lines that describe the fabric (rings / trees / channels)."""
    if not log_dir:
        return {"picked": {}, "init_lines": [], "how": "not captured (R3D_BENCH_NCCL_DEBUG=0, or NCCL_DEBUG_FILE set by the caller)"}
    picks, init = {}, []
    for path in glob.glob(os.path.join(log_dir, "rccl.%d.log*" % os.getpid())) or glob.glob(os.path.join(log_dir, "rccl.*")):
        try:
            with open(path, errors="replace") as f:
                for ln in f:
                    m = re.search(r"(\w+): (\d+) Bytes -> Algo (\S+) proto (\S+)(?: channel\{Lo\.\.Hi\}=\{(\d+)\.\.(\d+)\})?", ln)
                    if m:
                        key = "%s %s B" % (m.group(1), m.group(2))
                        picks[key] = {"algo": m.group(3), "proto": m.group(4),
                                      "channels": (int(m.group(6)) - int(m.group(5)) + 1) if m.group(5) else None}
                    elif re.search(r"Connected all|nChannels|comm 0x\w+ rank \d+ nranks|RCCL version|NCCL version|Trees|Channel 00", ln) \
                            and len(init) < 24:
                        init.append(ln.strip()[-200:])
        except OSError:
            pass
    return {"picked": picks, "init_lines": init, "how": "NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,TUNING into a file, this rank's log"}


class Assembly:
    """State of the exchange step of one rank.  `B` is bench.py's namespace of the job: torch, dist, r3d, ctx, cam, stream,
    dev, dev_index, backend, rank, world, F, depth (this step's raster), table, shard, full, depth_all, pose_all, out_np,
    xyz_bytes, n_local, fuse, fuse_all, fence, max_over_ranks."""

    def __init__(self, B):
        self.B = B
        self.D = importlib.import_module("3d_reconstruction_system_amd.dist")
        self.transport = self.side = self.side_transport = self.ctx2 = None
        self.note = ""
        self.frames_pr, self.points_pr = [B.F] * B.world, [B.n_local] * B.world
        self.beat = {"t": time.monotonic(), "what": "start", "armed": False}
        self.results = {}
        self.reference = None          # strided sample of the world cloud from ONE launch over every rank's frames

    # -- transports -------------------------------------------------------------------------------------------------
    def setup_transports(self):
        """The library's own RCCL communicator (C ABI, r3d_comm_*) when it comes up on every rank, torch.distributed otherwise
        (always for gloo rehearsals); and a second channel on a side stream for the pipelined strategy."""
        B, D = self.B, self.D
        want = os.environ.get("R3D_BENCH_TRANSPORT", "r3d" if B.backend == "nccl" else "torch")
        if want == "r3d":
            try:
                CM = importlib.import_module("3d_reconstruction_system_amd.comm")
                box = [CM.Comm.unique_id() if B.rank == 0 else None]
                B.dist.broadcast_object_list(box, src=0)
                self.transport = D.R3dTransport(CM.Comm(B.ctx, box[0], B.rank, B.world))
                self.note = "r3d_comm over RCCL (%s)" % self.transport.comm.rccl_origin()
            except Exception as e:     # e.g. no librccl to dlopen: every rank takes the same way out
                self.transport, self.note = None, "r3d_comm unavailable (%s: %s); " % (type(e).__name__, str(e)[:120])
        if self.transport is None:
            self.transport = D.TorchTransport()
            self.note += "torch.distributed (%s)" % B.backend
        if B.F % OVERLAP_CHUNKS == 0:
            try:
                self.side = B.torch.cuda.Stream(B.dev)
                if isinstance(self.transport, D.R3dTransport):
                    CM = importlib.import_module("3d_reconstruction_system_amd.comm")
                    self.ctx2 = B.r3d.Context(B.dev_index, stream=self.side.cuda_stream)
                    box = [CM.Comm.unique_id() if B.rank == 0 else None]
                    B.dist.broadcast_object_list(box, src=0)
                    self.side_transport = D.R3dTransport(CM.Comm(self.ctx2, box[0], B.rank, B.world))
                else:
                    self.side_transport = self.transport        # torch collectives follow torch's current stream
            except Exception:
                self.side = self.side_transport = None

    def comm_report(self, log_dir):
        """`comm` object of the line: what the communicator itself says about the job (RCCL's own count / rank / device, not the
        launcher's environment) and what RCCL logged about its choices."""
        B = self.B
        rep = {"launcher_world": B.world, "transport": self.note}
        if isinstance(self.transport, self.D.R3dTransport):
            rep["rccl"] = self.transport.comm.rccl_report()
        else:
            rep["torch_world"] = B.dist.get_world_size()
        rep["rccl_log"] = parse_rccl_log(log_dir)
        return rep

    # -- steps ------------------------------------------------------------------------------------------------------
    def _algo(self, algo):
        if isinstance(self.transport, self.D.R3dTransport):
            self.transport.algo = algo

    def make_step(self, m):
        B, D = self.B, self.D
        algo = 2 if m.endswith("_direct") else 0
        if m == "none":
            return B.fuse
        if m.startswith("outputs"):
            def step_outputs():
                B.fuse()
                self._algo(algo)
                self.transport.allgather_rows(B.shard, self.points_pr, out=B.full)
            return step_outputs
        if m == "inputs_overlap":
            torch, C_, fc, per = B.torch, OVERLAP_CHUNKS, B.F // OVERLAP_CHUNKS, H * W
            d_chunks = [torch.empty((B.world * fc, H, W), dtype=torch.uint8, device=B.dev) for _ in range(C_)]
            p_chunks = [torch.empty((B.world * fc, 12), dtype=torch.float64, device=B.dev) for _ in range(C_)]
            events = [torch.cuda.Event() for _ in range(C_)]

            def step_overlap():
                depth = B.next_raster()
                self.side.wait_stream(B.stream)               # inputs are ready / last step's fuses have read the chunks
                with torch.cuda.stream(self.side):
                    for c in range(C_):
                        self.side_transport.allgather_rows(depth[c * fc:(c + 1) * fc], [fc] * B.world, out=d_chunks[c])
                        self.side_transport.allgather_rows(B.table[c * fc:(c + 1) * fc], [fc] * B.world, out=p_chunks[c])
                        events[c].record(self.side)
                for c in range(C_):
                    B.stream.wait_event(events[c])
                    if not isinstance(self.side_transport, D.R3dTransport):
                        B.ctx.inputs_fresh()
                    for r in range(B.world):
                        B.r3d.fuse_frames_device(B.ctx, B.cam, d_chunks[c][r * fc:].data_ptr(), np.uint8, fc,
                                                 p_chunks[c][r * fc:].data_ptr(), B.full[(r * B.F + c * fc) * per:].data_ptr(), B.out_np)
            return step_overlap

        def step_inputs():
            self._algo(algo)
            self.transport.allgather_rows(B.next_raster(), self.frames_pr, out=B.depth_all)
            self.transport.allgather_rows(B.table, self.frames_pr, out=B.pose_all)
            if not isinstance(self.transport, D.R3dTransport):
                B.ctx.inputs_fresh()      # torch's collective wrote the rasters: a foreign producer (r3d_comm tracks its own)
            B.fuse_all()
        return step_inputs

    # -- the bitwise check of a strategy ------------------------------------------------------------------------------
    def _check(self, m, step):
        """Every strategy must leave the single-launch cloud on this rank, bit for bit: a strided sample (every 997th point of
        the world cloud; for 'none' this rank's shard) against ONE fused launch over every rank's frames of raster 0."""
        B = self.B
        B.reset_rasters()
        if self.reference is None:
            self._algo(0)
            self.transport.allgather_rows(B.next_raster(), self.frames_pr, out=B.depth_all)
            self.transport.allgather_rows(B.table, self.frames_pr, out=B.pose_all)
            B.ctx.inputs_fresh()
            B.fuse_all()
            B.torch.cuda.synchronize(B.dev)
            self.reference = B.full[::997].clone()
            B.reset_rasters()
        B.full.zero_()
        step()
        B.torch.cuda.synchronize(B.dev)
        got = B.full[::997]
        if m == "none":                   # only this rank's slot is written
            idx = B.torch.arange(0, B.full.shape[0], 997, device=B.dev)
            mine = (idx >= B.rank * B.n_local) & (idx < (B.rank + 1) * B.n_local)
            return bool(B.torch.equal(self.reference[mine], got[mine]))
        return bool(B.torch.equal(self.reference, got))

    def check_none(self):
        return self._check("none", self.B.fuse)

    # -- the survey -------------------------------------------------------------------------------------------------
    def _watchdog(self, fallback_line):
        while self.beat["armed"]:
            time.sleep(1.0)
            if self.beat["armed"] and time.monotonic() - self.beat["t"] > WATCHDOG_S:
                if self.B.rank == 0:
                    line = fallback_line()
                    line["watchdog"] = "assembly strategy '%s' made no progress for %d s; this line is the shards-stay-resident job " \
                                       "measured before the survey" % (self.beat["what"], WATCHDOG_S)
                    print(json.dumps(line), flush=True)
                os._exit(4)      # a wedged exchange is a failed multi-GPU run even though a line went out

    def arm(self, fallback_line):
        self.beat.update(t=time.monotonic(), armed=True, what="communicator set-up")
        threading.Thread(target=self._watchdog, args=(fallback_line,), daemon=True).start()

    def disarm(self):
        self.beat["armed"] = False

    def survey(self, choice):
        """Times every strategy briefly (3 + 10 steps) and checks its bits; returns the mode of the headline step."""
        B = self.B
        modes = ["outputs", "inputs"]
        if isinstance(self.transport, self.D.R3dTransport):
            modes += ["outputs_direct", "inputs_direct"]
        if self.side_transport is not None:
            modes.append("inputs_overlap")
        for m in modes:
            self.beat.update(t=time.monotonic(), what=m)
            try:   # a side measurement must never cost the headline line
                st = self.make_step(m)
                same = self._check(m, st)
                self.beat["t"] = time.monotonic()
                for _ in range(3):
                    st()
                B.fence()
                self.beat["t"] = time.monotonic()
                t1 = time.perf_counter()
                for _ in range(10):
                    st()
                B.fence()
                sec = B.max_over_ranks((time.perf_counter() - t1) / 10)
                self.beat["t"] = time.monotonic()
                fabric_in = (B.world - 1) * (B.n_local * B.xyz_bytes if m.startswith("outputs") else B.F * (H * W + 96))
                entry = {"ms_per_step": round(sec * 1e3, 4), "Mpoints_s": round(B.world * B.n_local / sec / 1e6, 1),
                         "fabric_bytes_in_per_gpu": fabric_in, "same_bits_as_single_launch": same}
                if fabric_in and B.world > 1:
                    gbs = fabric_in / sec / 1e9          # whole step time, compute included: a lower bound on the links
                    entry["xgmi_GBps_in_per_gpu"] = round(gbs, 1)
                    entry["xgmi_GBps_per_link"] = round(gbs / (B.world - 1), 1)
                    entry["frac_of_link_peak"] = round(gbs / (B.world - 1) / XGMI_LINK_GBS, 4)
                self.results[m] = entry
            except Exception as e:  # pragma: no cover
                self.results[m] = {"failed": "%s: %s" % (type(e).__name__, str(e)[:100])}
        ok = {m: v["ms_per_step"] for m, v in self.results.items()
              if "ms_per_step" in v and m != "none" and v.get("same_bits_as_single_launch", True)}
        if choice == "auto":
            mode = min(ok, key=ok.get) if ok else "none"
        else:
            mode = choice if (choice in ok or choice == "none") else "none"
        self.beat.update(t=time.monotonic(), what="headline (%s)" % mode)
        return mode

    def close(self):
        if self.ctx2 is not None:
            self.ctx2.close()
