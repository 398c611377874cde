"""benchlib.common — constants and process-wide state shared by the pieces of bench.py (the entry point at the repository root).

benchlib/contigs.py   the assembly scan (configs[1] / [2] / [4]): synthetic assembly, N = 1 and N > 1 steps, the sub-records
benchlib/reads.py     the read filter (configs[3])
benchlib/verify.py    the untimed full-size parity checks (--verify)
benchlib/launcher.py  arguments, rank processes, NUMA binding, the watchdog, main()
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # the repository root (bench.py lives there)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# (HIP streams share the runtime's hardware queues round-robin — 4 by default, GPU_MAX_HW_QUEUES — and kernels of two streams
# that landed on one queue run one after the other.  With a side stream per batch a sharded step used seven streams and 8 queues
# measured 0.174 against 0.197 ms at the N = 8 size (profiles/r04/hwq_sweep.txt); since the terminal walks of every batch share
# the context's one side stream a step uses four, and 4 and 8 queues measure the same: the runtime's default is left alone.)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FLAGS = "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -x 1 -w 1000 -s 500 -r -g -e -m -i"
LIB = os.path.join(ROOT, "teloscope_amd", "libteloscan.so")
ORACLE = os.path.join(ROOT, "oracle", "libteloscope_oracle.so")
READ_FLAGS = "--fastq-subset -l 42"
READ_CHUNK = 100_000            # reads per generation chunk: the read set does not depend on how it is dealt to ranks
HOST_NUMA_NODE = None
ORIG_AFFINITY = None            # the CPUs the process was started with (before it bound itself to the GPU's NUMA node)
SETTLE_LAUNCHES = 96            # untimed scans before the warm-up steps, see run_scan (profiles/summarize.py drops them too)



class Watchdog:
    """A deadline around a step that may never return (first contact of N ranks over RCCL: a rank that never posts its send leaves
    rank 0's grouped receive waiting for ever, and the driver's clock with it).  When `seconds` pass before cancel(), the process
    says — one JSON line on stderr: which rank, which stage, what was posted — and ends itself with os._exit (no exec, no
    clean-up that could block behind the very call that hangs); the launcher (torch.distributed.run, or bench.py's own spawn)
    then stops the other ranks.  `stage` is a free-text note the watched code keeps up to date."""

    def __init__(self, seconds, rank, what):
        import threading
        self.rank, self.what, self.seconds = rank, what, float(seconds)
        self.stage = "started"
        self.t0 = time.perf_counter()
        self._timer = threading.Timer(self.seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()

    def _fire(self):
        sys.stderr.write(json.dumps({"bench_watchdog": {"rank": self.rank, "what": self.what, "stage": self.stage,
                                                        "waited_s": round(time.perf_counter() - self.t0, 1),
                                                        "limit_s": self.seconds}}) + "\n")
        sys.stderr.flush()
        os._exit(3)

    def cancel(self):
        self._timer.cancel()
        return time.perf_counter() - self.t0


__all__ = ["Watchdog", "argparse", "C", "json", "os", "socket", "subprocess", "sys", "time", "ROOT", "HBM_PEAK_GBS", "FLAGS", "LIB", "ORACLE",
           "READ_FLAGS", "READ_CHUNK", "SETTLE_LAUNCHES"]
