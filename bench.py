#!/usr/bin/env python3
"""bench.py — Gbases/s scanned by the HIP telomeric-motif scan on MI355X.

Workload (BASELINE.json configs[1] / configs[2]): synthetic 3.0 Gb human-scale assembly, 200 contigs
(log-uniform 1-250 Mb, seed 42), telomeric arrays + TVRs at contig ends, planted ITS blocks, an N-run in 1 % of
the contigs, 0.1 % soft-masked bases; flags -c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -x 1 -w 1000 -s 500
-r -g -e -m -i  (124 patterns, k = 6).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--gbases G]

N = 1: one "step" = one full scan of the resident assembly: window records (8 x u32 per window) and the packed
match stream are produced in HBM.
N > 1 (strong scaling, configs[2]): the SAME assembly; every rank builds the same plan and scans the p-th of N
consecutive tile ranges plus a few context tiles (teloscope_amd/distributed.py), calls its blocks on its own device
and packs ONE message of a size both sides know from the plan — bit-packed window records, the match records a
writer reads, its blocks — and ONE grouped send / recv per step over RCCL brings the messages to rank 0, whose host
merge (ts_shards_finalize, untimed, before and after the timed steps) leaves what the reference's writers read of a
single-GPU scan.  A step = scan + block calling + pack + exchange; nothing is read back to the host inside a step,
the pack and the exchange of step i run beside the scans of the steps after it (buffer slots), and every exchange
is complete before the clock stops.  `--full-exchange` is round 2's form (every record to rank 0).  Ranks are
started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, when those are absent, by
this script itself (`python bench.py --gpus N` spawns N rank processes before anything touches a GPU).

  python bench.py --reads [--gpus N]        the read filter (configs[3]) instead of the assembly scan
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The pieces live in benchlib/ (common, contigs, reads, verify, launcher); the names below are what tests and profile scripts take
# from `import bench`.
from benchlib.common import FLAGS, HBM_PEAK_GBS, READ_FLAGS, SETTLE_LAUNCHES  # noqa: E402,F401
from benchlib.contigs import contig_lengths, fill_synthetic  # noqa: E402,F401
from benchlib.reads import fill_read_range, read_lengths  # noqa: E402,F401
from benchlib.launcher import main  # noqa: E402

if __name__ == "__main__":
    main()
