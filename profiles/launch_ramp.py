"""Diagnostic: how does the time of one ts_scan_tiles launch over the 3 Gb bench workload change with the number of launches
the process has already made?  Prints the per-launch time of the first 128 launches (each bracketed by its own pair of HIP
events on the launch stream; the host waits for each before the next, then not at all), then the average of 80 back-to-back
launches.  bench.py's SETTLE_LAUNCHES comes from this."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import teloscope_amd as ta
from teloscope_amd import _capi as K
from teloscope_amd import distributed as D
from teloscope_amd.cli import parse_cli, user_input

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench.bind_to_gpu_node(0)
tel = ta.Teloscope(user_input(parse_cli("x.fa " + bench.FLAGS), device=0))
L = K.lib()
lens = bench.contig_lengths(int(3e9), 200, 42)
plan = D.ShardPlan(tel, lens, world=1)
offsets = plan.segment_offsets()
buf = torch.zeros(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
bench.fill_synthetic(buf, offsets, lens, 42, dev)
torch.cuda.synchronize()
stream = torch.cuda.current_stream()
sptr = C.c_void_p(stream.cuda_stream)
dptr = C.c_void_p(buf.data_ptr())


def launches(n, synced):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(stream)
        assert L.ts_batch_scan(plan.batch, dptr, sptr) == 0
        b.record(stream)
        if synced:
            b.synchronize()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


for title, synced in (("host waits for every launch", True), ("launches queued back to back", False)):
    t = launches(128, synced)
    print("%s -- ms per launch, launches 1..128 of this phase:" % title)
    for i in range(0, 128, 8):
        print("  %3d  " % (i + 1) + " ".join("%.3f" % x for x in t[i:i + 8]))
t = launches(80, False)
print("80 more back-to-back launches: mean %.4f ms, min %.4f, max %.4f" % (sum(t) / len(t), min(t), max(t)))
assert L.ts_batch_sync(plan.batch) == 0
