#!/usr/bin/env python3
"""The sharded step of bench.py's scan_plus_block_calling (emitting scan + pack beside the next scan, four slots) with the pack
left out (PACK=0) or with a library variant whose pack skips kernels (-DTS_PACK_ABL, TELOSCAN_LIB): what each of the pack's kernels
costs the scan it runs beside.  No result is checked here (the variants' results are wrong).  python3 profiles/pack_abl_time.py [steps]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import teloscope_amd as ta
import teloscope_amd.distributed as D
from teloscope_amd.cli import parse_cli, user_input

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gb = float(os.environ.get("GBASES", "3.0"))
ncontig = int(os.environ.get("CONTIGS", "200"))
dev = torch.device("cuda", 0)
tel = ta.Teloscope(user_input(parse_cli("x.fa " + bench.FLAGS), device=0))
lens = bench.contig_lengths(int(gb * 1e9), ncontig, 42)
slots = int(os.environ.get("SLOTS", "4"))
out = []
# EV: "torch" torch.cuda.Event between the streams (bench.py's way), "hip" HIP events made with hipEventDisableTiming |
# hipEventDisableSystemFence, "none" no events at all (only meaningful with PACK=0); PRIO=1: the pack streams at high priority
EV = os.environ.get("EV", "torch")
PRIO = int(os.environ.get("PRIO", "0"))
hip = C.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]


class HipEvent:
    def __init__(self):
        self.e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(self.e), 0x2 | 0x20000000) == 0

    def record(self, st):
        assert hip.hipEventRecord(self.e, C.c_void_p(st.cuda_stream)) == 0


def wait(st, ev):
    if isinstance(ev, HipEvent):
        assert hip.hipStreamWaitEvent(C.c_void_p(st.cuda_stream), ev.e, 0) == 0
    else:
        st.wait_event(ev)


for do_pack in ([1, 0, 1, 0] if os.environ.get("PACK", "both") == "both" else [int(os.environ["PACK"])] * 2):
    plan = D.ShardPlan(tel, lens, world=1)
    buf = torch.zeros(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
    bench.fill_synthetic(buf, plan.segment_offsets(), lens, 42, dev)
    shard = D.PackedShard(plan, 0, dev, slots=slots, scale=1)
    if "TIME_EVERY" in os.environ:
        shard.set_timing(int(os.environ["TIME_EVERY"]))
    NSCAN, NPACK = int(os.environ.get("SCAN_STREAMS", "1")), int(os.environ.get("PACK_STREAMS", "2"))
    if os.environ.get("STREAMS") == "probe":               # scan + pack streams on hardware queues of their own
        got = D.concurrent_streams(tel, dev, NSCAN + NPACK)
        scan_streams, pack_streams = got[:NSCAN], got[NSCAN:]
    else:
        scan_streams = [torch.cuda.Stream(device=dev) for _ in range(NSCAN)]
        pack_streams = [torch.cuda.Stream(device=dev, priority=-1 if PRIO else 0) for _ in range(NPACK)]
    main = scan_streams[0]
    mk = HipEvent if EV == "hip" else torch.cuda.Event
    scanned = [mk() for _ in range(slots)]
    packed = [mk() for _ in range(slots)]
    used = [False] * slots
    in_ptr = buf.data_ptr()

    def step(i):
        j = i % slots
        ps = pack_streams[j % len(pack_streams)]
        stream = scan_streams[i % len(scan_streams)]
        sptr = C.c_void_p(stream.cuda_stream)
        if EV == "none":
            shard.scan(in_ptr, sptr, j)
            return
        if used[j] and EV != "record-only":
            wait(stream, packed[j])
        shard.scan(in_ptr, sptr, j)
        if EV == "wait-only":
            used[j] = True
            packed[j].record(ps)
            return
        if EV != "lib":
            scanned[j].record(stream)
        with torch.cuda.stream(ps):
            if EV == "lib":                         # ts_batch_wait_scan: the library's own event behind the scan
                shard.wait_scan(C.c_void_p(ps.cuda_stream), j)
            else:
                wait(ps, scanned[j])
            if do_pack:
                shard.pack(C.c_void_p(ps.cuda_stream), j)
            packed[j].record(ps)
        used[j] = True

    with torch.cuda.stream(main):
        for _ in range(2):
            for i in range(slots):
                step(i)
            torch.cuda.synchronize()
            for j in range(slots):
                shard.sync(j)
        for i in range(40 * slots // 4):
            step(i)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        out.append("%s %.4f" % ("scan+pack" if do_pack else "scan only", (time.perf_counter() - c0) / steps * 1e3))
    shard.close()
    plan.close()
    del buf
print(os.environ.get("LABEL", "?"), " | ".join(out), flush=True)
