#!/usr/bin/env python3
"""Plain and emitting scan of bench.py's 3 Gb assembly, back to back on one box: ms per launch of each (HIP events over N launches after
a warm-up), for A/B runs of library variants (TELOSCAN_LIB) and planner knobs.  python3 profiles/emit_time.py [launches]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
tel = ta.Teloscope(user_input(parse_cli("x.fa " + os.environ.get("TS_FLAGS", bench.FLAGS)), device=0))
lens = bench.contig_lengths(int(3e9), 200, 42)
L = K.lib()
out = []
for emit in (0, 1, 0, 1):
    plan = D.ShardPlan(tel, lens, world=1)
    buf = torch.zeros(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
    bench.fill_synthetic(buf, plan.segment_offsets(), lens, 42, dev)
    L.ts_batch_set_emit(plan.batch, emit)
    if "TIME_EVERY" in os.environ:                         # ts_batch_set_timing: one scan in n has a start event
        L.ts_batch_set_timing(plan.batch, int(os.environ["TIME_EVERY"]))
    st = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(st.cuda_stream)
    dp = C.c_void_p(buf.data_ptr())
    with torch.cuda.stream(st):
        for _ in range(100):
            assert L.ts_batch_scan(plan.batch, dp, sp) == 0
        torch.cuda.synchronize()
        assert L.ts_batch_sync(plan.batch) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n):
            L.ts_batch_scan(plan.batch, dp, sp)
        e1.record(st)
        torch.cuda.synchronize()
    out.append("%s %.4f" % ("emit" if emit else "plain", e0.elapsed_time(e1) / n))
    plan.close()
    del buf
print(os.environ.get("LABEL", "?"), " | ".join(out), flush=True)
