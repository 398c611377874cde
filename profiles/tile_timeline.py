"""Diagnostic: when does every wave of ts_scan_tiles finish?  Needs a library built with -DTS_EXP=8 (profiles/abx.sh build 8;
TELOSCAN_LIB=profiles/abx_8.so): every tile then leaves the 100 MHz timestamp of its end in its tile_stats row.  Prints the
distribution of the waves' last-tile times against the kernel's span, per XCD, and of the time per tile."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import teloscope_amd as ta
from teloscope_amd import _capi as K
from teloscope_amd import distributed as D
from teloscope_amd.cli import parse_cli, user_input

flags = sys.argv[1] if len(sys.argv) > 1 else bench.FLAGS
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
opts = parse_cli("x.fa " + flags)
tel = ta.Teloscope(user_input(opts, device=0))
L = K.lib()
lens = bench.contig_lengths(int(3e9), 200, 42)
plan = D.ShardPlan(tel, lens, world=1)
offsets = plan.segment_offsets()
buf = torch.zeros(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
bench.fill_synthetic(buf, offsets, lens, 42, dev)
sptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(6):
    assert L.ts_batch_scan(plan.batch, C.c_void_p(buf.data_ptr()), sptr) == 0
assert L.ts_batch_sync(plan.batch) == 0
info = plan.info
L.ts_batch_get_info(plan.batch, C.byref(info))
nt = int(info.n_tiles)
host = np.zeros(nt * 4, dtype=np.uint32)
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
assert hip.hipMemcpy(host.ctypes.data, L.ts_batch_tile_stats_ptr(plan.batch), host.nbytes, 2) == 0
word = host[3::4].astype(np.int64)
wave_of = word >> 20
end = word & 0xFFFFF
# the 20-bit clock wraps every 10.5 ms: unwrap around the median
med = int(np.median(end))
end = ((end - med + 0x80000) & 0xFFFFF) - 0x80000
end = end - end.min()                                    # 10 ns ticks since the first tile end
nwaves = int(wave_of.max()) + 1
print("kernel %.4f ms, %d tiles, %d waves, TS_DEALT_TILES=%s" % (info.avg_kernel_ms, nt, nwaves, os.environ.get("TS_DEALT_TILES", "")))
last = np.zeros(nwaves); first = np.zeros(nwaves); cnt = np.zeros(nwaves, dtype=np.int64)
order = np.argsort(wave_of, kind="stable")
bounds = np.searchsorted(wave_of[order], np.arange(nwaves + 1))
for w in range(nwaves):
    e = end[order[bounds[w]:bounds[w + 1]]]
    if len(e):
        last[w] = e.max(); first[w] = e.min(); cnt[w] = len(e)
print("tiles per wave: min %d median %d max %d" % (cnt.min(), np.median(cnt), cnt.max()))
span = last.max()
print("span of tile ends: %.1f us; a wave's last tile ends at: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us" % (
    span / 100, last.min() / 100, np.percentile(last, 10) / 100, np.median(last) / 100, np.percentile(last, 90) / 100, last.max() / 100))
print("a wave's FIRST tile ends at: min %.1f median %.1f p90 %.1f max %.1f us" % (first.min() / 100, np.median(first) / 100, np.percentile(first, 90) / 100, first.max() / 100))
per_tile = (last - first) / np.maximum(cnt - 1, 1)
print("time per tile by wave: min %.2f median %.2f p90 %.2f max %.2f us" % (per_tile.min() / 100, np.median(per_tile) / 100, np.percentile(per_tile, 90) / 100, per_tile.max() / 100))
# workgroup w / 16 -> XCD = workgroup % 8 (round-robin dispatch)
wg = np.arange(nwaves) // 16
for x in range(8):
    m = (wg % 8) == x
    print("  XCD %d: last tile end median %.1f max %.1f us, per tile %.2f us" % (x, np.median(last[m]) / 100, last[m].max() / 100, np.median(per_tile[m]) / 100))
# which wave slot of its SIMD is fast?  wave w of a workgroup sits on SIMD w % 4 (slot w / 4)
wslot = (np.arange(nwaves) % 16) // 4
simd = np.arange(nwaves) % 4
for q in range(4):
    print("  waves %d..%d of a workgroup: %.2f us per tile, %.1f tiles;   SIMD %d: %.2f us per tile" % (
        4 * q, 4 * q + 3, np.median(per_tile[wslot == q]) / 100, cnt[wslot == q].mean(), q, np.median(per_tile[simd == q]) / 100))
idle = (span - last).sum() / (span * nwaves)
print("share of wave-time idle at the tail (after a wave's last tile, before the kernel's): %.1f %%" % (100 * idle))
