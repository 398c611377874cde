#!/usr/bin/env python3
"""bench.py — Gbases/s scanned by the HIP telomeric-motif scan on MI355X.

Workload (BASELINE.json configs[1]): synthetic 3.0 Gb human-scale assembly, 200 contigs
(log-uniform 1-250 Mb), telomeric arrays + TVRs at contig ends, planted ITS blocks, an N-run
in 1 % of the contigs, 0.1 % soft-masked bases; flags -c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -x 1 -w 1000 -s 500
-r -g -e -m -i  (124 patterns, k = 6).  One "step" = one full scan of the resident batch:
window records (8 x u32 per window) and the packed match stream are produced in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--gbases G]

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank scans its own
3.0 Gb shard of contigs (weak scaling, contigs are independent units) and the per-segment
hit summaries are gathered to rank 0 over RCCL inside the timed region (asynchronously, overlapping the next step's scan).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (imported before libteloscan so both share one HIP runtime)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FLAGS = "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -x 1 -w 1000 -s 500 -r -g -e -m -i"


def contig_lengths(total, n, seed):
    rng = np.random.default_rng(seed)
    raw = np.exp(rng.uniform(np.log(1e6), np.log(250e6), size=n))
    lens = np.maximum((raw * (total / raw.sum())).astype(np.int64), 20000)
    lens[-1] += total - lens.sum()
    return [int(x) for x in lens]


def fill_synthetic(buf, offsets, lens, seed, dev):
    """Random ACGT + telomeres/TVRs at both ends of every contig + ITS blocks + N-runs + soft-masking,
    generated on the device (model of src/get-mock-chr.cpp:96-136)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    chunk = 1 << 28
    for a in range(0, buf.numel(), chunk):
        b = min(buf.numel(), a + chunk)
        idx = torch.randint(0, 4, (b - a,), dtype=torch.uint8, device=dev, generator=g)
        buf[a:b] = lut[idx.long()]
        m = torch.rand(b - a, device=dev, generator=g) < 0.001          # 0.1 % lower case
        buf[a:b] |= (m.to(torch.uint8) << 5)
        del idx, m
    rng = np.random.default_rng(seed + 1)

    def tract(unit, reps, rate):
        t = np.tile(np.frombuffer(unit, dtype=np.uint8), reps).copy()
        k = rng.random(len(t)) < rate
        t[k] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(k.sum()))]
        return torch.from_numpy(t).to(dev)

    for off, n in zip(offsets, lens):
        p = torch.cat([tract(b"CCCTAA", 2000, 0.0), tract(b"CCCTAA", 100, 1.0 / 6)])
        q = torch.cat([tract(b"TTAGGG", 100, 1.0 / 6), tract(b"TTAGGG", 2000, 0.0)])
        if len(p) + len(q) < n:
            buf[off:off + len(p)] = p
            buf[off + n - len(q):off + n] = q
    n_its = 50
    for _ in range(n_its):
        ci = int(rng.integers(0, len(lens)))
        ln = int(rng.integers(200, 2000)) // 6
        unit = b"TTAGGG" if rng.random() < 0.5 else b"CCCTAA"
        at = int(rng.integers(20000, max(20001, lens[ci] - 20000 - 6 * ln)))
        t = tract(unit, ln, 0.02)
        buf[offsets[ci] + at:offsets[ci] + at + len(t)] = t
    for ci in rng.choice(len(lens), size=max(1, len(lens) // 100), replace=False):    # 1 % of contigs: an N-run
        ln = int(rng.integers(100, 10001))
        at = int(rng.integers(20000, max(20001, lens[ci] - 20000 - ln)))
        buf[offsets[ci] + at:offsets[ci] + at + ln] = ord("N")


def verify_full_size(L, batch, tel, buf, offsets, lens, ui, dev):
    """Size-independent parity properties at the full bench size, with an INDEPENDENT computation
    in torch (rolling 2-bit k-mer code + table lookup, nothing shared with the HIP kernels):
      * per contig: matches / canonical / forward counts == ts_batch_segment_summary;
      * per contig: A,C,G,T totals == sum of the nucleotide counts of the windows that tile the contig
        (w = 2s: the even-indexed ones; w = s: all of them);
      * w = s: matches straddling a window end are excluded, as the reference loses them.
    Returns a dict for the bench line; raises on any mismatch."""
    n = len(lens)
    k = len(ui.patternInfo[0][0])
    code_of = {"A": 0, "C": 1, "T": 2, "G": 3}
    tbl = torch.zeros(3, 4 ** k, dtype=torch.bool)
    for pat, fwd in ui.patternInfo:
        x = sum(code_of[ch] << (2 * i) for i, ch in enumerate(pat))
        tbl[0, x] = True
        tbl[1, x] = bool(fwd)
        tbl[2, x] = pat in (ui.canonicalFwd, ui.canonicalRev)
    tbl = tbl.to(dev)
    lut = torch.full((256,), 4, dtype=torch.int32)
    for ch, c in code_of.items():
        lut[ord(ch)] = c
        lut[ord(ch.lower())] = c
    lut = lut.to(dev)
    summ = torch.zeros(n * 4, dtype=torch.int64, device=dev)
    if L.ts_batch_segment_summary(batch, C.c_void_p(summ.data_ptr()), None) != 0:
        raise RuntimeError(tel._ctx.error())
    torch.cuda.synchronize()
    summ = summ.view(n, 4).cpu().numpy()
    info = __import__("teloscope_amd")._capi.BatchInfo()
    L.ts_batch_get_info(batch, C.byref(info))
    wins = torch.empty(int(info.n_windows) * 8, dtype=torch.int32, device=dev)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    wp = L.ts_batch_windows_ptr(batch)
    assert hip.hipMemcpy(C.c_void_p(wins.data_ptr()), C.c_void_p(wp), C.c_size_t(wins.numel() * 4), 3) == 0
    wins = wins.view(-1, 8)
    step, window = ui.step, ui.windowSize
    assert window in (step, 2 * step), "--verify knows the window tilings of w = s and w = 2s"
    stride = window // step                                  # every stride-th window: together they tile a contig
    wbase = 0
    chunk = 1 << 27
    for ci in range(n):
        nb, off = lens[ci], offsets[ci]
        cnt = torch.zeros(3, dtype=torch.int64, device=dev)
        nuc = torch.zeros(4, dtype=torch.int64, device=dev)
        for a in range(0, nb, chunk):
            b = min(nb, a + chunk + k - 1)
            c = lut[buf[off + a:off + b].long()]
            own = min(nb, a + chunk) - a
            nuc += torch.bincount(c[:own], minlength=5)[:4]
            m = b - a - k + 1
            if m > 0:
                code = torch.zeros(m, dtype=torch.int32, device=dev)
                bad = torch.zeros(m, dtype=torch.bool, device=dev)
                for i in range(k):
                    ci_ = c[i:i + m]
                    code += (ci_ & 3) << (2 * i)
                    bad |= ci_ == 4
                take = min(m, own)
                code, bad = code[:take].long(), bad[:take]
                if window == step:                           # w == s: a match that straddles a window end is lost
                    pos = torch.arange(a, a + take, device=dev)
                    bad = bad | ((pos % step) + k > step)
                for f in range(3):
                    cnt[f] += (tbl[f][code] & ~bad).sum()
            del c
        nwin = -(-nb // step)
        w = wins[wbase:wbase + nwin]
        wbase += nwin
        got_nuc = w[0::stride, [0, 1, 3, 2]].sum(dim=0, dtype=torch.int64)  # records are A C G T; codes A C T G
        assert summ[ci].tolist() == [nwin, int(cnt[0]), int(cnt[2]), int(cnt[1])], \
            ("match counts differ on contig %d" % ci, summ[ci].tolist(), cnt.tolist())
        assert got_nuc.tolist() == nuc.tolist(), ("nucleotide totals differ on contig %d" % ci)
        cov = w[:, 4:8].sum(dim=0, dtype=torch.int64)                       # covered bases, each match in <= 2 windows
        assert int(cov[0] + cov[1]) == int(cov[2] + cov[3])
    return {"contigs_checked": n, "matches_checked": int(summ[:, 1].sum()),
            "properties": "per-contig match/canonical/forward counts vs independent torch k-mer lookup; "
                          "A/C/G/T totals vs the sums of the windows that tile each contig"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--verify", action="store_true", help="untimed full-size parity properties (torch)")
    ap.add_argument("--blocks", action="store_true", help="also time device block calling (untimed in value)")
    ap.add_argument("--e2e", action="store_true",
                    help="also time the host-buffer C-ABI calls on the same workload (PCIe-inclusive; never in value)")
    ap.add_argument("--flags", default=FLAGS, help="Teloscope flags of the workload (default: configs[1])")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gbases", type=float, default=3.0, help="bases per GPU (Gb); 3.0 = BASELINE config")
    ap.add_argument("--contigs", type=int, default=200)
    ap.add_argument("--cpu-sample-mb", type=float, default=384.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("TS_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse N > 1 on one GPU
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # a fresh checkout has no built artefacts: rank 0 builds them (hipcc, gcc), the others wait; nothing here is
    # a fallback — without the HIP library the import below raises
    if not (os.path.exists(os.path.join(ROOT, "teloscope_amd", "libteloscan.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "libteloscope_oracle.so"))):
        if rank == 0:
            import __graft_entry__ as entry
            entry.build()
    if world > 1:
        dist.barrier()
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from tests import harness as H
    from tests.backends import _user_input

    opts = H.parse_cli("x.fa " + args.flags)
    ui = _user_input(opts)
    ui.device = dev_index
    tel = ta.Teloscope(ui)
    L = K.lib()

    total = int(args.gbases * 1e9)
    lens = contig_lengths(total, args.contigs, 42 + rank)
    n = len(lens)
    lens_c = (C.c_uint64 * n)(*lens)
    batch = L.ts_batch_create(tel._ctx.ptr, lens_c, None, n, 0, 0)
    if not batch:
        raise RuntimeError(tel._ctx.error())
    info = K.BatchInfo()
    L.ts_batch_get_info(batch, C.byref(info))
    offsets = [int(L.ts_batch_segment_offset(batch, i)) for i in range(n)]

    buf = torch.zeros(int(info.input_bytes), dtype=torch.uint8, device=dev)
    fill_synthetic(buf, offsets, lens, 42 + rank, dev)
    # the per-segment hit summaries of a step are gathered while the next step scans: two buffers in turn,
    # the gather of step i is waited for before its buffer is written again (and at the end of the timed region)
    xdev = dev if backend == "nccl" else torch.device("cpu")
    summaries = [torch.zeros(n * 4, dtype=torch.int64, device=dev) for _ in range(2)]
    gathered = [[torch.zeros(n * 4, dtype=torch.int64, device=xdev) for _ in range(world)]
                if (world > 1 and rank == 0) else None for _ in range(2)]
    pending = [None, None]
    step_no = [0]
    gather_mode = ["async"]
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    dptr = C.c_void_p(buf.data_ptr())

    def step():
        rc = L.ts_batch_scan(batch, dptr, sptr)
        if rc != 0:
            raise RuntimeError(tel._ctx.error())
        if world > 1:
            j = step_no[0] & 1
            step_no[0] += 1
            if pending[j] is not None:
                pending[j].wait()
            rc = L.ts_batch_segment_summary(batch, C.c_void_p(summaries[j].data_ptr()), sptr)
            if rc != 0:
                raise RuntimeError(tel._ctx.error())
            src = summaries[j] if backend == "nccl" else summaries[j].cpu()
            if gather_mode[0] == "async":
                try:
                    pending[j] = dist.gather(src, gathered[j], dst=0, async_op=True)
                    return
                except (RuntimeError, NotImplementedError, TypeError):   # a backend without an asynchronous gather
                    gather_mode[0] = "sync"
            pending[j] = None
            dist.gather(src, gathered[j], dst=0)

    def drain_gathers():
        for j in range(2):
            if pending[j] is not None:
                pending[j].wait()
                pending[j] = None

    for _ in range(args.warmup):
        step()
    drain_gathers()
    if L.ts_batch_sync(batch) != 0:          # also grows the match buffer if it overflowed
        raise RuntimeError(tel._ctx.error())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    drain_gathers()
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps

    if L.ts_batch_sync(batch) != 0:
        raise RuntimeError(tel._ctx.error())
    L.ts_batch_get_info(batch, C.byref(info))

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())

    if rank == 0:
        ms_per_step = tmax / args.steps * 1e3
        value = world * total / (tmax / args.steps) / 1e9
        alg_bytes = int(info.algorithmic_bytes)
        # HIP events around every timed launch of the scan kernel, on its stream, averaged
        kern_ms = float(info.avg_kernel_ms)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
        if args.gbases == 3.0 and args.contigs == 200 and args.flags == FLAGS and os.path.exists(tpath):
            # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
        out = {
            "metric": "Gbases/s scanned (whole node), 3 Gb FASTA TTAGGG w=1000 s=500",
            "value": round(value, 3), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: synthetic %.2f Gb / %d contigs per GPU, %s, %d patterns k=%d"
                                   % ("configs[1]" if args.flags == FLAGS else "custom", args.gbases, n, args.flags,
                                      len(ui.patternInfo), len(ui.patternInfo[0][0])),
                       "bases_per_gpu": total, "windows": int(info.n_windows),
                       "matches": int(info.n_matches), "tiles": int(info.n_tiles),
                       "timed_region": "resident ASCII in HBM -> window records + packed match stream in HBM"
                                       + (" + RCCL gather of per-segment hit summaries" if world > 1 else ""),
                       "device_ms_per_step_events": round(dev_ms, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "ts_scan_tiles", "kernel_ms": round(kern_ms, 4), "launches_timed": int(info.kernel_launches),
                         "algorithmic_bytes": alg_bytes},
        }
        if not args.no_cpu_baseline and world == 1:           # the host-core baseline is timed at N = 1 only
            # Bounded sample of the same workload on the host cores through the oracle port, with the
            # reference's own parallel decomposition: one job per path (src/input.cpp:719-724), here
            # one contig slice per thread (ctypes releases the GIL inside the C call).
            from concurrent.futures import ThreadPoolExecutor
            from tests.backends import OracleBackend
            cores = max(1, min(16, os.cpu_count() or 1, n))
            per_job = int(args.cpu_sample_mb * 1e6 / 4)                 # 96 Mb per job by default
            order = sorted(range(n), key=lambda i: -lens[i])[:cores]
            jobs = []
            for ci in order:
                nb = int(min(per_job, lens[ci]))
                jobs.append(buf[offsets[ci]:offsets[ci] + nb].cpu().numpy().tobytes().upper())
            backends = [OracleBackend(opts) for _ in jobs]

            def run(i):
                return backends[i].oracle.bench_scan(jobs[i])

            c0 = time.perf_counter()
            r1 = run(0)                                                 # one core, one job
            t1 = time.perf_counter() - c0
            c0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                res = list(ex.map(run, range(len(jobs))))
            tn = time.perf_counter() - c0
            took = sum(len(j) for j in jobs)
            out["cpu_baseline"] = {
                "value": round(took / tn / 1e9, 5), "unit": "Gbases/s", "cores": cores, "kind": "port",
                "single_core_value": round(len(jobs[0]) / t1 / 1e9, 5),
                "sample": "first %.0f Mb of each of the %d largest contigs of the same synthetic assembly "
                          "(%.0f Mb), same flags, scan stage only incl. block calling "
                          "(oracle/teloscope_oracle.c: trie walk + carry loop), one job per contig as the "
                          "reference's -j N does; %d windows, %d matches, %.1f s wall (%.1f s for one job on "
                          "one core)" % (per_job / 1e6, len(jobs), took / 1e6, sum(r[1] for r in res),
                                         sum(r[2] for r in res), tn, t1)}
            assert r1[0] == res[0][0]
        if args.verify:
            out["verify"] = verify_full_size(L, batch, tel, buf, offsets, lens, ui, dev)
        if args.blocks:
            seg_out = (K.SegmentOut * n)()
            b0 = time.perf_counter()
            rc = L.ts_batch_download_blocks(batch, seg_out)
            bdt = time.perf_counter() - b0
            if rc != 0:
                raise RuntimeError(tel._ctx.error())
            out["device_block_calling"] = {
                "wall_ms_incl_alloc_d2h_windows": round(bdt * 1e3, 2),
                "terminal_blocks": int(sum(seg_out[i].n_terminal_blocks for i in range(n))),
                "interstitial_blocks": int(sum(seg_out[i].n_interstitial_blocks for i in range(n)))}
            L.ts_free_segments(seg_out, n)
        if args.e2e:
            # PCIe-inclusive: ASCII in host memory in, results in host memory out, through the drop-in entry points
            # (upload through the pinned ring, scan, block calling, D2H, host post-processing) — SURVEY 8d's second
            # figure; reported beside `value`, never in it.
            host = buf.cpu().numpy()
            segs = (K.SegmentIn * n)()
            base = host.ctypes.data
            for i in range(n):
                segs[i].seq = C.cast(C.c_void_p(base + offsets[i]), C.c_char_p)
                segs[i].len = lens[i]
                segs[i].abs_pos = 0
                segs[i].tips_only = 0
            e2e = {}
            for name, with_matches in (("blocks_windows_counts", False), ("with_match_vectors", True)):
                res = (K.SegmentOut * n)()
                cnts = (K.SegmentCounts * n)()
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    rc = (L.ts_scan_segments(tel._ctx.ptr, segs, n, res) if with_matches
                          else L.ts_scan_segments_blocks(tel._ctx.ptr, segs, n, res, cnts))
                    dt = time.perf_counter() - t0
                    if rc != 0:
                        raise RuntimeError(tel._ctx.error())
                    nm = int(sum(res[i].n_matches for i in range(n))) if with_matches else int(sum(c.n_matches for c in cnts))
                    L.ts_free_segments(res, n)
                    best = dt if best is None else min(best, dt)
                e2e[name] = {"seconds": round(best, 4), "gbases_per_s": round(total / best / 1e9, 3), "matches": nm}
            out["pcie_inclusive"] = {"entry_points": "ts_scan_segments_blocks / ts_scan_segments (pageable host buffers in, "
                                                     "host results out, best of 2)", **e2e}
        print(json.dumps(out))
    L.ts_batch_destroy(batch)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
