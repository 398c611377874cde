"""benchlib.reads — the read filter of bench.py (configs[3]): synthetic HiFi-like reads, the `reads` sub-record, --reads."""
from . import common
from .common import *  # noqa: F401,F403  (the standard modules bench.py always imported, and its constants)


def read_lengths(n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    return np.clip(rng.normal(15000, 3000, size=n), 1000, 40000).astype(np.int64)

def fill_read_range(buf, all_lens, g0, g1, dev):
    """configs[3]'s synthetic HiFi reads [g0, g1) of the global read set into `buf`, back to back in the batch layout
    (every read at a 16-byte boundary): uniform ACGT, 0.5 % of the reads carry a 300-8000 b terminal TTAGGG / CCCTAA tract
    with 1 % substitutions.  Generated per chunk of READ_CHUNK reads (seeded by the chunk's index), so that the same read
    has the same bases whichever rank, sub-batch or read count it is generated for.  Returns the indices (relative to
    g0) of the reads that carry a tract."""
    import numpy as np
    import torch
    pad = (all_lens + 15) & ~15
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    carriers, at = [], 0
    for c in range(g0 // READ_CHUNK, (g1 - 1) // READ_CHUNK + 1):
        c0, c1 = c * READ_CHUNK, min((c + 1) * READ_CHUNK, len(all_lens))
        cl = all_lens[c0:c1]
        coffs = np.concatenate(([0], np.cumsum(pad[c0:c1])))
        g = torch.Generator(device=dev)
        g.manual_seed(43 + c)
        tmp = torch.empty(int(coffs[-1]), dtype=torch.uint8, device=dev)
        step = 1 << 28
        for a in range(0, tmp.numel(), step):
            b = min(tmp.numel(), a + step)
            tmp[a:b] = lut[torch.randint(0, 4, (b - a,), dtype=torch.uint8, device=dev, generator=g).long()]
        rng = np.random.default_rng(1043 + c)
        a, z = max(g0, c0) - c0, min(g1, c1) - c0
        for i in np.flatnonzero(rng.random(len(cl)) < 0.005):
            ln = int(min(rng.integers(300, 8001), cl[i]))
            unit = b"TTAGGG" if rng.random() < 0.5 else b"CCCTAA"
            t = np.tile(np.frombuffer(unit, dtype=np.uint8), ln // 6 + 1)[:ln].copy()
            k = rng.random(ln) < 0.01
            t[k] = acgt[rng.integers(0, 4, size=int(k.sum()))]
            if a <= i < z:
                p = int(coffs[i]) if unit == b"CCCTAA" else int(coffs[i]) + int(cl[i]) - ln
                tmp[p:p + ln] = torch.from_numpy(t).to(dev)
                carriers.append(int(i) + c0 - g0)
        nbytes = int(coffs[z] - coffs[a])
        buf[at:at + nbytes] = tmp[int(coffs[a]):int(coffs[z])]
        at += nbytes
        del tmp
    return np.asarray(carriers, dtype=np.int64)

def reads_sub_record(args, dev):
    """configs[3] in small inside the default line: 500 k synthetic HiFi reads (7.5 Gb) resident in HBM, whole-read tips
    scan + terminal-block predicate on the device, one pass byte per read.  Roofline by SURVEY 8(d): 1 B per base + 1 bit
    per read over the WHOLE step (scan and predicate; the match stream is an intermediate).  A sample is checked against
    the oracle's ReadTelomereFilter::matches."""
    import numpy as np
    import torch
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    from tests.backends import OracleReadFilter
    opts = parse_cli(READ_FLAGS)
    ui = user_input(opts, device=dev.index)
    rf = ta.ReadTelomereFilter(ui)
    L = K.lib()
    n = 500_000
    lens = read_lengths(n, 43)
    bases = int(lens.sum())
    arr = (C.c_uint64 * n)(*[int(x) for x in lens])
    # two batch objects over the same reads: the predicate of step i (few registers, no LDS) runs on a stream of its own
    # beside the scan of step i + 1, as bench.py --reads does with its sub-batches
    nslots = 2
    batches = []
    for _ in range(nslots):
        b = L.ts_batch_create(rf._ctx.ptr, arr, None, n, 1, bases // 8 + 4096)
        if not b:
            raise RuntimeError(rf._ctx.error())
        if os.environ.get("TS_REC32") != "1":
            L.ts_batch_set_record_bits(b, 16)                          # (the predicate is the records' only reader: ts_filter_reads does the same)
        batches.append(b)
    info = K.BatchInfo()
    L.ts_batch_get_info(batches[0], C.byref(info))
    offs = np.concatenate(([0], np.cumsum((lens + 15) & ~15)))[:-1]
    buf = torch.zeros(int(info.input_bytes), dtype=torch.uint8, device=dev)
    carriers = fill_read_range(buf, lens, 0, n, dev)
    d_passes = [torch.zeros(n + 16, dtype=torch.uint8, device=dev) for _ in range(nslots)]
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    import teloscope_amd.distributed as D
    pred_stream = D.concurrent_streams(rf, dev, 2, first=stream)[1]     # (not on the scan stream's hardware queue)
    pptr = C.c_void_p(pred_stream.cuda_stream)
    judged = [torch.cuda.Event() for _ in range(nslots)]
    used = [False] * nslots

    def step(i):
        j = i % nslots
        if used[j]:
            stream.wait_event(judged[j])                           # the predicate that last read this slot's records
        if L.ts_batch_scan(batches[j], C.c_void_p(buf.data_ptr()), sptr) != 0:
            raise RuntimeError(rf._ctx.error())
        if L.ts_batch_wait_scan(batches[j], pptr) != 0:            # (the library's own event behind the scan)
            raise RuntimeError(rf._ctx.error())
        if L.ts_batch_read_pass(batches[j], C.c_void_p(d_passes[j].data_ptr()), pptr) != 0:
            raise RuntimeError(rf._ctx.error())
        judged[j].record(pred_stream)
        used[j] = True

    def drain():
        stream.wait_stream(pred_stream)
        torch.cuda.synchronize()

    def overflowed():
        any_flag = False
        for b in batches:
            flag = C.c_int(0)
            if L.ts_batch_read_pass_status(b, C.byref(flag)) != 0:
                raise RuntimeError(rf._ctx.error())
            any_flag = any_flag or bool(flag.value)
        return any_flag

    steps = 6
    sec = None
    for attempt in range(3):
        for i in range(nslots):
            step(i)
        drain()
        for b in batches:
            if L.ts_batch_sync(b) != 0:                            # (grows the record regions and rescans if the scan overflowed)
                raise RuntimeError(rf._ctx.error())
        overflowed()
        for i in range(nslots):
            step(i)
        drain()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        drain()
        sec = (time.perf_counter() - t0) / steps
        if not overflowed():
            break
    else:
        raise RuntimeError("the read batch kept overflowing its record regions")
    for b in batches:
        if L.ts_batch_sync(b) != 0:
            raise RuntimeError(rf._ctx.error())
    L.ts_batch_get_info(batches[0], C.byref(info))
    d_pass = d_passes[(steps - 1) % nslots]
    assert all(bool(torch.equal(d_passes[0][:n], x[:n])) for x in d_passes[1:]), "the slots' pass bytes differ"
    got = d_pass[:n].cpu().numpy()
    assert got[carriers].all(), "a read with a planted terminal telomere tract was not kept"
    sample = sorted(set(range(150)) | set(int(i) for i in carriers[:100]))
    host = {i: bytes(buf[int(offs[i]):int(offs[i]) + int(lens[i])].cpu().numpy()) for i in sample}
    want = OracleReadFilter(opts).filter([host[i] for i in sample])
    assert [bool(got[i]) for i in sample] == want, "oracle and HIP read filter disagree on the sample"
    alg = bases + (n + 7) // 8
    out = {"workload": "configs[3] at %d reads (%s; lengths N(15000, 3000^2) clipped to [1000, 40000], 0.5 %% with a terminal tract), "
                       "resident in HBM -> one pass byte per read in HBM" % (n, READ_FLAGS),
           "reads": n, "bases": bases, "steps": steps, "ms_per_step": round(sec * 1e3, 4), "gbases_per_s": round(bases / sec / 1e9, 3),
           "reads_per_s": round(n / sec, 1), "kept": int(got.sum()), "planted_carriers": int(len(carriers)),
           "oracle_checked_reads": len(sample),
           "roofline": {"bound": "hbm", "achieved": round(alg / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": alg,
                        "over": "the whole step: tips scan (%.3f ms alone, HIP events) + predicate; the predicate of step i runs on a "
                                "second stream beside the scan of step i + 1 (two batch objects)" % float(info.avg_kernel_ms)}}
    for b in batches:
        L.ts_batch_destroy(b)
    del buf
    torch.cuda.empty_cache()
    return out


def run_reads(args, rank, local_rank, world, dev, backend):
    """configs[3]: --fastq-subset -l 42 on synthetic HiFi reads (~15 kb), the reads dealt to the ranks in consecutive
    shards of equal count (the reference deals a batch's records to its workers in chunks and writes the chunk
    outputs in chunk order, src/input.cpp:753-812); the only exchange is one gather of the pass bytes, in input
    order, to rank 0.  `value` = resident rate (reads in HBM -> pass bytes in HBM [-> rank 0]); the streaming
    PCIe-inclusive rate through ts_filter_reads is reported beside it."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input

    opts = parse_cli(READ_FLAGS)
    ui = user_input(opts, device=dev.index)
    rf = ta.ReadTelomereFilter(ui)
    L = K.lib()
    n_total = int(args.n_reads)
    lo, hi = rank * n_total // world, (rank + 1) * n_total // world       # this rank's shard of the reads
    all_lens = read_lengths(n_total, 43)
    lens = all_lens[lo:hi]
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    xdev = dev if backend == "nccl" else torch.device("cpu")

    # resident sub-batches of <= 500 k reads (~7.5 Gb each): input, match stream and pass bytes stay in HBM
    sub = int(os.environ.get("TS_BENCH_READ_SUB", "500000"))     # (reads per resident sub-batch)
    batches = []
    for a in range(0, len(lens), sub):
        sl = lens[a:a + sub]
        n = len(sl)
        arr = (C.c_uint64 * n)(*[int(x) for x in sl])
        b = L.ts_batch_create(rf._ctx.ptr, arr, None, n, 1, int(sl.sum()) // 8 + 4096)
        if not b:
            raise RuntimeError(rf._ctx.error())
        if os.environ.get("TS_REC32") != "1":
            L.ts_batch_set_record_bits(b, 16)                          # (the predicate is the records' only reader: ts_filter_reads does the same)
        info = K.BatchInfo()
        L.ts_batch_get_info(b, C.byref(info))
        offs = np.concatenate(([0], np.cumsum((sl + 15) & ~15)))[:-1]
        buf = torch.zeros(int(info.input_bytes), dtype=torch.uint8, device=dev)
        carriers = fill_read_range(buf, all_lens, lo + a, lo + a + n, dev)
        batches.append(dict(b=b, n=n, buf=buf, lens=sl, offs=offs, carriers=carriers,
                            d_pass=torch.zeros(n + 16, dtype=torch.uint8, device=dev)))
    my_bases = int(lens.sum())
    total_bases = int(all_lens.sum())
    max_n = max((rank_hi - rank_lo) for rank_lo, rank_hi in ((r * n_total // world, (r + 1) * n_total // world) for r in range(world)))
    pass_local = torch.zeros(max_n, dtype=torch.uint8, device=dev)
    gathered = [torch.zeros(max_n, dtype=torch.uint8, device=xdev) for _ in range(world)] if (world > 1 and rank == 0) else None

    # Two streams: the predicate of sub-batch i (no LDS, 64 VGPRs) runs beside the tips scan of sub-batch i + 1 (a persistent
    # kernel that leaves wave slots and a fifth of the issue cycles free), as it does between the stages of ts_filter_reads.
    import teloscope_amd.distributed as D
    pred_stream = D.concurrent_streams(rf, dev, 2, first=stream)[1]     # (not on the scan stream's hardware queue)
    pptr = C.c_void_p(pred_stream.cuda_stream)
    for e in batches:
        e["judged"] = torch.cuda.Event()
    overlap = [True]

    def step(_i=0):
        at = 0
        for e in batches:
            if overlap[0]:
                stream.wait_event(e["judged"])                   # the last predicate over this sub-batch's records is done
            if L.ts_batch_scan(e["b"], C.c_void_p(e["buf"].data_ptr()), sptr) != 0:
                raise RuntimeError(rf._ctx.error())
            if overlap[0] and L.ts_batch_wait_scan(e["b"], pptr) != 0:          # (the library's own event behind the scan)
                raise RuntimeError(rf._ctx.error())
            if L.ts_batch_read_pass(e["b"], C.c_void_p(e["d_pass"].data_ptr()), pptr if overlap[0] else sptr) != 0:
                raise RuntimeError(rf._ctx.error())
            with torch.cuda.stream(pred_stream if overlap[0] else stream):
                pass_local[at:at + e["n"]] = e["d_pass"][:e["n"]]
            if overlap[0]:
                e["judged"].record(pred_stream)
            at += e["n"]
        # No join of the two streams at the end of a pass: the next pass's first scans run beside this pass's last predicates,
        # as the groups of ts_filter_reads do (a sub-batch's scan only waits for the predicate over its own records: "judged").
        if world > 1:                                            # the 1-byte-per-read gather, in input order, behind the predicates
            with torch.cuda.stream(pred_stream if overlap[0] else stream):
                dist.gather(pass_local if backend == "nccl" else pass_local.cpu(), gathered, dst=0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def any_overflow():
        """Did a predicate since the last call find its scan overflowed (ts_batch_read_pass_status)?  Agreed over the ranks."""
        over = 0
        for e in batches:
            flag = C.c_int(0)
            if L.ts_batch_read_pass_status(e["b"], C.byref(flag)) != 0:
                raise RuntimeError(rf._ctx.error())
            over |= flag.value
        t = torch.tensor([over], dtype=torch.int32, device=xdev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    tmax = None
    for attempt in range(3):
        for _ in range(max(1, args.warmup)):
            step()
        for e in batches:                                        # (a sync grows a match buffer that overflowed, and rescans)
            if L.ts_batch_sync(e["b"]) != 0:
                raise RuntimeError(rf._ctx.error())
        step()
        barrier()
        any_overflow()                                           # (what the warm-up raised is dealt with: the syncs regrew)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
        # tiles are taken on demand, so a wave's record count differs from pass to pass: a timed pass whose scan overflowed a
        # region judged nothing — the measurement is repeated after the regions have been regrown
        if not any_overflow():
            break
    else:
        raise RuntimeError("the read batches kept overflowing their record regions")

    # the tips kernel's own time, for the roofline: one more pass with nothing beside it (the syncs harvest the event times
    # of the scans since the sync before: first those of the timed, overlapped loop, then this pass's)
    for e in batches:
        if L.ts_batch_sync(e["b"]) != 0:
            raise RuntimeError(rf._ctx.error())
    overlap[0] = False
    torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()
    overlap[0] = True
    kern_ms = alg = nm = launches = 0
    for e in batches:
        if L.ts_batch_sync(e["b"]) != 0:
            raise RuntimeError(rf._ctx.error())
        info = K.BatchInfo()
        L.ts_batch_get_info(e["b"], C.byref(info))
        kern_ms += float(info.avg_kernel_ms)
        launches += int(info.kernel_launches)
        alg += int(info.algorithmic_bytes)
        nm += int(info.n_matches)
    kept_local = int(pass_local[:len(lens)].sum().item())
    n_carriers = int(sum(len(e["carriers"]) for e in batches))
    # every read that carries a planted terminal tract of >= 300 b must pass; (a random 15 kb read passes with negligible probability)
    at = 0
    for e in batches:
        got = e["d_pass"][:e["n"]].cpu().numpy()
        assert got[e["carriers"]].all(), "a read with a planted terminal telomere tract was not kept"
        at += e["n"]

    assert not any_overflow(), "the verification pass overflowed its record regions"
    if rank == 0:
        shard_sizes = [(r + 1) * n_total // world - r * n_total // world for r in range(world)]
        all_pass = pass_local[:len(lens)].cpu().numpy() if world == 1 else \
            np.concatenate([gathered[r][:shard_sizes[r]].cpu().numpy() for r in range(world)])
        kept = int(all_pass.sum())
        sec = tmax / args.steps
        achieved = alg / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "Gbases/s filtered (whole node), --fastq-subset -l 42 on synthetic ~15 kb HiFi reads",
            "value": round(total_bases / sec / 1e9, 3), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[3] at %d reads: %s, lengths N(15000, 3000^2) clipped to [1000, 40000], seed 43, 0.5 %% of the reads with a "
                                   "300-8000 b terminal tract; %d patterns k=%d" % (n_total, READ_FLAGS, len(ui.patternInfo), len(ui.patternInfo[0][0])),
                       "reads": n_total, "reads_per_s": round(n_total / sec, 1), "bases": total_bases, "kept": kept,
                       "planted_carriers_rank0": n_carriers, "matches_rank0": nm,
                       "timed_region": "reads resident in HBM -> whole-read tips scan + terminal-block predicate on the device -> one pass byte "
                                       "per read in HBM" + (" -> one gather of the pass bytes to rank 0" if world > 1 else "")
                                       + "; the predicate of sub-batch i runs beside the scan of sub-batch i + 1 (two streams); roofline.kernel_ms is "
                                         "the tips kernel alone, from a pass without that overlap"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "ts_scan_tiles (tips mode, rank 0's reads)", "kernel_ms": round(kern_ms, 4), "launches_timed": launches,
                         "algorithmic_bytes": alg},
        }
        import hashlib
        out["config"]["pass_bytes_sha1"] = hashlib.sha1(all_pass.tobytes()).hexdigest()     # (the same for any number of ranks)
        if args.verify:
            # a sample over the WHOLE read set — the first reads and the last ones, wherever they were filtered — against
            # the oracle's ReadTelomereFilter::matches (the last chunk is generated again here)
            from tests.backends import OracleReadFilter
            orf = OracleReadFilter(opts)
            checked = 0
            for g0, g1 in ((0, min(n_total, 3000)), (max(0, n_total - 3000), n_total)):
                pad = (all_lens[g0:g1] + 15) & ~15
                o = np.concatenate(([0], np.cumsum(pad)))
                tmpb = torch.zeros(int(o[-1]) + 64, dtype=torch.uint8, device=dev)
                car = fill_read_range(tmpb, all_lens, g0, g1, dev)
                pick = sorted(set(range(0, g1 - g0, 40)) | set(int(i) for i in car))
                hostb = tmpb.cpu().numpy()
                seqs = [bytes(hostb[int(o[i]):int(o[i]) + int(all_lens[g0 + i])]) for i in pick]
                assert [bool(all_pass[g0 + i]) for i in pick] == orf.filter(seqs), "oracle and the gathered pass bytes disagree"
                checked += len(pick)
            out["verify"] = {"reads_checked_against_oracle": checked, "where": "the first and the last 3000 reads of the set (every 40th + every carrier)"}
        # streaming, PCIe-inclusive: host reads through ts_filter_reads (groups pipelined through upload / scan / predicate);
        # a pool of 200 k host reads is cycled, so that any read count streams through bounded host memory
        e = batches[0]
        npool = min(200_000, e["n"])
        host = e["buf"][:int(e["offs"][npool - 1] + e["lens"][npool - 1])].cpu().numpy()
        ptrs = (C.c_char_p * npool)()
        base = host.ctypes.data
        for i in range(npool):
            ptrs[i] = C.cast(C.c_void_p(base + int(e["offs"][i])), C.c_char_p)
        hl = (C.c_uint64 * npool)(*[int(x) for x in e["lens"][:npool]])
        hp = (C.c_uint8 * npool)()
        pool_bases = int(e["lens"][:npool].sum())
        rounds = max(1, min(25, int(round(n_total / world / npool))))
        assert L.ts_filter_reads(rf._ctx.ptr, ptrs, hl, npool, hp) == 0, rf._ctx.error()        # warm: pool, pinned rings
        assert bytes(hp) == bytes(e["d_pass"][:npool].cpu().numpy().tobytes()), "streaming and resident filters disagree"
        c0 = time.perf_counter()
        for _ in range(rounds):
            if L.ts_filter_reads(rf._ctx.ptr, ptrs, hl, npool, hp) != 0:
                raise RuntimeError(rf._ctx.error())
        dt = time.perf_counter() - c0
        out["pcie_inclusive"] = {"host_placement": ("process and library threads on NUMA node %d, the GPU's" % common.HOST_NUMA_NODE
                                                    if common.HOST_NUMA_NODE is not None else "not bound to a NUMA node"),
                                 "entry_point": "ts_filter_reads (pageable host reads in, pass bytes out; groups of ~256 MB pipelined)",
                                 "reads": rounds * npool, "seconds": round(dt, 4), "reads_per_s": round(rounds * npool / dt, 1),
                                 "gbases_per_s": round(rounds * pool_bases / dt / 1e9, 3),
                                 "note": "a pool of %d host reads filtered %d times on one GPU" % (npool, rounds)}
        if world == 1 and not args.no_cpu_baseline:
            from concurrent.futures import ThreadPoolExecutor
            from tests.backends import OracleReadFilter
            cores = max(1, os.cpu_count() or 1)
            per = 600                                             # reads per core: ~9 Mb each, ~10-30 s of CPU work in all
            take = min(npool, per * cores)
            seqs = [bytes(host[int(e["offs"][i]):int(e["offs"][i]) + int(e["lens"][i])]) for i in range(take)]
            filt = [OracleReadFilter(opts) for _ in range(cores)]
            share = -(-take // cores)

            def job(ci):
                return filt[ci].filter(seqs[ci * share:(ci + 1) * share])
            c0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                res = [x for part in ex.map(job, range(cores)) for x in part]
            tn = time.perf_counter() - c0
            assert res == [bool(x) for x in hp[:take]], "oracle and HIP read filter disagree on the sample"
            sb = sum(len(x) for x in seqs)
            out["cpu_baseline"] = {"value": round(sb / tn / 1e9, 5), "unit": "Gbases/s", "cores": cores, "kind": "port",
                                   "calibrated_against_reference": False,
                                   "sample": "%d of the same reads (%.0f Mb), ReadTelomereFilter::matches through the oracle port, one chunk of "
                                             "reads per thread as the reference's -j N does; %.1f s wall; identical pass bits"
                                             % (take, sb / 1e6, tn)}
        print(json.dumps(out), flush=True)
    for e in batches:
        L.ts_batch_destroy(e["b"])
    barrier()
