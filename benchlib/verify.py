"""benchlib.verify — the untimed full-size parity checks of bench.py (--verify)."""
from . import common
from .common import *  # noqa: F401,F403  (the standard modules bench.py always imported, and its constants)


def verify_full_size(L, batch, tel, buf, offsets, lens, ui, dev):
    """Size-independent parity properties at the full bench size, with an INDEPENDENT computation
    in torch (rolling 2-bit k-mer code + table lookup, nothing shared with the HIP kernels):
      * per contig: matches / canonical / forward counts == ts_batch_segment_summary;
      * per contig: A,C,G,T totals == sum of the nucleotide counts of the windows that tile the contig
        (w = 2s: the even-indexed ones; w = s: all of them);
      * w = s: matches straddling a window end are excluded, as the reference loses them.
    `batch` holds the results (a scanned batch, or one that adopted the ranks' shards).
    Returns a dict for the bench line; raises on any mismatch."""
    import torch
    from teloscope_amd import _capi as K
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
    info = K.BatchInfo()
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
    # ---- per WINDOW, on a seeded sample: every field of the record recomputed from the bases with the same torch k-mer
    # lookup — A/C/G/T over the window, and k x the matches that lie fully inside it, by kind (the closed form of
    # analyzeWindow's carry loop; with w == s this is also the straddle rule).  Nothing of the oracle or of the
    # HIP kernels is involved: the per-contig sums above cannot see a count that moved from one window to its neighbour.
    import numpy as np
    n_sample = int(os.environ.get("TS_VERIFY_WINDOWS", "10000"))
    rng = np.random.default_rng(1234)
    nwins = np.array([-(-nb // step) for nb in lens], dtype=np.int64)
    wstart = np.concatenate([[0], np.cumsum(nwins)])
    total_w = int(wstart[-1])
    pick = np.unique(rng.integers(0, total_w, size=min(n_sample, total_w)))
    # every contig's last window (the short tail) and first window are always in
    pick = np.unique(np.concatenate([pick, wstart[:-1][nwins > 0], (wstart[1:] - 1)[nwins > 0]]))
    ci_of = np.searchsorted(wstart, pick, side="right") - 1
    widx = pick - wstart[ci_of]
    starts = widx * step
    sizes = np.minimum(window, np.array(lens, dtype=np.int64)[ci_of] - starts)
    base_off = np.array(offsets, dtype=np.int64)[ci_of] + starts
    span = window + k - 1
    checked = 0
    for a in range(0, len(pick), 2048):
        z = min(len(pick), a + 2048)
        bo = torch.as_tensor(base_off[a:z], device=dev).view(-1, 1)
        sz = torch.as_tensor(sizes[a:z], device=dev).view(-1, 1)
        col = torch.arange(span, device=dev).view(1, -1)
        idx = torch.minimum(bo + col, torch.tensor(buf.numel() - 1, device=dev))
        c = lut[buf[idx].long()]                                 # [m, span] codes, 4 = not A/C/G/T
        inside = col < sz
        want = torch.zeros(z - a, 8, dtype=torch.int64, device=dev)
        for code_v, field in ((0, 0), (1, 1), (3, 2), (2, 3)):   # records are A C G T; codes A0 C1 T2 G3
            want[:, field] = ((c == code_v) & inside).sum(dim=1)
        code = torch.zeros(z - a, window, dtype=torch.int64, device=dev)
        bad = torch.zeros(z - a, window, dtype=torch.bool, device=dev)
        for i in range(k):
            ci_ = c[:, i:i + window]
            code += (ci_ & 3).long() << (2 * i)
            bad |= ci_ == 4
        fully = (col[:, :window] + k) <= sz                      # the match ends inside the window
        hit = tbl[0][code] & ~bad & fully
        is_fwd, is_can = tbl[1][code] & hit, tbl[2][code] & hit
        want[:, 4] = k * is_can.sum(dim=1)
        want[:, 5] = k * (hit & ~is_can).sum(dim=1)
        want[:, 6] = k * is_fwd.sum(dim=1)
        want[:, 7] = k * (hit & ~is_fwd).sum(dim=1)
        got = wins[torch.as_tensor(pick[a:z], device=dev)].long()
        if not torch.equal(got, want):
            bad_row = int((got != want).any(dim=1).nonzero()[0])
            raise AssertionError("window record differs from the independent recomputation: contig %d window %d: got %s want %s"
                                 % (int(ci_of[a + bad_row]), int(widx[a + bad_row]), got[bad_row].tolist(), want[bad_row].tolist()))
        checked += z - a
    return {"contigs_checked": n, "matches_checked": int(summ[:, 1].sum()), "windows_checked_field_by_field": checked,
            "properties": "per-contig match/canonical/forward counts vs independent torch k-mer lookup; "
                          "A/C/G/T totals vs the sums of the windows that tile each contig; all eight fields of %d sampled "
                          "window records (every contig's first and last window among them) vs the same lookup" % checked}


def compare_sharded_with_single_gpu(L, K, tel, batch, sharded, n, with_matches):
    """The merged shard results (ts_shards_finalize on rank 0) against the downloads of a single-GPU scan of the whole
    assembly: window records and blocks byte for byte, the per-segment counts, and — when the assembly is small enough
    to bring every match record to the host — the visible match records.  Raises on any difference."""
    import numpy as np
    seg_out, seg_cnt = sharded["seg_out"], sharded["seg_cnt"]
    ref = (K.SegmentOut * n)()
    rc = L.ts_batch_download(batch, None, ref) if with_matches else L.ts_batch_download_blocks(batch, ref)
    if rc != 0:
        raise RuntimeError(tel._ctx.error())

    def raw(ptr, count, dt):
        return np.frombuffer(C.string_at(C.cast(ptr, C.c_void_p), int(count) * dt.itemsize), dtype=np.uint8) if count else np.zeros(0, np.uint8)

    nwin = nblk = nvis = 0
    for i in range(n):
        g, e = seg_out[i], ref[i]
        assert g.n_windows == e.n_windows and np.array_equal(raw(g.windows, g.n_windows, K.WINDOW_DT), raw(e.windows, e.n_windows, K.WINDOW_DT)), ("windows", i)
        assert g.n_terminal_blocks == e.n_terminal_blocks and g.n_interstitial_blocks == e.n_interstitial_blocks, ("block counts", i)
        assert np.array_equal(raw(g.terminal_blocks, g.n_terminal_blocks, K.BLOCK_DT), raw(e.terminal_blocks, e.n_terminal_blocks, K.BLOCK_DT)), ("terminal blocks", i)
        assert np.array_equal(raw(g.interstitial_blocks, g.n_interstitial_blocks, K.BLOCK_DT), raw(e.interstitial_blocks, e.n_interstitial_blocks, K.BLOCK_DT)), ("interstitial blocks", i)
        nwin += int(g.n_windows)
        nblk += int(g.n_terminal_blocks + g.n_interstitial_blocks)
        if with_matches:
            em = np.frombuffer(raw(e.matches, e.n_matches, K.MATCH_DT), dtype=K.MATCH_DT) if e.n_matches else np.zeros(0, K.MATCH_DT)
            vis = em[(em["flags"] & (K.MATCH_CANONICAL | K.MATCH_TERMINAL)) != 0]
            gm = np.frombuffer(raw(g.matches, g.n_matches, K.MATCH_DT), dtype=K.MATCH_DT) if g.n_matches else np.zeros(0, K.MATCH_DT)
            assert len(gm) == len(vis) and np.array_equal(gm["position"], vis["position"]) and np.array_equal(gm["flags"], vis["flags"]), ("visible matches", i)
            assert int(seg_cnt[i].n_matches) == len(em) and int(seg_cnt[i].n_canonical) == int(((em["flags"] & K.MATCH_CANONICAL) != 0).sum()) \
                and int(seg_cnt[i].n_forward) == int(((em["flags"] & K.MATCH_FORWARD) != 0).sum()), ("counts", i)
            nvis += len(gm)
    L.ts_free_segments(ref, n)
    return {"segments": n, "windows": nwin, "blocks": nblk, "visible_matches": nvis if with_matches else None,
            "compared": "window records and blocks byte for byte" + (", visible match records, per-segment counts" if with_matches else "")}
