"""Backend-agnostic test harness: drives either the CPU oracle or the HIP product
through the same caller logic the reference wraps around its scan path, so that
the reference's own `.tst` manifests (stdout of the CLI) can be checked.

What is restated here (test infrastructure, host/python only):
  * option parsing of the flags the manifests use       (src/main.cpp:186-660)
  * FASTA -> path components (segments / N-gaps)        (gfalibs behaviour, pinned by
                                                         testFiles/expected/*_gaps.bed)
  * Teloscope::walkPath                                 (src/input.cpp:942-1041)
  * Path/Assembly summary printing                      (src/teloscope.cpp:687-694,815-857,959-1055)
  * FASTQ 4-line record reader of --fastq-subset        (src/input.cpp:96-149,737-832)

The scan itself (scanSegment, block calling, labelTerminalBlocks, read filter) is
delegated to a backend object: tests.backends.OracleBackend or ProductBackend.
"""
import gzip
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SCAFFOLD_NAMES = ["t2t", "gapped_t2t", "misassembly", "gapped_misassembly", "incomplete",
                  "gapped_incomplete", "none", "gapped_none", "discordant", "gapped_discordant"]


# ----------------------------------------------------------------------------- CLI
# (the option parser lives in the package: bench.py describes its workload with the same flag strings)
from teloscope_amd.cli import Options, parse_cli  # noqa: E402,F401

# --------------------------------------------------------------------------- FASTA
def _open(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "rt")


def read_fasta(path):
    """-> [(header_token, sequence)] ; header = first whitespace-delimited token."""
    out, name, chunks = [], None, []
    with _open(path) as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(chunks)))
                name = line[1:].split()[0] if len(line) > 1 else ""
                chunks = []
            elif name is not None:
                chunks.append(line)
    if name is not None:
        out.append((name, "".join(chunks)))
    return out


def path_components(seq):
    """Split a record into segments and gaps the way gfalibs builds a path: every run of
    N/n (also X/x) is a gap component, the rest are '+' segments.  Returns
    [('S', start, str) | ('G', start, length)]."""
    comps = []
    b = np.frombuffer(seq.encode(), dtype=np.uint8)
    if len(b) == 0:
        return comps
    isgap = (b == ord("N")) | (b == ord("n")) | (b == ord("X")) | (b == ord("x"))
    edges = np.flatnonzero(np.diff(isgap.astype(np.int8))) + 1
    starts = np.concatenate(([0], edges))
    ends = np.concatenate((edges, [len(b)]))
    for s, e in zip(starts, ends):
        if isgap[s]:
            comps.append(("G", int(s), int(e - s)))
        else:
            comps.append(("S", int(s), seq[s:e]))
    return comps


# ------------------------------------------------------------------------ walkPath
def walk_path(backend, opts, seq_pos, header, seq):
    """Teloscope::walkPath, src/input.cpp:942-1041."""
    pd = dict(seq_pos=seq_pos, header=header.replace("\r", ""), path_size=len(seq), gaps=[],
              windows=[], terminal_blocks=[], interstitial_blocks=[],
              canonical_matches=[], non_canonical_matches=[])
    abs_pos = 0
    for comp in path_components(seq):
        if comp[0] == "S":
            s = comp[2].upper()                    # unmaskSequence
            r = backend.scan_segment(s, abs_pos, opts.ultra_fast)
            for k in ("windows", "terminal_blocks", "interstitial_blocks",
                      "canonical_matches", "non_canonical_matches"):
                pd[k].append(r[k])
            abs_pos += len(s)
        else:
            pd["gaps"].append((abs_pos, comp[2]))
            abs_pos += comp[2]
    for k in ("windows", "terminal_blocks", "interstitial_blocks",
              "canonical_matches", "non_canonical_matches"):
        pd[k] = np.concatenate(pd[k]) if pd[k] else None
    tb = pd["terminal_blocks"]
    if tb is None:
        tb = backend.empty_blocks()
    tb, label, stype = backend.label_terminal_blocks(tb, len(pd["gaps"]) & 0xFFFF, pd["path_size"],
                                                     opts.terminal_limit)
    pd["terminal_blocks"], pd["terminal_label"], pd["scaffold_type"] = tb, label, stype
    return pd


def _n(a):
    return 0 if a is None else len(a)


def _fmt_float(x):
    """operator<<(float) at default precision 6"""
    return "%g" % float(np.float32(x))


def _n50(lengths):
    """Teloscope::computeN50, include/teloscope.h:224-235"""
    if not lengths:
        return 0
    ls = sorted(lengths, reverse=True)
    total, cum = sum(ls), 0
    for v in ls:
        cum += v
        if cum * 2 >= total:
            return v
    return ls[-1]


def _stats(values):
    """getStats, src/tools.cpp:23-51 (float32 accumulation)"""
    v = [np.float32(x) for x in values]
    s = np.float32(0.0)
    for x in v:
        s = np.float32(s + x)
    mean = np.float32(s / np.float32(len(v)))
    srt = sorted(v)
    mid = len(v) // 2
    if len(v) % 2 == 0:
        med = np.float32((srt[mid] + srt[mid - 1]) / np.float32(2))
    else:
        med = srt[mid]
    return mean, med, min(v), max(v)


def run_assembly(backend, opts, fasta_path):
    """Input::read + handleBEDFile/printSummary: returns the CLI's stdout as a string."""
    recs = read_fasta(fasta_path)
    paths = [walk_path(backend, opts, i, h, s) for i, (h, s) in enumerate(recs)]
    return format_report(paths, opts), paths


def format_report(paths, opts):
    out = ["", "+++ Path Summary Report +++"]
    if not opts.ultra_fast:
        out.append("pos\theader\ttelomeres\tlabels\tgaps\ttype\tgranular\tits\tcanonical\twindows")
    else:
        out.append("pos\theader\ttelomeres\tlabels\tgaps\ttype\tgranular")
    tot_telo = tot_gaps = tot_its = tot_can = tot_win = 0
    lengths = []
    counts = [0] * 10
    scaf_lens, contig_lens = [], []
    for pd in paths:
        longest, labels = 0, ""
        for b in pd["terminal_blocks"]:
            if b["is_longest"]:
                longest += 1
                labels += b["block_label"].decode()
                lengths.append(float(b["block_len"]))
        line = "%d\t%s\t%d\t%s\t%d\t%s\t%s" % (
            pd["seq_pos"] + 1, pd["header"], longest, labels if labels else "none",
            len(pd["gaps"]) & 0xFFFF, SCAFFOLD_NAMES[pd["scaffold_type"]], pd["terminal_label"])
        tot_telo += longest
        tot_gaps += len(pd["gaps"]) & 0xFFFF
        if not opts.ultra_fast:
            line += "\t%d\t%d\t%d" % (_n(pd["interstitial_blocks"]), _n(pd["canonical_matches"]),
                                      _n(pd["windows"]))
            tot_win += _n(pd["windows"])
            tot_its += _n(pd["interstitial_blocks"])
            tot_can += _n(pd["canonical_matches"])
        out.append(line)
        counts[pd["scaffold_type"]] += 1
        scaf_lens.append(pd["path_size"])
        prev_end = 0
        for gs, gl in sorted(pd["gaps"]):
            if gs > prev_end:
                contig_lens.append(gs - prev_end)
            prev_end = gs + gl
        if pd["path_size"] > prev_end:
            contig_lens.append(pd["path_size"] - prev_end)

    out += ["", "+++ Assembly Summary Report +++", "Total paths:\t%d" % len(paths),
            "Total gaps:\t%d" % tot_gaps, "Scaffold N50:\t%d" % _n50(scaf_lens),
            "Contig N50:\t%d" % _n50(contig_lens), "Total telomeres:\t%d" % tot_telo]
    if not opts.ultra_fast:
        out += ["Total ITS blocks:\t%d" % tot_its, "Total canonical matches:\t%d" % tot_can,
                "Total windows analyzed:\t%d" % tot_win]
    out += ["", "+++ Telomere Statistics +++"]
    if tot_telo > 0:
        mean, med, mn, mx = _stats(lengths)
        out += ["Mean length:\t" + _fmt_float(mean), "Median length:\t" + _fmt_float(med),
                "Min length:\t" + _fmt_float(mn), "Max length:\t" + _fmt_float(mx)]
    else:
        out.append("No telomeres found for statistics.")
    T = counts
    out += ["", "+++ Chromosome Telomere Counts+++",
            "Two telomeres:\t%d" % (T[0] + T[1] + T[2] + T[3]),
            "One telomere:\t%d" % (T[4] + T[5]), "Zero telomeres:\t%d" % (T[6] + T[7]),
            "", "+++ Chromosome Telomere/Gap Completeness+++",
            "T2T:\t%d" % T[0], "Gapped T2T:\t%d" % T[1], "Misassembled:\t%d" % T[2],
            "Gapped misassembled:\t%d" % T[3], "Incomplete:\t%d" % T[4],
            "Gapped incomplete:\t%d" % T[5], "No telomeres:\t%d" % T[6],
            "Gapped no telomeres:\t%d" % T[7], "Discordant:\t%d" % T[8],
            "Gapped discordant:\t%d" % T[9]]
    return "\n".join(out) + "\n"


# ------------------------------------------------------------------ output files (row f2)
BED_SUFFIXES = ("_window_repeat_density.bedgraph", "_window_canonical_ratio.bedgraph",
                "_window_strand_ratio.bedgraph", "_window_gc.bedgraph", "_window_entropy.bedgraph",
                "_canonical_matches.bed", "_noncanonical_matches.bed", "_terminal_telomeres.bed",
                "_interstitial_telomeres.bed", "_gaps.bed", "_report.tsv")


def format_bed_files(paths, records, opts):
    """Independent restatement of handleBEDFile / writeBEDFile / printSummary
    (src/teloscope.cpp:661-957, 994-1055; formats in docs/outputs.md): {suffix: text} for the files
    the flags switch on.  `records` are the FASTA records the paths came from (matchSeq is a
    substring of the upper-cased record, src/teloscope.cpp:466-468)."""
    f32 = np.float32
    out = {"_terminal_telomeres.bed": [], "_gaps.bed": [], "_report.tsv": []}
    if opts.out_win_repeats:
        out["_window_repeat_density.bedgraph"] = [
            'track type=bedGraph name="Repeat Density" description="Total repeat density per window"']
        out["_window_canonical_ratio.bedgraph"] = [
            'track type=bedGraph name="Canonical Ratio" description="Canonical fraction of repeat density per window"']
        out["_window_strand_ratio.bedgraph"] = [
            'track type=bedGraph name="Strand Ratio" description="Forward-strand fraction of repeat density per window"']
    if opts.out_entropy:
        out["_window_entropy.bedgraph"] = [
            'track type=bedGraph name="Shannon Entropy" description="Shannon entropy per window"']
    if opts.out_gc:
        out["_window_gc.bedgraph"] = ['track type=bedGraph name="GC Content" description="GC content per window"']
    if opts.out_matches:
        out["_canonical_matches.bed"] = []
        out["_noncanonical_matches.bed"] = []
    if opts.out_its:
        out["_interstitial_telomeres.bed"] = []
    report = format_report(paths, opts).split("\n")
    # the report file holds the column header + path rows, then everything printSummary prints
    out["_report.tsv"] = report[2:-1]
    for pd, (_, seq) in zip(paths, records):
        h, size = pd["header"], pd["path_size"]
        for b in pd["terminal_blocks"]:
            start, end = int(b["start"]), int(b["start"]) + int(b["block_len"])
            limit = opts.terminal_limit
            scaffold = start < limit or end > ((size - limit) & 0xFFFFFFFFFFFFFFFF)
            if scaffold or opts.manual_curation:
                out["_terminal_telomeres.bed"].append("\t".join(map(str, [
                    h, start, end, int(b["block_len"]), b["block_label"].decode(), int(b["forward_count"]),
                    int(b["reverse_count"]), int(b["canonical_count"]), int(b["non_canonical_count"]), size,
                    "scaffold" if scaffold else "contig"])))
        if opts.out_its and pd["interstitial_blocks"] is not None:
            for b in pd["interstitial_blocks"]:
                out["_interstitial_telomeres.bed"].append("\t".join(map(str, [
                    h, int(b["start"]), int(b["start"]) + int(b["block_len"]), int(b["block_len"]),
                    b["block_label"].decode(), int(b["forward_count"]), int(b["reverse_count"]),
                    int(b["canonical_count"]), int(b["non_canonical_count"]), size])))
        for gs, gl in pd["gaps"]:
            out["_gaps.bed"].append("%s\t%d\t%d" % (h, gs, gs + gl))
        if opts.out_matches:
            up = seq.upper()
            for key, name in (("canonical_matches", "_canonical_matches.bed"),
                              ("non_canonical_matches", "_noncanonical_matches.bed")):
                if pd[key] is None:
                    continue
                for m in pd[key]:
                    p0, ln = int(m["position"]), int(m["match_size"])
                    out[name].append("%s\t%d\t%d\t%s" % (h, p0, p0 + ln, up[p0:p0 + ln]))
        if pd["windows"] is not None:
            for w in pd["windows"]:
                ws, sz = int(w["window_start"]), int(w["current_window_size"])
                pre = "%s\t%d\t%d\t" % (h, ws, ws + sz)
                if opts.out_win_repeats:
                    fwd, rev = int(w["fwd_covered"]), int(w["rev_covered"])
                    can, non = int(w["canonical_covered"]), int(w["non_canonical_covered"])
                    tot = (fwd + rev) & 0xFFFFFFFF
                    dens = f32(f32(tot) / f32(sz))
                    cr = f32(f32(can) / f32((can + non) & 0xFFFFFFFF)) if tot > 0 else f32(-1.0)
                    sr = f32(f32(fwd) / f32((fwd + rev) & 0xFFFFFFFF)) if tot > 0 else f32(-1.0)
                    out["_window_repeat_density.bedgraph"].append(pre + _fmt_float(dens))
                    out["_window_canonical_ratio.bedgraph"].append(pre + _fmt_float(cr))
                    out["_window_strand_ratio.bedgraph"].append(pre + _fmt_float(sr))
                if opts.out_entropy:
                    out["_window_entropy.bedgraph"].append(pre + _fmt_float(w["shannon_entropy"]))
                if opts.out_gc:
                    out["_window_gc.bedgraph"].append(pre + _fmt_float(w["gc_content"]))
    return {k: "".join(l + "\n" for l in v) for k, v in out.items()}


# --------------------------------------------------------------------------- FASTQ
def read_fastq_records(data):
    """4-line reader of src/input.cpp:113-138: blank lines between records are skipped,
    line terminators are kept verbatim in the echoed record, the sequence handed to the
    filter is the second line without its '\\n' (a trailing '\\r' is the filter's job).
    Returns [(raw_record_bytes, sequence_bytes)] or raises ValueError on a malformed file."""
    if not data:
        raise ValueError("FASTQ input is empty")
    if data[:1] != b"@":
        raise ValueError("FASTQ input must start with '@'")
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":          # std::getline yields no line after a final '\n'
        lines.pop()

    def logical_len(l):
        return len(l) - 1 if l.endswith(b"\r") else len(l)

    recs, i, n = [], 0, len(lines)
    while i < n:
        if logical_len(lines[i]) == 0:      # blank line (or lone '\r') before a header
            i += 1
            continue
        if i + 3 >= n:
            raise ValueError("truncated FASTQ record")
        h, s, p, q = lines[i:i + 4]
        if not h.startswith(b"@"):
            raise ValueError("expected header line starting with '@'")
        if not p.startswith(b"+"):
            raise ValueError("expected separator line starting with '+'")
        if logical_len(s) != logical_len(q):
            raise ValueError("sequence and quality length differ")
        recs.append((b"\n".join((h, s, p, q)) + b"\n", s))
        i += 4
    return recs


def run_fastq_subset(read_filter, data):
    """Input::readFastqSubset: returns (stdout bytes, kept, total)."""
    recs = read_fastq_records(data)
    passes = read_filter.filter([s for _, s in recs])
    out = b"".join(r for (r, _), ok in zip(recs, passes) if ok)
    return out, int(sum(passes)), len(recs)


# ------------------------------------------------------------------------ manifests
def load_manifest(path):
    with open(path) as fh:
        lines = fh.read().split("\n")
    command = lines[0]
    rest = lines[1:]
    first = next((j for j, l in enumerate(rest) if l.strip()), None)
    if first is None:
        return dict(command=command, mode="empty")
    if rest[first].startswith("expect_") or rest[first].startswith("gfa_"):
        directives = []
        for l in rest[first:]:
            if l.strip():
                k, _, v = l.partition(" ")
                directives.append((k, v))
        return dict(command=command, mode="directive", directives=directives)
    if rest[first] == "embedded":
        body = rest[first + 1:]
        return dict(command=command, mode="embedded", expected="\n".join(body))
    return dict(command=command, mode="file", expected_path=rest[first].strip())


def golden_path(rel):
    """testFiles/x -> tests/golden/testFiles/x"""
    return os.path.join(GOLDEN, rel)


# ---------------------------------------------------------------------------------------- GFA mode (tips-only pins)
def parse_gfa(path):
    """Segments and paths of a GFA 1.x / 2.0 file, as far as the telomere annotation reads them: S lines (name ->
    sequence, None for '*'; GFA 2 carries the length before the sequence), P lines (GFA 1: comma- or semicolon-separated
    oriented segments) and O lines (GFA 2 ordered groups)."""
    segs, paths = {}, []
    with _open(path) as fh:
        for line in fh:
            if isinstance(line, bytes):
                line = line.decode()
            f = line.rstrip("\r\n").split("\t")
            if f[0] == "S" and len(f) >= 3:
                seq = f[3] if (len(f) >= 4 and f[2].isdigit()) else f[2]
                segs[f[1]] = None if seq == "*" else seq
            elif f[0] == "P" and len(f) >= 3:
                comps = [c for c in f[2].replace(";", ",").split(",") if c]
                paths.append((f[1], [(c[:-1], c[-1]) for c in comps]))
            elif f[0] == "O" and len(f) >= 3:
                comps = [c for c in f[2].split(" ") if c]
                paths.append((f[1], [(c[:-1], c[-1]) for c in comps if c[-1] in "+-"]))
    return segs, paths


def gfa_annotations(backend, opts, gfa_path):
    """The telomere nodes the reference's GFA mode adds (src/input.cpp:625-716 job resolution; walkSegment :835-881,
    walkSegmentForPath :884-939): one tips-only scanSegment per terminal end, the terminal blocks at that end, the
    longest one's length as the node's tl_bp.  Returns the set of (segment, terminal_role, path_orient, node_name, tl_bp)."""
    segs, paths = parse_gfa(gfa_path)
    rows = set()

    def blocks_of(name):
        seq = segs.get(name)
        if seq is None:
            return None, 0
        res = backend.scan_segment(seq.upper().encode(), 0, True)          # unmaskSequence, then scanSegment(seq, 0, true)
        return res["terminal_blocks"], len(seq)

    if paths:
        ends = set()
        for _, comps in paths:
            comps = [c for c in comps if c[0] in segs]
            if comps:
                ends.add((comps[0][0], comps[0][1], True))
                ends.add((comps[-1][0], comps[-1][1], False))
        for name, orient, is_first in sorted(ends):
            blocks, n = blocks_of(name)
            if blocks is None:
                continue
            scan_start = is_first == (orient == "+")
            best = 0
            for b in blocks:
                at_start = int(b["start"]) <= n - (int(b["start"]) + int(b["block_len"]))
                if at_start == scan_start:
                    best = max(best, int(b["block_len"]))
            if best:
                role = "start" if is_first else "end"
                rows.add((name, role, orient, "telomere_%s%s_%s" % (name, orient, role), best))
    else:
        for name in segs:
            blocks, n = blocks_of(name)
            if blocks is None:
                continue
            best = {}
            for b in blocks:
                at_start = int(b["start"]) <= n - (int(b["start"]) + int(b["block_len"]))
                best[at_start] = max(best.get(at_start, 0), int(b["block_len"]))
            for at_start, ln in best.items():
                role = "start" if at_start else "end"
                rows.add((name, role, ".", "telomere_%s+_%s" % (name, role), ln))
    return rows


def read_gfa_expectation(tsv_path):
    rows = set()
    with open(tsv_path) as fh:
        header = fh.readline().rstrip("\n").split("\t")
        for line in fh:
            f = dict(zip(header, line.rstrip("\n").split("\t")))
            rows.add((f["segment"], f["terminal_role"], f["path_orient"], f["node_name"], int(f["tl_bp"])))
    return rows
