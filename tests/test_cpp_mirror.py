"""The C++17 host-side mirror (include/teloscope_mi355x.hpp: Teloscope::scanSegment,
ReadTelomereFilter::matches, expandPatternsWithOrientation, labelTerminalBlocks above the C-ABI)
driven by tests/cpp/manifest_cli.cpp: on a GPU box every legacy `.tst` manifest of the reference is
replayed through it and must reproduce the expected CLI stdout; on CPU it must build and refuse
to run without a HIP device."""
import glob
import os
import shlex
import subprocess

import pytest

from tests import harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANIFESTS = sorted(glob.glob(os.path.join(H.GOLDEN, "validateFiles", "*.tst")))
LEGACY = [m for m in MANIFESTS if H.load_manifest(m)["mode"] == "embedded"]


@pytest.fixture(scope="module")
def cli(tmp_path_factory):
    import teloscope_amd  # noqa: F401  (makes sure libteloscan.so is built)
    out = tmp_path_factory.mktemp("cpp") / "manifest_cli"
    libdir = os.path.join(ROOT, "teloscope_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "manifest_cli.cpp"), "-L", libdir, "-lteloscan",
                           "-Wl,-rpath," + libdir, "-pthread", "-lz", "-o", str(out)])
    return str(out)


def test_cpp_mirror_builds_and_refuses_without_gpu(cli):
    from teloscope_amd import _capi as K
    if K.lib().ts_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([cli, "-f", H.golden_path("testFiles/t2t.fa"), "-i"], capture_output=True, text=True)
    assert r.returncode == 1 and "no usable HIP device" in r.stderr and r.stdout == ""


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["", "0,0", "0,0,0"], ids=["D1", "D2", "D3"])
def test_cpp_mirror_replays_all_legacy_manifests(cli, tmp_path, devices):
    """D2 / D3: the assembly scan over two / three contexts (here on one GPU) — Teloscope(ui, devices), every batch of
    segments cut into one shard per context (ts_scan_segments_multi) — must print what one context prints."""
    failures = []
    for path in LEGACY:
        m = H.load_manifest(path)
        args = []
        for tok in shlex.split(m["command"]):
            if tok.startswith("testFiles/"):
                tok = H.golden_path(tok)                      # .fa and .fa.gz alike (zlib in readFasta)
            args.append(tok)
        if devices:
            args += ["--devices", devices]
        r = subprocess.run([cli] + args, capture_output=True, text=True, timeout=120)
        if r.returncode != 0 or r.stdout.split("\n") != m["expected"].split("\n"):
            failures.append((os.path.basename(path), r.returncode, r.stderr[-200:]))
    assert not failures, failures[:5]


FASTQ = [m for m in MANIFESTS if os.path.basename(m).startswith("fastq_subset")]


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["", "0,0"], ids=["D1", "D2"])
@pytest.mark.parametrize("path", FASTQ, ids=[os.path.basename(p) for p in FASTQ])
def test_cpp_mirror_replays_fastq_manifests(cli, path, devices):
    """--fastq-subset through include/teloscope_mi355x_io.hpp (fastqSubset + ReadTelomereFilter on the
    GPU): stdout must equal the reference's expected subset file byte for byte, the kept/total line and
    the malformed-input message must match (src/input.cpp:737-832, 113-138).  D2: the same with the read shard —
    two read-filter contexts (here on one GPU), every batch dealt to them in consecutive shards, pass bits and
    output order unchanged (the reference's -j 1 / -j 8 manifests expect identical bytes)."""
    m = H.load_manifest(path)
    d = {}
    for k, v in m["directives"]:
        d.setdefault(k, []).append(v)
    if " -o " in m["command"]:
        pytest.skip("-o (file instead of stdout) belongs to the reference's front end")
    args, stdin = [], None
    toks = shlex.split(m["command"])
    i = 0
    while i < len(toks):
        tok = toks[i]
        if tok == "<":
            stdin = open(H.golden_path(toks[i + 1]), "rb")
            i += 2
            continue
        if tok.startswith("testFiles/"):
            tok = H.golden_path(tok)
        args.append(tok)
        i += 1
    if devices:
        args += ["--read-devices", devices, "--reads-per-batch", "64"]      # several batches, each cut into two shards
    r = subprocess.run([cli] + args, stdin=stdin, capture_output=True, timeout=120)
    assert (r.returncode != 0) == (int(d["expect_exit"][0]) != 0), r.stderr
    so = d.get("expect_stdout", ["ignore"])[0]
    if so != "ignore":
        assert r.stdout == open(H.golden_path(so), "rb").read()
    for sub in d.get("expect_stderr_substr", []):
        assert sub.encode() in r.stderr, (sub, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("block", [64, 257, 1000, 4096, 70000])
@pytest.mark.parametrize("name", ["fastq_subset_large.fq", "fastq_subset_crlf.fq", "fastq_subset_blanklines.fq"])
def test_fastq_reader_block_boundaries(cli, name, block):
    """fastqSubset parses records in place.  A stream (stdin, gzip) comes in fixed-size blocks: records that
    straddle a block end are carried over to the next block, a record larger than the block grows it; a regular
    file is mapped and cut into GPU batches by record bytes.  Any block size, through either reader, must give
    the bytes the one-block run gives (which the manifests pin against the reference's expected files)."""
    src = H.golden_path("testFiles/" + name)
    flags = ["--fastq-subset", "-x", "0", "-l", "18", "-y", "0.8", "-k", "10", "-d", "10"]
    ref = subprocess.run([cli] + flags + [src], capture_output=True, timeout=120)
    assert ref.returncode == 0 and len(ref.stdout) > 0, ref.stderr
    with open(src, "rb") as fh:
        data = fh.read()
    runs = [subprocess.run([cli] + flags + ["--fastq-block", str(block), src], capture_output=True, timeout=120),       # mapped file
            subprocess.run([cli] + flags + ["--fastq-block", str(block)], input=data, capture_output=True, timeout=120)]  # stdin: block reader
    for got in runs:
        assert got.returncode == 0, got.stderr
        assert got.stdout == ref.stdout
        assert got.stderr.splitlines()[-1] == ref.stderr.splitlines()[-1]        # kept N of M reads


SUFFIXES = ["_window_repeat_density.bedgraph", "_window_canonical_ratio.bedgraph", "_window_strand_ratio.bedgraph",
            "_window_gc.bedgraph", "_window_entropy.bedgraph", "_canonical_matches.bed", "_noncanonical_matches.bed",
            "_terminal_telomeres.bed", "_interstitial_telomeres.bed", "_gaps.bed", "_report.tsv"]


@pytest.mark.gpu
@pytest.mark.parametrize("flags", ["-r -g -e -m -i -w 1000 -s 500 -t 3000", "-r -g -e -i", "-r -m -t 20000"])
def test_output_files_do_not_depend_on_the_number_of_devices(cli, tmp_path, flags):
    """Every output file of a run over a real 4.2 Mb chromosome (windows, match BEDs, blocks, report) byte for byte the
    same from one context and from two and three (a boundary of the split falls inside the chromosome; -t 3000 lets it
    fall anywhere but the outermost tiles)."""
    src = H.golden_path("testFiles/bTaeGut7_chr33_mat.fa.gz")
    outs = {}
    for devices in ("", "0,0", "0,0,0"):
        base = str(tmp_path / ("run" + devices.replace(",", "")))
        args = [cli, "-f", src] + shlex.split(flags) + ["--out-base", base] + (["--devices", devices] if devices else [])
        r = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        outs[devices] = (r.stdout, {sfx: open(base + sfx, "rb").read() if os.path.exists(base + sfx) else None for sfx in SUFFIXES})
    ref = outs[""]
    assert any(v for v in ref[1].values())
    for devices in ("0,0", "0,0,0"):
        assert outs[devices][0] == ref[0], devices
        for sfx in SUFFIXES:
            assert outs[devices][1][sfx] == ref[1][sfx], (devices, sfx)


def _build_rank_scan(tmp_path):
    import teloscope_amd  # noqa: F401
    exe = tmp_path / "rank_scan"
    libdir = os.path.join(ROOT, "teloscope_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "rank_scan.cpp"), "-L", libdir, "-lteloscan",
                           "-Wl,-rpath," + libdir, "-pthread", "-o", str(exe)])
    return exe


def test_cpp_rank_program_builds_and_refuses_without_gpu(tmp_path):
    from teloscope_amd import _capi as K
    if K.lib().ts_device_count() > 0:
        pytest.skip("a GPU is present")
    exe = _build_rank_scan(tmp_path)
    r = subprocess.run([str(exe), "--rank", "0", "--ranks", "1", "--id-file", str(tmp_path / "id")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no usable HIP device" in r.stderr and r.stdout == ""


@pytest.mark.gpu
def test_cpp_rank_program_scans_packs_and_gathers_over_rccl(tmp_path):
    """tests/cpp/rank_scan.cpp: one rank of a sharded scan written against the C-ABI alone (plan, shard, scan, pack,
    ts_exchange_gather over RCCL, ts_shards_finalize), run here as the only rank — its message loops back through RCCL — and
    compared inside the program with the whole-batch result of ts_scan_segments_blocks."""
    exe = _build_rank_scan(tmp_path)
    r = subprocess.run([str(exe), "--rank", "0", "--ranks", "1", "--id-file", str(tmp_path / "id"), "--mbases", "60"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rank_scan ok: 1 rank(s)" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-600:])
