"""Row f4: --bam-subset through include/teloscope_mi355x_io.hpp (bamSubset, the device read filter):
subsetBam of src/bam.cpp:188-259 — the header and every record whose read has a terminal telomere
block are copied byte for byte into a new BGZF-compressed BAM closed by the EOF marker.  The reference
ships no BAM fixtures (its scripts/test_bam_subset.py builds them on the fly); so does this test, with
its own BGZF/BAM encoder, and the kept set is checked against the CPU oracle's read filter."""
import struct
import subprocess
import zlib

import numpy as np
import pytest

from tests import harness as H
from tests import seqgen
from tests.backends import OracleReadFilter
from tests.test_cpp_mirror import cli  # noqa: F401  (fixture: builds tests/cpp/manifest_cli.cpp)

EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
NIBBLE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def bgzf(data, chunk):
    out = []
    for a in range(0, len(data), chunk):
        piece = data[a:a + chunk]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = co.compress(piece) + co.flush()
        total = 18 + len(payload) + 8
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", total - 1) + payload +
                   struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
    return b"".join(out) + EOF_BLOCK


def gunzip_members(data):
    out, pos = bytearray(), 0
    while pos < len(data):
        d = zlib.decompressobj(31)
        out += d.decompress(data[pos:])
        pos = len(data) - len(d.unused_data)
    return bytes(out)


def bam_record(name, seq):
    packed = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= NIBBLE[ch] << (0 if i & 1 else 4)
    body = struct.pack("<iiBBHHHiiii", -1, -1, len(name) + 1, 0, 4680, 0, 4, len(seq), -1, -1, 0)
    body += name.encode() + b"\0" + bytes(packed) + b"\xff" * len(seq)
    return struct.pack("<i", len(body)) + body


def build_bam(reads, chunk):
    text = b"@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000\n"
    header = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + \
        struct.pack("<i", 1000)
    records = [bam_record(n, s) for n, s in reads]
    return header, records, bgzf(header + b"".join(records), chunk)


def make_reads():
    rng = np.random.default_rng(99)
    reads = [("kat_pass", "TTAGGG" * 60), ("kat_short", "TTAGGG" * 6), ("kat_rev", "CCCTAA" * 70), ("acgt", "ACGT" * 100),
             ("noseq", ""), ("iupac", "TTAGGG" * 30 + "NRYK" + "TTAGGG" * 30), ("odd", "TTAGGG" * 55 + "A")]
    for i in range(400):
        n = int(rng.integers(50, 20000))
        s = bytearray(seqgen.random_dna(rng, n).tobytes())
        if i % 5 == 0:
            t = seqgen.mutate(rng, seqgen.repeat_array("TTAGGG" if i % 2 else "CCCTAA", int(rng.integers(8, 400))), 0.02).tobytes()
            t = t[:n]
            if i % 3:
                s[:len(t)] = t
            else:
                s[n - len(t):] = t
        reads.append(("r%d" % i, s.decode()))
    return reads


@pytest.mark.gpu
@pytest.mark.parametrize("flags,chunk,via_stdin", [("", 60000, False), ("-l 42", 1000, False), ("-x 0 -l 18 -y 0.8 -k 10 -d 10", 64000, True)])
def test_bam_subset_matches_oracle(cli, tmp_path, flags, chunk, via_stdin):  # noqa: F811
    reads = make_reads()
    header, records, bam = build_bam(reads, chunk)
    path = tmp_path / "in.bam"
    path.write_bytes(bam)
    args = [cli, "--bam-subset"] + flags.split()
    if via_stdin:
        r = subprocess.run(args, stdin=open(path, "rb"), capture_output=True, timeout=300)
    else:
        r = subprocess.run(args + [str(path)], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr
    opts = H.parse_cli("--fastq-subset " + flags)
    with_seq = [i for i, (_, s) in enumerate(reads) if s]
    passes = OracleReadFilter(opts).filter([reads[i][1].encode() for i in with_seq])
    keep = [i for i, ok in zip(with_seq, passes) if ok]
    assert 0 < len(keep) < len(with_seq)
    out = r.stdout
    assert out.endswith(EOF_BLOCK) and out[:4] == b"\x1f\x8b\x08\x04"
    plain = gunzip_members(out)
    assert plain[:len(header)] == header
    assert plain[len(header):] == b"".join(records[i] for i in keep)          # kept records, in order, byte for byte
    err = r.stderr.decode()
    assert "BAM subset: kept %d of %d records." % (len(keep), len(reads)) in err
    assert "BAM subset: skipped 1 record without SEQ." in err


@pytest.mark.gpu
def test_bam_subset_rejects_garbage_and_flags_missing_eof(cli, tmp_path):  # noqa: F811
    bad = tmp_path / "bad.bam"
    bad.write_bytes(bgzf(b"NOTBAM" + b"\0" * 100, 60000))
    r = subprocess.run([cli, "--bam-subset", str(bad)], capture_output=True, timeout=120)
    assert r.returncode != 0 and b"not a BAM" in r.stderr
    header, records, bam = build_bam(make_reads()[:8], 60000)
    noeof = tmp_path / "noeof.bam"
    noeof.write_bytes(bam[:-len(EOF_BLOCK)])
    r = subprocess.run([cli, "--bam-subset", str(noeof)], capture_output=True, timeout=120)
    assert r.returncode == 0 and b"missing the BGZF EOF marker" in r.stderr


def _kept_names(plain, header):
    """read names of the records of an uncompressed BAM payload (validating its structure on the way)."""
    assert plain[:len(header)] == header
    names, pos = [], len(header)
    while pos < len(plain):
        (bs,) = struct.unpack_from("<i", plain, pos)
        assert 32 <= bs and pos + 4 + bs <= len(plain)
        lname = plain[pos + 4 + 8]
        names.append(plain[pos + 36:pos + 36 + lname - 1].decode())
        pos += 4 + bs
    return names


def _run_bam(cli, tmp_path, reads, flags, tag):
    header, records, bam = build_bam(reads, 60000)
    path = tmp_path / ("%s.bam" % tag)
    path.write_bytes(bam)
    r = subprocess.run([cli, "--bam-subset"] + flags + [str(path)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return _kept_names(gunzip_members(r.stdout), header)


@pytest.mark.gpu
def test_bam_threshold_known_answers(cli, tmp_path):  # noqa: F811
    """The exact threshold cases the reference pins through its BAM entry (scripts/test_bam_subset.py:339-381:
    default -l 42 boundary, exact 12 / 18 bp lengths at -y 1, the 2/3 density boundary, a custom plant canonical),
    here through bamSubset on the GPU."""
    default = [("short", "TTAGGG" * 6), ("default_pass", "TTAGGG" * 7), ("long", "CCCTAA" * 15), ("fail", "ACGT" * 20)]
    assert _run_bam(cli, tmp_path, default, [], "d") == ["default_pass", "long"]
    lengths = [("one_repeat", "TTAGGG"), ("exact_12", "TTAGGG" * 2), ("flanked_exact", "ACGT" + "CCCTAA" * 2 + "TGCA"),
               ("exact_18", "TTAGGG" * 3)]
    l12 = ["-x", "0", "-l", "12", "-y", "1", "-k", "10", "-d", "10"]
    l18 = ["-x", "0", "-l", "18", "-y", "1", "-k", "10", "-d", "10"]
    assert _run_bam(cli, tmp_path, lengths, l12, "l12") == ["exact_12", "flanked_exact", "exact_18"]
    assert _run_bam(cli, tmp_path, lengths, l18, "l18") == ["exact_18"]
    density = [("two_thirds", "TTAGGGAAAAAATTAGGG")]
    assert _run_bam(cli, tmp_path, density, ["-x", "0", "-l", "18", "-y", "0.666", "-k", "20", "-d", "10"], "y1") == ["two_thirds"]
    assert _run_bam(cli, tmp_path, density, ["-x", "0", "-l", "18", "-y", "0.667", "-k", "20", "-d", "10"], "y2") == []
    plant = [("plant_pass", "TTTAGGG" * 3), ("vertebrate_fail", "TTAGGG" * 4)]
    assert _run_bam(cli, tmp_path, plant, ["-c", "CCCTAAA", "-x", "0", "-l", "21", "-y", "1"], "p") == ["plant_pass"]


@pytest.mark.gpu
def test_bam_mutation_robustness(cli, tmp_path):  # noqa: F811
    """Deterministic damage to a small BAM — random bytes, truncated payloads, single bit flips, a corrupt
    block_size / l_read_name / n_cigar_op, dropped records (the idea of the reference's mutation suite,
    scripts/test_bam_subset.py:589-633) — must end in a clean error (exit 1 + message) or in a valid BAM; never in a
    crash, a hang or a structurally broken output."""
    import random
    gen = random.Random(91)
    reads = [("record_%d" % i, "TTAGGG" * (3 + i % 5)) for i in range(8)]
    header, records, _ = build_bam(reads, 60000)
    payload = header + b"".join(records)
    roff = len(header)
    for index in range(70):
        mode = index % 7
        if mode == 0:
            data = bytes(gen.getrandbits(8) for _ in range(gen.randrange(0, 2048)))
        elif mode == 1:
            data = bgzf(payload[:gen.randrange(len(payload) + 1)], 60000)
        elif mode == 2:
            m = bytearray(payload)
            m[gen.randrange(roff + 36, len(m))] ^= 1 << gen.randrange(8)
            data = bgzf(bytes(m), 60000)
        elif mode == 3:
            m = bytearray(payload)
            struct.pack_into("<i", m, roff, gen.randrange(-16, 129))
            data = bgzf(bytes(m), 60000)
        elif mode == 4:
            m = bytearray(payload)
            m[roff + 12] = gen.randrange(256)
            data = bgzf(bytes(m), 60000)
        elif mode == 5:
            m = bytearray(payload)
            struct.pack_into("<H", m, roff + 16, gen.randrange(65536))
            data = bgzf(bytes(m), 60000)
        else:
            data = bgzf(header + b"".join(records[:gen.randrange(len(records) + 1)]), 60000)
        path = tmp_path / "m.bam"
        path.write_bytes(data)
        r = subprocess.run([cli, "--bam-subset", "-x", "0", "-l", "18", str(path)], capture_output=True, timeout=60)
        assert r.returncode in (0, 1), (index, mode, r.returncode, r.stderr[-300:])       # a signal would be negative
        if r.returncode == 0:
            plain = gunzip_members(r.stdout)
            assert r.stdout.endswith(EOF_BLOCK)
            pos = plain.index(b"chr1\0") + 5 + 4                                           # past the reference list
            while pos < len(plain):
                (bs,) = struct.unpack_from("<i", plain, pos)
                assert 32 <= bs and pos + 4 + bs <= len(plain), (index, mode)
                pos += 4 + bs
        else:
            assert r.stderr.startswith(b"Error:") or b"Error:" in r.stderr, (index, mode, r.stderr[-300:])
            assert r.stdout == b"" or not r.stdout.endswith(EOF_BLOCK), (index, mode)      # no complete-looking output on failure


def bgzf_fancy(data, chunk, gen):
    """BGZF the way the format allows and htslib does not write it: further extra subfields around BC, a file name, a
    comment, a header checksum (FHCRC = low 16 bits of the CRC32 of the header), empty blocks in between."""
    out = []
    for n, a in enumerate(range(0, len(data), chunk)):
        piece = data[a:a + chunk]
        co = zlib.compressobj(gen.choice([1, 6, 9]), zlib.DEFLATED, -15)
        payload = co.compress(piece) + co.flush()
        flags, extra_pre, extra_post, tail = 4, b"", b"", b""
        if n % 2:
            extra_pre = b"XY" + struct.pack("<H", 3) + b"abc"
        if n % 3 == 0:
            extra_post = b"ZZ" + struct.pack("<H", 0)
        if n % 4 == 1:
            flags |= 8; tail += b"name.bam\0"
        if n % 5 == 2:
            flags |= 16; tail += b"a comment\0"
        xlen = len(extra_pre) + 6 + len(extra_post)
        hcrc = 2 if n % 3 == 1 else 0
        if hcrc:
            flags |= 2
        total = 12 + xlen + len(tail) + hcrc + len(payload) + 8
        assert total <= 65536
        head = b"\x1f\x8b\x08" + bytes([flags]) + b"\0\0\0\0\0\xff" + struct.pack("<H", xlen) + extra_pre + \
            b"BC\x02\x00" + struct.pack("<H", total - 1) + extra_post + tail
        if hcrc:
            head += struct.pack("<H", zlib.crc32(head) & 0xFFFF)
        out.append(head + payload + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
        if n % 7 == 3:
            out.append(EOF_BLOCK)                                   # an empty block in mid-stream is legal
    return b"".join(out) + EOF_BLOCK


@pytest.mark.gpu
def test_bam_blocks_chunks_and_large_records(cli, tmp_path):  # noqa: F811
    """The parallel BGZF front end: blocks with every optional gzip header field the format allows, records that span
    blocks and inflate chunks (--bam-chunk-bytes 1 MiB against ~10 MB of records), one record larger than the headroom in
    front of a chunk (a 1.4 Mb read), small filter batches — kept records byte for byte, in order, as the oracle says."""
    import random
    gen = random.Random(5)
    rng = np.random.default_rng(123)
    reads = make_reads()
    big = bytearray(seqgen.random_dna(rng, 1_400_000).tobytes())
    t = seqgen.repeat_array("CCCTAA", 2000).tobytes()
    big[-len(t):] = t
    reads.insert(200, ("big_telomeric", big.decode()))
    reads.insert(300, ("big_plain", seqgen.random_dna(rng, 1_200_000).tobytes().decode()))
    header, records, _ = build_bam(reads, 60000)
    payload = header + b"".join(records)
    path = tmp_path / "fancy.bam"
    path.write_bytes(bgzf_fancy(payload, 30011, gen))
    r = subprocess.run([cli, "--bam-subset", "--bam-chunk-bytes", str(1 << 20), "--reads-per-batch", "97", str(path)],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    opts = H.parse_cli("--fastq-subset")
    with_seq = [i for i, (_, s) in enumerate(reads) if s]
    passes = OracleReadFilter(opts).filter([reads[i][1].encode() for i in with_seq])
    keep = [i for i, ok in zip(with_seq, passes) if ok]
    assert reads.index(("big_telomeric", big.decode())) in keep
    plain = gunzip_members(r.stdout)
    assert plain[:len(header)] == header
    assert plain[len(header):] == b"".join(records[i] for i in keep)
    assert b"missing the BGZF EOF marker" not in r.stderr
    # the same bytes through a pipe
    r2 = subprocess.run([cli, "--bam-subset", "--bam-chunk-bytes", str(1 << 20)], stdin=open(path, "rb"), capture_output=True, timeout=600)
    assert r2.returncode == 0 and gunzip_members(r2.stdout) == plain
    # a wrong header checksum, a BC subfield of the wrong length, reserved flag bits: clean errors
    good = bgzf_fancy(payload, 30011, random.Random(5))
    for what, damage in (("flags", lambda b: b[:3] + bytes([b[3] | 0x20]) + b[4:]),
                         ("bc length", lambda b: b[:14] + b"\x03" + b[15:]),
                         ("not gzip", lambda b: b"\x1f\x8c" + b[2:])):
        bad = tmp_path / "bad.bam"
        bad.write_bytes(damage(good))
        r3 = subprocess.run([cli, "--bam-subset", str(bad)], capture_output=True, timeout=120)
        assert r3.returncode == 1 and b"Error:" in r3.stderr, (what, r3.returncode, r3.stderr[-200:])


@pytest.mark.gpu
def test_bam_mutation_suite_512(cli, tmp_path):  # noqa: F811
    """512 deterministic mutations of a small BAM (the size of the reference's fast suite, scripts/test_bam_subset.py:589-633:
    random bytes, truncations of the compressed file and of the payload, bit flips in the compressed bytes and in the
    payload, corrupt block_size / l_read_name / n_cigar_op / l_seq, a damaged BGZF header field, dropped and duplicated
    blocks), all through ONE process and one filter (--bam-subset-each): every input ends in a clean error or in a
    structurally valid BAM whose records are a subset of the input's, in order — never a crash, never a hang."""
    import random
    gen = random.Random(20260)
    reads = [("record_%d" % i, ("TTAGGG" * (3 + i % 9)) if i % 3 else "ACGT" * (5 + i % 7)) for i in range(24)]
    header, records, _ = build_bam(reads, 60000)
    payload = header + b"".join(records)
    roff = len(header)
    good = bgzf(payload, 700)                                     # several blocks
    blocks = []
    pos = 0
    while pos < len(good):
        size = struct.unpack_from("<H", good, pos + 16)[0] + 1
        blocks.append(good[pos:pos + size]); pos += size
    paths = []
    for index in range(512):
        mode = index % 11
        if mode == 0:
            data = bytes(gen.getrandbits(8) for _ in range(gen.randrange(0, 3000)))
        elif mode == 1:
            data = good[:gen.randrange(len(good) + 1)]
        elif mode == 2:
            data = bgzf(payload[:gen.randrange(len(payload) + 1)], 700)
        elif mode == 3:
            m = bytearray(good); m[gen.randrange(len(m))] ^= 1 << gen.randrange(8); data = bytes(m)
        elif mode == 4:
            m = bytearray(payload); m[gen.randrange(len(m))] ^= 1 << gen.randrange(8); data = bgzf(bytes(m), 700)
        elif mode == 5:
            m = bytearray(payload); struct.pack_into("<i", m, roff, gen.randrange(-64, 4096)); data = bgzf(bytes(m), 700)
        elif mode == 6:
            m = bytearray(payload); m[roff + 12] = gen.randrange(256); data = bgzf(bytes(m), 700)
        elif mode == 7:
            m = bytearray(payload); struct.pack_into("<H", m, roff + 16, gen.randrange(65536)); data = bgzf(bytes(m), 700)
        elif mode == 8:
            m = bytearray(payload); struct.pack_into("<i", m, roff + 20, gen.randrange(-8, 1 << 20)); data = bgzf(bytes(m), 700)
        elif mode == 9:
            m = bytearray(good); at = gen.randrange(18); m[at] = gen.randrange(256); data = bytes(m)      # first block's gzip / BGZF header
        else:
            bl = list(blocks)
            k = gen.randrange(len(bl))
            if gen.random() < 0.5: del bl[k]
            else: bl.insert(k, bl[k])
            data = b"".join(bl)
        path = tmp_path / ("m%03d.bam" % index)
        path.write_bytes(data)
        paths.append(path)
    lst = tmp_path / "list.txt"
    lst.write_text("".join(str(p) + "\n" for p in paths))
    r = subprocess.run([cli, "--bam-subset-each", str(lst), "-x", "0", "-l", "18"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    record_set = {bytes(x) for x in records}
    n_ok = n_err = 0
    for index, path in enumerate(paths):
        ok, err, out = (tmp_path / (path.name + ext) for ext in (".ok", ".err", ".out"))
        assert ok.exists() != err.exists(), (index, "neither / both verdicts")
        if err.exists():
            n_err += 1
            assert err.read_text().strip() and not out.exists(), index
            continue
        n_ok += 1
        data = out.read_bytes()
        assert data.endswith(EOF_BLOCK), index
        plain = gunzip_members(data)
        at = plain.index(b"chr1\0") + 5 + 4 if b"chr1\0" in plain[:len(header) + 8] else None
        assert at is not None, index
        prev = -1
        while at < len(plain):                                    # a valid record stream ...
            (bs,) = struct.unpack_from("<i", plain, at)
            assert 32 <= bs and at + 4 + bs <= len(plain), index
            rec = plain[at:at + 4 + bs]
            if index % 11 in (1, 2):                              # ... of the input's own records, in order, where the damage kept them whole
                assert rec in record_set, index
                where = records.index(rec)
                assert where > prev, index
                prev = where
            at += 4 + bs
    assert n_ok >= 40 and n_err >= 200, (n_ok, n_err)             # both outcomes occur in numbers
