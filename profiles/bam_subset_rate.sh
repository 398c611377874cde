#!/bin/bash
# Row f4 at scale: --bam-subset -l 42 on an unaligned HiFi-like BAM (N(15 kb, 3 kb) reads, 0.5 % telomeric, qualities
# and a few tags; BGZF at zlib level 1 like a basecaller's output) through the C++ mirror
# (tests/cpp/manifest_cli.cpp --bam-subset).  Run on the GPU box.   usage: profiles/bam_subset_rate.sh [reads]
set -e
cd "$(dirname "$0")/.."
N=${1:-60000}
python3 - $N <<'PY'
import numpy as np, struct, sys, zlib, time
sys.path.insert(0, '.')
from tests import seqgen
n = int(sys.argv[1])
rng = np.random.default_rng(43)
lens = np.clip(rng.normal(15000, 3000, size=n), 1000, 40000).astype(np.int64)
pool = seqgen.random_dna(rng, int(lens.sum()))
offs = np.concatenate(([0], np.cumsum(lens)))
for i in np.flatnonzero(rng.random(n) < 0.005):
    ln = int(rng.integers(300, 8000))
    t = seqgen.mutate(rng, seqgen.repeat_array("TTAGGG", ln // 6 + 1), 0.01)[:min(ln, lens[i])]
    pool[offs[i]:offs[i] + len(t)] = t
code = np.zeros(256, dtype=np.uint8)
for ch, v in zip(b"=ACMGRSVTWYHKDBN", range(16)):
    code[ch] = v
nib = code[pool]
text = b"@HD\tVN:1.6\tSO:unknown\n"
t0 = time.time()
raw_bytes = 0
with open('/tmp/reads.bam', 'wb') as fh:
    pend = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0))
    def flush(final=False):
        global pend
        while len(pend) >= 65280 or (final and pend):
            piece = bytes(pend[:65280]); del pend[:65280]
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            payload = co.compress(piece) + co.flush()
            fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload +
                     struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
    for i in range(n):
        s = nib[offs[i]:offs[i + 1]]
        L = len(s)
        if L & 1:
            s = np.concatenate((s, np.zeros(1, dtype=np.uint8)))
        packed = ((s[0::2] << 4) | s[1::2]).astype(np.uint8).tobytes()
        name = b"m64011_%d/ccs" % i
        qual = rng.integers(20, 60, size=L, dtype=np.uint8).tobytes()       # noisy qualities: compress like real ones (badly)
        tags = b"npC\x08rqf" + struct.pack("<f", 0.999)
        body = struct.pack("<iiBBHHHiiii", -1, -1, len(name) + 1, 255, 4680, 0, 4, L, -1, -1, 0) + name + b"\0" + packed + qual + tags
        pend += struct.pack("<i", len(body)) + body
        raw_bytes += 4 + len(body)
        flush()
    flush(True)
    fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
print("reads %d, bases %d, uncompressed BAM %.2f GB, written in %.0f s" % (n, int(lens.sum()), raw_bytes / 1e9, time.time() - t0))
PY
g++ -std=c++17 -O2 -I include tests/cpp/manifest_cli.cpp -L teloscope_amd -lteloscan -Wl,-rpath,$PWD/teloscope_amd -pthread -lz -o /tmp/manifest_cli
/tmp/manifest_cli --bam-subset -l 42 /tmp/reads.bam > /tmp/kept.bam     # warm-up (page cache, device init)
for r in 1 2; do
t0=$(date +%s%N); TS_TIMING=1 /tmp/manifest_cli --bam-subset -l 42 /tmp/reads.bam > /tmp/kept.bam; t1=$(date +%s%N); echo "wall $(( (t1 - t0) / 1000000 )) ms"
done
ls -la /tmp/reads.bam /tmp/kept.bam | awk '{print $5, $9}'
