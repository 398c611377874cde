#!/bin/bash
# read filter (configs[3] at 5e5 reads per step): round 4's tree against this one, and tilings of this one (TS_GEOMETRY = waves,chunks)
cd "$(dirname "$0")/.."
rr() {   # rr <label> <dir> [geometry]
  (cd $2 && TS_GEOMETRY=$3 timeout -k 10 240 python3 bench.py --reads --n-reads ${NREADS:-5e5} --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline 2> /tmp/rg.err \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-14s value %.0f Gbases/s  ms_per_step %.4f  frac %.4f  scan kernel %.4f ms' % ('$1', d['value'], d['ms_per_step'], d['value'] / 8000.0, d['roofline']['kernel_ms']))" || tail -n 3 /tmp/rg.err)
}
for r in 1 2; do
  rr r04 ab_old
  rr r05 .
  for g in $GEOMS; do rr "r05 $g" . $g; done
done
