#!/bin/bash
# One workgroup of 16 waves per CU against two of 10 (the 80-VGPR build of the scan kernel; each workgroup has its own table
# copy and half of the LDS, so the tiles are shorter): kernel_ms per pinned geometry (TS_GEOMETRY=waves,chunks; 10 waves = the
# two-workgroup mode), three configurations and the read batches.  "auto" = what plan_geometry picks.
cd "$(dirname "$0")/.."
PLANT="-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i"
DEFLT="-c TTAGGG -r -g -e -m -i"
run() {   # geometry label [flags]
  if [ "$1" = auto ]; then unset TS_GEOMETRY; else export TS_GEOMETRY=$1; fi
  TS_TIMING=1 timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-e2e ${3:+--flags "$3"} 2> /tmp/occ.err \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-5s %-9s kernel %.4f ms  frac %.4f' % ('$1', '$2', d['roofline']['kernel_ms'], d['roofline']['frac']))" || tail -n 2 /tmp/occ.err
  grep -a ts_batch_create /tmp/occ.err | head -1 | sed 's/.*pair table), /      /'
}
rr() {
  if [ "$1" = auto ]; then unset TS_GEOMETRY; else export TS_GEOMETRY=$1; fi
  timeout -k 10 200 python3 bench.py --reads --no-cpu-baseline 2> /tmp/occ.err \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-5s reads     value %.0f Gbases/s  ms_per_step %.3f  scan kernel %.3f ms' % ('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" || tail -n 2 /tmp/occ.err
}
for r in $(seq 1 ${ROUNDS:-2}); do
  for g in 16,8 10,6 auto; do run $g headline; done
  for g in 16,8 10,6 auto; do run $g default "$DEFLT"; done
  for g in 16,8 10,5 auto; do run $g plant "$PLANT"; done
  for g in 16,6 10,6 auto; do rr $g; done
done
