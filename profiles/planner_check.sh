#!/bin/bash
# Does plan_geometry's choice between one workgroup of 16 waves and two of 10 hold on parameter sets it was not fitted on?
# kernel_ms on a 1 Gb synthetic assembly: the planner's own choice against the search pinned to 16 waves ("16,0").
cd "$(dirname "$0")/.."
run() {   # geometry flags
  if [ "$1" = auto ]; then unset TS_GEOMETRY; else export TS_GEOMETRY=$1; fi
  TS_TIMING=1 timeout -k 10 120 python3 bench.py --gbases 1 --contigs 67 --no-cpu-baseline --no-e2e --flags "$2" 2> /tmp/pc.err \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %-5s kernel %.4f ms  frac %.4f' % ('$1', d['roofline']['kernel_ms'], d['roofline']['frac']), end='')" || tail -n 2 /tmp/pc.err
  grep -a ts_batch_create /tmp/pc.err | head -1 | sed 's/.*pair table), /   /; s/ per tile.*//'
}
while read -r flags; do
  echo "$flags"
  run auto "$flags"; run 16,0 "$flags"; run auto "$flags"; run 16,0 "$flags"
done <<'LIST'
-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -x 1 -w 1000 -s 500 -r -g -e -m -i
-c TTAGGG -r -g -e -m -i
-w 500 -s 250 -r -g -e -i
-w 200 -s 200 -r -g
-w 1000 -s 100 -g -e -i
-w 1000 -s 300 -r -g -e -i
-w 2000 -s 1000 -r -g -e -m -i
-w 5000 -s 2500 -r -g -e -m -i
-c TTAGG -w 1000 -s 500 -g -e -i
-x 0 -w 1000 -s 500 -g -e -i
LIST
