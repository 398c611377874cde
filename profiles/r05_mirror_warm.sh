set -e
cd $GRAFT_REPO_ROOT
KEEP=1 bash profiles/writers_rate.sh > gpurun_out/writers_rate_warm.txt 2>&1 || { tail -5 gpurun_out/writers_rate_warm.txt; exit 1; }
grep -E '^manifest_cli' gpurun_out/writers_rate_warm.txt | head -2 | cut -c1-200
FLAGS="-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -i"
for rep in 1 2 3; do for w in 1 0; do
  echo -n "TS_MIRROR_WARMUP=$w: "
  TS_MIRROR_WARMUP=$w TS_MIRROR_TRACE=1 TS_TIMING=1 /tmp/manifest_cli -f /tmp/writers_rate.fa --out-base /tmp/writers_rate $FLAGS 2> gpurun_out/mirror_warm_$w.txt >/dev/null
  grep -E "^manifest_cli" gpurun_out/mirror_warm_$w.txt | cut -c26-110 | tr -d '\n'; echo -n " | "; grep -E "^trace (warm|scan)" gpurun_out/mirror_warm_$w.txt | head -3 | tr '\n' ';'; echo
done; done
rm -f /tmp/writers_rate*
