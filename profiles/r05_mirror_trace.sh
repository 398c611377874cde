# scanFastaToFiles at 3 Gb with its per-group stage intervals (TS_MIRROR_TRACE=1): where the wall time beyond the slowest stage goes
set -e
cd $GRAFT_REPO_ROOT
KEEP=1 bash profiles/writers_rate.sh > gpurun_out/writers_rate_now.txt 2>&1 || { tail -5 gpurun_out/writers_rate_now.txt; exit 1; }
grep -E '^manifest_cli' gpurun_out/writers_rate_now.txt | head -2 | cut -c1-260
FLAGS="-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -i"
for run in 1 2; do
TS_MIRROR_TRACE=1 /tmp/manifest_cli -f /tmp/writers_rate.fa --out-base /tmp/writers_rate $FLAGS 2> gpurun_out/mirror_trace_$run.txt >/dev/null
done
grep -E "^trace" gpurun_out/mirror_trace_2.txt > gpurun_out/mirror_trace_2_lines.txt || true; df /tmp | tail -1; mount | grep -E " /tmp | / " | head -3
head -70 gpurun_out/mirror_trace_2_lines.txt
grep -E "manifest_cli" gpurun_out/mirror_trace_2.txt | cut -c1-260 || true
rm -f /tmp/writers_rate*
