# the read step with smaller resident sub-batches (TS_BENCH_READ_SUB): do a sub-batch's records, written by its scan and read by its
# predicate right behind it, stay in the 256 MB memory-side cache?  bench.py --reads at 1e6 reads, interleaved on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for sub in 500000 125000 62500 250000; do
  echo -n "sub=$sub: "
  TS_BENCH_READ_SUB=$sub timeout -k 10 280 python bench.py --reads --n-reads 1000000 --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
