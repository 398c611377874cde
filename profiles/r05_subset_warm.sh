# --fastq-subset through the C++ mirror with and without the library's first call made beside the first block's read (TS_MIRROR_WARMUP),
# whole-process wall times (date around the process), interleaved; and the subset tests
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_io_selftest.py tests/test_bam_subset.py tests/test_cpp_mirror.py -q -x -m gpu -k "fastq or bam or subset or selftest" > gpurun_out/subset_tests.log 2>&1 || { tail -20 gpurun_out/subset_tests.log; exit 1; }
tail -1 gpurun_out/subset_tests.log
bash profiles/fastq_subset_rate.sh > gpurun_out/fastq_rate_warm.txt 2>&1 || { tail gpurun_out/fastq_rate_warm.txt; exit 1; }
grep -E "wall|reads" gpurun_out/fastq_rate_warm.txt | head -4
for rep in 1 2 3; do for w in 1 0; do
  t0=$(date +%s%N); TS_MIRROR_WARMUP=$w /tmp/manifest_cli --fastq-subset -l 42 /tmp/reads.fq > /tmp/kept_$w.fq; t1=$(date +%s%N)
  echo "TS_MIRROR_WARMUP=$w wall $(( (t1 - t0) / 1000000 )) ms"
done; done
cmp /tmp/kept_0.fq /tmp/kept_1.fq && echo "# identical output"
