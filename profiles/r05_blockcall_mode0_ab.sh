# the sharded step (emitting scan + pack beside the next scan) with the library of commit 659adff (before blockcall.hip's kernels
# became templates with a MODE 1 for the general path) against this tree's, interleaved on one box: MODE 0 must cost what it did
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_old.so; else unset TELOSCAN_LIB; fi
    echo -n "$lib: "; timeout -k 10 200 python profiles/pack_abl_time.py 60 2>/dev/null | tail -1 | cut -c1-300
  done
done
