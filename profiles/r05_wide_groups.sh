# the wide form's group size (TS_WIDE_GROUP_MB; default 256): end to end on the nine-length set, interleaved on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for mb in 256 512 128; do
  echo "== TS_WIDE_GROUP_MB=$mb"
  TS_WIDE_GROUP_MB=$mb TS_GEN_ONLY=wide_9_lengths timeout -k 10 200 python profiles/general_path_rate.py 3.0 2>/dev/null | grep -E "gbases_per_s" | tr -d '\n'; echo
done; done
