-f testFiles/gfa_internal_only_small.gfa -j 1 -x 0 -l 60 -t 100 -o %OUTDIR%
expect_exit 0
expect_stdout ignore
expect_output_name gfa_internal_only_small.gfa.telo.annotated.gfa
expect_gfa_header 1.2
gfa_expect testFiles/expected/gfa/empty.tsv
gfa_preserve_input strict
expect_gfa_colors 1
