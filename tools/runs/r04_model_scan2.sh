# second model scan: combinations, on two disjoint sets of 1024 decoys per map (seeds 1000.. and 3000..)
O=gpurun_out/r04_model_scan2
mkdir -p $O
for s in 1000 3000; do
timeout -k 10 300 python3 tools/outcome_sample.py . 16 $s --fastrelax > $O/base_$s.txt 2>&1; cat $O/base_$s.txt
for v in o02c02g13 o02c01g13 o03c03g13 o02c04g13 o015c02g13; do
  TRX2FOLD_LIB=$PWD/trrosettax2-dynamics_amd/_scan/libtrx2fold_$v.so timeout -k 10 300 python3 tools/outcome_sample.py . 16 $s --fastrelax > $O/${v}_$s.txt 2>&1; cat $O/${v}_$s.txt
done; done
