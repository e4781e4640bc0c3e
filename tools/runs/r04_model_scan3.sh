# third model scan: the softer omega / bonded surrogates in the RELAX stage only (weights of its runs scaled, TRX2_SF_FA_SCALE), the centroid
# stage at the constants of rounds 1-3; with and without the guard offset; default protocol and --no-fastrelax
O=gpurun_out/r04_model_scan3
mkdir -p $O
S=$PWD/trrosettax2-dynamics_amd/_scan
for s in 1000 3000; do
for spec in "orig 1,1" "orig 0.4,0.4" "origg13 0.4,0.4" "orig 0.4,1" "orig 1,0.4" "origg13 0.25,0.4"; do
  set -- $spec
  echo "== lib $1 relax scale (omega,bonded) $2 seeds $s"
  TRX2_SF_FA_SCALE=$2 TRX2FOLD_LIB=$S/libtrx2fold_$1.so timeout -k 10 300 python3 tools/outcome_sample.py . 16 $s --fastrelax 2>&1 | grep "n=1024"
done; done > $O/scan.txt 2>&1
echo "== --no-fastrelax, lib origg13" >> $O/scan.txt
TRX2FOLD_LIB=$S/libtrx2fold_origg13.so timeout -k 10 300 python3 tools/outcome_sample.py . 16 5000 2>&1 | grep "n=1024" >> $O/scan.txt
echo "== --no-fastrelax, lib orig" >> $O/scan.txt
TRX2FOLD_LIB=$S/libtrx2fold_orig.so timeout -k 10 300 python3 tools/outcome_sample.py . 16 5000 2>&1 | grep "n=1024" >> $O/scan.txt
cat $O/scan.txt
