# Round 4, run 29: relax-stage scale of the omega / bonded surrogates (protocol.SF_FA_SCALE) against global AND torsion-level outcome
O=gpurun_out/r04_run29
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for sc in "1.0,0.4" "1.0,1.0" "0.4,1.0" "2.0,0.4"; do
  echo "## SF_FA_SCALE (omega, bonded) = $sc" >> $O/outcome.txt
  TRX2_SF_FA_SCALE=$sc run 900 python3 tools/outcome_sample.py . 8 1000 --fastrelax >> $O/outcome.txt 2>> $O/err.txt || exit 1
done
cut -c1-330 $O/outcome.txt
