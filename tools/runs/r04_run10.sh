O=gpurun_out/r04_run10
mkdir -p $O
for ev in 3 30 300 0; do for w in 4 1; do for sh in 0 1; do for c in 0 1; do
  echo "evals=$ev waves=$w shared=$sh cache=$c"; TRX2_SEG_CACHE=$c timeout -k 5 120 python3 tools/diag_segcache.py . $sh $w $ev 2>&1 | grep "^(" | tr '\n' ' '; echo
done; done; done; done > $O/diag.txt 2>&1
cat $O/diag.txt
