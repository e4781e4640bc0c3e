# Round 4, run 15: pooled contact walk (one contact list per decoy over the four waves and sub-lanes that hold it): parity, then A/B
O=gpurun_out/r04_run15
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_relax.py tests/test_gpu_selfcheck.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
for v in pool nopool; do
  unset TRX2FOLD_LIB; if [ $v = nopool ]; then export TRX2FOLD_LIB=$PWD/trrosettax2-dynamics_amd/_ab/libtrx2fold_nopool.so; fi
  for c in 2 3 4; do l=2; if [ $c = 3 ]; then l=1; fi; run 300 python3 tools/percall.py . $c $l 4 >> $O/percall_$v.txt 2>&1; echo "$v $(tail -1 $O/percall_$v.txt)"; done
  run 300 python3 tools/pool_sweep.py . 2 1280 640 >> $O/pool_$v.txt 2>&1; echo "$v $(tail -1 $O/pool_$v.txt)"
  run 300 python3 tools/pool_sweep.py . 3 640 320 >> $O/pool_$v.txt 2>&1; echo "$v $(tail -1 $O/pool_$v.txt)"
done
