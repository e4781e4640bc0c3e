# Cartesian role in Gram form: parity tests, stamps, A/B against the two-loop build
O=gpurun_out/r14
mkdir -p $O
X=$PWD/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 -m pytest tests/test_gpu_cartesian.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_selfcheck.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
TRX2FOLD_LIB=$X/libtrx2fold_stamp.so run 300 python3 tools/stamp_single_decoy.py $PWD 150 > $O/stamp150.txt 2>&1; echo "stamp rc=$?"; tail -14 $O/stamp150.txt
for lib in "" $X/libtrx2fold_twoloop.so; do
  echo "== lib=$lib"
  for cfg in "2 2" "3 1" "4 2"; do
    TRX2FOLD_LIB=$lib run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1
  done
  TRX2FOLD_LIB=$lib run 300 python3 tools/single_decoy_trace.py $PWD 150 1 8 2>&1 | tail -2
done > $O/ab.txt 2>&1; cat $O/ab.txt
for lib in "" $X/libtrx2fold_twoloop.so; do
  TRX2FOLD_LIB=$lib run 600 python3 bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2> $O/bench_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',round(d['value'],1),'pooled',round(d['pooled_queue']['value'],1),'inflight',round(d['in_flight_B']['value'],1),'single',round(d['single_stream']['value'],1),'c3',round(d['sub_records']['config3']['value'],1),'c4',round(d['sub_records']['config4']['value'],1),'step_ms',d['roofline_step']['avg_launch_ms'],'pooled_step_ms',d['pooled_queue']['roofline_step']['avg_launch_ms'])"
done > $O/bench_ab.txt 2>&1; cat $O/bench_ab.txt
