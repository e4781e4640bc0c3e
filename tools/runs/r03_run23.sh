# candidates extension: its test, and the bench line with the e2e records
O=gpurun_out/r23
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 -m pytest tests/test_gpu_boundary.py -m gpu -x -q -s > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
run 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; echo "bench rc=$?"
python3 - $O/bench_s20.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value',round(d['value'],1),'pooled',round(d['pooled_queue']['value'],1),'inflight',round(d['in_flight_B']['value'],1),'single',round(d['single_stream']['value'],1),'c3',round(d['sub_records']['config3']['value'],1),'c4',round(d['sub_records']['config4']['value'],1))
for k,v in d['e2e'].items(): print(k, round(v['value'],1), 'decoys/s', v['decoys_written'], 'decoys', round(v['wall_s'],1), 's', round(v['ms_per_iteration'],1), 'ms/iter')
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'traffic', d['roofline']['traffic'], 'step', d['roofline_step']['avg_launch_ms'])
PY
