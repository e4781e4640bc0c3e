O=gpurun_out/r38
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k "pooled_shape" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E 'eval of|pooled fold|passed|failed|Error' $O/pytest.txt | cut -c1-300
run 900 python3 bench.py --no-cpu-baseline --no-e2e --no-sub-records --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=d['pooled_queue']
print('value',round(d['value'],1),'pooled',round(p['value'],1), p['workload'][:120])
print(' pooled pair', p['roofline']); print(' pooled step', p['roofline_step'])
PY
