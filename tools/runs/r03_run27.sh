# two-rank rehearsal of bench.py on the one GPU of the box (gloo, both ranks on device 0): the N > 1 control flow of the bench line
O=gpurun_out/r27
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
export TRX2_BENCH_FORCE_DEVICE=0
run 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_n2.json 2> $O/bench_n2.err; echo "bench n2 rc=$?"
tail -c 1500 $O/bench_n2.json; tail -5 $O/bench_n2.err
