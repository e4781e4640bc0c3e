# Round-4 closing sequence on the GPU box: the GPU suite, the profile records (copied to profiles/ afterwards), the bench lines.
R=$PWD
O=gpurun_out/r04_final
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1150 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
run 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-200
rm -rf gpurun_out/r04_profiles
bash tools/runs/r04_profiles.sh > $O/profiles.log 2>&1; echo "profiles rc=$?"; tail -12 $O/profiles.log
cp gpurun_out/r04_profiles/r04_traffic.json profiles/r04_traffic.json   # on the box: the bench below reads it
run 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.json
run 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; echo "bench rc=$?"; tail -c 300 $O/bench_s20.json
