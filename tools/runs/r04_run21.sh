# Round 4, run 21: self-checking build over every step-kernel instantiation (incl. shared launches and the 256-register kernels); shape soak on the closing build
O=gpurun_out/r04_run21
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_selfcheck.py tests/test_context_cache.py -q -s > $O/selfcheck.txt 2>&1; echo "selfcheck rc=$?"; grep -E "^ [0-9]+ |passed|failed" $O/selfcheck.txt | cut -c1-260
run 900 python3 tools/soak_shapes.py . 30 7 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -3 $O/soak.txt | cut -c1-200
run 600 python3 tools/soak_shapes.py . 8 9 big > $O/soak_big.txt 2>&1; echo "soak big rc=$?"; tail -2 $O/soak_big.txt | cut -c1-200
