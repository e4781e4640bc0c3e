R=$PWD
O=gpurun_out/r8
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1150 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
rm -rf gpurun_out/r03_profiles
bash tools/runs/r03_profiles.sh > $O/profiles.log 2>&1; echo "profiles rc=$?"; tail -9 $O/profiles.log
