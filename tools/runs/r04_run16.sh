# Round 4, run 16: torsion history of 257-512-residue chains staged in LDS (Cartesian role's arrays in the dynamic buffer): suite + config 4 timing + kernel trace
O=gpurun_out/r04_run16
mkdir -p $O
R=$PWD
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1150 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt | cut -c1-200
run 300 python3 tools/percall.py . 4 2 4 >> $O/percall.txt 2>&1; tail -1 $O/percall.txt
run 300 python3 tools/percall.py . 4 2 4 >> $O/percall.txt 2>&1; tail -1 $O/percall.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt4
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -- python3 $R/tools/percall.py $R 4 2 3 > $R/$O/prof.log 2>&1
f=$(find /tmp/kt4 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/c4_kernel_stats.csv && cut -d, -f1-4 $R/$O/c4_kernel_stats.csv | head -4
