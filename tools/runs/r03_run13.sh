# tolerance schedules (outcome vs evaluations), single-decoy stamps at L=150
O=gpurun_out/r13
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 300 python3 tools/tol_sweep.py $PWD 2 1000 cum > $O/tol_cum.txt 2>&1; echo "cum rc=$?"; cat $O/tol_cum.txt
run 900 python3 tools/tol_sweep.py $PWD 8 1000 sweep > $O/tol_sweep.txt 2>&1; echo "sweep rc=$?"; cat $O/tol_sweep.txt
TRX2FOLD_LIB=$PWD/trrosettax2-dynamics_amd/csrc/_exp/libtrx2fold_stamp.so run 300 python3 tools/stamp_single_decoy.py $PWD 150 > $O/stamp150.txt 2>&1; echo "stamp rc=$?"; cat $O/stamp150.txt
