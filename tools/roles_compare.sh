# k_step launch lengths with and without the Cartesian stage, this build against another one.  usage: roles_compare.sh [other .so]
set -e
R=$PWD; export TMPDIR=/tmp; cd /tmp
for v in new other; do
  if [ $v = other ]; then [ -n "$1" ] || continue; export TRX2FOLD_LIB=$R/$1; else unset TRX2FOLD_LIB; fi
  rm -rf /tmp/roles_$v
  timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/roles_$v -- python3 $R/tools/step_roles.py $R > $R/gpurun_out/roles_$v.txt 2>&1
  f=$(find /tmp/roles_$v -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/step_roles.py --report $f >> $R/gpurun_out/roles_$v.txt
done
