# Round 4: does a surrogate closer to Rosetta's own per-residue energies (profiles/r04_pose_energies.txt: omega 13 x, bonded 25 x Rosetta's) fold
# closer to the reference's decoys?  1024 decoys per map and variant, default protocol.  Variants are builds of the same library with one constant
# changed (trrosettax2-dynamics_amd/_scan/, made by hand: hipcc ... -DTRX2_OMEGA_K=.. / -DTRX2_CART_KSCALE=.. / -DTRX2_RAMA_GUARD_OFFSET=..).
O=gpurun_out/r04_model_scan
mkdir -p $O
timeout -k 10 300 python3 tools/outcome_sample.py . 16 1000 --fastrelax > $O/base.txt 2>&1; cat $O/base.txt
for v in omega02 omega01 cart02 cart004 guard13 omega02cart02; do
  TRX2FOLD_LIB=$PWD/trrosettax2-dynamics_amd/_scan/libtrx2fold_$v.so timeout -k 10 300 python3 tools/outcome_sample.py . 16 1000 --fastrelax > $O/$v.txt 2>&1; cat $O/$v.txt
done
