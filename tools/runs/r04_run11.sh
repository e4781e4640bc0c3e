O=gpurun_out/r04_run11
mkdir -p $O
for w in 4 1; do for c in 0 1; do TRX2_SEG_CACHE=$c timeout -k 5 120 python3 tools/diag_segcache2.py . $w $O/ev_w${w}_c$c.npy > $O/log_w${w}_c$c.txt 2>&1; echo "w=$w c=$c rc=$?"; done; done
python3 - <<'PY'
import numpy as np
for w in (4, 1):
    a = np.load(f"gpurun_out/r04_run11/ev_w{w}_c0.npy"); b = np.load(f"gpurun_out/r04_run11/ev_w{w}_c1.npy")
    d = np.abs(a - b)
    print("waves", w, "rows", a.shape, "max abs diff", d.max(), "rows differing", int((d.max(1) > 0).sum()), "first terms c0", a[0, :9], "c1", b[0, :9])
    rel = d / (np.abs(a) + 1e-30)
    print("   max rel diff over entries with |a|>1e-3:", rel[np.abs(a) > 1e-3].max())
    i, j = np.unravel_index(d.argmax(), d.shape); print("   worst at row", i, "col", j, a[i, j], b[i, j])
PY
