"""What runs beside what: from a rocprofv3 --kernel-trace CSV (kernel_trace.csv: Kernel_Name, Start_Timestamp, End_Timestamp), the share of
wall time in every (pair kernels running, step kernels running) state between the first and the last shared launch, and the mean durations.
Only the LAST <tail fraction> of the span between the first and the last such launch is read (the warm-up of a measurement script comes first).
usage: overlap_timeline.py <kernel_trace.csv> [tail fraction = 0.6] [name fragment of the pair kernel = k_pair1_multi] [step = k_step_multi]"""
import csv, sys
from collections import defaultdict
path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
pk = sys.argv[3] if len(sys.argv) > 3 else "k_pair1_multi"
sk = sys.argv[4] if len(sys.argv) > 4 else "k_step_multi"
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        kind = "P" if pk in n else "S" if sk in n else None
        if kind is not None:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
t_lo, t_hi = min(r[0] for r in rows), max(r[1] for r in rows)
cut = t_hi - frac * (t_hi - t_lo)
ev, dur = [], defaultdict(list)
for s, e, kind in rows:
    if s >= cut:
        ev.append((s, 1, kind)); ev.append((e, -1, kind)); dur[kind].append(e - s)
ev.sort()
cnt = {"P": 0, "S": 0}
state_ns = defaultdict(int)
prev = ev[0][0]
for t, d, kind in ev:
    state_ns[(min(cnt["P"], 2), min(cnt["S"], 2))] += t - prev
    prev = t
    cnt[kind] += d
tot = sum(state_ns.values())
print(f"{len(dur['P'])} pair launches (mean {sum(dur['P']) / len(dur['P']) / 1e3:.1f} us), {len(dur['S'])} step launches (mean {sum(dur['S']) / len(dur['S']) / 1e3:.1f} us), span {tot / 1e6:.1f} ms")
for (p, s), ns in sorted(state_ns.items()):
    print(f"  pair kernels running {p}{'+' if p == 2 else ' '} step kernels running {s}{'+' if s == 2 else ' '}: {100.0 * ns / tot:5.1f} % of the span")
