"""Assemble profiles/rNN_kernel_trace.json from the `rocprofv3 --kernel-trace --stats` summaries under profiles/ (tools/r06_runs.sh traces):
per traced command the fold kernels' average duration, calls and share of kernel time, keyed by the kernel sources' SHA like the PMC records
(bench.py attaches `rocprof_avg_launch_ms` to its roofline records from it and drops a stale file).
usage: make_kernel_trace_json.py <round prefix, e.g. r06> <out.json>"""
import csv, hashlib, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre, out = sys.argv[1], sys.argv[2]
SOURCES = {"k_pair": ("kernel_pair.h", "trx2_device.h"), "k_step": ("kernel_step.h", "trx2_device.h")}
def sha(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(root, "trrosettax2-dynamics_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]
COMMANDS = {"metric_job": "bench.py --steps 4 --warmup 0 --nmax 40 --no-cpu-baseline --no-sub-records --no-e2e", "c2": "bench.py --config 2 --steps 5 --warmup 1 --no-cpu-baseline --no-legs",
            "c3": "bench.py --config 3 --steps 5 --warmup 1 --no-cpu-baseline --no-legs", "c4": "bench.py --config 4 --steps 5 --warmup 1 --no-cpu-baseline --no-legs",
            "batch16": "tools/e2e_batch.py . 150 16 40 16"}
rec = {"kernel_src_sha": {k: sha(v) for k, v in SOURCES.items()}, "method": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 <command>; the tool's own kernel_stats.csv", "traces": {}}
for tag, cmd in COMMANDS.items():
    p = os.path.join(root, "profiles", f"{pre}_{tag}_kernel_stats.csv")
    if not os.path.exists(p):
        continue
    rows = [r for r in csv.DictReader(open(p)) if r["Name"].startswith("void k_")]
    rec["traces"][tag] = {"command": cmd, "file": os.path.basename(p),
                          "kernels": [{"kernel": r["Name"].replace("void ", "").split("(")[0], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                       "share_of_kernel_time": float(r["Percentage"]) / 100.0} for r in rows[:6]]}
json.dump(rec, open(out, "w"), indent=1)
for tag, t in rec["traces"].items():
    print(tag, [(k["kernel"], round(k["avg_us"], 2), round(k["share_of_kernel_time"], 3)) for k in t["kernels"][:3]])
