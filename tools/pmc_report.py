"""Mean per-launch counter values of the last N dispatches of one kernel in a rocprofv3 --pmc run.
usage: pmc_report.py <counter_collection.csv> [N=40] [kernel name part = k_pair] [skip the last M dispatches = 0]   (prints one JSON object)
(skip: the last chunks an engine launches for a fold that has already reported leave at once -- no traffic, no instructions)"""
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
part = sys.argv[3] if len(sys.argv) > 3 else "k_pair"
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sel = [r for r in rows if part in r["Kernel_Name"]]
if part == "k_pair":  # the narrow instantiations of a fold's tail are not the replayed shape: keep the kernel of the LAST dispatch
    last = max(sel, key=lambda r: int(r["Dispatch_Id"]))["Kernel_Name"]
    sel = [r for r in sel if r["Kernel_Name"] == last]
ids = sorted({int(r["Dispatch_Id"]) for r in sel})
ids = (ids[:-skip] if skip else ids)[-n:]
tab = collections.defaultdict(dict)
for r in sel:
    d = int(r["Dispatch_Id"])
    if d in ids:
        tab[d][r["Counter_Name"]] = tab[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({k for v in tab.values() for k in v})
print(json.dumps({"kernel": sel[-1]["Kernel_Name"].split("(")[0].replace("void ", ""), "launches": len(ids), **{k: sum(tab[d].get(k, 0.0) for d in ids) / len(ids) for k in names}}))
