"""Mean per-launch counter values of the last N k_pair dispatches in a rocprofv3 --pmc run.
usage: pmc_report.py <counter_collection.csv> [N=40] [out.csv]   (prints one JSON object; out.csv = per-dispatch table)"""
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pair = [r for r in rows if "k_pair" in r["Kernel_Name"]]
ids = sorted({int(r["Dispatch_Id"]) for r in pair})[-n:]
tab = collections.defaultdict(dict)
for r in pair:
    d = int(r["Dispatch_Id"])
    if d in ids:
        tab[d][r["Counter_Name"]] = tab[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({k for v in tab.values() for k in v})
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        f.write("dispatch_id," + ",".join(names) + "\n")
        for d in ids:
            f.write(str(d) + "," + ",".join(str(tab[d].get(k, "")) for k in names) + "\n")
print(json.dumps({"kernel": pair[0]["Kernel_Name"].split("(")[0], "launches": len(ids), **{k: sum(tab[d].get(k, 0.0) for d in ids) / len(ids) for k in names}}))
