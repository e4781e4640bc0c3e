"""Rewrites the number-carrying rows of DESIGN.md section 5 and README.md's summary from profiles/r02_bench.json, r02_traffic.json and
r02_c2_kernel_stats.csv (the prose around them is edited by hand).  usage: refresh_docs.py"""
import csv, json, os, re
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(os.path.join(root, "profiles", "r02_bench.json")).read().strip().splitlines()[-1])
sr = d["sub_records"]
t = json.load(open(os.path.join(root, "profiles", "r02_traffic.json")))
st = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(os.path.join(root, "profiles", "r02_c2_kernel_stats.csv")))}
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
def sub(pattern, repl):
    global s
    new, n = re.subn(pattern, lambda m: repl, s, count=1, flags=re.S)
    assert n == 1, pattern[:60]
    s = new
c3, c4 = sr["config3"], sr["config4"]
sub(r"\| \*\*decoys/sec\*\* \(round 1 → first half of round 2 → now\) \|[^\n]*\n",
    f"| **decoys/sec** (round 1 → first half of round 2 → now) | 264.6 → 396.7 → **{d['value']:.1f}** | 396 → 486 → **{c3['value']:.0f}** | 40 → 69.8 → **{c4['value']:.1f}** |\n")
sub(r"\| same queue, init_num decoys in flight: 2 lanes × 32 slots / one stream of 64 \|[^\n]*\n",
    f"| same queue, init_num decoys in flight: 2 lanes × 32 slots / one stream of 64 | {d['in_flight_B']['value']:.1f} / {d['single_stream']['value']:.1f} (384 / 326 at the start of the second half) | | |\n")
sub(r"\| K separate calls, one slot per decoy \(round 1's definition\) \|[^\n]*\n", f"| K separate calls, one slot per decoy (round 1's definition) | {d['per_call']['value']:.1f} | | |\n")
sub(r"\| slot efficiency = [^\n]*\n",
    f"| slot efficiency = Σ evaluations / Σ (launch pairs × slots they served); tail compaction off → on | 0.67 → {d['slot_efficiency']:.2f} | 0.68 → {c3['slot_efficiency']:.2f} | 0.71 → {c4['slot_efficiency']:.2f} |\n")
sub(r"\| `k_pair` launch on final coordinates \(HIP events\) / live average over the fold \|[^\n]*\n",
    f"| `k_pair` launch on final coordinates (HIP events) / live average over the fold | `<64>`, 160 decoys: {d['roofline']['avg_launch_ms']*1e3:.1f} / {d['roofline']['avg_launch_ms_over_fold']*1e3:.1f} µs (the fold shrinks to 128, 64, 32 .. decoys per launch) | `<64>`, 128 decoys: {c3['roofline']['avg_launch_ms']*1e3:.1f} µs | `<32>`, 32 decoys: {c4['roofline']['avg_launch_ms']*1e3:.1f} µs |\n")
sub(r"\| `k_step` live average \|[^\n]*\n",
    f"| `k_step` live average | {d['roofline_step']['avg_launch_ms']*1e3:.1f} µs | {c3['roofline_step']['avg_launch_ms']*1e3:.1f} µs | {c4['roofline_step']['avg_launch_ms']*1e3:.1f} µs |\n")
sub(r"\| algorithmic bytes per launch → achieved / 8000 GB/s \(`roofline.frac`\) \|[^\n]*\n",
    f"| algorithmic bytes per launch → achieved / 8000 GB/s (`roofline.frac`) | {d['roofline']['algorithmic_bytes_per_launch']/1e6:.1f} MB → {d['roofline']['achieved']:.0f} GB/s = **{100*d['roofline']['frac']:.1f} %** | {c3['roofline']['algorithmic_bytes_per_launch']/1e6:.1f} MB → {c3['roofline']['achieved']:.0f} GB/s = **{100*c3['roofline']['frac']:.1f} %** | {c4['roofline']['algorithmic_bytes_per_launch']/1e6:.1f} MB → {c4['roofline']['achieved']:.0f} GB/s = **{100*c4['roofline']['frac']:.1f} %** |\n")
sub(r"\| HBM-side traffic per launch \(PMC; `roofline.traffic`\) \|[^\n]*\n",
    f"| HBM-side traffic per launch (PMC; `roofline.traffic`) | {t['2']['hbm_bytes_per_launch']/1e6:.1f} MB | {t['3']['hbm_bytes_per_launch']/1e6:.1f} MB | {t['4']['hbm_bytes_per_launch']/1e6:.1f} MB |\n")
sub(r"\| VALU instructions per wave / wave time on `s_waitcnt` \(PMC\) \|[^\n]*\n",
    "| VALU instructions per wave / wave time on `s_waitcnt` (PMC) | " + " | ".join(f"{t[k]['valu_insts_per_launch']/t[k]['waves']:.0f} / {100*t[k]['wait_any_quad_cycles']/t[k]['wave_quad_cycles']:.0f} %" for k in "234") + " |\n")
sub(r"\| CPU oracle \(`kind: port`\): 1 thread / OpenMP over decoys on the 16 usable cores \(256 present\) \|[^\n]*\n",
    f"| CPU oracle (`kind: port`): 1 thread / OpenMP over decoys on the 16 usable cores (256 present) | {d['cpu_baseline']['single_thread']['value']:.2f} / **{d['cpu_baseline']['value']:.1f} decoys/s** | | |\n")
kp, ks = st["k_pair<64, 7>"], st["k_step<1, 256, 256>"]
a = s.index("Kernel trace of the headline queue alone (`tools/bench_stats.sh`")
b = s.index("**Where the gain of the second half came from**")
s = s[:a] + f"""Kernel trace of the headline queue alone (`tools/bench_stats.sh`: `rocprofv3 --kernel-trace --stats -- python3 bench.py
--no-cpu-baseline --no-sub-records --no-legs`, `profiles/r02_c2_kernel_stats.csv`): `k_pair<64>` {float(kp['AverageNs'])/1e3:.1f} µs average over {int(kp['Calls'])}
launches (three-, two- and one-group launches of the fold under the profiler and the 200 three-group replays on final
coordinates at {d['roofline']['avg_launch_ms']*1e3:.1f} µs that `roofline.achieved` is taken from), {float(kp['Percentage']):.0f} % of kernel time; fused `k_step` {float(ks['AverageNs'])/1e3:.1f} µs, {float(ks['Percentage']):.0f} %;
`k_pair<32>` / `<16>` / `<1>` .. are the warm-up step and the narrowing tail. `roofline_step`: bytes the step must move — pair
records, history, state in and out, coordinates — over its live average: {d['roofline_step']['algorithmic_bytes_per_launch']/1e6:.1f} MB / {d['roofline_step']['avg_launch_ms']*1e3:.0f} µs = {d['roofline_step']['achieved']:.0f} GB/s = {100*d['roofline_step']['frac']:.1f} % of
the HBM roof; it is a chain of ~25 dependent phases on one workgroup per slot, bound by latency, not bandwidth.

""" + s[b:]
s = re.sub(r"`s_waitcnt`\. The CPU ratio: GPU \d+ decoys/s against [\d.]+ on the 16 cores this process may use \(\d+×; [\d.]+ on one\ncore: \d+×\);",
           f"`s_waitcnt`. The CPU ratio: GPU {d['value']:.0f} decoys/s against {d['cpu_baseline']['value']:.1f} on the 16 cores this process may use ({d['value']/d['cpu_baseline']['value']:.0f}×; {d['cpu_baseline']['single_thread']['value']:.2f} on one\ncore: {d['value']/d['cpu_baseline']['single_thread']['value']:.0f}×);", s, count=1)
open(p, "w").write(s)
p = os.path.join(root, "README.md")
s = open(p).read()
s = re.sub(r"\d+ decoys/s for a queue of 64-decoy batches at L=150 with distance restraints\n\(two lanes of 160 decoy slots; \d+ with 64 decoys in flight, \d+ as separate calls\), \d+ for two 64-decoy chains with all four\nchannels \(every decoy within 2 Å of the map's structure\), \d+ at L=400;",
           f"{d['value']:.0f} decoys/s for a queue of 64-decoy batches at L=150 with distance restraints\n(two lanes of 160 decoy slots; {d['in_flight_B']['value']:.0f} with 64 decoys in flight, {d['per_call']['value']:.0f} as separate calls), {c3['value']:.0f} for two 64-decoy chains with all four\nchannels (every decoy within 2 Å of the map's structure), {c4['value']:.0f} at L=400;", s, count=1)
open(p, "w").write(s)
print("refreshed:", round(d["value"], 1), round(c3["value"], 1), round(c4["value"], 1))
