"""Assemble profiles/rNN_traffic.json from the PMC passes of tools/pmc_run.sh.
usage: make_traffic_json.py <dir holding c<cfg>_<pair|step>_B<decoys>_pass{1..4}.json> <out.json>
HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half: MI355X_MICROARCH.md, HBM section) + WRITE_SIZE,
both reported by rocprofv3 in KiB.  Records are keyed by (config, kernel family, decoys per launch) and by a hash of the kernel's
sources, so that bench.py drops them when the kernel changes or when it launches another shape."""
import glob, hashlib, json, os, re, sys
src, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = {"k_pair": ("kernel_pair.h", "trx2_device.h"), "k_step": ("kernel_step.h", "trx2_device.h")}
def sha(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(root, "trrosettax2-dynamics_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]
rec = {"kernel_src_sha": {k: sha(v) for k, v in SOURCES.items()}, "records": []}
for p1 in sorted(glob.glob(os.path.join(src, "c*_*_B*_pass1.json"))):
    m = re.match(r"c(\w)_(pair|step)_B(\d+)_pass1\.json", os.path.basename(p1))
    ps = [p1.replace("pass1", f"pass{i}") for i in (1, 2, 3, 4)]
    if not m or not all(os.path.exists(q) and os.path.getsize(q) for q in ps):
        continue
    a, b, f, w = (json.load(open(q)) for q in ps)
    rec["records"].append({
        "config": "e2e_single" if m.group(1) == "e" else int(m.group(1)), "kernel_family": "k_" + m.group(2), "kernel": a["kernel"], "decoys_per_launch": int(m.group(3)),
        "fetch_bytes_raw": f["FETCH_SIZE"] * 1024.0, "write_bytes": w["WRITE_SIZE"] * 1024.0,
        "hbm_bytes_per_launch": 2.0 * f["FETCH_SIZE"] * 1024.0 + w["WRITE_SIZE"] * 1024.0,
        "method": "rocprofv3 --pmc in separate runs of tools/pmc_kernel.py (tools/pmc_run.sh: two SQ groups, FETCH_SIZE, WRITE_SIZE; mean of the last "
                  f"{a['launches']} dispatches, tools/pmc_report.py); KiB -> bytes; FETCH_SIZE doubled (gfx950 counts wide coalesced reads at half)",
        "valu_insts_per_launch": a["SQ_INSTS_VALU"], "valu_active_quad_cycles": b["SQ_ACTIVE_INST_VALU"], "waves": a["SQ_WAVES"],
        "wave_quad_cycles": b["SQ_WAVE_CYCLES"], "wait_any_quad_cycles": b["SQ_WAIT_ANY"], "sq_busy_cycles_sum": a["SQ_BUSY_CYCLES"],
        "lds_insts_per_launch": a["SQ_INSTS_LDS"], "vmem_rd_insts_per_launch": a["SQ_INSTS_VMEM_RD"]})
# the shared-launch shape: sixteen single-decoy folds in one engine's launches (profiles/history/runs_r01_r04.sh.txt section r04_profiles.sh: shared16_{pair,step}_<group>.json
# beside the pmc directory), keyed config "shared16"
up = os.path.dirname(os.path.abspath(src))
for fam, kern in (("pair", "k_pair1_multi"), ("step", "k_step_multi"), ("half", "k_half_multi")):   # half: the same folds in half-evaluation form (eight per role and launch)
    ps = [os.path.join(up, f"shared16_{fam}_{g}.json") for g in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "SQ_WAVES")]
    if all(os.path.exists(q) and os.path.getsize(q) for q in ps):
        f, w, t, q = (json.load(open(x)) for x in ps)
        rec["records"].append({
            "config": "shared16", "kernel_family": kern, "kernel": f["kernel"], "decoys_per_launch": 16,
            "fetch_bytes_raw": f["FETCH_SIZE"] * 1024.0, "write_bytes": w["WRITE_SIZE"] * 1024.0,
            "hbm_bytes_per_launch": 2.0 * f["FETCH_SIZE"] * 1024.0 + w["WRITE_SIZE"] * 1024.0,
            "l2_requests": t.get("TCC_REQ_sum"), "l2_hits": t.get("TCC_HIT_sum"), "l2_misses": t.get("TCC_MISS_sum"),
            "method": "rocprofv3 --pmc in separate runs of tools/shared_scaling.py (16 folds, one engine, one wave per row; mean of "
                      f"{f['launches']} dispatches before the engine's last three chunks, which name folds that have already reported); KiB -> bytes; FETCH_SIZE doubled as for the other records (the guide's correction is calibrated for wide "
                      "coalesced reads; these are 64-byte gathers: L2 misses x 64 B equal the RAW FETCH_SIZE, so the doubled figure is an upper bound)",
            "valu_insts_per_launch": q.get("SQ_INSTS_VALU"), "waves": q.get("SQ_WAVES"), "wave_quad_cycles": q.get("SQ_WAVE_CYCLES"), "wait_any_quad_cycles": q.get("SQ_WAIT_ANY")})
json.dump(rec, open(out, "w"), indent=1)
for r in rec["records"]:
    print(r["config"], r["kernel"], r["decoys_per_launch"], "hbm MB", round(r["hbm_bytes_per_launch"] / 1e6, 2), "VALU M", round((r["valu_insts_per_launch"] or 0) / 1e6, 2),
          "wait", round(r["wait_any_quad_cycles"] / r["wave_quad_cycles"], 2) if r.get("wave_quad_cycles") else None)
