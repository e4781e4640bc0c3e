"""Assemble profiles/r02_traffic.json from the PMC passes of tools/pmc_run.sh (one directory per config under gpurun_out/).
usage: make_traffic_json.py <gpurun_out dir holding c2_pass*.json, c3_.., c4_..> <out.json>
HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half: MI355X_MICROARCH.md, HBM section) + WRITE_SIZE,
both reported by rocprofv3 in KiB."""
import hashlib, json, os, sys
src, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in ("kernel_pair.h", "trx2_device.h"):
    h.update(open(os.path.join(root, "trrosettax2-dynamics_amd", "csrc", f), "rb").read())
rec = {"kernel_src_sha": h.hexdigest()[:16]}
for cfg in (2, 3, 4):
    p = [os.path.join(src, f"c{cfg}_pass{i}.json") for i in (1, 2, 3, 4)]
    if not all(os.path.exists(q) for q in p):
        continue
    a, b, f, w = (json.load(open(q)) for q in p)
    rec[str(cfg)] = {
        "kernel": a["kernel"].replace("void ", ""), "decoys_per_launch": {2: 160, 3: 128, 4: 32}[cfg], "fetch_bytes_raw": f["FETCH_SIZE"] * 1024.0, "write_bytes": w["WRITE_SIZE"] * 1024.0,
        "hbm_bytes_per_launch": 2.0 * f["FETCH_SIZE"] * 1024.0 + w["WRITE_SIZE"] * 1024.0,
        "method": "rocprofv3 --pmc in separate runs of tools/pmc_pair.py (tools/pmc_run.sh: two SQ groups, FETCH_SIZE, WRITE_SIZE; mean of the last "
                  f"{a['launches']} replays on final coordinates, tools/pmc_report.py); KiB -> bytes; FETCH_SIZE doubled (gfx950 counts wide coalesced reads at half)",
        "valu_insts_per_launch": a["SQ_INSTS_VALU"], "valu_active_quad_cycles": b["SQ_ACTIVE_INST_VALU"], "waves": a["SQ_WAVES"],
        "wave_quad_cycles": b["SQ_WAVE_CYCLES"], "wait_any_quad_cycles": b["SQ_WAIT_ANY"], "sq_busy_cycles_sum": a["SQ_BUSY_CYCLES"],
        "lds_insts_per_launch": a["SQ_INSTS_LDS"], "vmem_rd_insts_per_launch": a["SQ_INSTS_VMEM_RD"]}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: (v if isinstance(v, str) else {q: v[q] for q in ("kernel", "hbm_bytes_per_launch", "valu_insts_per_launch")}) for k, v in rec.items()}))
