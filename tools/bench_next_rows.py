"""Measurement of the rows SURVEY.md section 8(f) marks "next" and this tree has built on the device: each device path timed on an MI355X with the package's own
host mirror of the same arithmetic timed beside it on the box's CPU (the mirrors are what the parity tests pin to the reference and compare the device output with: tests/test_oracle_golden.py,
tests/test_gpu_feedback.py, tests/test_host_boundary.py) -- never the oracle, which is test infrastructure.  One JSON line per row.
  f1  the feedback step between two folds of a chain (utils_trX2dy/utils.py:294-403, run_inference.py:75-131): realised bins of the new decoy, re-weighting of the four
      maps, cumulative tmp array, convergence number, restraint tables rebuilt -- on the maps resident in the context (trx2_feedback_step)
  f1b the reference's ranking of the initial decoys (utils.py:352-372), n decoys at once (trx2_reliability_scores)
  f4  all-pairs C-alpha RMSD and TM-score of a target's decoys (cluster.py / evaluate.py; the reference: one ./bin/TMscore subprocess per pair) (trx2_superpose_matrix)
  f4b the GloCon matrix of a target's decoys (utils.py:543-569) (trx2_glocon_matrix)
Bytes are the algorithmic ones (inputs read once + outputs written once); the fraction of the 8 TB/s HBM roof is printed for the record: these are small, launch-
and latency-bound calls and none of them is a roofline story.
usage: bench_next_rows.py <repo> [L = 150] [decoys = 600]"""
import importlib, json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd")
S = importlib.import_module("trrosettax2-dynamics_amd.synth")
FB = importlib.import_module("trrosettax2-dynamics_amd.feedback")
PD = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
EV = importlib.import_module("trrosettax2-dynamics_amd.evaluate")
CL = importlib.import_module("trrosettax2-dynamics_amd.cluster")
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
n = int(sys.argv[3]) if len(sys.argv) > 3 else 600
HBM = 8000.0


def best_of(f, reps):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts)


m = S.make_map(L, seed=L)
seq = m["seq"]
ctx = T.Context(0, lanes=2)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
r = ctx.fold_batch(64, T.protocol.build_runs(L, 2, fastrelax=True), seed=5)
rng = np.random.default_rng(0)
decoys = np.stack([r["xyz"][i % 64] + rng.normal(size=(L, 5, 3)).astype(np.float32) * 0.3 * (i // 64) for i in range(n)]).astype(np.float32)   # n distinct structures around 64 folds
work = tempfile.mkdtemp(prefix="trx2_next_")
pdb = os.path.join(work, "d0.pdb"); PD.write_pdb(pdb, seq, decoys[0])
xyz_pdb = PD.as_read_from_pdb(seq, decoys[0])[0]

# ---- f1: one feedback step
host = {k: m[k] for k in ("dist", "theta", "omega", "phi")}
t_host = best_of(lambda: FB.feedback_labels(dict(host), pdb, 1.0, True), 3)
ctx.feedback_step(xyz_pdb, seq)                      # warm-up (allocations)
t_dev = best_of(lambda: ctx.feedback_step(xyz_pdb, seq), 20)
by = 2 * 4 * L * L * (37 + 25 + 25 + 13) + 4 * L * L * 100        # four maps read and written + the tmp array
print(json.dumps({"row": "f1 feedback step", "L": L, "device_ms": round(1e3 * t_dev, 3), "host_mirror_ms": round(1e3 * t_host, 1), "ratio": round(t_host / t_dev, 1),
                  "algorithmic_MB": round(by / 1e6, 1), "GBps": round(by / t_dev / 1e9, 1), "frac_of_hbm_roof": round(by / t_dev / 1e9 / HBM, 4),
                  "includes": "device: upload of the decoy, bins, re-weighting, tmp array, convergence number, rebuilt spline tables and row lists; host mirror: the same without the table rebuild, PDB parsed from a file as the reference does"}), flush=True)

# ---- f1b: reliability scores of n decoys
stack = np.stack([PD.as_read_from_pdb(seq, d)[0] for d in decoys[:64]])
paths = []
for i in range(64):
    q = os.path.join(work, f"r{i}.pdb"); PD.write_pdb(q, seq, decoys[i]); paths.append(q)
t_host = best_of(lambda: [FB.calculate_reliability_score(q) for q in paths], 2)
ctx.reliability_scores(stack)
t_dev = best_of(lambda: ctx.reliability_scores(stack), 20)
print(json.dumps({"row": "f1b reliability scores", "decoys": 64, "L": L, "device_ms": round(1e3 * t_dev, 3), "host_mirror_ms": round(1e3 * t_host, 1), "ratio": round(t_host / t_dev, 1),
                  "includes": "host mirror parses 64 PDB files (the reference's way); device takes the coordinates in memory"}), flush=True)

# ---- f4: all-pairs RMSD + TM-score
ca = decoys[:, :, 1].copy()
ctx.superpose_matrix(ca[:32])
t_dev = best_of(lambda: ctx.superpose_matrix(ca), 3)
ns = 40                                              # bounded host sample: ns x ns pairs
t0 = time.perf_counter()
for i in range(ns):
    for j in range(ns):
        EV.rmsd_common(ca[i].astype(np.float64), ca[j].astype(np.float64)); EV.tm_score(ca[i].astype(np.float64), ca[j].astype(np.float64))
t_pair_host = (time.perf_counter() - t0) / (ns * ns)
pairs = n * n
print(json.dumps({"row": "f4 all-pairs RMSD + TM-score", "decoys": n, "L": L, "pairs": pairs, "device_s": round(t_dev, 4), "device_us_per_pair": round(1e6 * t_dev / pairs, 3),
                  "host_mirror_ms_per_pair": round(1e3 * t_pair_host, 3), "host_mirror_s_for_all_pairs_extrapolated": round(t_pair_host * pairs, 1), "ratio": round(t_pair_host * pairs / t_dev),
                  "host_sample": f"{ns} x {ns} pairs, one thread (evaluate.rmsd_common + evaluate.tm_score, the functions the device output is tested equal to)",
                  "reference": "one ./bin/TMscore subprocess per pair (utils.py:514-541): not runnable here, process start alone is milliseconds per pair"}), flush=True)

# ---- f4b: GloCon matrix
ng = min(n, 200)
xs = np.stack([PD.as_read_from_pdb(seq, d)[0] for d in decoys[:ng]])
ctx.glocon_matrix(xs[:8], [seq] * 8)
t_dev = best_of(lambda: ctx.glocon_matrix(xs, [seq] * ng), 3)
gdir = os.path.join(work, "g"); os.makedirs(gdir)
nh = 24
for i in range(nh):
    PD.write_pdb(os.path.join(gdir, f"conf_{i}.pdb"), seq, decoys[i])
t_host = best_of(lambda: CL.get_glocon_matrix(gdir, device=None), 1)
print(json.dumps({"row": "f4b GloCon matrix", "decoys": ng, "L": L, "device_s": round(t_dev, 4), "host_mirror_s": round(t_host, 2), "host_decoys": nh,
                  "host_mirror_s_extrapolated_to_device_size": round(t_host * (ng / nh) ** 2, 1), "ratio": round(t_host * (ng / nh) ** 2 / t_dev),
                  "includes": "host mirror reads PDB files and scales with the square of the decoy count"}), flush=True)
ctx.close()
import shutil
shutil.rmtree(work, ignore_errors=True)
