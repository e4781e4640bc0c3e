"""Where does one chain of run_inference spend its time?  initial batch (B=N), then per iteration: fold (B=1), PDB write,
feedback (numpy).  Example NMR map, L=90.  usage: time_iteration_loop.py <repo>"""
import importlib, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, sys.argv[1])
FO = importlib.import_module("trrosettax2-dynamics_amd.fold"); FB = importlib.import_module("trrosettax2-dynamics_amd.feedback")
g = os.path.join(sys.argv[1], "tests", "golden"); npz, fa = os.path.join(g, "seq_NMR.npz"), os.path.join(g, "seq.fasta")
d = tempfile.mkdtemp()
FO.folding_with_pred_npz(npz, fa, d, "warm", repeat=2, seed=1)
t = time.perf_counter(); FO.folding_with_pred_npz(npz, fa, d, "initial", repeat=10, seed=2); t_init = time.perf_counter() - t
t = time.perf_counter(); s = [FB.calculate_reliability_score(os.path.join(d, f"initial{i}.pdb")) for i in range(10)]; t_rel = time.perf_counter() - t
cur = npz; rows = []
for it in range(1, 6):
    t0 = time.perf_counter(); r = FO.folding_with_pred_npz(cur, fa, d, f"seq{it}", seed=10 + it); t1 = time.perf_counter()
    pdb = os.path.join(d, f"seq{it}.pdb")
    dd, oo, tt, pp = FB.get_npz_from_pred_pdb(cur, pdb); tm = FB.get_npz_from_pred_pdb(cur, pdb, tmp=True); t2 = time.perf_counter()
    nxt = os.path.join(d, f"it{it+1}.npz"); np.savez_compressed(nxt, dist=dd, omega=oo, theta=tt, phi=pp, tmp=tm); t3 = time.perf_counter()
    rows.append((t1 - t0, r["seconds"], t2 - t1, t3 - t2, int(r["n_evals"][0]))); cur = nxt
print(f"initial batch of 10 decoys: {t_init*1e3:.0f} ms (incl. npz load, table build, PDB writes); reliability scores of 10 PDBs: {t_rel*1e3:.0f} ms")
print("per iteration (ms): fold call total | of which GPU fold | feedback (2 calls) | savez_compressed | evals")
for q in rows: print("   %7.0f | %7.0f | %7.0f | %7.0f | %d" % (q[0]*1e3, q[1]*1e3, q[2]*1e3, q[3]*1e3, q[4]))
a = np.array(rows); print("mean: fold call %.0f ms (GPU %.0f), feedback %.0f ms, savez %.0f ms -> %.0f ms per iteration" % (a[:,0].mean()*1e3, a[:,1].mean()*1e3, a[:,2].mean()*1e3, a[:,3].mean()*1e3, a[:,[0,2,3]].sum(1).mean()*1e3))
