"""Wall time of one chain of run_inference (10 initial decoys + 5 iterations) on the example NMR map.  usage: time_chain.py <repo>"""
import importlib, os, sys, tempfile, time, io, contextlib
sys.path.insert(0, sys.argv[1])
PL = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
g = os.path.join(sys.argv[1], "tests", "golden"); d = tempfile.mkdtemp()
with contextlib.redirect_stdout(io.StringIO()):
    PL.generate_npz_and_pdb("w", d + "/t0", d + "/p0", os.path.join(g, "seq_NMR.npz"), os.path.join(g, "seq.fasta"), N=2, Nmax=1, seed=1)   # warm
    t0 = time.perf_counter(); n0 = PL.generate_npz_and_pdb("a", d + "/t1", d + "/p1", os.path.join(g, "seq_NMR.npz"), os.path.join(g, "seq.fasta"), N=10, Nmax=0, seed=2); t1 = time.perf_counter()
    n5 = PL.generate_npz_and_pdb("b", d + "/t2", d + "/p2", os.path.join(g, "seq_NMR.npz"), os.path.join(g, "seq.fasta"), N=10, Nmax=5, seed=2); t2 = time.perf_counter()
print(f"10 initial decoys + selection + 1 fold: {1e3*(t1-t0):.0f} ms;  with 5 iterations: {1e3*(t2-t1):.0f} ms  -> {(t2-t1-(t1-t0))/max(n5-1,1)*1e3:.0f} ms per extra iteration (was 190 ms)")
