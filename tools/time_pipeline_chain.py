"""Wall time of one chain of run_inference through pipeline.generate_npz_and_pdb (example NMR map, L=90): 10 initial decoys, then
feedback iterations of one decoy each.  usage: time_pipeline_chain.py <repo> [Nmax=20]"""
import importlib, os, sys, tempfile, time, io, contextlib
sys.path.insert(0, sys.argv[1])
PL = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
g = os.path.join(sys.argv[1], "tests", "golden"); npz, fa = os.path.join(g, "seq_NMR.npz"), os.path.join(g, "seq.fasta")
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 20
def chain(n, **kw):
    d = tempfile.mkdtemp(); t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        last = PL.generate_npz_and_pdb("s", os.path.join(d, "tmp"), os.path.join(d, "pdb"), npz, fa, N=10, Nmax=n, seed=5, **kw)
    return time.perf_counter() - t, last
chain(2)
t0, _ = chain(0 + 1)
for kw in ({}, {"device_feedback": "arrays"}, {"device_feedback": False}, {"device_feedback": False, "write_tmp_npz": True}):
    t, last = chain(nmax, **kw)
    print(f"{kw or 'default'}: {last} iterations in {t*1e3:.0f} ms; first iteration + 10 initial decoys {t0*1e3:.0f} ms; "
          f"{(t - t0) / max(last - 1, 1) * 1e3:.1f} ms per further iteration")
