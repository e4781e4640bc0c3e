"""GPU: shared launches (csrc/launch_engine.h, trx2_set_shared_launches).  The single-decoy folds of many
contexts -- the iteration phase of run_inference.py:97-139, one chain per (target, model) -- advance in ONE (pair, step) launch
pair per evaluation instead of one pair per chain.  A fold's arithmetic must not depend on what shares its launches: every
result bit for bit the fold's own launches', at mixed chain lengths (all three step-kernel classes) and mixed channel sets
(both pair-kernel instantiations), with folds joining and leaving the launches at different times."""
import importlib
import os
import threading

import numpy as np
import pytest
import torch  # noqa: F401  -- before libtrx2fold.so (see test_gpu_boundary.py)

pytestmark = pytest.mark.gpu

T = importlib.import_module("trrosettax2-dynamics_amd")
L_ = importlib.import_module("trrosettax2-dynamics_amd._lib")
S = importlib.import_module("trrosettax2-dynamics_amd.synth")
PL = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
KEYS = ("xyz", "tors", "e_terms", "f", "status", "n_evals", "n_iters")


@pytest.fixture(autouse=True)
def restore_mode():
    yield
    L_.set_shared_launches(-1)
    L_.set_shared_launch_halves(-1)


@pytest.mark.parametrize("waves", [4, 1], ids=["four-waves-per-row", "one-wave-per-row"])
def test_shared_launches_are_bitwise_the_folds_own(golden_dir, seq, waves):
    """both pair-kernel shapes of a single-decoy fold (Context.set_single_decoy_waves): the default, and the one batch mode sets -- the latter
    also in half-evaluation form (trx2_set_shared_launch_halves: one kernel steps one half of an engine's folds beside the pair terms of the
    other half; the L=300 folds do not qualify, so the same chunks also hold a launch class in pair | step form)"""
    real = np.load(os.path.join(golden_dir, "seq_NMR.npz"))
    cases = [("real90", dict(dist=real["dist"], omega=real["omega"], theta=real["theta"], phi=real["phi"], seq=seq), True, 0),
             ("L150", S.make_map(150, seed=150), True, 900), ("L150d", S.make_map(150, seed=151), False, 900),
             ("L120", S.make_map(120, seed=120, n_moves=150), True, 700), ("L300", S.make_map(300, seed=300, n_moves=150), True, 500)]
    ctxs = []
    try:
        for name, m, orient, cap in cases:
            for rep in range(2):                       # two contexts per map: ten folds share the launches
                c = T.Context(0)
                c.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
                c.set_single_decoy_waves(waves)
                ctxs.append((name, len(m["seq"]), cap, rep, c))
        out = {}
        for mode in (0, 1) + ((2,) if waves == 1 else ()):
            L_.set_shared_launches(min(mode, 1))
            L_.set_shared_launch_halves(1 if mode == 2 else 0)
            res = [None] * len(ctxs)

            def work(i):
                name, L, cap, rep, c = ctxs[i]
                rs = []
                for it in range(2):                    # two folds in a row per context: joins and leaves at different times
                    rs.append(c.fold_batch(1, T.protocol.build_runs(L, 2, fastrelax=True), seed=40 + i, decoy0=it, max_evals=cap))
                res[i] = rs

            if mode == 0:
                for i in range(len(ctxs)):
                    work(i)
            else:
                th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
                [t.start() for t in th]
                [t.join() for t in th]
            assert all(r is not None for r in res)
            out[mode] = res
        for i, (name, L, cap, rep, c) in enumerate(ctxs):
            for it in range(2):
                a, b = out[0][i][it], out[1][i][it]
                assert np.all(np.isfinite(a["xyz"])) and (cap > 0 or a["status"][0] == 0)
                for k in KEYS:
                    assert np.array_equal(a[k], b[k]), (name, rep, it, k)
                    if 2 in out:
                        assert np.array_equal(a[k], out[2][i][it][k]), ("half-evaluation launches", name, rep, it, k)
        ev = [int(out[1][i][1]["n_evals"][0]) for i in range(len(ctxs))]
        print("\nshared launches: %d folds on %d contexts bit for bit equal to their own launches; evaluations of the second folds %s" % (2 * len(ctxs), len(ctxs), ev))
    finally:
        for *_, c in ctxs:
            c.close()


def test_batch_mode_files_do_not_depend_on_shared_launches(golden_dir, tmp_path):
    """run_inference's batch mode (pipeline.run_batch), three targets x two models: every PDB byte for byte the same whether the
    chains' single-decoy folds share launches (all six chains in flight) or every fold launches for itself, one target at a time."""
    import shutil
    fdir = tmp_path / "fasta"
    fdir.mkdir()
    names = ["ta", "tb", "tc"]
    for n in names:
        shutil.copyfile(os.path.join(golden_dir, "seq.fasta"), fdir / f"{n}.fasta")
    kw = dict(init_num=2, Nmax=4, angle=True, mult_two_models=True, seed=5,
              npz_nmr=os.path.join(golden_dir, "seq_NMR.npz"), npz_xray=os.path.join(golden_dir, "seq_Xray.npz"))
    out = {}
    for mode, inflight in ((0, 1), (1, 3), (2, 3)):      # 2: shared launches in half-evaluation form
        L_.set_shared_launches(min(mode, 1))
        L_.set_shared_launch_halves(1 if mode == 2 else 0)
        save = str(tmp_path / f"out{mode}")
        res = PL.run_batch(names, str(fdir), save, targets_in_flight=inflight, **kw)
        assert res["failed"] == 0 and res["decoys"] == 3 * (2 * 2 + 2 * 4), res
        out[mode] = {(n, f): open(os.path.join(save, n, "pred_pdb", f), "rb").read() for n in names for f in sorted(os.listdir(os.path.join(save, n, "pred_pdb")))}
    assert out[0].keys() == out[1].keys() == out[2].keys() and len(out[0]) == 36
    assert all(out[0][k] == out[1][k] == out[2][k] for k in out[0])


def test_segment_cache_changes_no_bit(golden_dir, tmp_path):
    """The pair kernel's segment cache (csrc/kernel_pair.h, PairArgs) serves a spline lookup from the decoy's own copy of the segment it
    used last; the cached numbers are the table's, so every result must be bit for bit that of the gathers (TRX2_SEG_CACHE=0; the
    switch is read once per process: two child processes).  Shapes: a batch of 40 on two lanes through tail compaction and wave
    halving (all channels), 5 decoys of a distance-only map, a single decoy followed by a feedback step and another single decoy
    (the tables change under the cache: its tags must have been reset)."""
    import subprocess
    import sys
    code = r'''
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
T = importlib.import_module("trrosettax2-dynamics_amd")
g = os.path.join(sys.argv[1], "tests", "golden")
seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
m = np.load(os.path.join(g, "seq_NMR.npz"))
runs = T.protocol.build_runs(90, 2, fastrelax=True)
out = {}
c = T.Context(0, lanes=2)
c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
r = c.fold_batch(40, runs, seed=3); out["b40_xyz"], out["b40_f"], out["b40_ev"] = r["xyz"], r["f"], r["n_evals"]
r = c.fold_batch(1, runs, seed=4); out["s1_xyz"], out["s1_ev"] = r["xyz"], r["n_evals"]
out["delta"] = np.array(c.feedback_step(np.nan_to_num(np.load(os.path.join(g, "ref_decoys.npz"))["conf_2_1"]), seq))
r = c.fold_batch(1, runs, seed=5); out["s2_xyz"], out["s2_ev"] = r["xyz"], r["n_evals"]
c.set_map(m["dist"], seq=seq)
r = c.fold_batch(5, T.protocol.build_runs(90, 2), seed=6); out["d5_xyz"], out["d5_f"] = r["xyz"], r["f"]
c.close()
np.savez(sys.argv[2], **out)
'''
    res = {}
    for mode in ("1", "0"):
        path = str(tmp_path / f"seg{mode}.npz")
        env = dict(os.environ, TRX2_SEG_CACHE=mode)
        p = subprocess.run([sys.executable, "-c", code, os.path.dirname(golden_dir.rstrip("/")).rsplit("/tests", 1)[0], path], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = dict(np.load(path))
    assert res["1"].keys() == res["0"].keys()
    for k in res["1"]:
        assert np.array_equal(res["1"][k], res["0"][k]), k
    assert np.all(np.isfinite(res["1"]["b40_xyz"])) and res["1"]["b40_ev"].min() > 500


def test_who_launches_follows_the_number_of_live_contexts(golden_dir, seq):
    """The library's rule (include/trx2fold.h, trx2_set_shared_launches(-1)): with fewer than five contexts alive a single-decoy fold
    launches for itself (the engines' chunk counter does not move); from the fifth on it goes to the engines.  Same bits either way."""
    if os.environ.get("TRX2_SHARED_LAUNCH") is not None:
        pytest.skip("TRX2_SHARED_LAUNCH forces one route")
    L_.set_shared_launches(-1)
    real = np.load(os.path.join(golden_dir, "seq_NMR.npz"))
    runs = T.protocol.build_runs(90, 2, fastrelax=True)
    ctxs = []
    try:
        def add(n):
            for _ in range(n):
                c = T.Context(0)
                c.set_map(real["dist"], real["omega"], real["theta"], real["phi"], seq=seq)
                ctxs.append(c)
        add(2)
        s0 = L_.shared_launch_stats(0)["chunks"]
        few = [c.fold_batch(1, runs, seed=77, max_evals=400) for c in ctxs]
        s1 = L_.shared_launch_stats(0)["chunks"]
        assert s1 == s0, (s0, s1)                       # two contexts: nobody went to an engine
        add(4)                                           # six alive
        many = [None] * 6
        th = [threading.Thread(target=lambda i=i: many.__setitem__(i, ctxs[i].fold_batch(1, runs, seed=77, max_evals=400))) for i in range(6)]
        [t.start() for t in th]
        [t.join() for t in th]
        s2 = L_.shared_launch_stats(0)["chunks"]
        assert s2 > s1, (s1, s2)                         # six contexts: the engines stepped them
        for r in many:
            for k in KEYS:
                assert np.array_equal(r[k], few[0][k]), k     # same map, same seed: the same fold whoever launched it
    finally:
        for c in ctxs:
            c.close()


def test_mixed_chain_lengths_in_flight_files_do_not_depend_on_the_schedule():
    """tools/soak_batch.py, small: twelve targets of six chain lengths (all three step-kernel classes in the same shared launches),
    folded with eight and with three targets in flight: every PDB byte for byte the same."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_batch.py"), root, "12", "3", "8", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "files identical between the two passes: True (120 files)" in out.stdout, out.stdout[-800:]
