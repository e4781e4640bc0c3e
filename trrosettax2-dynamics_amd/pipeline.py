"""Iteration orchestrator and output layout of run_inference.py, on top of the batched GPU fold.

Mirrors /root/reference/run_inference.py:16-143 (generate_npz_and_pdb), :145-278 (flatten + rename to conf_1_k /
conf_2_k) and :280-337 (run_single).  Same file names at every stage: initial{i}.pdb, {name}{iter}.pdb,
tmp_npz/{name}{iter}.npz with keys dist/theta/omega/phi/tmp, final conf_1_k.pdb / conf_2_k.pdb.

The trX2 network front-end (pred_2d_geometry, utils_trX2dy/utils.py:768-797) is out of scope for this package: the
distograms must already exist as pred_npz/{name}_NMR.npz (and _Xray.npz) or be passed in explicitly.

Deliberate divergence: the reference's final numbering depends on os.walk / os.listdir order
and sorts iteration files lexicographically (seq10 < seq2).  Here the order is fixed -- Xray/ is flattened before NMR/,
which reproduces the provenance of the reference's committed example (conf_1_1 = Xray/initial0, conf_2_1 = NMR/initial0)
-- and iteration files are ordered by their number.
"""
import os
import re
import shutil
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import protocol, sched
from .feedback import calculate_reliability_score, feedback_labels
from .fold import FoldError, close_contexts, fold_arrays_to_pdb, fold_resident_to_pdb, get_context

_BATCH_CALLS = 0   # run_batch calls of this process (part of the shared queue's key)
from .pdbio import as_read_from_pdb, read_backbone, read_fasta


def resident_ok(device_feedback, sigma):
    return device_feedback is True and float(sigma) == 1.0


def generate_npz_and_pdb(pdb_name, processed_npz_dir, pred_pdb_dir, initial_npz, fasta, N=10, Nmax=500, begin_num=0,
                         sigma=1.0, tta_opt="-m 2 -r no-idp --orient ", angle=True, device=0, seed=None, lanes=2,
                         write_tmp_npz=False, device_feedback=True, timing=None, candidates=1, single_decoy_waves=4, profile_every=0):
    """N initial decoys as one GPU batch -> best by reliability -> feedback -> one decoy per iteration until the
    cumulative `tmp` array moves by < 0.01 or Nmax iterations (run_inference.py:97-139).  Returns the last index.

    The reference hands every intermediate distogram to the next fold through tmp_npz/{name}{k}.npz because that fold is
    another process.  Here the arrays stay in memory; the files are still written, with the same names and keys, but with
    np.savez instead of np.savez_compressed: compressing 3 MB of float32 took 95 ms per iteration at L=90, more than the
    fold (54 ms) or the feedback (25 ms), and run_single deletes tmp_npz/ at the end (run_inference.py:334).  Nothing reads
    those files any more, and even uncompressed they cost 9 ms of a 50 ms iteration: they are written only with
    write_tmp_npz=True (`run_inference.py --keep_tmp_npz`).

    device_feedback: the feedback step (decoy -> bins -> re-weighted distograms) runs on the GPU (csrc/kernel_feedback.h)
    instead of in numpy (feedback.feedback_labels, 20 ms per iteration at L=90).  True: on the distograms RESIDENT in the
    fold context (Context.feedback_step: only the decoy's coordinates go in, the convergence measure comes back, the next
    fold uses the rebuilt tables without any upload).  "arrays": same kernels on host arrays (Context.feedback_labels).
    Both are tested bitwise equal to the numpy path, which is pinned bit for bit to the reference; sigma other than 1
    falls back to numpy.

    candidates: K > 1 is an EXTENSION, off by default (hard part 5): every iteration folds K
    decoys of the same re-weighted distograms as one batch instead of one -- decoy identities (seed, 0 .. K-1), so candidate 0
    starts where the reference-order chain's decoy starts -- writes all K ({name}{(it-1) K + c + 1}.pdb) and feeds candidate 0
    back.  The chain of maps is the reference's (up to the rounding a batch of another width brings); the ensemble holds K decoys
    per state of the maps instead of one, for ~1.3 x (K = 8) the time of an iteration: a single-decoy fold leaves the chip idle.
    Returns the number of iteration files (K per iteration).  Needs the resident feedback path.

    single_decoy_waves: the pair kernel's shape for the iteration phase's single-decoy folds (Context.set_single_decoy_waves): 4 for a
    job with few chains in flight, 1 when many chains share launches (run_batch passes 1).

    timing: a dict that receives initial_s (initial batch: table build, fold, files, ranking), iteration_s (everything after),
    iteration_fold_s (the single-decoy folds alone), iterations, iteration_evals (energy / gradient evaluations of those folds), tmp_change (the convergence measure after every iteration: max
    |tmp_new - tmp_old|, run_inference.py:133) and converged (the chain took the < 0.01 exit) -- bench.py's e2e legs.

    profile_every: n > 0 brackets every n-th evaluation of the iteration phase's folds by HIP events on the fold's own stream
    (Context.set_profiling); `timing` then also receives kernel = dict(pair_ms, step_ms, samples: live averages over the sampled launches;
    evals; selected_terms of the last fed-back map and replay_pair_ms, 200 replays of the pair kernel on the last decoy) -- bench.py's roofline
    record of the metric's job.  Measurement only: a profiled job is never the timed one."""
    import time
    os.makedirs(processed_npz_dir, exist_ok=True)
    t_start = time.perf_counter()
    tm = dict(initial_s=0.0, iteration_s=0.0, iteration_fold_s=0.0, iterations=0, iteration_evals=0, tmp_change=[], converged=False)

    def done(iter_n):
        tm["iterations"] = iter_n - begin_num
        tm["iteration_s"] = time.perf_counter() - t_start - tm["initial_s"]
        if timing is not None:
            timing.update(tm)
        return iter_n

    resident = resident_ok(device_feedback, sigma)
    K = int(candidates)
    if K < 1 or (K > 1 and not resident):
        raise ValueError("candidates: a positive number; more than one needs the device feedback on resident distograms (sigma = 1)")

    def feedback(arrays, pdb):
        if device_feedback and float(sigma) == 1.0:
            xyz, s_pdb = read_backbone(pdb)                 # the decoy as the reference sees it: through its PDB file
            return get_context(device, lanes).feedback_labels(arrays, xyz, s_pdb, sigma, angle)
        return feedback_labels(arrays, pdb, sigma, angle)

    seq = read_fasta(fasta)
    # initial_npz: the path of the network's npz (the reference's hand-off), or the arrays themselves -- a dict of numpy arrays or of
    # CUDA tensors straight from the front-end (device pointers go into the table build, fold.set_restraints)
    init = dict(np.load(initial_npz)) if isinstance(initial_npz, (str, os.PathLike)) else dict(initial_npz)
    if not resident_ok(device_feedback, sigma):
        init = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in init.items()}      # the host feedback path works on numpy
    print("Start generating the initial structures")
    r_init = fold_arrays_to_pdb(init, seq, pred_pdb_dir, [f"initial{i}.pdb" for i in range(N)], tta_opt, device=device, seed=seed, lanes=lanes)
    print("Done generating initial structures")
    best_score, best_pdb, best_i = -np.inf, None, 0
    if device_feedback:
        # the decoys as the reference would read them back from their PDB files (bit-exact "%8.3f" round trip, no file parse),
        # scored in one device call (trx2_reliability_scores); tested equal to the per-file host function
        scores = get_context(device, lanes).reliability_scores(np.stack([as_read_from_pdb(seq, r_init["xyz"][i])[0] for i in range(N)]))
    else:
        scores = [calculate_reliability_score(os.path.join(pred_pdb_dir, f"initial{i}.pdb")) for i in range(N)]
    for i in range(N):                                   # strict '>' : the first maximum wins (run_inference.py:67)
        if scores[i] > best_score:
            best_score, best_pdb, best_i = scores[i], os.path.join(pred_pdb_dir, f"initial{i}.pdb"), i

    pattern = os.path.join(processed_npz_dir, pdb_name + "{}.npz")
    tm["initial_s"] = time.perf_counter() - t_start
    if resident:
        # The initial batch left its map in this thread's context.  From here on the distograms never leave the device.
        ctx = get_context(device, lanes)
        ctx.set_single_decoy_waves(single_decoy_waves)
        prof = dict(pair=0.0, step=0.0, n=0, evals=0)
        if profile_every:
            ctx.set_profiling(int(profile_every))

        def step(xyz_fold, k):
            # the decoy as the reference sees it -- through its PDB file -- without reading the file back
            xyz, s_pdb = as_read_from_pdb(seq, xyz_fold)
            delta = ctx.feedback_step(xyz, s_pdb, sigma, angle)
            if write_tmp_npz:
                np.savez(pattern.format(k), **{c: ctx.get_map(c) for c in ((("dist", "theta", "omega", "phi") if angle else ("dist",)) + ("tmp",))})
            return delta

        step(r_init["xyz"][best_i], begin_num + 1)
        iter_n = begin_num
        while True:
            iter_n += 1
            print(f"Start generating structure {iter_n}")
            t_f = time.perf_counter()
            first = begin_num + (iter_n - begin_num - 1) * K + 1      # K = 1: iter_n
            try:
                r = fold_resident_to_pdb(ctx, seq, pred_pdb_dir, [f"{pdb_name}{first + c}.pdb" for c in range(K)], tta_opt,
                                         seed=None if seed is None else seed + iter_n)
            except FoldError as e:
                # Only candidate 0 is fed back: a diverged EXTRA candidate must not end a chain that K = 1 would have continued
                # (ADVICE r3).  Its file slot is filled with a copy of candidate 0, so that the numbering every later stage
                # relies on (K files per iteration) holds; the failure is reported.  Candidate 0 itself failing ends the chain.
                if 0 in e.bad:
                    raise
                r = e.result
                for c in e.bad:
                    shutil.copyfile(os.path.join(pred_pdb_dir, f"{pdb_name}{first}.pdb"), os.path.join(pred_pdb_dir, f"{pdb_name}{first + c}.pdb"))
                print(f"warning: candidates {list(e.bad)} of iteration {iter_n} failed to fold; their files repeat candidate 0")
            tm["iteration_fold_s"] += time.perf_counter() - t_f
            tm["iteration_evals"] += int(r["n_evals"].sum())
            if profile_every:
                a, b, n = ctx.last_fold_kernel_times()
                prof["pair"] += a * n; prof["step"] += b * n; prof["n"] += n; prof["evals"] += int(r["n_evals"].sum())
            print("Done generating structure", iter_n)
            if iter_n - begin_num >= Nmax:
                break
            delta = step(r["xyz"][0], iter_n + 1)
            tm["tmp_change"].append(float(delta))                  # max |tmp_new - tmp_old|: the reference's convergence measure
            if delta < 0.01:
                tm["converged"] = True
                break
        if profile_every:
            ctx.set_profiling(0)
            w = np.array(protocol.SF, np.float32)
            ctx.eval_batch(r["tors"][:1], w)                      # lays the last decoy out for one slot
            rep_ms, terms = ctx.time_pair_kernel(1, w, 1, len(seq), n_rep=200)
            tm["kernel"] = dict(pair_ms=prof["pair"] / max(prof["n"], 1), step_ms=prof["step"] / max(prof["n"], 1), samples=prof["n"], evals=prof["evals"],
                                selected_terms=float(terms), replay_pair_ms=float(rep_ms), record_bytes=ctx.info(1), history_pairs=int(ctx.info(2)),
                                pair_workgroups=int(ctx.info(4)))
        done(iter_n)
        return begin_num + (iter_n - begin_num) * K
    base = {k: init[k] for k in (("dist", "theta", "omega", "phi") if angle else ("dist",))}   # no "tmp": falls back to dist
    old_tmp = init["dist"]
    cur = feedback(base, best_pdb)
    if write_tmp_npz:
        np.savez(pattern.format(begin_num + 1), **cur)
    iter_n = begin_num
    while True:
        iter_n += 1
        old_tmp = cur["tmp"]                             # what np.load(current_npz)["tmp"] gives (run_inference.py:101-102)
        print(f"Start generating structure {iter_n}")
        t_f = time.perf_counter()
        fold_arrays_to_pdb(cur, seq, pred_pdb_dir, [f"{pdb_name}{iter_n}.pdb"], tta_opt, device=device,
                           seed=None if seed is None else seed + iter_n)
        tm["iteration_fold_s"] += time.perf_counter() - t_f
        print("Done generating structure", iter_n)
        if iter_n - begin_num >= Nmax:
            break
        cur = feedback(cur, os.path.join(pred_pdb_dir, f"{pdb_name}{iter_n}.pdb"))
        if write_tmp_npz:
            np.savez(pattern.format(iter_n + 1), **cur)
        if np.max(np.abs(old_tmp - cur["tmp"])) < 0.01:
            break
    return done(iter_n)


def flatten_and_rename(save_pdb_dir, num_conf1_others):
    """pred_pdb/{Xray,NMR}/ -> pred_pdb/conf_1_k.pdb, conf_2_k.pdb (run_inference.py:145-278, order made deterministic)."""
    moved = []
    for sub in ("Xray", "NMR"):
        d = os.path.join(save_pdb_dir, sub)
        if not os.path.isdir(d):
            continue
        for name in sorted(os.listdir(d)):
            target = os.path.join(save_pdb_dir, name)
            base, ext = os.path.splitext(name)
            k = 1
            while os.path.exists(target):                 # name clash -> _1, _2 ... (run_inference.py:154-159)
                target = os.path.join(save_pdb_dir, f"{base}_{k}{ext}")
                k += 1
            shutil.move(os.path.join(d, name), target)
            moved.append(os.path.basename(target))
        os.rmdir(d)
    init1, init2, others = [], [], []
    for name in os.listdir(save_pdb_dir):
        m1, m2 = re.fullmatch(r"initial(\d+)\.pdb", name), re.fullmatch(r"initial(\d+)_1\.pdb", name)
        if m1:
            init1.append((int(m1.group(1)), name))
        elif m2:
            init2.append((int(m2.group(1)), name))
        elif re.fullmatch(r".*?(\d+)\.pdb", name) and not re.fullmatch(r"conf_[12]_\d+\.pdb", name):
            others.append((int(re.fullmatch(r".*?(\d+)\.pdb", name).group(1)), name))
    plan, c1, c2 = [], 0, 0
    for x, name in sorted(init1):
        plan.append((name, f"conf_1_{x + 1}.pdb")); c1 = max(c1, x + 1)
    for x, name in sorted(init2):
        plan.append((name, f"conf_2_{x + 1}.pdb")); c2 = max(c2, x + 1)
    others.sort()
    for _, name in others[:num_conf1_others]:
        c1 += 1; plan.append((name, f"conf_1_{c1}.pdb"))
    for _, name in others[num_conf1_others:]:
        c2 += 1; plan.append((name, f"conf_2_{c2}.pdb"))
    for old, new in plan:
        os.rename(os.path.join(save_pdb_dir, old), os.path.join(save_pdb_dir, new))
    return dict(plan)


def run_single(name, fasta_file, save_dir, init_num=10, Nmax=300, angle=True, mult_two_models=True, npz_nmr=None,
               npz_xray=None, device=0, seed=None, keep_tmp_npz=False, phase_times=None, candidates=1, single_decoy_waves=4, arrays=None,
               write_pred_npz=True, profile_every=0):
    """run_inference.py:280-337 without the network front-end: expects the distograms to exist -- as pred_npz files (npz_nmr / npz_xray or
    already in place) or IN MEMORY: arrays = {"NMR": {dist, omega, theta, phi}, "Xray": {...}} of numpy arrays or of float32 CUDA tensors, what
    pred_2d_geometry computes before it writes them (run_inference.py:301-310, utils_trX2dy/utils.py:783-796).  CUDA tensors are handed to the
    table build by pointer (Context.set_map_device).  The reference's output layout keeps pred_npz/{name}_{tag}.npz, so the files
    are still written from the arrays (in the background, beside the fold) unless write_pred_npz is False.
    phase_times: a dict that receives, per chain ("NMR" / "Xray"), generate_npz_and_pdb's timing record.
    candidates: decoys folded and written per feedback iteration (extension, default 1: generate_npz_and_pdb)."""
    content = os.path.join(save_dir, name)
    npz_dir, pdb_dir, tmp_dir = (os.path.join(content, d) for d in ("pred_npz", "pred_pdb", "tmp_npz"))
    for d in (npz_dir, pdb_dir, tmp_dir):
        os.makedirs(d, exist_ok=True)
    tta_opt = "-m 2 --orient -r no-idp" if angle else "-m 2 --no-orient -r no-idp"      # run_inference.py:295
    maps = [("NMR", npz_nmr)] + ([("Xray", npz_xray)] if mult_two_models else [])
    paths = {}
    writers = []
    for tag, given in maps:
        path = os.path.join(npz_dir, f"{name}_{tag}.npz")
        if arrays is not None:
            if tag not in arrays:
                raise KeyError(f"arrays has no entry for the {tag} model")
            paths[tag] = dict(arrays[tag])
            if write_pred_npz:      # the reference's file, written beside the fold (np.savez_compressed of 3-64 MB takes 0.1-2 s)
                host = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in paths[tag].items()}
                th = threading.Thread(target=np.savez_compressed, args=(path,), kwargs=host)
                th.start(); writers.append(th)
            continue
        if given and os.path.abspath(given) != os.path.abspath(path):
            shutil.copyfile(given, path)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: the trX2 network front-end (pred_2d_geometry) is not part of this "
                                    f"package -- produce the distogram with the reference's model or pass --npz_{tag.lower()}")
        paths[tag] = path

    def chain(tag, name_offset):
        try:
            return chain_(tag, name_offset)
        finally:
            close_contexts()     # this worker thread's context: its stream, maps and buffers go with the thread

    def chain_(tag, name_offset):
        # The reference runs the X-ray chain after the NMR one and continues its file numbering (begin_num = number of NMR
        # iterations, run_inference.py:315-318).  The chains are otherwise independent, so here they are folded
        # CONCURRENTLY (one context = one stream each; x1.7 on one GPU) with provisional names, and the X-ray iteration
        # files are renumbered afterwards to what the sequential order would have produced.
        return generate_npz_and_pdb(name + name_offset, os.path.join(tmp_dir, tag), os.path.join(pdb_dir, tag), paths[tag], fasta_file,
                                    N=init_num, Nmax=Nmax, begin_num=0, angle=angle, tta_opt=tta_opt, device=device,
                                    seed=None if seed is None else seed + 100000 * len(tag),
                                    lanes=1 if len(maps) == 2 else 2,   # two chains already occupy two streams
                                    write_tmp_npz=keep_tmp_npz, candidates=candidates, single_decoy_waves=single_decoy_waves, profile_every=profile_every,
                                    timing=None if phase_times is None else phase_times.setdefault(tag, {}))

    if len(maps) == 2:
        with ThreadPoolExecutor(max_workers=2) as ex:
            fut = {tag: ex.submit(chain, tag, "" if tag == "NMR" else "__x") for tag, _ in maps}
            num = fut["NMR"].result()
            nx = fut["Xray"].result()
        xdir = os.path.join(pdb_dir, "Xray")
        for k in range(1, nx + 1):                       # {name}__x{k}.pdb -> {name}{num+k}.pdb
            os.rename(os.path.join(xdir, f"{name}__x{k}.pdb"), os.path.join(xdir, f"{name}{num + k}.pdb"))
        total = num + nx
    else:
        num = total = chain_("NMR", "")   # on the calling thread: its context stays cached for the next target
    for th in writers:
        th.join()
    n_out = total + init_num * len(maps)
    print("All structures generation finished.")
    print(f"Total structures generated: {n_out}")
    shutil.rmtree(tmp_dir)
    flatten_and_rename(pdb_dir, num)
    print(f"Inference for sample '{name}' completed. Results in {content}")
    return n_out


def run_batch(names, fasta_dir, save_dir, rank=0, world=1, dist=None, device=0, run=None, targets_in_flight=None, store=None, **kw):
    """Batch mode (run_inference.py:339-348: `for name in names: run_single(...)`) sharded over ranks, one process per GPU.

    Targets are independent, so there is no data-path collective.  Every rank orders the name list longest first by the modelled
    seconds of a target (sched.MODEL: initial batches + sequential single-decoy iterations; latency-bound, not n L^2) and PULLS
    the next target from a shared counter when one of its worker threads is free (sched.DynamicQueue on a TCPStore: one atomic
    integer add per target).  How long a target takes cannot be known beforehand -- its chains stop when their maps converge (53-64
    iterations on the reference's example) or at Nmax = 300 -- so a static plan straggles; the queue does not.
    A target is not split: its iteration loop is sequential and the choice of the best initial decoy needs all of them
    (run_inference.py:60-73).  A failing target is recorded and the rest of the list still runs (the reference's loop dies at the
    first exception); the summary -- gathered with all_gather_object -- carries the failures and the caller turns them into a
    non-zero exit code.  `run` stands in for run_single in the CPU tests; `store` for the TCPStore (default: sched.queue_store).

    targets_in_flight: a rank's targets are folded that many at a time on host threads (default: as many as keep SIXTY-FOUR chains in
    flight -- thirty-two targets with both models, fewer for very long chains; measured on one MI355X at L=150: 8 / 16 / 32 / 64 targets
    in flight -> 102 / 134-136 / 153-154 / 154 decoys/s).  A chain's iteration phase folds one decoy at a time and leaves the chip idle;
    with shared launches (csrc/launch_engine.h, the library's rule from five live contexts on) the single-decoy folds of all chains in flight advance in
    one launch pair per evaluation, which costs about what one chain's launch pair costs (round 3, without them: four chains on
    four streams were the ceiling, 3.6 x one chain).  A target's files do not depend on what folds beside it: its decoys are
    identified by (seed, index), its contexts are its threads' own, and a fold's arithmetic does not depend on what shares its
    launches (tests/test_gpu_shared_launch.py)."""
    import time
    if run is None:
        # many chains in flight: the single-decoy folds run the one-wave-per-row pair kernel (include/trx2fold.h,
        # trx2_ctx_set_single_decoy_waves); set for every target alike, whatever targets_in_flight is: files do not depend on it
        kw.setdefault("single_decoy_waves", 1)
    run = run or run_single
    n_chain = 2 if kw.get("mult_two_models", True) else 1
    init_num = kw.get("init_num", 10)
    fasta = {n: os.path.join(fasta_dir, n + ".fasta") for n in names}
    # Order of the queue: modelled seconds of a target, longest first.  ITER_EST stands in for the unknowable iteration count
    # (the convergence test fired after 47-80 iterations on the reference's example; never more than Nmax).
    ITER_EST = 60
    items = [sched.Item(n, "all", len(read_fasta(fasta[n])), 0, init_num * n_chain, iterations=min(int(kw.get("Nmax", 300)), ITER_EST)) for n in names]
    items.sort(key=lambda it: (-it.cost, it.target))
    group = sched.summary_group(dist)                                # created up front: new_group is itself a collective
    # the counter's key names THIS job: every rank calls run_batch the same number of times with the same name list, so the call
    # count and a checksum of the names agree across ranks; a second job on the same store starts from its own zero
    global _BATCH_CALLS
    _BATCH_CALLS += 1
    import zlib
    key = f"trx2_next_item/{_BATCH_CALLS}/{zlib.crc32(' '.join(it.target for it in items).encode()):08x}"
    shared = store if store is not None else (sched.queue_store(dist, group=group) if world > 1 else None)
    static_split = False
    if world > 1 and shared is None:
        # No store to share a counter on (a library caller without an initialised process group): a process-local counter would hand
        # EVERY target to EVERY rank -- duplicated work and ranks racing on the same PDB files.  Static split instead: every
        # rank computes the same longest-first assignment from the same list and folds its own share.
        # Whole targets only: run_single folds a target with both its chains, so a decoy-block split (which lpt_assign makes when
        # there are fewer targets than ranks or no iterations) would make two ranks fold the SAME target.
        items = sched.lpt_assign(items, world, min_block=1 << 30)[rank]
        static_split = True
    queue = sched.DynamicQueue(len(items), shared, key=key)
    local = dict(decoys=0, seconds=0.0, failed=0, targets=[], errors=[])
    if targets_in_flight is None:
        # sixty-four chains in flight (measured on one MI355X, 64 targets of L=150: 16 / 32 / 64 targets in flight -> 136 / 153 / 154
        # decoys/s), fewer where the chains' tables would not leave room: a chain's context holds ~L^2 x 100 knots x 16 B of spline
        # tables (0.04 GB at L=150, 0.3 GB at L=400, 1.7 GB at L=1000) -- at most ~100 GB of the 288 for them
        per_chain = 1.3 * max(it.L for it in items) ** 2 * 100 * 16 if items else 1.0
        targets_in_flight = max(1, min(64, int(100e9 // per_chain)) // n_chain)
    lock = threading.Lock()
    t_all = time.perf_counter()

    def one(it):
        try:
            n = run(it.target, fasta[it.target], save_dir, device=device, **kw)
            with lock:
                local["decoys"] += n
                local["targets"].append(it.target)
        except Exception as e:  # noqa: BLE001 -- recorded, reported, and reflected in the exit code
            with lock:
                local["failed"] += 1
                local["errors"].append(f"{it.target}: {type(e).__name__}: {e}")
        finally:
            close_contexts()    # this worker thread's cached context (the one-model path folds on the calling thread)

    def worker():
        while True:
            i = queue.next()
            if i is None:
                return
            one(items[i])

    # never more workers than a rank's fair share of the list: with a list shorter than world x targets_in_flight the first ranks
    # to arrive would otherwise take every target and leave the others idle
    # (after a static split `items` is already this rank's private share)
    n_workers = max(1, min(int(targets_in_flight), len(items) if static_split else -(-len(items) // max(1, int(world)))))
    if n_workers == 1:
        worker()
    else:
        with ThreadPoolExecutor(max_workers=n_workers) as ex:
            for f in [ex.submit(worker) for _ in range(n_workers)]:
                f.result()
    local["seconds"] = time.perf_counter() - t_all
    per = sched.gather_stats(local, dist, group)   # gloo, 24 h timeout: ranks arrive as they finish
    return dict(decoys=sum(p["decoys"] for p in per), seconds=max(p["seconds"] for p in per), failed=sum(p["failed"] for p in per),
                errors=[e for p in per for e in p["errors"]], per_rank=per)
