#!/usr/bin/env python
"""bench.py -- decoys/sec of the MI355X-native fold, with kernel rooflines, a CPU baseline and sub-records of the other configs.

Contract:  python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run, one rank per GPU)
prints ONE JSON line on rank 0.

`value` (default, --config 0) = the job BASELINE.json's metric is quoted on: "decoys/sec (L=150, init_num=10)" = run_inference.py on one
target end to end (run_inference.py:280-337): 10 initial decoys per model as one batch, the best one fed back, then one decoy per feedback
iteration up to Nmax = 300, both models, all channels, default protocol (-m 2 --fastrelax), PDB files written; distograms resident in HBM
as float32 tensors when the clock starts.  The timed region is that ONE job (metric_job); a step is a K-th of it.  N > 1: one target per
GPU (weak scaling, no data-path collective).  `roofline` = the job's pair-energy kernel (one decoy per launch), `roofline_step` = its
dominant kernel by time (the fused minimiser step), both from HIP events around live launches; `cpu_baseline` = the same job on the CPU
port (a bounded sample run end to end + the figure for the whole job derived from its seconds per iteration).

Sub-records on the same line (N=1): config2 = BASELINE.json configs[1] (ONE trx2_fold_batch call of 64 decoys, distances only: rounds 1-5's
`value`) with its legs (pooled_queue: 1280 decoys in one call on 2 x 640 slots; in_flight_B; single_stream; no_fastrelax) and its OpenMP CPU
figure; config3 (all channels, two models), config4 (L=400, 32 decoys), config5_one_gpu; e2e: the same job on the reference's own L=90
example, run_inference's batch mode on one GPU (16 and 8 targets in flight); shared_launches; multi_gpu_plan (a prediction, unmeasured).
N>1 sub-records: config2 weak-scaling calls, config 5 (eight targets, strong scaling), batch mode over 16 N targets pulled from the shared queue.
--config 2..5 make that config `value` (what tools/ and the profiles use).

Nothing here reads /root/reference.  The oracle is imported ONLY for the cpu_baseline legs (rank 0, N=1).
"""
import argparse
import hashlib
import importlib
import json
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
SIMDS = 256 * 4        # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # max shader clock (same table); the in-kernel clock under load is lower, so busy_frac reads low
CONFIGS = {
    2: dict(L=150, B=64, orient=False, name="L=150 single target, init_num=64, dist-only, synthetic map seed 150"),
    3: dict(L=150, B=64, orient=True, chains=2,
            name="L=150 single target, init_num=64 per model, dist+omega+theta+phi, --mult_two_models: two independent chains "
                 "(synthetic maps seed 150 and 151) folded concurrently on two streams"),
    4: dict(L=400, B=32, orient=True, name="L=400 single target, init_num=32, dist+omega+theta+phi, synthetic map seed 400"),
    5: dict(L=400, B=32, orient=True, targets=(100, 140, 180, 220, 260, 300, 350, 400),
            name="eight targets L in {100,140,180,220,260,300,350,400}, init_num=32 each, dist+omega+theta+phi, synthetic maps "
                 "seed L; (target, decoy-block) items assigned to ranks longest-processing-time-first"),
}
MAX_SLOTS = 640  # decoy slots per lane of the pooled leg = every decoy of its queue in flight (the library's policy, fold.SLOTS_PER_LANE: up to 960;
                 # tools/pool_sweep.py, round 3: 1280 decoys on 2 x 192 / 320 / 640 slots -> 1121 / 1425 / 1675 decoys/s)
POOLED_QUEUE = 1280  # decoys of the pooled_queue leg (fixed: the leg does not depend on --steps)
LEG_QUEUE = 320      # decoys of the in_flight_B / single_stream legs
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_traffic.json")
TRACE_FILE = os.path.join(ROOT, "profiles", "r06_kernel_trace.json")   # rocprofv3 --kernel-trace --stats averages of the same commands (tools/make_kernel_trace_json.py)
KERNEL_SOURCES = {"k_pair": ("kernel_pair.h", "trx2_device.h"), "k_step": ("kernel_step.h", "trx2_device.h")}


def kernel_source_sha(kernel="k_pair"):
    """identifies the kernel build a committed PMC record belongs to (records must not go stale silently)"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[kernel]:
        with open(os.path.join(ROOT, "trrosettax2-dynamics_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(B, n_terms_per_decoy, L):
    """16 B per term-eval (one cubic segment) + 96*L B per decoy-eval (4 atoms x L x 12 B in and out)."""
    return B * (16.0 * n_terms_per_decoy + 96.0 * L)


def step_algorithmic_bytes(B, L, record_bytes, m):
    """Step kernel (torsion role), bytes one launch must move per active decoy and residue: the pair kernel's gradient / energy
    records in (record_bytes: 96 B per tile of the residue's row + 80 B per tile of its column, library-reported average), trial
    coordinates in (80 B: six atoms), accepted point / gradient / direction / trial torsions in and out (4 + 4 float4), the stored
    correction pairs in (m pairs x 2 x 16 B) and the new pair out (2 x 16 B), internal geometry in (48 B), coordinates out
    (80 B).  DESIGN.md section 4."""
    per_res = record_bytes + 80.0 + 8 * 16.0 + m * 32.0 + 32.0 + 48.0 + 80.0
    return B * L * per_res


def cpu_baseline(m, cfg, runs, budget_s=10.0):
    """oracle (CPU restatement of the same fold) on a bounded sample of the same workload: one thread, then OpenMP over decoys on
    every core this process may use.  `value` is the all-core figure."""
    from oracle import oracle as O
    Tb = O.Tables(m["dist"], *([m["omega"], m["theta"], m["phi"]] if cfg["orient"] else [None, None, None]))
    L = cfg["L"]
    n, t0 = 0, time.time()
    while True:
        O.fold(Tb, O.random_torsions(L, 12345, n), runs)
        n += 1
        el = time.time() - t0
        if el + el / n > budget_s or n >= 16:
            break
    one = dict(value=n / el, unit="decoys/sec", cores=1, sample=f"{n} decoys, 1 thread, {el:.1f} s")
    cores = O.usable_cores()
    per = max(1, min(4, int(budget_s / (el / n))))  # decoys per thread within the budget
    nb = cores * per
    t0 = time.time()
    _, _, st, used = O.fold_batch(Tb, np.stack([O.random_torsions(L, 12345, 100 + d) for d in range(nb)]), runs, nthreads=cores)
    el2 = time.time() - t0
    # BASELINE.md section 3 / SURVEY.md 8d: the PyRosetta leg is timed only where `import pyrosetta` succeeds; say which
    pyro_leg = None
    try:
        importlib.import_module("pyrosetta")
        pyro, pyro_note = True, "pyrosetta is importable on this host: the build's own driver (tools/pyrosetta_driver.py) was timed, see pyrosetta_leg"
        pyro_leg = pyrosetta_leg(cores)
    except Exception as e:  # noqa: BLE001 -- ModuleNotFoundError on every host seen so far
        pyro, pyro_note = False, (f"PyRosetta unavailable ({type(e).__name__}); CPU baseline = the build's own C restatement (kind: port); the build's own "
                                  "PyRosetta driver (tools/pyrosetta_driver.py) runs when a host has it")
    host = os.cpu_count() or used
    return dict(value=nb / el2, unit="decoys/sec", cores=used, kind="port",
                sample=f"{nb} decoys of the same map and protocol, oracle/trx2_oracle.c (gcc -O3 -march=native -fopenmp), "
                       f"OpenMP over decoys on {used} threads, {el2:.1f} s",
                single_thread=one, host_cores_available=host, host_cores_usable=cores,
                all_core_extrapolation={"value": nb / el2 * host / max(used, 1), "unit": "decoys/sec", "cores": host,
                                        "note": "NOT measured: the usable-core figure scaled linearly to every core the host reports "
                                                "(decoys are independent; an upper bound for this port)"},
                pyrosetta_available=pyro, pyrosetta_note=pyro_note, pyrosetta_leg=pyro_leg)


def pyrosetta_leg(cores, budget_s=120.0):
    """Only where PyRosetta is importable: the build's own driver (tools/pyrosetta_driver.py, the reference's protocol) on the
    reference's L=90 example map, one process per usable core (at most 16), with and without the full-atom stage.  decoys/sec =
    processes / wall of the slowest.  The restraint tables come from the GPU library (the same tables the GPU folds use)."""
    import subprocess
    T = importlib.import_module("trrosettax2-dynamics_amd")
    drv = importlib.import_module("tools.pyrosetta_driver") if os.path.isdir(os.path.join(ROOT, "tools")) else None
    gd = os.path.join(ROOT, "tests", "golden")
    seq = "".join(l.strip() for l in open(os.path.join(gd, "seq.fasta")) if not l.startswith(">"))
    m = np.load(os.path.join(gd, "seq_NMR.npz"))
    work = tempfile.mkdtemp(prefix="trx2_pyr_")
    try:
        ctx = T.Context(0)
        ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
        tables = os.path.join(work, "tables.npz")
        drv.dump_tables(ctx, seq, tables)
        ctx.close()
        n = max(1, min(16, cores))
        out = {}
        for tag, extra in (("no_fastrelax", ["--no-fastrelax"]), ("default", [])):
            t0 = time.time()
            procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "pyrosetta_driver.py"), "--tables", tables, "--fasta", os.path.join(gd, "seq.fasta"),
                                       "--out", os.path.join(work, f"{tag}{i}.pdb"), "--seed", str(i)] + extra, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in range(n)]
            ok = sum(p.wait(timeout=budget_s * 10) == 0 for p in procs)
            el = time.time() - t0
            out[tag] = {"value": ok / el, "unit": "decoys/sec", "cores": n, "kind": "pyrosetta", "processes_ok": ok, "seconds": el,
                        "sample": f"{n} decoys of the L=90 example map, one PyRosetta process each"}
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def traffic_record(config, decoys_per_launch, kernel="k_pair"):
    """HBM-side bytes of a kernel per launch: PMC counters cannot be collected from inside this process, so the value is the one
    measured with rocprofv3 for THIS kernel source, config and launch shape and committed under profiles/ (tools/pmc_run.sh;
    null when the sources changed since, or when no record exists for the shape this run launches)"""
    if not os.path.exists(TRAFFIC_FILE):
        return None
    rec = json.load(open(TRAFFIC_FILE))
    src = "k_step" if kernel.startswith("k_step") else "k_pair"      # which sources the kernel is built from
    if rec.get("kernel_src_sha", {}).get(src) != kernel_source_sha(src):
        return None
    for r in rec.get("records", []):
        if r["config"] == config and r["kernel_family"] == kernel and r["decoys_per_launch"] == decoys_per_launch:
            return r
    return None


def trace_avg_ms(tag, kernel_prefix):
    """average duration (ms) of the first kernel whose name starts with `kernel_prefix` in the committed rocprofv3 kernel trace of command `tag`
    (metric_job, c2, c3, c4, batch16); None when the kernel sources changed since the trace was taken"""
    if not os.path.exists(TRACE_FILE):
        return None
    rec = json.load(open(TRACE_FILE))
    src = "k_step" if kernel_prefix.startswith("k_step") else "k_pair"
    if rec.get("kernel_src_sha", {}).get(src) != kernel_source_sha(src):
        return None
    for k in rec.get("traces", {}).get(tag, {}).get("kernels", []):
        if k["kernel"].startswith(kernel_prefix):
            return k["avg_us"] * 1e-3
    return None


def pair_roofline(ctx, T, tors, L, config, fold_times=None):
    """k_pair on the FINAL coordinates of the timed decoys at the launch shape of the timed calls: one evaluation batch of their
    final torsions lays the coordinates out for exactly that many slots (full layout, nothing left over from a compacted fold:
    ADVICE r2), then 200 replays bracketed by HIP events on the kernel's own stream (trx2_time_pair_kernel)."""
    w = np.array(T.protocol.SF, np.float32)
    B = tors.shape[0]
    ctx.eval_batch(tors, w)
    ms, term_evals = ctx.time_pair_kernel(B, w, 1, L, n_rep=200)
    n_terms = term_evals / B
    abytes = algorithmic_bytes(B, n_terms, L)
    achieved = abytes / (ms * 1e-3) / 1e9
    rec = traffic_record(config, B, "k_pair")
    traffic, valu = None, None
    if rec:
        traffic = rec["hbm_bytes_per_launch"]
        # The kernel computes in f32 on the vector ALUs (nothing to contract on MFMA): next to the contract's HBM figure, how busy
        # the SIMDs' VALUs are.  SQ_ACTIVE_INST_VALU counts units of 4 cycles summed over waves; one wave64 f32 instruction holds
        # its SIMD for 4 cycles.
        cyc = rec["valu_active_quad_cycles"] * 4.0 / SIMDS
        valu = {"insts_per_launch": rec["valu_insts_per_launch"], "busy_cycles_per_simd": cyc, "busy_frac": cyc / (ms * 1e-3 * CLOCK_HZ),
                "clock_hz_assumed": CLOCK_HZ, "wave_time_waiting_frac": rec["wait_any_quad_cycles"] / rec["wave_quad_cycles"]}
    bw = int(ctx.info(0))
    out = {"bound": "hbm", "kernel": f"k_pair<{bw}> ({int(ctx.info(4))} workgroups, {B} decoys per launch)", "achieved": achieved, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": (rec or {}).get("method"),
           "avg_launch_ms": ms, "algorithmic_bytes_per_launch": abytes, "selected_terms_per_decoy": n_terms, "valu": valu,
           "binding_limit": "vector-ALU issue + dependent-load latency, not HBM bandwidth (DESIGN.md section 5)"}
    rp = trace_avg_ms(f"c{config}", f"k_pair<{bw},") if B <= 64 else None
    if rp:
        out["rocprof_avg_launch_ms"] = rp
        out["frac_rocprof"] = abytes / (rp * 1e-3) / 1e9 / HBM_PEAK_GBS
    if fold_times and fold_times[2]:
        out["avg_launch_ms_over_fold"] = fold_times[0]
        # the conservative reading: the same bytes over the fold's own average launch, which includes the narrower launches of the tail
        # (fewer decoys, fewer bytes, shorter) -- `frac` uses replays of the full-width launch on the fold's final coordinates; the
        # rocprofv3 average of the full-width instantiation (profiles/r05_c*_kernel_stats.csv) lies between the two
        out["frac_over_fold"] = abytes / (fold_times[0] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return out


def step_roofline(ctx, B, L, config, fold_times):
    """second roofline record: the fused step kernel, live average over a whole (untimed, event-sampled) fold"""
    if not fold_times or not fold_times[2]:
        return None
    rec_bytes, m = ctx.info(1), int(ctx.info(2))
    ms = fold_times[1]
    abytes = step_algorithmic_bytes(B, L, rec_bytes, m)
    ach = abytes / (ms * 1e-3) / 1e9
    rec = traffic_record(config, B, "k_step")
    return {"bound": "hbm", "kernel": f"k_step (torsion + Cartesian roles, one workgroup per decoy and role, {B} decoys per launch)", "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": rec["hbm_bytes_per_launch"] if rec else None,
            "traffic_source": (rec or {}).get("method"), "avg_launch_ms": ms,
            "algorithmic_bytes_per_launch": abytes, "samples": fold_times[2],
            "binding_limit": "latency of ~25 dependent phases on one workgroup per decoy (DESIGN.md section 4), not bandwidth"}


def sampled_fold(ctx, B, runs, seed, decoy0):
    """one extra, UNTIMED call of the same shape with every 8th evaluation bracketed by HIP events -> (pair ms, step ms, samples)"""
    ctx.set_profiling(8)
    try:
        ctx.fold_batch(B, runs, seed=seed, decoy0=decoy0)
        return ctx.last_fold_kernel_times()
    finally:
        ctx.set_profiling(0)


def metric_job(args, T, synth, rank, local_rank, world, dist, forced, L=150, init_num=10):
    """`value`: the job BASELINE.json's metric is quoted on -- "decoys/sec (L=150, init_num=10)" = run_inference.py on ONE target, end to end
    (/root/reference/run_inference.py:280-337 -> pipeline.run_single): init_num = 10 initial decoys per model as one batch, the best one fed back,
    then ONE decoy per feedback iteration until the cumulative array converges or Nmax = 300 (the CLI default; the synthetic maps never take the
    convergence exit), both models (--mult_two_models), all four channels, the reference's default protocol (-m 2 --fastrelax), every decoy written
    as a PDB file, final renaming included.  The distograms are float32 CUDA tensors -- what the trX2 front-end holds when it is done
    (utils_trX2dy/utils.py:783-796) -- so the inputs are resident in HBM when the clock starts (pointers go into the table build).
    The TIMED REGION IS THAT ONE JOB, run once; --steps K reports it as K equal slices (a step = Nmax / K feedback iterations of both chains plus a
    K-th of the initial batches: ms_per_step x K = the job's wall), --warmup W runs W such slices as a shorter job of the same shape beforehand (same
    kernels, buffers and code paths; untimed).  N > 1: every rank folds its OWN target on its own GPU (independent targets one per GPU, no data-path
    collective): weak scaling, value = decoys written by all ranks / the slowest rank's wall."""
    import contextlib
    import io
    import torch
    pipe_mod = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
    K, W, nmax = max(1, args.steps), max(0, args.warmup), int(args.nmax)
    dev = torch.device("cuda", local_rank)
    maps = [synth.make_map(L, seed=L + c) for c in range(2)]
    seq = maps[0]["seq"]
    work = tempfile.mkdtemp(prefix=f"trx2_metric_r{rank}_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    fasta = os.path.join(work, "t.fasta")
    with open(fasta, "w") as f:
        f.write(f">t\n{seq}\n")
    arrays = {tag: {k: torch.from_numpy(np.ascontiguousarray(m[k], np.float32)).to(dev) for k in ("dist", "omega", "theta", "phi")}
              for tag, m in zip(("NMR", "Xray"), maps)}
    torch.cuda.synchronize(dev)

    def job(tag, n_iter, seed, phases=None, profile_every=0):
        out_dir = os.path.join(work, tag)
        with contextlib.redirect_stdout(io.StringIO()):     # the pipeline prints the reference's progress lines
            n = pipe_mod.run_single("t", fasta, out_dir, init_num=init_num, Nmax=n_iter, angle=True, mult_two_models=True, arrays=arrays,
                                    device=local_rank, seed=seed, phase_times=phases, profile_every=profile_every)
        files = len([f for f in os.listdir(os.path.join(out_dir, "t", "pred_pdb")) if f.endswith(".pdb")])
        shutil.rmtree(out_dir, ignore_errors=True)
        return n, files

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    try:
        if W:
            job("warm", max(1, (W * nmax) // K), 900 + rank)
        sync()
        phases = {}
        t0 = time.perf_counter()
        n_out, n_files = job("timed", nmax, 7 + 1000 * rank, phases)
        sync()
        elapsed = time.perf_counter() - t0
        per_rank, decoys = [elapsed], n_out
        if dist is not None:
            tt = torch.tensor([elapsed, float(n_out)], dtype=torch.float64, device="cpu" if forced is not None else dev)
            gathered = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(gathered, tt)
            per_rank = [float(g[0].item()) for g in gathered]
            decoys = int(sum(float(g[1].item()) for g in gathered))
            elapsed = max(per_rank)
        # kernel records: one more, UNTIMED, shorter job of the same shape with every 4th evaluation of the iteration folds bracketed by HIP events
        # on the fold's own stream (rank 0)
        kern = {}
        if rank == 0:
            ph2 = {}
            job("prof", min(nmax, 16), 7, ph2, profile_every=4)
            kern = {k: v.get("kernel") for k, v in ph2.items() if v.get("kernel")}
        if dist is not None:
            dist.barrier()
        if rank != 0:
            return None
        n_iter = sum(v["iterations"] for v in phases.values())
        t_init = max(v["initial_s"] for v in phases.values())
        roof, roof_step = None, None
        if kern:
            ks = list(kern.values())
            n = sum(k["samples"] for k in ks)
            pair_ms = sum(k["pair_ms"] * k["samples"] for k in ks) / max(n, 1)
            step_ms = sum(k["step_ms"] * k["samples"] for k in ks) / max(n, 1)
            terms = float(np.mean([k["selected_terms"] for k in ks]))
            abytes = algorithmic_bytes(1, terms, L)
            tr = traffic_record("e2e_single", 1, "k_pair")
            roof = {"bound": "hbm", "kernel": f"k_pair_c<all channels, segment cache> -- ONE decoy per launch, four waves per row, {ks[0]['pair_workgroups']} workgroups; "
                                              "the pair-energy kernel of the job's iteration phase (99 % of its wall)",
                    "achieved": abytes / (pair_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": tr["hbm_bytes_per_launch"] if tr else None, "traffic_source": (tr or {}).get("method"),
                    "avg_launch_ms": pair_ms, "avg_launch_source": f"HIP events on the fold's own stream around every 4th live launch of {min(nmax, 16)} iteration folds per chain ({n} samples), untimed job of the same shape",
                    "replay_launch_ms": float(np.mean([k["replay_pair_ms"] for k in ks])), "rocprof_avg_launch_ms": trace_avg_ms("metric_job", "k_pair_c<"),
                    "algorithmic_bytes_per_launch": abytes, "selected_terms_per_decoy": terms,
                    "share_of_kernel_time": pair_ms / (pair_ms + step_ms),
                    "binding_limit": "one decoy per launch: a few hundred waves on 256 CUs for ~9 us; dependent-load latency, not HBM bandwidth (DESIGN.md section 5)"}
            if roof["rocprof_avg_launch_ms"]:      # the kernel's own duration (profiles/r06_metric_job_kernel_stats.csv): an event pair around ONE 9-us launch also spans its dispatch gaps
                roof["frac_rocprof"] = abytes / (roof["rocprof_avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            sb = step_algorithmic_bytes(1, L, ks[0]["record_bytes"], ks[0]["history_pairs"])
            trs = traffic_record("e2e_single", 1, "k_step")
            roof_step = {"bound": "hbm", "kernel": "k_step<1,256,256> -- one decoy: one workgroup in the slot's current role (torsion or Cartesian); the job's DOMINANT kernel by time",
                         "achieved": sb / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sb / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": trs["hbm_bytes_per_launch"] if trs else None, "traffic_source": (trs or {}).get("method"),
                         "avg_launch_ms": step_ms, "rocprof_avg_launch_ms": trace_avg_ms("metric_job", "k_step<1, 256, 256, false>"),
                         "algorithmic_bytes_per_launch": sb, "samples": n, "share_of_kernel_time": step_ms / (pair_ms + step_ms),
                         "binding_limit": "latency of ~25 dependent phases on ONE workgroup (DESIGN.md section 4), not bandwidth"}
        evals = sum(v.get("iteration_evals", 0) for v in phases.values()) / max(n_iter, 1)
        return {
            "metric": "decoys/sec", "value": decoys / elapsed, "unit": "decoys/sec", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE.json's metric job: run_inference end to end on one target per GPU -- L={L}, init_num={init_num} per model, two models (synthetic maps seed {L}, {L + 1}, "
                                   f"resident in HBM as float32 tensors), dist+omega+theta+phi, Nmax={nmax}" + ("" if nmax == 300 else " (NOT the CLI default 300: --nmax)") +
                                   ", default protocol (-m 2 --fastrelax), PDB files written and renamed",
                       "L": L, "init_num": init_num, "Nmax": nmax, "decoys_per_job": n_out, "pdb_files": n_files,
                       "step": f"a step is 1/{K} of the ONE timed job ({nmax}/{K} feedback iterations of both chains + 1/{K} of the initial batches); the job is run once, unsliced",
                       "warmup": f"{W} such slices as a shorter job of the same shape (Nmax={max(1, (W * nmax) // K) if W else 0}), untimed",
                       "parallelism": f"one target per GPU over {world} rank(s), no collective on the data path"},
            "roofline": roof, "roofline_step": roof_step,
            "job": {"wall_s": elapsed, "initial_phase_s": t_init, "iteration_phase_s": max(v["iteration_s"] for v in phases.values()),
                    "iterations": {k: v["iterations"] for k, v in phases.items()}, "converged": {k: bool(v.get("converged")) for k, v in phases.items()},
                    "ms_per_iteration": 1e3 * sum(v["iteration_s"] for v in phases.values()) / max(n_iter, 1),
                    "ms_per_iteration_fold": 1e3 * sum(v["iteration_fold_s"] for v in phases.values()) / max(n_iter, 1),
                    "evaluations_per_iteration_fold": evals, "us_per_evaluation": 1e3 * (1e3 * sum(v["iteration_fold_s"] for v in phases.values()) / max(n_iter, 1)) / max(evals, 1.0),
                    "note": "the two chains (models) run concurrently on two streams; within a chain the iterations are sequential single-decoy folds, each from a fresh random start"},
            "per_rank_seconds": per_rank,
        }
    finally:
        shutil.rmtree(work, ignore_errors=True)


def e2e_leg(pipe_mod, synth, L, init_num, seed=7, candidates=1, real=False, nmax=300):
    """The job the reference runs (run_inference.py:16-143,280-337; VERDICT r2 missing 2): both models of one target end to end --
    init_num initial decoys per model as one batch, the best one fed back, then ONE decoy per iteration (fold + feedback on the
    resident distograms) until the cumulative array moves by < 0.01 or Nmax = 300 (the CLI default) -- every decoy written as a
    PDB file, final renaming included.  Synthetic pair of maps (seeds L and L + 1), or with real=True the reference's own example
    (tests/golden/seq_{NMR,Xray}.npz, L=90: the only real distograms there are).  decoys/sec = files written / wall.  Per chain:
    iterations run, whether the convergence exit (max |tmp_new - tmp_old| < 0.01, run_inference.py:133-137) was taken, and the
    trace of that measure (first five, the minimum, the last)."""
    work = tempfile.mkdtemp(prefix="trx2_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        name = "t"
        fasta = os.path.join(work, name + ".fasta")
        if real:
            gd = os.path.join(ROOT, "tests", "golden")
            seq = "".join(l.strip() for l in open(os.path.join(gd, "seq.fasta")) if not l.startswith(">"))
            paths = [os.path.join(gd, "seq_NMR.npz"), os.path.join(gd, "seq_Xray.npz")]
            L = len(seq)
        else:
            maps = [synth.make_map(L, seed=L + c) for c in range(2)]
            seq = maps[0]["seq"]
            paths = []
            for tag, m in zip(("NMR", "Xray"), maps):
                q = os.path.join(work, f"{name}_{tag}.npz")
                np.savez(q, dist=m["dist"], omega=m["omega"], theta=m["theta"], phi=m["phi"])
                paths.append(q)
        with open(fasta, "w") as f:
            f.write(f">{name}\n{seq}\n")
        phases = {}
        t0 = time.perf_counter()
        n_out = pipe_mod.run_single(name, fasta, os.path.join(work, "out"), init_num=init_num, Nmax=nmax, angle=True, mult_two_models=True,
                                    npz_nmr=paths[0], npz_xray=paths[1], device=0, seed=seed, phase_times=phases, candidates=candidates)
        wall = time.perf_counter() - t0
        n_files = len([f for f in os.listdir(os.path.join(work, "out", name, "pred_pdb")) if f.endswith(".pdb")])
        it = {k: v for k, v in phases.items()}
        n_iter = sum(v["iterations"] for v in it.values())
        t_init = max(v["initial_s"] for v in it.values())           # the two chains run concurrently
        t_iter = max(v["iteration_s"] for v in it.values())

        def trace(v):
            d = v.get("tmp_change", [])
            return {"iterations": v["iterations"], "converged": bool(v.get("converged")), "first": [round(x, 4) for x in d[:5]],
                    "min": round(min(d), 4) if d else None, "last": round(d[-1], 4) if d else None}
        src = "the reference's example maps tests/golden/seq_{NMR,Xray}.npz" if real else f"synthetic maps seed {L}, {L + 1}"
        return {"workload": f"run_inference end to end: L={L}, init_num={init_num} per model, two models ({src}), all channels, "
                            f"Nmax={nmax}, default protocol (--fastrelax), PDB files written" + ("" if candidates == 1 else f"; EXTENSION (off by default): {candidates} decoys folded and written per "
                            "feedback iteration, candidate 0 fed back"), "value": n_out / wall, "unit": "decoys/sec", "decoys_written": n_out, "pdb_files": n_files,
                "wall_s": wall, "initial_phase_s": t_init, "iteration_phase_s": t_iter, "iterations": {k: v["iterations"] for k, v in it.items()},
                "convergence": {k: trace(v) for k, v in it.items()},
                "ms_per_iteration": 1e3 * sum(v["iteration_s"] for v in it.values()) / max(n_iter, 1),
                "ms_per_iteration_fold": 1e3 * sum(v["iteration_fold_s"] for v in it.values()) / max(n_iter, 1),
                "note": "chains (models) run concurrently; within a chain the iterations are sequential single-decoy folds"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_e2e_baseline(synth, L, init_num, iterations, runs, n_folds=4, t_iteration=None):
    """CPU figure for the e2e job: the job is init_num initial decoys per model, then `iterations` SEQUENTIAL
    single-decoy folds per model (run_inference.py:97-139) -- on a CPU a chain's iteration cannot use more than one core of this
    port, so the job's wall is (initial batch over the cores) + iterations x (seconds of one fold).  Measured here: `n_folds`
    oracle folds of the all-channel L-residue map, one per thread; DERIVED (stated, not measured end to end): the job's rate."""
    from oracle import oracle as O
    m = synth.make_map(L, seed=L)
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
    cores = O.usable_cores()
    nt = max(1, min(n_folds, cores))
    t0 = time.time()
    O.fold_batch(Tb, np.stack([O.random_torsions(L, 777, d) for d in range(nt)]), runs, nthreads=nt)
    t_fold = time.time() - t0                                     # nt folds side by side: the seconds of one (the slowest)
    n_chain = 2
    t_init = t_fold * -(-init_num * n_chain // cores)            # initial decoys of both models over the usable cores
    # seconds of one iteration: measured end to end on the bounded sample (cpu_e2e_measured: feedback step, tables, fold on the fed-back map, PDB file)
    # when the caller has it, else the seconds of one fold of the initial map
    t_it = t_iteration if t_iteration else t_fold
    wall = t_init + iterations * t_it                             # the two chains side by side on two cores
    n = n_chain * (init_num + iterations)
    return {"value": n / wall, "unit": "decoys/sec", "kind": "port", "cores": cores,
            "seconds_per_single_decoy_fold": t_fold, "seconds_per_iteration_used": t_it,
            "sample": f"{nt} oracle folds (L={L}, all channels, default protocol) on {nt} threads, {t_fold:.1f} s"
                      + (f"; one iteration run end to end (measured_sample): {t_it:.1f} s" if t_iteration else ""),
            "derivation": f"DERIVED for the full job from measured parts: {n} decoys / ({t_init:.1f} s initial batches + {iterations} sequential iterations x {t_it:.2f} s); "
                          "a chain's iterations are sequential, so more cores do not shorten them"}


def cpu_e2e_measured(synth, L, runs, init_num=2, nmax=2):
    """The e2e job RUN on the CPU port, end to end, on a bounded sample (the figure above is derived): both models side by side (two threads),
    per model init_num initial decoys (OpenMP over them), the reference's ranking (calculate_reliability_score on the PDB files), then nmax feedback
    iterations of { host feedback step on the best / latest PDB, tables from the fed-back maps, ONE oracle fold from a fresh random start, PDB file } --
    run_inference.py:50-139 with the oracle in the fold's place and the package's host mirror of the feedback (itself pinned bit for bit to the reference)."""
    import threading
    from oracle import oracle as O
    FB = importlib.import_module("trrosettax2-dynamics_amd.feedback")
    PD = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
    work = tempfile.mkdtemp(prefix="trx2_cpue2e_")
    maps = [synth.make_map(L, seed=L + c) for c in range(2)]
    t_iter = [[], []]

    def chain(c):
        m = maps[c]
        host = {k: m[k] for k in ("dist", "theta", "omega", "phi")}
        Tb = O.Tables(host["dist"], host["omega"], host["theta"], host["phi"], seq=m["seq"])
        _, xyz, _, _ = O.fold_batch(Tb, np.stack([O.random_torsions(L, 900 + c, d) for d in range(init_num)]), runs, nthreads=init_num)
        paths = []
        for d in range(init_num):
            q = os.path.join(work, f"c{c}_initial{d}.pdb"); PD.write_pdb(q, m["seq"], xyz[d]); paths.append(q)
        last = paths[int(np.argmax([FB.calculate_reliability_score(q) for q in paths]))]
        for it in range(nmax):
            t0 = time.time()
            host = FB.feedback_labels(host, last, 1.0, True)
            Tb = O.Tables(host["dist"], host["omega"], host["theta"], host["phi"], seq=m["seq"])
            _, xyz, _, _ = O.fold_batch(Tb, O.random_torsions(L, 950 + c, it)[None], runs, nthreads=1)
            last = os.path.join(work, f"c{c}_it{it}.pdb"); PD.write_pdb(last, m["seq"], xyz[0])
            t_iter[c].append(time.time() - t0)

    try:
        t0 = time.time()
        th = [threading.Thread(target=chain, args=(c,)) for c in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        wall = time.time() - t0
    finally:
        shutil.rmtree(work, ignore_errors=True)
    n = 2 * (init_num + nmax)
    return {"value": n / wall, "unit": "decoys/sec", "kind": "port", "cores": 2 * init_num, "seconds": wall, "decoys": n,
            "seconds_per_iteration": float(np.mean(t_iter[0] + t_iter[1])),
            "sample": f"run end to end: both models of the synthetic L={L} pair, init_num={init_num}, Nmax={nmax} (the job itself has 10 and up to 300), default protocol, PDB files written"}


def e2e_batch_leg(pipe_mod, synth, L, n_targets=16, nmax=40, seed=3, in_flight=(16,)):
    """Batch mode of run_inference.py (:339-348) on ONE GPU: n_targets targets (the same synthetic pair of maps under different names),
    init_num=10, both models, Nmax shortened to `nmax`, `in_flight` targets at a time (pipeline.run_batch's default: up to thirty-two =
    sixty-four chains whose single-decoy folds share launch pairs, csrc/launch_engine.h; the files are byte-identical to one target after the
    other with every fold launching for itself: tests/test_gpu_shared_launch.py).  Round 3's way (four streams, two targets in
    flight, every fold its own launches): 64 decoys/s without the relax stage, 41.5 with it (profiles/README.md, round 4)."""
    work = tempfile.mkdtemp(prefix="trx2_e2eb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        maps = [synth.make_map(L, seed=L + c) for c in range(2)]
        paths = []
        for tag, m in zip(("NMR", "Xray"), maps):
            q = os.path.join(work, f"m_{tag}.npz")
            np.savez(q, dist=m["dist"], omega=m["omega"], theta=m["theta"], phi=m["phi"])
            paths.append(q)
        names = [f"t{i}" for i in range(n_targets)]
        fdir = os.path.join(work, "fasta")
        os.makedirs(fdir)
        for nm in names:
            with open(os.path.join(fdir, nm + ".fasta"), "w") as f:
                f.write(f">{nm}\n{maps[0]['seq']}\n")
        out = {"workload": f"run_inference batch mode on one GPU: {n_targets} targets of L={L}, init_num=10, two models, all channels, Nmax={nmax}, PDB files written"}
        for k in in_flight:
            save = os.path.join(work, f"out{k}")
            t0 = time.perf_counter()
            res = pipe_mod.run_batch(names, fdir, save, targets_in_flight=k, init_num=10, Nmax=nmax, angle=True, mult_two_models=True, seed=seed,
                                     npz_nmr=paths[0], npz_xray=paths[1])
            el = time.perf_counter() - t0
            out[f"targets_in_flight_{k}"] = {"value": res["decoys"] / el, "unit": "decoys/sec", "decoys_written": res["decoys"], "wall_s": el, "failed": res["failed"]}
            shutil.rmtree(save, ignore_errors=True)
        out["best"] = max((out[f"targets_in_flight_{k}"] for k in in_flight), key=lambda r: r["value"])
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def fold_quality(synth, m, results):
    """Does the workload FOLD?  C-alpha RMSD of every decoy of the timed steps to the synthetic map's own target structure and to
    its mirror image (computed after the timed region): a throughput figure on a fold that fails is a cost-per-evaluation
    figure, not a fold."""
    ca = synth.nerf_backbone(m["tors"])[1]
    xyz = np.concatenate([r["xyz"][:, :, 1] for r in results]).astype(np.float64)

    def rmsd(P, Q):
        P = P - P.mean(0); Q = Q - Q.mean(0)
        U, S_, Vt = np.linalg.svd(P.T @ Q)
        dsign = np.sign(np.linalg.det(U @ Vt))
        return float(np.sqrt(max(0.0, ((P ** 2).sum() + (Q ** 2).sum() - 2 * (S_[0] + S_[1] + dsign * S_[2])) / len(P))))

    rm = np.array([rmsd(x, ca) for x in xyz])
    mir = np.array([rmsd(x * np.array([1.0, 1.0, -1.0]), ca) for x in xyz])
    return {"decoys": len(rm), "rmsd_to_target_median": float(np.median(rm)), "frac_within_2A_of_target": float((rm < 2.0).mean()),
            "frac_within_3.5A_of_mirror_image": float((mir < 3.5).mean()),
            "note": "synthetic helical-bundle target (synth.py); distance-only maps cannot fix handedness"}


def multi_target(args, cfg, T, synth, rank, local_rank, world, dist, forced, with_cpu):
    """a list of targets of different length.  Work item = (target, decoy block); every rank
    derives the same longest-first plan (sched.lpt_assign splits decoy blocks while ranks would idle or the load is uneven)
    and folds its items, up to three at a time on separate contexts (streams).  Total work is fixed: strong scaling."""
    sched = importlib.import_module("trrosettax2-dynamics_amd.sched")
    B = cfg["B"]
    items = sched.make_items([(f"L{L}", L) for L in cfg["targets"]], chains=("NMR",), init_num=B)
    mine = sched.lpt_assign(items, world)[rank]
    maps = {it.L: synth.make_map(it.L, seed=it.L) for it in mine}
    ctxs = []
    for it in mine:  # tables of every item resident before the timed region
        c = T.Context(local_rank)
        m = maps[it.L]
        c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
        ctxs.append(c)

    def fold(k, i):
        it = mine[k]
        return ctxs[k].fold_batch(it.n, T.protocol.build_runs(it.L, 2, fastrelax=True), seed=it.L, decoy0=i * B + it.decoy0)

    def step(i):
        with ThreadPoolExecutor(max_workers=3) as ex:
            return list(ex.map(lambda k: fold(k, i), range(len(mine))))

    def sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(900 + i)
    sync()
    t0 = time.perf_counter()
    res = [r for i in range(args.steps) for r in step(i)]
    t_work = time.perf_counter() - t0   # this rank's own folds, before it waits for the others
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device="cpu" if forced is not None else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ok = all(np.all(r["status"] == 0) and np.all(np.isfinite(r["xyz"])) for r in res)
    stats = sched.gather_stats(dict(decoys=sum(it.n for it in mine) * args.steps, seconds=t_work, failed=0 if ok else 1,
                                    items=[(i.target, i.decoy0, i.n) for i in mine]), dist)
    out = None
    if rank == 0:
        total = len(cfg["targets"]) * B
        out = {
            "metric": "decoys/sec", "value": args.steps * total / elapsed, "unit": "decoys/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "decoys_per_step": total, "protocol": "the reference's default: -m 2 --fastrelax",
                       "parallelism": f"{len(items)} targets -> {sum(len(p) for p in sched.lpt_assign(items, world))} items over {world} rank(s), "
                                      "no collective on the data path",
                       "items_rank0": [(i.target, i.decoy0, i.n) for i in mine]},
            "all_decoys_converged": bool(all(p["failed"] == 0 for p in stats)),
            # the longest-first plan and every rank's own seconds (before the closing barrier): the 8-GPU claim can be read off
            # this one record
            "per_rank": [{"rank": k, "decoys": p["decoys"], "seconds": p["seconds"], "items": p["items"]} for k, p in enumerate(stats)],
        }
        if mine:
            it = mine[0]  # rank 0's heaviest item: its pair kernel on the final coordinates of its last batch
            out["roofline"] = pair_roofline(ctxs[0], T, res[-len(mine)]["tors"], it.L, 5)
            out["roofline"]["kernel"] += f" of target {it.target}"
        if with_cpu:
            L0 = min(cfg["targets"])
            out["cpu_baseline"] = cpu_baseline(synth.make_map(L0, seed=L0), dict(L=L0, orient=True), T.protocol.build_runs(L0, 2, fastrelax=True))
            out["cpu_baseline"]["sample"] += f" (the L={L0} target only)"
    for c in ctxs:
        c.close()
    return out


def shared_launch_leg(T, synth, L, n_folds=32, evals=1500):
    """The kernels of the product's throughput mode (batch mode's iteration phase): n_folds single-decoy folds of n_folds contexts --
    each with its own tables of the synthetic all-channel map, as the chains of a batch job have -- at the same time, sharing launches
    (csrc/launch_engine.h), one wave per row, a fixed evaluation budget from near the structure (the state a fold spends its time in).
    Two passes: under the library's rule (32 contexts: half-evaluation launches, k_half_multi) -> fold-evaluations per second and the durations of
    the two launches of an evaluation; then with a pair launch and a step launch per evaluation (trx2_set_shared_launch_halves(0)) -> the same figure
    and a roofline record of the shared pair kernel from launch pairs the engines bracket with HIP events on their own streams (one per chunk of
    16): algorithmic bytes of the folds a sampled launch held / its duration."""
    import threading
    LB = importlib.import_module("trrosettax2-dynamics_amd._lib")
    m = synth.make_map(L, seed=L)
    runs = T.protocol.build_runs(L, 2, fastrelax=True)
    rng = np.random.default_rng(1)
    ctxs = [T.Context(0) for _ in range(n_folds)]
    try:
        for c in ctxs:
            c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
            c.set_single_decoy_waves(1)
        t0s = [(m["tors"] + rng.normal(size=m["tors"].shape) * 0.3).astype(np.float32)[None] for _ in range(n_folds)]
        for c, t in zip(ctxs, t0s):
            c.fold_batch(1, runs[5:], tors0=t, max_evals=20)
        w = np.array(T.protocol.SF, np.float32)
        ctxs[0].eval_batch(t0s[0], w)
        _, term_evals = ctxs[0].time_pair_kernel(1, w, 1, L, n_rep=2)     # selected terms of one decoy of this map
        def measure(halves):
            """one pass: halves = -1 the library's rule (half-evaluation launches from twelve live contexts on), 0 = a pair launch and a step launch per evaluation"""
            LB.set_shared_launch_halves(halves)
            LB.set_shared_launch_profiling(True)
            s0 = LB.shared_launch_stats(0)
            out = [None] * n_folds

            def work(i):
                out[i] = ctxs[i].fold_batch(1, runs[5:], tors0=t0s[i], max_evals=evals)
            th = [threading.Thread(target=work, args=(i,)) for i in range(n_folds)]
            t0 = time.perf_counter()
            [t.start() for t in th]
            [t.join() for t in th]
            el = time.perf_counter() - t0
            LB.set_shared_launch_profiling(False)
            s1 = LB.shared_launch_stats(0)
            ev = sum(int(r["n_evals"][0]) for r in out)
            ns = s1["samples"] - s0["samples"]
            r = {"fold_evaluations_per_sec": ev / el, "us_per_fold_evaluation": 1e6 * el / ev, "seconds": el}
            if ns > 0:
                r["_folds"] = (s1["sampled_folds"] - s0["sampled_folds"]) / ns
                r["_a_ms"] = (s1["pair_ms_sum"] - s0["pair_ms_sum"]) / ns
                r["_b_ms"] = (s1["step_ms_sum"] - s0["step_ms_sum"]) / ns
                r["_ns"] = int(ns)
            return r

        rec = {"workload": f"{n_folds} single-decoy folds (L={L}, all channels, own tables each) in flight at once, sharing launches; {evals} evaluations each from near the structure"}
        h = measure(-1)
        rec.update({k: v for k, v in h.items() if not k.startswith("_")})
        rec["form"] = ("half-evaluation launches (k_half_multi: one kernel steps one half of an engine's folds beside the pair terms of the other half; the library's rule "
                       "from twelve live contexts on)")
        if "_ns" in h:
            rec["half_launch_ms"] = [h["_a_ms"], h["_b_ms"]]
            rec["folds_per_sampled_launch_class"] = h["_folds"]
        p = measure(0)
        rec["pair_step_form"] = {k: v for k, v in p.items() if not k.startswith("_")}
        if "_ns" in p:
            folds, pair_ms, step_ms, ns = p["_folds"], p["_a_ms"], p["_b_ms"], p["_ns"]
            abytes = algorithmic_bytes(folds, term_evals, L)
            tr = traffic_record("shared16", 16, "k_pair1_multi")
            rec["roofline"] = {"bound": "hbm", "kernel": f"k_pair1_multi<all channels, segment cache> ({folds:.1f} folds per sampled launch, {int(ns)} samples; pair | step form)",
                               "achieved": abytes / (pair_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "avg_launch_ms": pair_ms, "algorithmic_bytes_per_launch": abytes, "selected_terms_per_decoy": term_evals,
                               "traffic": (tr["hbm_bytes_per_launch"] * folds / 16.0) if tr else None,
                               "traffic_source": (tr or {}).get("method"),
                               "step_avg_launch_ms": step_ms}
        return rec
    finally:
        LB.set_shared_launch_profiling(False)
        LB.set_shared_launch_halves(-1)
        for c in ctxs:
            c.close()


def multi_gpu_plan(T, synth, local_rank, one_gpu_seconds):
    """What 2 / 4 / 8 ranks would make of BASELINE config 5 (eight targets, L = 100 .. 400, 32 decoys each), PREDICTED from seconds
    measured on this one GPU -- no multi-GPU node has been available to anyone in four rounds: every figure here is marked unmeasured.
    Each item is timed folding ALONE (one context, one lane, default protocol); the prediction is list scheduling longest-first, what
    sched.DynamicQueue / lpt_assign do.  Stated beside it: the bound (a job is never shorter than its longest item: the L=400
    target), the same with the long items split into decoy blocks where sched.lpt_assign's model says a split pays, and the fit of
    the seconds model (sched.CostModel) to these samples.  one_gpu_seconds: the same eight items on ONE GPU three at a time
    (bench.py --config 5's way, measured by the caller) -- the denominator of the speed-up a scaling run would report."""
    sched = importlib.import_module("trrosettax2-dynamics_amd.sched")
    cfg = CONFIGS[5]
    B = cfg["B"]
    solo, samples = {}, []
    for L in cfg["targets"]:
        m = synth.make_map(L, seed=L)
        c = T.Context(local_rank)
        try:
            c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
            runs = T.protocol.build_runs(L, 2, fastrelax=True)
            for n in (B, B // 2):
                c.fold_batch(n, runs, seed=L, decoy0=10 ** 6, max_evals=2)     # buffers of this shape
                t0 = time.perf_counter()
                c.fold_batch(n, runs, seed=L, decoy0=0)
                samples.append((L, n, time.perf_counter() - t0))
            solo[L] = samples[-2][2]
        finally:
            c.close()
    half = {L: t for (L, n, t) in samples if n == B // 2}
    fit = sched.CostModel.fit(samples)
    serial = sum(solo.values())
    out = {"unmeasured": True, "note": "PREDICTION from single-GPU timings; no multi-GPU run has been observed (SCALE_r01..r03 skipped)",
           "workload": cfg["name"], "item_seconds_alone": {f"L{L}": round(t, 4) for L, t in solo.items()},
           "half_block_seconds_alone": {f"L{L}": round(t, 4) for L, t in half.items()},
           "one_gpu_seconds_serial": serial, "one_gpu_seconds_three_in_flight": one_gpu_seconds,
           "cost_model": {"form": "seconds(L, n) = c0 + c1 L + n (c2 L + c3 L^2)", "shipped": [sched.MODEL.c0, sched.MODEL.c1, sched.MODEL.c2, sched.MODEL.c3],
                          "shipped_rel_error_max": float(np.max(np.abs(sched.MODEL.rel_errors(samples)))),
                          "refit": [fit.c0, fit.c1, fit.c2, fit.c3], "refit_rel_error_max": float(np.max(np.abs(fit.rel_errors(samples))))},
           "predicted": {}}
    for N in (2, 4, 8):
        mk, _ = sched.predict_makespan(list(solo.values()), N)
        # with decoy-block splits where the model says they pay: measured half-block seconds for the halves
        items = sched.make_items([(f"L{L}", L) for L in cfg["targets"]], chains=("NMR",), init_num=B)
        plan = sched.lpt_assign(items, N)
        secs = []
        for its in plan:
            secs.append(sum(solo[it.L] if it.n == B else half[it.L] if it.n == B // 2 else solo[it.L] * sched.MODEL.call_seconds(it.L, it.n) / sched.MODEL.call_seconds(it.L, B) for it in its))
        out["predicted"][str(N)] = {"makespan_s": mk, "speedup_vs_serial": serial / mk, "speedup_vs_three_in_flight": (one_gpu_seconds / mk) if one_gpu_seconds else None,
                                    "with_block_splits": {"items": sum(len(p) for p in plan), "makespan_s": max(secs), "speedup_vs_serial": serial / max(secs),
                                                          "speedup_vs_three_in_flight": (one_gpu_seconds / max(secs)) if one_gpu_seconds else None}}
    out["bound"] = {"longest_item_s": max(solo.values()), "max_speedup_vs_serial_without_splits": serial / max(solo.values()),
                    "north_star_target": ">= 6 x at 8 GPUs in batch mode",
                    "reading": "a call is latency-bound: the job cannot end before its L=400 target does; splitting that target's decoys helps as far as a half block "
                               "is quicker than a whole one (measured above); beyond that only more targets than GPUs raise the ratio"}
    return out


def single_target(args, cfg, config, T, synth, rank, local_rank, world, dist, forced, steps, warmup, full):
    """one target (configs 2, 3, 4): every step is ONE call of B decoys per chain on every rank.  full: legs"""
    L, B = cfg["L"], cfg["B"]
    n_chains = cfg.get("chains", 1)
    ms_ = [synth.make_map(L, seed=L + c) for c in range(n_chains)]
    m = ms_[0]
    # the reference's DEFAULT protocol: mode 2 with --fastrelax on (folding/utils_ros/arguments.py:12,24-25; run_inference.py:295 never
    # disables it) = staged minimisation + the backbone-visible part of the full-atom refinement (protocol.relax_runs)
    runs = T.protocol.build_runs(L, 2, fastrelax=True)
    runs_norelax = T.protocol.build_runs(L, 2)
    # the library's defaults for a call of B decoys (fold.fold_arrays): two lanes, unless two chains already occupy two streams
    # (pipeline.run_single); one slot per decoy; default tail compaction
    lanes = 2 if n_chains == 1 else 1
    ctxs = [T.Context(local_rank, lanes=lanes) for _ in range(n_chains)]
    for c_, m_ in zip(ctxs, ms_):
        c_.set_map(m_["dist"], *([m_["omega"], m_["theta"], m_["phi"]] if cfg["orient"] else []), seq=m_["seq"])
    ctx = ctxs[0]

    def step(i, n=B, runs_=None):
        # distinct decoys for every step, rank and chain (timed steps use indices 0.., warm-up steps 900..); the chains of a
        # step are independent (run_inference.py:310-318) and run concurrently, one context = one stream each
        def one(c):
            return ctxs[c].fold_batch(n, runs_ or runs, seed=150 + c, decoy0=((rank * 1000 + i) * B))
        if n_chains == 1:
            return [one(0)]
        with ThreadPoolExecutor(max_workers=n_chains) as ex:
            return list(ex.map(one, range(n_chains)))

    def sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(warmup):
        step(900 + i)
    sync()
    t0 = time.perf_counter()
    res = [step(i) for i in range(steps)]   # K separate calls; each returns with the coordinates of its decoys on the host
    sync()
    my_elapsed = elapsed = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device="cpu" if forced is not None else "cuda")
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)
        per_rank = [float(g.item()) for g in gathered]
        elapsed = max(per_rank)
    flat = [r for st in res for r in st]
    ok = all(np.all(r["status"] == 0) and np.all(np.isfinite(r["xyz"])) for r in flat)
    evals = np.concatenate([r["n_evals"] for r in flat])
    launches = sum(r["launches"] for r in flat)   # per lane: both lanes of a context make about the same number
    legs = {}
    if full and rank == 0 and dist is None and not args.no_legs:
        def leg(lanes_, pool_, n_queue, note):
            for c_ in ctxs:
                c_.set_lanes(lanes_); c_.set_pool(pool_)
            # slot buffers sized before the clock starts (one evaluation of as many decoys as there are slots: an allocation, nothing
            # the timed fold could reuse)
            for c_ in ctxs:
                c_.fold_batch(min(n_queue, lanes_ * pool_ if pool_ else n_queue), runs, seed=149, decoy0=10 ** 6, max_evals=1)
            t1 = time.perf_counter()
            rs = step(700, n_queue)
            e1 = time.perf_counter() - t1
            return {"value": n_queue * n_chains / e1, "unit": "decoys/sec", "seconds": e1, "workload": note,
                    "pair_launches": sum(r["launches"] for r in rs) / n_chains, "slot_efficiency": float(np.mean([r["slot_efficiency"] for r in rs])),
                    "all_decoys_converged": bool(all(np.all(r["status"] == 0) for r in rs))}, rs
        # round 3's `value`: the same calls with --no-fastrelax (the protocol every earlier round benched)
        kk = max(1, min(5, steps))
        step(899, B, runs_norelax)
        t1 = time.perf_counter()
        rs_nr = [r for i in range(kk) for r in step(800 + i, B, runs_norelax)]
        e1 = time.perf_counter() - t1
        legs["no_fastrelax"] = {"value": kk * B * n_chains / e1, "unit": "decoys/sec", "steps": kk, "ms_per_step": 1e3 * e1 / kk,
                                "workload": "the same calls with --no-fastrelax: mode 2 without the relax stage (what rounds 1-3 reported as `value`)",
                                "evals_per_decoy_median": float(np.median(np.concatenate([r["n_evals"] for r in rs_nr]))),
                                "all_decoys_converged": bool(all(np.all(r["status"] == 0) for r in rs_nr))}
        if lanes == 2:
            legs["pooled_queue"], rs_pool = leg(2, MAX_SLOTS, POOLED_QUEUE, f"ONE call over a queue of {POOLED_QUEUE} decoys of the same map on 2 lanes x {MAX_SLOTS} "
                                                f"decoy slots, the library's slot policy (round 2's headline mode ran the same queue on 2 x 192 slots)")
            # kernel records at the pooled shape: MAX_SLOTS decoys per launch on lane 0
            ctx.set_lanes(1); ctx.set_pool(MAX_SLOTS)
            ftp = sampled_fold(ctx, 2 * MAX_SLOTS, runs, 150, 902 * B)
            legs["pooled_queue"]["roofline"] = compact_roofline(pair_roofline(ctx, T, rs_pool[0]["tors"][:MAX_SLOTS], L, config, ftp))
            legs["pooled_queue"]["roofline_step"] = compact_roofline(step_roofline(ctx, MAX_SLOTS, L, config, ftp))
            legs["in_flight_B"], _ = leg(2, (B + 1) // 2, LEG_QUEUE, f"a queue of {LEG_QUEUE} decoys with {B} in flight: two lanes of {(B + 1) // 2} slots")
            legs["single_stream"], _ = leg(1, B, LEG_QUEUE, f"a queue of {LEG_QUEUE} decoys with {B} in flight: ONE stream of {B} slots")
        for c_ in ctxs:
            c_.set_lanes(lanes); c_.set_pool(0)
    out = None
    if rank == 0:
        # live per-kernel averages over one more (untimed) call of the same shape: lane 0's launches
        b_lane = (B + 1) // 2 if (lanes == 2 and B >= 32) else B     # decoys per launch on lane 0 (trx2_fold_batch's split)
        ft = sampled_fold(ctx, B, runs, 150, 901 * B)
        out = {
            "metric": "decoys/sec", "value": world * steps * B * n_chains / elapsed, "unit": "decoys/sec",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "L": L, "decoys_per_step": B * n_chains,
                       "protocol": "the reference's default: -m 2 --fastrelax (staged minimisation + the backbone-visible part of the full-atom "
                                   f"refinement, {len(runs)} minimiser runs); the no_fastrelax leg carries the {len(runs_norelax)}-run protocol of rounds 1-3",
                       "call": f"one trx2_fold_batch of {B} decoys per chain and step, library defaults: {lanes} lane(s) "
                               f"({b_lane} decoys per launch), one slot per decoy, default tail compaction",
                       "parallelism": f"calls sharded over {world} rank(s), no collective on the data path"},
            "roofline": pair_roofline(ctx, T, flat[-n_chains]["tors"][:b_lane], L, config, ft),
            "roofline_step": step_roofline(ctx, b_lane, L, config, ft),
            "all_decoys_converged": bool(ok), "evals_per_decoy": {"min": int(evals.min()), "median": float(np.median(evals)), "max": int(evals.max())},
            "fold_quality": fold_quality(synth, m, [r for r in flat[::n_chains]]),
            "pair_launches_per_step": launches / steps / n_chains,
            "slot_efficiency": float(np.mean([r["slot_efficiency"] for r in flat])),  # evaluations of the decoys / (launch pairs x slots), per lane
        }
        if per_rank is not None:
            out["per_rank_seconds"] = per_rank
        out.update(legs)
    for c_ in ctxs:
        c_.close()
    return out, m, runs


def batch_mode_multi(args, T, synth, rank, local_rank, world, dist, forced, L=150, targets_per_gpu=16, nmax=10):
    """N > 1: the north star's multi-GPU mode IS `value`: run_inference.py's batch mode (:339-348) over 16 N independent
    targets, ranks PULLING targets from the shared counter (pipeline.run_batch, sched.DynamicQueue; no collective on the data path, one gather of
    the summaries at the end).  Sixteen targets per GPU -- 32 chains in flight -- is where a GPU's shared launches fill the chip (one GPU: 8 / 16 / 32
    targets in flight 95 / 162 / 179 decoys/s).  A step = one such job: 16 N targets of L = 150, init_num = 10, both models, all channels, the default protocol,
    Nmax shortened to `nmax` feedback iterations per chain so that W + K steps fit the driver's run (stated in `config.workload`), PDB files
    written.  Sixteen targets PER GPU: the work grows with N, so the record says `scaling: weak`.  A sub-record of the N > 1 line since
    round 6 (`value` is the metric's job, one target per GPU).  UNMEASURED until a multi-GPU node runs it: no such node has been available
    to the builder in six rounds; the two-rank rehearsal on one GPU (tests/test_gpu_bench.py) exercises the control flow only."""
    import contextlib
    import io
    import torch
    pipe_mod = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
    n_targets = targets_per_gpu * world
    work = [None]
    if rank == 0:
        work[0] = tempfile.mkdtemp(prefix="trx2_bm_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        maps = [synth.make_map(L, seed=L + c) for c in range(2)]
        for tag, m in zip(("NMR", "Xray"), maps):
            np.savez(os.path.join(work[0], f"m_{tag}.npz"), dist=m["dist"], omega=m["omega"], theta=m["theta"], phi=m["phi"])
        os.makedirs(os.path.join(work[0], "fasta"))
        for i in range(n_targets):
            with open(os.path.join(work[0], "fasta", f"t{i}.fasta"), "w") as f:
                f.write(f">t{i}\n{maps[0]['seq']}\n")
    dist.broadcast_object_list(work, src=0)          # one node: every rank sees rank 0's /dev/shm
    work = work[0]
    names = [f"t{i}" for i in range(n_targets)]
    paths = [os.path.join(work, f"m_{tag}.npz") for tag in ("NMR", "Xray")]

    def step(i):
        save = os.path.join(work, f"out{i}")
        with contextlib.redirect_stdout(io.StringIO()):
            res = pipe_mod.run_batch(names, os.path.join(work, "fasta"), save, rank=rank, world=world, dist=dist, device=local_rank, init_num=10, Nmax=nmax,
                                     angle=True, mult_two_models=True, seed=1000 * i + 3, npz_nmr=paths[0], npz_xray=paths[1])
        dist.barrier()
        if rank == 0:
            shutil.rmtree(save, ignore_errors=True)
        return res

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    try:
        for i in range(args.warmup):
            step(900 + i)
        sync()
        t0 = time.perf_counter()
        res = [step(i) for i in range(args.steps)]
        sync()
        elapsed = time.perf_counter() - t0
        tt = torch.tensor([elapsed], device="cpu" if forced is not None else "cuda")
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)
        per_rank = [float(g.item()) for g in gathered]
        elapsed = max(per_rank)
        decoys = sum(r["decoys"] for r in res)
        if rank != 0:
            return None
        return {
            "metric": "decoys/sec", "value": decoys / elapsed, "unit": "decoys/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(1, args.steps), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"run_inference batch mode: {n_targets} independent targets of L={L} ({targets_per_gpu} per GPU), init_num=10, both models, all channels, default protocol "
                                   f"(-m 2 --fastrelax), Nmax={nmax} feedback iterations per chain (the CLI's default is 300: shortened so that warm-up + timed steps fit the run), PDB files written",
                       "L": L, "decoys_per_step": decoys // max(1, args.steps),
                       "parallelism": f"targets pulled from a shared counter by {world} rank(s), one process per GPU; no collective on the data path; one summary gather per job",
                       "measured_on_hardware": "this record is the first measurement of the queue's multi-GPU path whenever the driver produces it: the builder has had no multi-GPU node in six rounds"},
            "all_targets_folded": all(r["failed"] == 0 for r in res), "per_rank_seconds": per_rank,
            "per_rank_decoys_last_step": [p["decoys"] for p in res[-1]["per_rank"]] if res else None,
        }
    finally:
        dist.barrier()
        if rank == 0:
            shutil.rmtree(work, ignore_errors=True)


def pooled_all_channels(T, synth, local_rank, config=3, L=150):
    """the pair kernel's ceiling -- ONE call over a queue of 1280 ALL-CHANNEL decoys of config 3's map on 2 lanes x 640 slots
    (k_pair<64, all channels>, ten decoy groups per launch: the chip full, three waves per SIMD), and the kernel records at that shape."""
    m = synth.make_map(L, seed=L)
    runs = T.protocol.build_runs(L, 2, fastrelax=True)
    ctx = T.Context(local_rank, lanes=2)
    try:
        ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
        ctx.set_pool(MAX_SLOTS)
        ctx.fold_batch(2 * MAX_SLOTS, runs, seed=149, decoy0=10 ** 6, max_evals=1)      # slot buffers sized before the clock starts
        t1 = time.perf_counter()
        r = ctx.fold_batch(POOLED_QUEUE, runs, seed=150, decoy0=7000)
        e1 = time.perf_counter() - t1
        out = {"value": POOLED_QUEUE / e1, "unit": "decoys/sec", "seconds": e1,
               "workload": f"ONE call over a queue of {POOLED_QUEUE} decoys of config 3's first map (L={L}, dist+omega+theta+phi) on 2 lanes x {MAX_SLOTS} decoy slots, default protocol",
               "pair_launches": r["launches"], "slot_efficiency": r["slot_efficiency"], "all_decoys_converged": bool(np.all(r["status"] == 0)),
               "evals_per_decoy_median": float(np.median(r["n_evals"]))}
        ctx.set_lanes(1); ctx.set_pool(MAX_SLOTS)
        ftp = sampled_fold(ctx, 2 * MAX_SLOTS, runs, 150, 9000)
        out["roofline"] = compact_roofline(pair_roofline(ctx, T, r["tors"][:MAX_SLOTS], L, config, ftp))
        out["roofline_step"] = compact_roofline(step_roofline(ctx, MAX_SLOTS, L, config, ftp))
        rec = traffic_record(config, MAX_SLOTS, "k_pair")
        if rec and out["roofline"]:
            terms = out["roofline"]["algorithmic_bytes_per_launch"] / MAX_SLOTS      # bytes per decoy = 16 n_terms + 96 L
            n_terms = (terms - 96.0 * L) / 16.0
            out["roofline"]["valu_lane_ops_per_selected_term"] = rec["valu_insts_per_launch"] * 64.0 / (MAX_SLOTS * n_terms)
            out["roofline"]["wave_time_waiting_frac"] = rec["wait_any_quad_cycles"] / rec["wave_quad_cycles"]
        return out
    finally:
        ctx.close()


def compact_roofline(r):
    return {q: r[q] for q in ("kernel", "achieved", "frac", "frac_over_fold", "frac_rocprof", "unit", "avg_launch_ms", "avg_launch_ms_over_fold", "rocprof_avg_launch_ms",
                              "algorithmic_bytes_per_launch", "traffic") if q in r} if r else None


def compact(rec):
    """sub-record of another config inside the default line: value, time per step, convergence, the kernel records"""
    keys = ("value", "unit", "steps", "ms_per_step", "all_decoys_converged", "evals_per_decoy", "pair_launches_per_step", "slot_efficiency", "fold_quality")
    out = {k: rec[k] for k in keys if k in rec}
    out["workload"] = rec["config"]["workload"]
    out["call"] = rec["config"]["call"]
    for k in ("roofline", "roofline_step"):
        if rec.get(k):
            out[k] = compact_roofline(rec[k])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=[0] + sorted(CONFIGS),
                    help="0 (default): the job BASELINE.json's metric is quoted on (run_inference end to end, L=150, init_num=10); 2-5: BASELINE.json's configs[1..4] as `value`")
    ap.add_argument("--nmax", type=int, default=300, help="feedback iterations per chain of the metric's job (300 = the CLI default of run_inference.py; anything else is named in config.workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the pooled-queue / B-in-flight / single-stream legs (profiling: every launch in the trace then belongs to `value`'s calls)")
    ap.add_argument("--no-sub-records", action="store_true", help="skip the sub-records of the other configs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs beside `value` (the L=90 example, batch mode on one GPU, shared launches)")
    ap.add_argument("--all-e2e", action="store_true", help="also run the init_num=64 and --candidates 8 end-to-end legs (another ~2 minutes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # Rehearsal on a box with fewer GPUs than ranks: TRX2_BENCH_FORCE_DEVICE=0 puts every rank on that GPU and uses gloo
    # (NCCL refuses two ranks on one device).  It exercises the multi-rank control flow, not multi-GPU performance.
    forced = os.environ.get("TRX2_BENCH_FORCE_DEVICE")
    if forced is not None:
        local_rank = int(forced)
    # Rehearsal of the N > 1 path on RCCL with the one GPU a box has: TRX2_BENCH_REHEARSE_NCCL=1 under torchrun with ONE rank takes every branch a
    # multi-GPU run takes (process group on nccl, barriers, object broadcast, CUDA all-gather, the queue's store and gloo group) -- world size 1, so
    # what it shows is that RCCL initialises and the collectives run with this code, nothing about scaling (tests/test_gpu_bench.py).
    multi = world > 1 or os.environ.get("TRX2_BENCH_REHEARSE_NCCL") == "1"
    if multi:
        import datetime
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if forced is not None:
            dist.init_process_group("gloo", timeout=datetime.timedelta(hours=2))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(hours=2))

    T = importlib.import_module("trrosettax2-dynamics_amd")
    synth = importlib.import_module("trrosettax2-dynamics_amd.synth")
    with_cpu = not multi and not args.no_cpu_baseline
    k_sub = max(1, min(5, args.steps))
    if args.config == 0:
        out = metric_job(args, T, synth, rank, local_rank, world, dist, forced)
        L = 150
        runs = T.protocol.build_runs(L, 2, fastrelax=True)
        if rank == 0 and with_cpu:
            # the same job on the CPU port: a bounded sample RUN end to end (both models, init_num = 2, Nmax = 2), and from its seconds per
            # iteration the figure for the whole job (a chain's iterations are sequential: more cores do not shorten them)
            its = max(out["job"]["iterations"].values())
            ms = cpu_e2e_measured(synth, L, runs)
            out["cpu_baseline"] = cpu_e2e_baseline(synth, L, 10, its, runs, t_iteration=ms["seconds_per_iteration"])
            out["cpu_baseline"]["measured_sample"] = ms
        sub = {}
        if not args.no_sub_records:
            a_sub = argparse.Namespace(**{**vars(args), "steps": k_sub, "warmup": 1})
            # BASELINE.json configs[1]: one call of 64 decoys, distances only (rounds 1-5's `value`), with its legs and kernel records
            r2, m2, runs2 = single_target(a_sub, CONFIGS[2], 2, T, synth, rank, local_rank, world, dist, forced, k_sub, 1, True)
            if rank == 0:
                sub["config2"] = compact(r2)
                for k in ("no_fastrelax", "pooled_queue", "in_flight_B", "single_stream", "per_rank_seconds", "scaling", "n_gpus"):
                    if k in r2:
                        sub["config2"][k] = r2[k]
                if with_cpu:
                    sub["config2"]["cpu_baseline"] = cpu_baseline(m2, CONFIGS[2], runs2)
            if not multi:
                for c in (3, 4):
                    r, _, _ = single_target(a_sub, CONFIGS[c], c, T, synth, rank, local_rank, world, dist, forced, k_sub, 1, False)
                    sub[f"config{c}"] = compact(r)
                if not args.no_legs:
                    sub["config3"]["pooled_queue"] = pooled_all_channels(T, synth, local_rank)
                a5 = argparse.Namespace(steps=1, warmup=1)
                bm = multi_target(a5, CONFIGS[5], T, synth, rank, local_rank, world, dist, forced, False)
                sub["config5_one_gpu"] = {"value": bm["value"], "unit": bm["unit"], "ms_per_step": bm["ms_per_step"], "workload": bm["config"]["workload"]}
                out["multi_gpu_plan"] = multi_gpu_plan(T, synth, local_rank, bm["ms_per_step"] * 1e-3)
            else:
                # the other multi-GPU shapes: config 5 (eight targets of different length, strong scaling) and run_inference's batch mode with sixteen
                # targets per GPU pulled from the shared queue (work grows with N: weak) -- one step each
                bm = multi_target(argparse.Namespace(steps=1, warmup=0), CONFIGS[5], T, synth, rank, local_rank, world, dist, forced, False)
                bq = batch_mode_multi(argparse.Namespace(steps=1, warmup=1), T, synth, rank, local_rank, world, dist, forced)
                if rank == 0:
                    sub["config5_batch_mode"] = {k: bm[k] for k in ("value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "all_decoys_converged", "per_rank")}
                    sub["config5_batch_mode"]["workload"] = bm["config"]["workload"]
                    sub["config5_batch_mode"]["parallelism"] = bm["config"]["parallelism"]
                    sub["batch_mode_queue"] = {k: bq[k] for k in ("value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "all_targets_folded", "per_rank_seconds", "per_rank_decoys_last_step")}
                    sub["batch_mode_queue"]["workload"] = bq["config"]["workload"]
                    sub["batch_mode_queue"]["decoys_per_step"] = bq["config"]["decoys_per_step"]
        if rank == 0:
            out["sub_records"] = sub
        if not multi and rank == 0 and not args.no_e2e:
            pipe_mod = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                # the same job on the reference's own example (L=90: the only real distograms there are; its chains take the convergence exit)
                out["e2e"] = {"example_L90_init_num_10": e2e_leg(pipe_mod, synth, 90, 10, real=True)}
                if args.all_e2e:
                    out["e2e"]["init_num_64"] = e2e_leg(pipe_mod, synth, L, 64)
                    out["e2e"]["init_num_10_candidates_8"] = e2e_leg(pipe_mod, synth, L, 10, candidates=8)
                out["e2e"]["batch_mode_one_gpu"] = e2e_batch_leg(pipe_mod, synth, L)
                # ... and the shape VERDICT r3 set its target on: 8 targets x 2 models, Nmax = 80, all eight in flight
                out["e2e"]["batch_mode_8_targets_nmax80"] = e2e_batch_leg(pipe_mod, synth, L, n_targets=8, nmax=80, in_flight=(8,))
            out["shared_launches"] = shared_launch_leg(T, synth, L)
    else:
        cfg = CONFIGS[args.config]
        if "targets" in cfg:
            out = multi_target(args, cfg, T, synth, rank, local_rank, world, dist, forced, with_cpu)
        else:
            out, m, runs = single_target(args, cfg, args.config, T, synth, rank, local_rank, world, dist, forced, args.steps, args.warmup, True)
            if rank == 0 and with_cpu:
                out["cpu_baseline"] = cpu_baseline(m, cfg, runs)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # Short numeric copies of the other records inside `config` (the driver's parsed record keeps `config` whole).  decoys/sec each.
        def num(*path):
            d = out
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
                if d is None:
                    return None
            return round(float(d), 2)
        for key, path in (("e2e_example_L90_init10", ("e2e", "example_L90_init_num_10", "value")),
                          ("cpu_same_job", ("cpu_baseline", "value")), ("batch_mode_one_gpu", ("e2e", "batch_mode_one_gpu", "best", "value")),
                          ("batch_8x2_nmax80", ("e2e", "batch_mode_8_targets_nmax80", "best", "value")),
                          ("shared_fold_evals_per_s", ("shared_launches", "fold_evaluations_per_sec")),
                          ("c2_call_of_64", ("sub_records", "config2", "value")), ("c2_pooled_1280", ("sub_records", "config2", "pooled_queue", "value")),
                          ("c2_no_fastrelax", ("sub_records", "config2", "no_fastrelax", "value")),
                          ("c3", ("sub_records", "config3", "value")), ("c3_pooled_1280", ("sub_records", "config3", "pooled_queue", "value")), ("c4", ("sub_records", "config4", "value")), ("c5_one_gpu", ("sub_records", "config5_one_gpu", "value")),
                          ("batch_mode_queue", ("sub_records", "batch_mode_queue", "value")),
                          ("predicted_speedup_8gpu_unmeasured", ("multi_gpu_plan", "predicted", "8", "with_block_splits", "speedup_vs_three_in_flight"))):
            v = num(*path)
            if v is not None:
                out["config"][key] = v
        print(json.dumps(out))


if __name__ == "__main__":
    main()
