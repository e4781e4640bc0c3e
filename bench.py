#!/usr/bin/env python
"""bench.py -- decoys/sec of the MI355X-native fold, with kernel rooflines and a CPU baseline.

Contract:  python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run, one rank per GPU)
prints ONE JSON line on rank 0.

A "step" = one pass of the hot path over one batch: B decoys of one distogram folded through the full staged protocol
(folding/folding.py:118-171 mode 2), timed from tables-resident-in-HBM to the last coordinates on the host.  The K steps of a
run are ONE queue of K x B decoys: every decoy of every step is folded inside the timed region.  The queue runs on decoy slots
(trx2_ctx_set_pool): a slot whose decoy has finished takes the next decoy on the device, as the reference's process pool starts
the next `folding.py` child when a worker frees up (utils_trX2dy/utils.py:501-503).  How many slots is a scheduling choice of the
library's user, not part of the workload: both kernels are latency-bound, a launch over 192 slots takes 2.7 x as long as one
over 32 (profiles/README.md), so `value` uses two lanes (two streams: one lane's step kernel overlaps the other's pair kernel,
trx2_ctx_set_lanes) of min(192, K B / 2) slots each.  Beside `value`, the same queue with only B decoys in flight:
`in_flight_B` (two lanes of B/2 slots: round 2's first `value`) and `single_stream` (one stream of B slots); and `per_call`
(K separate calls of B decoys, one slot per decoy, each call ending with its slowest decoy: round 1's `value`).
Workload at N=1 = BASELINE.json configs[1]: L=150 single target, init_num=64, dist-only restraints (synthetic map,
SURVEY.md 8d -- the reference ships data for L=90 only).  Other configs: --config 3 (all channels, two models),
--config 4 (L=400, B=32), --config 5 (eight targets L=100..400, 32 decoys each, assigned to ranks longest-first: strong
scaling).  N>1: every rank folds its own B decoys of the same target (independent units, no data-path collective): weak
scaling; the same line then also carries the config-5 batch-mode record ("batch_mode", strong scaling), which is the
north star's multi-GPU mode.  At N=1 the line carries compact sub-records for configs 3 and 4 ("sub_records").

Nothing here reads /root/reference.  The oracle is imported ONLY for the cpu_baseline leg (rank 0, N=1).
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
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
MAX_SLOTS = 192  # decoy slots per lane (three groups of 64 decoys in the pair kernel): tools/pool_sweep.py
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r02_traffic.json")
KERNEL_SOURCES = ("kernel_pair.h", "trx2_device.h")


def kernel_source_sha():
    """identifies the pair-kernel build a committed PMC record belongs to (records must not go stale silently)"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
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
    return dict(value=nb / el2, unit="decoys/sec", cores=used, kind="port",
                sample=f"{nb} decoys of the same map and protocol, oracle/trx2_oracle.c (gcc -O3 -march=native -fopenmp), "
                       f"OpenMP over decoys on {used} threads, {el2:.1f} s",
                single_thread=one, host_cores_available=os.cpu_count(), host_cores_usable=cores)


def traffic_record(config, decoys_per_launch):
    """HBM-side bytes of the pair kernel per launch: PMC counters cannot be collected from inside this process, so the value is the
    one measured with rocprofv3 for THIS kernel source, config and launch shape and committed under profiles/ (null when the
    sources changed, or when --steps gives the lanes another number of slots than the record's)"""
    if not os.path.exists(TRAFFIC_FILE):
        return None
    rec = json.load(open(TRAFFIC_FILE))
    if rec.get("kernel_src_sha") != kernel_source_sha():
        return None
    r = rec.get(str(config))
    return r if r and r.get("decoys_per_launch") == decoys_per_launch else None


def pair_roofline(ctx, T, B, L, config, fold_times=None):
    w = np.array(T.protocol.SF, np.float32)
    ms, term_evals = ctx.time_pair_kernel(B, w, 1, L, n_rep=200)
    n_terms = term_evals / B
    abytes = algorithmic_bytes(B, n_terms, L)
    achieved = abytes / (ms * 1e-3) / 1e9
    rec = traffic_record(config, B)
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
    out = {"bound": "hbm", "kernel": f"k_pair<{bw}> ({int(ctx.info(4))} workgroups)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": (rec or {}).get("method"),
           "avg_launch_ms": ms, "algorithmic_bytes_per_launch": abytes, "selected_terms_per_decoy": n_terms, "valu": valu,
           "binding_limit": "vector-ALU issue + dependent-load latency, not HBM bandwidth (DESIGN.md section 5)"}
    if fold_times and fold_times[2]:
        out["avg_launch_ms_over_fold"] = fold_times[0]
    return out


def step_roofline(ctx, B, L, fold_times):
    """second roofline record: the fused step kernel, live average over a whole (untimed, event-sampled) fold"""
    if not fold_times or not fold_times[2]:
        return None
    rec_bytes, m = ctx.info(1), int(ctx.info(2))
    ms = fold_times[1]
    abytes = step_algorithmic_bytes(B, L, rec_bytes, m)
    ach = abytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_step (torsion + Cartesian roles, one workgroup per decoy and role)", "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms,
            "algorithmic_bytes_per_launch": abytes, "samples": fold_times[2],
            "binding_limit": "latency of ~25 dependent phases on one workgroup per decoy (DESIGN.md section 4), not bandwidth"}


def sampled_fold(ctx, B, runs, seed, decoy0):  # B decoys through the context's slot pool
    """one extra, UNTIMED fold with every 8th evaluation bracketed by HIP events -> (pair ms, step ms, samples)"""
    ctx.set_profiling(8)
    try:
        ctx.fold_batch(B, runs, seed=seed, decoy0=decoy0)
        return ctx.last_fold_kernel_times()
    finally:
        ctx.set_profiling(0)


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
        return ctxs[k].fold_batch(it.n, T.protocol.build_runs(it.L, 2), seed=it.L, decoy0=i * B + it.decoy0)

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
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device="cpu" if forced is not None else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ok = all(np.all(r["status"] == 0) and np.all(np.isfinite(r["xyz"])) for r in res)
    stats = sched.gather_stats(dict(decoys=sum(it.n for it in mine) * args.steps, seconds=elapsed, failed=0 if ok else 1), dist)
    out = None
    if rank == 0:
        total = len(cfg["targets"]) * B
        out = {
            "metric": "decoys/sec", "value": args.steps * total / elapsed, "unit": "decoys/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "decoys_per_step": total, "protocol": "mode 2, full staged minimisation",
                       "parallelism": f"{len(items)} targets -> {sum(len(p) for p in sched.lpt_assign(items, world))} items over {world} rank(s), "
                                      "no collective on the data path",
                       "items_rank0": [(i.target, i.decoy0, i.n) for i in mine]},
            "all_decoys_converged": bool(all(p["failed"] == 0 for p in stats)),
        }
        if mine:
            it = mine[0]  # rank 0's heaviest item: its pair kernel on the coordinates of the last batch
            out["roofline"] = pair_roofline(ctxs[0], T, it.n, it.L, 5)
            out["roofline"]["kernel"] += f" of target {it.target}"
        if with_cpu:
            L0 = min(cfg["targets"])
            out["cpu_baseline"] = cpu_baseline(synth.make_map(L0, seed=L0), dict(L=L0, orient=True), T.protocol.build_runs(L0, 2))
            out["cpu_baseline"]["sample"] += f" (the L={L0} target only)"
    for c in ctxs:
        c.close()
    return out


def single_target(args, cfg, config, T, synth, rank, local_rank, world, dist, forced, steps, warmup, full):
    """one target (configs 2, 3, 4): B decoys per chain and step on every rank.  full: roofline records, two-lane leg"""
    L, B = cfg["L"], cfg["B"]
    n_chains = cfg.get("chains", 1)
    ms_ = [synth.make_map(L, seed=L + c) for c in range(n_chains)]
    m = ms_[0]
    runs = T.protocol.build_runs(L, 2)
    lanes = 2 if n_chains == 1 else 1          # two chains already occupy two streams
    per_lane = (steps * B + lanes - 1) // lanes  # decoys a lane folds in the timed region
    slots = min(MAX_SLOTS, per_lane)             # per lane
    ctxs = [T.Context(local_rank, lanes=lanes) for _ in range(n_chains)]
    for c_, m_ in zip(ctxs, ms_):
        c_.set_map(m_["dist"], *([m_["omega"], m_["theta"], m_["phi"]] if cfg["orient"] else []), seq=m_["seq"])
    ctx = ctxs[0]

    def step(i, n_steps=1):
        # distinct decoys for every step, rank and chain (timed steps use indices 0.., warm-up steps 900..); the chains of a
        # step are independent (run_inference.py:310-318) and run concurrently, one context = one stream each
        def one(c):
            return ctxs[c].fold_batch(n_steps * B, runs, seed=150 + c, decoy0=((rank * 1000 + i) * B))
        if n_chains == 1:
            return [one(0)]
        with ThreadPoolExecutor(max_workers=n_chains) as ex:
            return list(ex.map(one, range(n_chains)))

    def sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for c_ in ctxs:
        c_.set_pool(slots)       # B decoy slots per chain; the K steps' decoys are their queue
    for i in range(warmup):
        step(900 + i)
    if warmup > 0 and steps > 1:
        # the warm-up steps fold B decoys each; the timed queue runs on more slots: size the slot buffers for it now (one
        # evaluation of as many decoys as there are slots -- an allocation, nothing the timed folds could reuse), so that the
        # timed region holds folds and no hipMalloc
        for c_ in ctxs:
            c_.fold_batch(min(steps * B, lanes * slots), runs, seed=149, decoy0=10 ** 6, max_evals=1)
    sync()
    t0 = time.perf_counter()
    res = step(0, steps)         # fold_batch returns with the coordinates of every decoy on the host
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device="cpu" if forced is not None else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ok = all(np.all(r["status"] == 0) and np.all(np.isfinite(r["xyz"])) for r in res)
    evals = np.concatenate([r["n_evals"] for r in res])
    launches = sum(r["launches"] for r in res)   # per lane: both lanes of a context make about the same number
    per_call = single = wide = None
    if full and rank == 0 and world == 1 and not args.no_legs:
        def leg(lanes_, pool_, calls):
            for c_ in ctxs:
                c_.set_lanes(lanes_); c_.set_pool(pool_)
            step(900, 1)
            t1 = time.perf_counter()
            rs = [r for i in range(steps) for r in step(i)] if calls else step(0, steps)
            e1 = time.perf_counter() - t1
            nl = sum(r["launches"] for r in rs) / n_chains
            return {"value": steps * B * n_chains / e1, "unit": "decoys/sec", "ms_per_step": 1e3 * e1 / steps, "pair_launches_per_step": nl / steps,
                    "all_decoys_converged": bool(all(np.all(r["status"] == 0) for r in rs))}
        # the same decoys as K separate calls of B on one stream (one slot per decoy: every call ends with its slowest decoy)
        per_call = leg(1, 0, True)
        per_call["note"] = "K separate trx2_fold_batch calls of B decoys, one stream, one slot per decoy (round 1's `value`)"
        if lanes == 2:
            single = leg(1, B, False)
            single["note"] = f"the same queue with {B} decoys in flight: ONE stream of {B} slots"
            wide = leg(2, (B + 1) // 2, False)
            wide["note"] = f"the same queue with {B} decoys in flight: two lanes of {(B + 1) // 2} slots (round 2's first `value`)"
        for c_ in ctxs:
            c_.set_lanes(lanes); c_.set_pool(slots)
    out = None
    if rank == 0:
        ft = sampled_fold(ctx, steps * B, runs, 150, 901 * B)  # untimed; live per-kernel averages over the same queue once more: lane 0's launches
        out = {
            "metric": "decoys/sec", "value": world * steps * B * n_chains / elapsed, "unit": "decoys/sec",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "L": L, "decoys_per_step": B * n_chains, "protocol": "mode 2, full staged minimisation",
                       "slots": f"{lanes} lane(s) x {slots} decoy slots per chain (queue of {steps * B} decoys per chain)",
                       "parallelism": f"decoys sharded over {world} rank(s), no collective on the data path"},
            "roofline": pair_roofline(ctx, T, slots, L, config, ft),
            "roofline_step": step_roofline(ctx, slots, L, ft),
            "all_decoys_converged": bool(ok), "evals_per_decoy": {"min": int(evals.min()), "median": float(np.median(evals)), "max": int(evals.max())},
            "fold_quality": fold_quality(synth, m, res[:1]),
            "pair_launches_per_step": launches / steps / n_chains,
            "slot_efficiency": float(np.mean([r["slot_efficiency"] for r in res])),  # evaluations of the decoys / (launch pairs x slots), per lane
        }
        for k_, v_ in (("in_flight_B", wide), ("single_stream", single), ("per_call", per_call)):
            if v_:
                out[k_] = v_
    for c_ in ctxs:
        c_.close()
    return out, m, runs


def compact(rec):
    """sub-record of another config inside the default line: value, time per step, convergence, the kernel records"""
    keys = ("value", "unit", "steps", "ms_per_step", "all_decoys_converged", "evals_per_decoy", "pair_launches_per_step", "slot_efficiency", "fold_quality")
    out = {k: rec[k] for k in keys if k in rec}
    out["workload"] = rec["config"]["workload"]
    for k in ("roofline", "roofline_step"):
        r = rec.get(k)
        if r:
            out[k] = {q: r[q] for q in ("kernel", "achieved", "frac", "unit", "avg_launch_ms", "algorithmic_bytes_per_launch", "traffic") if q in r}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the B-in-flight / single-stream / per-call legs (profiling: every launch in the trace then belongs to `value`'s queue)")
    ap.add_argument("--no-sub-records", action="store_true", help="skip the config 3 / 4 sub-records (N=1) and the batch-mode record (N>1)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # Rehearsal on a box with fewer GPUs than ranks: TRX2_BENCH_FORCE_DEVICE=0 puts every rank on that GPU and uses gloo
    # (NCCL refuses two ranks on one device).  It exercises the multi-rank control flow, not multi-GPU performance.
    forced = os.environ.get("TRX2_BENCH_FORCE_DEVICE")
    if forced is not None:
        local_rank = int(forced)
    if world > 1:
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
    with_cpu = world == 1 and not args.no_cpu_baseline
    if "targets" in cfg:
        out = multi_target(args, cfg, T, synth, rank, local_rank, world, dist, forced, with_cpu)
    else:
        out, m, runs = single_target(args, cfg, args.config, T, synth, rank, local_rank, world, dist, forced, args.steps, args.warmup, True)
        if rank == 0 and with_cpu:
            out["cpu_baseline"] = cpu_baseline(m, cfg, runs)
        if args.config == 2 and not args.no_sub_records:
            if world == 1:
                # the other single-GPU configs of BASELINE.json, shorter legs of the same measurement
                sub = {}
                for c in (3, 4):
                    r, _, _ = single_target(args, CONFIGS[c], c, T, synth, rank, local_rank, world, dist, forced, max(1, min(2, args.steps)), 1, False)
                    sub[f"config{c}"] = compact(r)
                out["sub_records"] = sub
            else:
                # the north star's multi-GPU mode: independent targets sharded over the ranks (config 5, strong scaling)
                a5 = argparse.Namespace(steps=1, warmup=0)
                bm = multi_target(a5, CONFIGS[5], T, synth, rank, local_rank, world, dist, forced, False)
                if rank == 0:
                    out["batch_mode"] = {k: bm[k] for k in ("value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "all_decoys_converged")}
                    out["batch_mode"]["workload"] = bm["config"]["workload"]
                    out["batch_mode"]["parallelism"] = bm["config"]["parallelism"]
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
