"""CPU: the N>1 path -- partitioning and the stats gather, world_size 2 over gloo (no GPU, fold replaced by a stub)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

S = importlib.import_module("trrosettax2-dynamics_amd.sched")


def test_lpt_covers_every_decoy_once_and_its_makespan_is_the_longest_item():
    # BASELINE config 5: 8 targets L in {100..400}, 32 decoys each, 8 ranks
    items = S.make_items([(f"t{L}", L) for L in (100, 140, 180, 220, 260, 300, 350, 400)], chains=("NMR",), init_num=32)
    plan = S.lpt_assign(items, 8)
    seen = {}
    for r, its in enumerate(plan):
        for it in its:
            for d in range(it.decoy0, it.decoy0 + it.n):
                assert (it.target, it.chain, d) not in seen
                seen[(it.target, it.chain, d)] = r
    assert len(seen) == 8 * 32
    # A fold call is latency-bound (sched.CostModel): half a block of 32 decoys costs well over half the time, so the job is about
    # as long as its longest target whatever is split: between the L=400 target's half block and its whole.
    loads = [sum(it.cost for it in its) for its in plan]
    assert S.MODEL.call_seconds(400, 16) <= max(loads) <= S.MODEL.call_seconds(400, 32) * (1 + 1e-9)
    assert S.MODEL.call_seconds(400, 16) > 0.7 * S.MODEL.call_seconds(400, 32)
    assert S.lpt_assign(items, 8) == plan               # deterministic
    # the measured ratio the old n L^2 cost contradicted: L=400 x 32 decoys against L=150 x 64 is 1.7 x (265 / 152 ms), not 3.6 x
    ratio = S.MODEL.call_seconds(400, 32) / S.MODEL.call_seconds(150, 64)
    assert 1.3 < ratio < 2.3, ratio


def test_cost_model_fit_and_makespan_prediction():
    true = S.CostModel(0.01, 7e-4, 1e-5, 2e-8)
    samples = [(L, n, true.call_seconds(L, n)) for L in (100, 150, 260, 400) for n in (1, 16, 32, 64)]
    fit = S.CostModel.fit(samples)
    assert max(abs(e) for e in fit.rel_errors(samples)) < 1e-6
    with pytest.raises(ValueError):
        S.CostModel.fit(samples[:3])
    # list scheduling: never below the longest item, never above sum / world + longest
    secs = [true.call_seconds(L, 32) for L in (100, 140, 180, 220, 260, 300, 350, 400)]
    for world in (1, 2, 4, 8):
        mk, loads = S.predict_makespan(secs, world)
        assert max(secs) <= mk <= sum(secs) / world + max(secs) and len(loads) == world and sum(loads) == pytest.approx(sum(secs))
    assert S.predict_makespan(secs, 8)[0] == pytest.approx(max(secs))      # config 5 on 8 GPUs: the L=400 target is the job
    assert S.predict_makespan(secs, 1)[0] / S.predict_makespan(secs, 8)[0] < 6.0   # ... which caps the speed-up below the 6 x asked for
    q = S.DynamicQueue(3)
    assert [q.next() for _ in range(5)] == [0, 1, 2, None, None]


def test_fewer_items_than_ranks_are_split_into_decoy_blocks():
    plan = S.lpt_assign(S.make_items([("a", 150)], chains=("NMR",), init_num=64), 4)
    assert all(len(p) >= 1 for p in plan) and sorted(it.decoy0 for p in plan for it in p) == [0, 16, 32, 48]
    assert [S.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = S.make_items([("a", 100), ("b", 200), ("c", 300)], init_num=16)
        folded = []

        def fake_fold(it):  # stands in for Context.fold_batch: no GPU in this test
            folded.append((it.target, it.chain, it.decoy0, it.n))
            return dict(decoys=it.n, seconds=1e-6 * it.cost, failed=1 if (it.target == "b" and it.decoy0 == 0 and it.chain == "NMR") else 0)

        res = S.run_sharded(items, fake_fold, rank, world, dist)
        allf = [None] * world
        dist.all_gather_object(allf, folded)
        dist.barrier()
        if rank == 0:
            q.put((res, allf))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_sharded_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res, allf = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["decoys"] == 3 * 2 * 16 and res["failed"] == 1 and len(res["per_rank"]) == 2
    assert res["seconds"] == pytest.approx(max(p["seconds"] for p in res["per_rank"]))     # job time = slowest rank
    units = [(t, c, d) for f in allf for (t, c, d0, n) in f for d in range(d0, d0 + n)]
    assert len(units) == len(set(units)) == 96                                              # disjoint and complete
    assert all(len(f) > 0 for f in allf)


def _batch_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        P = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
        names = ["t%d" % L for L in (40, 90, 60, 120, 75, 50, 45)]
        # seconds a target REALLY takes here: nothing like what its length suggests (iteration counts are unknowable beforehand:
        # a chain stops when its maps converge) -- the longest chain is quick, two short ones are slow
        real = {"t120": 0.05, "t90": 0.05, "t75": 0.05, "t60": 0.05, "t50": 0.9, "t45": 0.05, "t40": 0.9}

        def fake_run_single(name, fasta_file, save_dir, device=0, **kw):  # no GPU here: stands in for pipeline.run_single
            assert os.path.exists(fasta_file) and device == rank and kw["init_num"] == 4
            if name == "t60":
                raise RuntimeError("fold failed for decoys [1]")
            time.sleep(real[name])
            open(os.path.join(save_dir, f"{name}.rank{rank}"), "w").close()
            return 2 * kw["init_num"] + 3

        res = P.run_batch(names, tmp, tmp, rank=rank, world=world, dist=dist, device=rank, run=fake_run_single, init_num=4,
                          mult_two_models=True, targets_in_flight=1)
        dist.barrier()
        # a second job in the same process group: its queue starts from its own zero (the key names the job), on the same store
        res2 = P.run_batch(["t90", "t120", "t45"], tmp, tmp + "/second", rank=rank, world=world, dist=dist, device=rank, run=fake_run_single,
                           init_num=4, mult_two_models=True, targets_in_flight=1)
        dist.barrier()
        q.put((rank, (res, res2)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_batch_mode_pulls_targets_from_a_shared_queue(tmp_path):
    """run_inference.py batch mode over two ranks: every target runs exactly once, handed out by the shared counter
    (sched.DynamicQueue on a TCPStore) as ranks free up; a failing target is reported by every rank's summary and does not stop
    the others; with durations that contradict the length-based order the job still ends near the best possible makespan, where a
    static longest-first plan by length would put both slow targets on one rank."""
    Ls = (40, 90, 60, 120, 75, 50, 45)
    for L in Ls:
        (tmp_path / f"t{L}.fasta").write_text(f">t{L}\n" + "A" * L + "\n")
    (tmp_path / "second").mkdir()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]                                   # every rank holds the same summaries
    res, res2 = got[0]
    assert res2["failed"] == 0 and res2["decoys"] == 3 * 11 and sorted(t for p in res2["per_rank"] for t in p["targets"]) == ["t120", "t45", "t90"]
    assert res["failed"] == 1 and res["errors"] == ["t60: RuntimeError: fold failed for decoys [1]"]
    assert res["decoys"] == 6 * 11 and len(res["per_rank"]) == 2
    assert len([f for f in os.listdir(tmp_path / "second") if ".rank" in f]) == 3
    done = sorted(f for f in os.listdir(tmp_path) if ".rank" in f)
    assert sorted(d.split(".")[0] for d in done) == sorted(f"t{L}" for L in Ls if L != 60)      # each surviving target exactly once
    by_rank = {r: sorted(p["targets"]) for r, p in enumerate(res["per_rank"])}
    assert by_rank[0] and by_rank[1]
    # the two slow targets (t50, t40: 0.9 s each) come late in the length order; whoever is free takes them: they end up on
    # DIFFERENT ranks and the job takes about one of them, not both (a static plan by length: t50 and t40 could share a rank)
    slow = {t: r for r in (0, 1) for t in by_rank[r] if t in ("t50", "t40")}
    assert len(slow) == 2 and slow["t50"] != slow["t40"], by_rank
    assert res["seconds"] < 1.6, res["seconds"]


def test_short_name_list_is_shared_fairly_between_ranks(tmp_path):
    """Four targets, two ranks, sixteen targets in flight allowed per rank: a rank starts at most its fair share (two) at once, so
    the rank that arrives first cannot take the whole list.  Two 'ranks' = two threads of this process on one counter."""
    import threading
    import time
    P = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
    names = [f"s{k}" for k in range(4)]
    for n in names:
        (tmp_path / f"{n}.fasta").write_text(f">{n}\n" + "A" * 50 + "\n")

    class Counter:                       # what sched.DynamicQueue needs of a TCPStore
        def __init__(self):
            self.v, self.lock = {}, threading.Lock()

        def add(self, key, n):
            with self.lock:
                self.v[key] = self.v.get(key, 0) + n
                return self.v[key]

    store, took = Counter(), {0: [], 1: []}

    def fake(name, fasta_file, save_dir, device=0, **kw):
        took[device].append(name)
        time.sleep(0.4)
        return 1

    def rank(r, delay):
        time.sleep(delay)                # rank 1 arrives late
        P._BATCH_CALLS = 0               # both 'ranks' live in one process: same call count, hence the same queue key
        P.run_batch(names, str(tmp_path), str(tmp_path), rank=r, world=2, dist=None, device=r, run=fake, store=store, targets_in_flight=16,
                    init_num=1, mult_two_models=True)

    th = [threading.Thread(target=rank, args=(r, 0.15 * r)) for r in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert sorted(took[0] + took[1]) == names and len(took[0]) == 2 and len(took[1]) == 2, took


def test_batch_mode_without_a_store_splits_statically(tmp_path):
    """world > 1 but no process group and no store (a library caller): the queue's counter would be process-local and every rank
    would fold every target.  run_batch then falls back to the deterministic longest-first static split: the ranks' shares are disjoint and
    complete.  (Two `ranks` simulated in one process: no collective is involved on this path.)"""
    P = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
    names = ["t%d" % L for L in (40, 90, 60, 120, 75, 50, 45)]
    for nm in names:
        with open(tmp_path / f"{nm}.fasta", "w") as f:
            f.write(f">{nm}\n" + "A" * int(nm[1:]) + "\n")
    took = {0: [], 1: []}

    def fake_run_single(name, fasta_file, save_dir, device=0, **kw):
        took[device].append(name)
        return 5

    out = [P.run_batch(names, str(tmp_path), str(tmp_path), rank=r, world=2, dist=None, store=None, device=r, run=fake_run_single, init_num=2,
                       targets_in_flight=1) for r in range(2)]
    assert sorted(took[0] + took[1]) == sorted(names) and not set(took[0]) & set(took[1]), took
    assert took[0] and took[1] and all(o["failed"] == 0 for o in out)
    assert out[0]["decoys"] == 5 * len(took[0]) and out[1]["decoys"] == 5 * len(took[1])


def test_static_split_never_cuts_a_target_in_two(tmp_path):
    """the static-split fallback used sched.lpt_assign's default, which cuts a target into two decoy blocks when there are fewer targets
    than ranks or no iterations (Nmax = 0) -- run_single folds a whole target, so both ranks folded the same one and raced on its files.  One target,
    two ranks, Nmax = 0: exactly one rank runs it; and with several workers allowed a rank's private share is not divided by the world again."""
    P = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
    with open(tmp_path / "t0.fasta", "w") as f:
        f.write(">t0\n" + "A" * 80 + "\n")
    took = {0: [], 1: []}

    def fake_run_single(name, fasta_file, save_dir, device=0, **kw):
        took[device].append(name)
        return 20

    out = [P.run_batch(["t0"], str(tmp_path), str(tmp_path), rank=r, world=2, dist=None, store=None, device=r, run=fake_run_single, init_num=10, Nmax=0)
           for r in range(2)]
    assert sorted(took[0] + took[1]) == ["t0"], took
    assert sum(o["decoys"] for o in out) == 20 and all(o["failed"] == 0 for o in out)
    # six targets, two ranks, four workers allowed: every target once, each rank its own three
    names = [f"u{k}" for k in range(6)]
    for nm in names:
        with open(tmp_path / f"{nm}.fasta", "w") as f:
            f.write(f">{nm}\n" + "A" * 60 + "\n")
    took = {0: [], 1: []}
    out = [P.run_batch(names, str(tmp_path), str(tmp_path), rank=r, world=2, dist=None, store=None, device=r, run=fake_run_single, init_num=10, Nmax=0,
                       targets_in_flight=4) for r in range(2)]
    assert sorted(took[0] + took[1]) == names and len(took[0]) == 3 and len(took[1]) == 3, took
