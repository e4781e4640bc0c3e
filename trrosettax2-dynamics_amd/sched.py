"""Sharding of fold work over ranks (one process per GPU).

The reference has no distributed code: batch mode is a serial `for name in names` loop (run_inference.py:345-348) and
decoys of a target are OS processes on one host (utils.py:501-503).  Targets, the NMR / X-ray chains of a target and the
initial decoys of a chain are independent (run_inference.py:50-57,310-318), so the path shards with NO data-path
collective: every rank folds its own items; torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only to
gather (decoys, seconds, failures) at the end, and a TCPStore counter hands out the next target (DynamicQueue).  The
iteration phase of a chain is sequential and stays on one rank.

Cost model.  A fold call is LATENCY-bound, not work-bound: its wall time is the evaluation count of its
slowest decoy times the length of one (pair, step) launch pair, and a launch pair over n decoys of an L-residue chain costs a
latency floor that grows with L plus a throughput term that grows with n L (restraint visits) and n L^2 (contact scan):
    seconds(L, n) = c0 + c1 L + n (c2 L + c3 L^2)
Rounds 1-3 used n L^2 alone, which the measurements contradict (L=400 x 32 decoys 265 ms against L=150 x 64 152 ms: ratio 1.7
where n L^2 says 3.6; halving a decoy block does not halve its time).  The constants below are a non-negative least-squares
fit to calls timed on one MI355X (tools/fit_cost_model.py -> profiles/r04_cost_model.json); `CostModel.fit` refits them from
any list of (L, n, seconds) samples -- bench.py does so from its own per-item seconds and reports both.
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class CostModel:
    # fit of 30 calls on one MI355X, L = 90 .. 400, n = 1 .. 128, default protocol, all channels (profiles/r04_cost_model.json):
    # median relative error 8.6 %, worst 27 % (what is left is the spread of the slowest decoy's evaluation count)
    c0: float = 0.0301
    c1: float = 3.42e-4
    c2: float = 6.45e-6
    c3: float = 1.87e-8

    def call_seconds(self, L, n):
        """one fold call: n decoys of an L-residue chain, all in flight, default protocol"""
        return self.c0 + self.c1 * L + n * (self.c2 * L + self.c3 * L * L)

    def target_seconds(self, L, init_num=10, chains=2, iterations=300):
        """one target of run_inference.py folded alone: the initial batches of its chains side by side, then `iterations`
        sequential single-decoy folds per chain (the chains side by side)"""
        return self.call_seconds(L, init_num * chains) + iterations * self.call_seconds(L, 1)

    @staticmethod
    def fit(samples):
        """samples: iterable of (L, n, seconds) -> CostModel by non-negative least squares (relative errors: every sample weighs
        the same whatever its length)"""
        import numpy as np
        from scipy.optimize import nnls
        s = [(float(L), float(n), float(t)) for L, n, t in samples if t > 0]
        if len(s) < 4:
            raise ValueError("need at least four (L, n, seconds) samples")
        A = np.array([[1.0, L, n * L, n * L * L] for L, n, _ in s])
        t = np.array([q[2] for q in s])
        x, _ = nnls(A / t[:, None], np.ones(len(s)))
        return CostModel(*[float(v) for v in x])

    def rel_errors(self, samples):
        return [(self.call_seconds(L, n) - t) / t for L, n, t in samples]


MODEL = CostModel()


@dataclass(frozen=True)
class Item:
    target: str
    chain: str        # "NMR" / "Xray" / "all" (a whole target of batch mode)
    L: int
    decoy0: int       # first initial decoy of this block
    n: int            # number of initial decoys in the block
    iterations: int = 0   # batch mode: expected sequential single-decoy folds per chain after the initial batch

    @property
    def cost(self):
        """modelled seconds of this item folded alone on one GPU (MODEL)"""
        return MODEL.call_seconds(self.L, self.n) + self.iterations * MODEL.call_seconds(self.L, 1)


def make_items(targets, chains=("NMR", "Xray"), init_num=10):
    """targets: iterable of (name, L) -> one item per (target, chain) holding all its initial decoys"""
    return [Item(name, c, int(L), 0, int(init_num)) for name, L in targets for c in chains]


def _order(its):
    return sorted(its, key=lambda i: (-i.cost, i.target, i.chain, i.decoy0))


def _plan(items, world):
    loads, out = [0.0] * world, [[] for _ in range(world)]
    for it in _order(items):
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(it)
        loads[r] += it.cost
    return out, loads


def lpt_assign(items, world, imbalance=0.20, min_block=8):
    """Longest-processing-time-first assignment by modelled seconds.  An item is split into two decoy blocks (legal: initial
    decoys are independent) while ranks would idle (fewer items than ranks) or while the split LOWERS the modelled makespan by
    more than 2 % -- with a latency-bound cost a half block costs well over half the time, so blocks of a few dozen decoys are
    not worth splitting and the makespan's floor is the longest item (predict_makespan reports it).  `imbalance` is kept for
    callers of the old signature: no split is tried while the heaviest rank is within (1 + imbalance) of the mean.
    Deterministic: every rank computes the same plan from the same list.  -> list (per rank) of lists of Item."""
    items = _order(items)
    for _ in range(64):
        out, loads = _plan(items, world)
        mean = sum(loads) / world
        idle = len(items) < world
        if not idle and (mean == 0 or max(loads) <= (1 + imbalance) * mean):
            break
        heavy = max(range(world), key=lambda k: (loads[k], -k))
        pool = items if idle else out[heavy]
        big = max((it for it in pool if it.n >= 2 * min_block and it.iterations == 0), key=lambda i: (i.cost, i.n), default=None)
        if big is None:
            break
        h = big.n // 2
        trial = [it for it in items if it is not big] + [Item(big.target, big.chain, big.L, big.decoy0, h),
                                                          Item(big.target, big.chain, big.L, big.decoy0 + h, big.n - h)]
        if not idle and max(_plan(trial, world)[1]) > 0.98 * max(loads):
            break
        items = _order(trial)
    return _plan(items, world)[0]


def predict_makespan(seconds, world):
    """List scheduling (longest first, next item to the rank that frees up first -- what DynamicQueue does) of items whose seconds
    are known -> (makespan, per-rank loads).  The makespan can never be below the longest item."""
    loads = [0.0] * world
    for t in sorted(seconds, reverse=True):
        loads[loads.index(min(loads))] += t
    return max(loads) if loads else 0.0, loads


def shard_range(n, rank, world):
    """contiguous split of n units: -> (start, count)"""
    base, extra = divmod(n, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


class DynamicQueue:
    """A shared counter that hands out item indices in order: every rank calls next() when it is free and gets the next index
    nobody has taken (torch.distributed TCPStore.add is atomic; no collective, no data on the wire but the integer).  With
    iteration counts that nobody can know beforehand (a chain stops when its maps converge: 47-80 iterations on the reference's
    example, 300 = Nmax on others) a static longest-first plan straggles; a queue ordered longest-first by the model does not.
    store = None: a process-local counter (single rank)."""

    def __init__(self, n_items, store=None, key="trx2_next_item"):
        import threading
        self.n, self.store, self.key, self._local, self._lock = int(n_items), store, key, 0, threading.Lock()

    def next(self):
        if self.store is None:
            with self._lock:        # a rank's worker threads pull from it too
                i = self._local
                self._local += 1
        else:
            i = int(self.store.add(self.key, 1)) - 1
        return i if i < self.n else None


def queue_store(dist, connect_timeout_s=120, group=None):
    """The store behind DynamicQueue for an initialised process group: a TCPStore of its own (the default group's store is private API), rank 0
    serving on a port IT FINDS FREE and broadcasts over the group (MASTER_PORT + 1 is reserved by nobody); a rank that cannot
    connect fails within connect_timeout_s instead of waiting out the job.  One store per process: a second job reuses it (its queue has a
    key of its own).  None for a single rank.  group: the group the port is broadcast on -- run_batch passes its gloo summary group, so that the
    job's first collective is not an RCCL one (object broadcasts on an NCCL group stage through the GPU)."""
    import datetime
    import os
    import socket
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    if "store" in _STORES:
        return _STORES["store"]
    from torch.distributed import TCPStore
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = [0]
    if dist.get_rank() == 0:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("", 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast_object_list(port, src=0, group=group)
    _STORES["store"] = TCPStore(host, int(port[0]), dist.get_world_size(), is_master=dist.get_rank() == 0,
                                timeout=datetime.timedelta(seconds=connect_timeout_s), wait_for_workers=False)
    return _STORES["store"]


_STORES = {}
GATHER_TIMEOUT_H = 24  # ranks finish hours apart on long name lists; the summary gather must outwait the slowest one


def summary_group(dist):
    """A gloo group with a long timeout for the end-of-job summary gather.  Ranks reach that gather as they finish, possibly
    hours apart; on the default NCCL group (10 min timeout) the watchdog would abort the early ranks and the summary and exit
    code would be lost although every PDB file is written.  Collective: every rank must call it, once."""
    import datetime
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    return dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=GATHER_TIMEOUT_H))


def gather_stats(local, dist=None, group=None):
    """local: dict(decoys, seconds, failed).  -> list of per-rank dicts on every rank (all_gather_object on `group`,
    a summary_group() for jobs whose ranks finish far apart; the default group otherwise)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [dict(local)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, dict(local), group=group)
    return out


def run_sharded(items, fold_fn, rank, world, dist=None):
    """Every rank folds its share with fold_fn(item) -> dict(decoys, seconds, failed); returns the job summary:
    total decoys, max-over-ranks seconds (the job's wall time), failures, and the per-rank breakdown."""
    mine = lpt_assign(items, world)[rank]
    tot = dict(decoys=0, seconds=0.0, failed=0)
    for it in mine:
        r = fold_fn(it)
        for k in tot:
            tot[k] += r[k]
    per = gather_stats(tot, dist)
    return dict(decoys=sum(p["decoys"] for p in per), seconds=max(p["seconds"] for p in per),
                failed=sum(p["failed"] for p in per), per_rank=per)
