"""Sharding of fold work over ranks (one process per GPU).

The reference has no distributed code: batch mode is a serial `for name in names` loop (run_inference.py:345-348) and
decoys of a target are OS processes on one host (utils.py:501-503).  Targets, the NMR / X-ray chains of a target and the
initial decoys of a chain are independent (run_inference.py:50-57,310-318), so the path shards with NO data-path
collective: every rank folds its own items; torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only to
gather (decoys, seconds, failures) at the end.  The iteration phase of a chain is sequential and stays on one rank.
"""
from dataclasses import dataclass, field


@dataclass(frozen=True)
class Item:
    target: str
    chain: str        # "NMR" / "Xray"
    L: int
    decoy0: int       # first initial decoy of this block
    n: int            # number of initial decoys in the block

    @property
    def cost(self):   # pair work per evaluation ~ n * L^2
        return self.n * self.L * self.L


def make_items(targets, chains=("NMR", "Xray"), init_num=10):
    """targets: iterable of (name, L) -> one item per (target, chain) holding all its initial decoys"""
    return [Item(name, c, int(L), 0, int(init_num)) for name, L in targets for c in chains]


def lpt_assign(items, world, imbalance=0.20, min_block=8):
    """Longest-processing-time-first assignment; items are split into decoy blocks (legal: initial decoys are
    independent) while there are fewer items than ranks or the heaviest rank exceeds the mean by `imbalance`.
    Deterministic: every rank computes the same plan from the same list.  -> list (per rank) of lists of Item."""
    items = sorted(items, key=lambda it: (-it.cost, it.target, it.chain, it.decoy0))

    def plan(its):
        loads, out = [0] * world, [[] for _ in range(world)]
        for it in sorted(its, key=lambda i: (-i.cost, i.target, i.chain, i.decoy0)):
            r = min(range(world), key=lambda k: (loads[k], k))
            out[r].append(it)
            loads[r] += it.cost
        return out, loads

    for _ in range(64):
        out, loads = plan(items)
        mean = sum(loads) / world
        if len(items) >= world and (mean == 0 or max(loads) <= (1 + imbalance) * mean):
            break
        big = max((it for it in items if it.n >= 2 * min_block), key=lambda i: i.cost, default=None)
        if big is None:
            break
        h = big.n // 2
        items.remove(big)
        items += [Item(big.target, big.chain, big.L, big.decoy0, h), Item(big.target, big.chain, big.L, big.decoy0 + h, big.n - h)]
    return plan(items)[0]


def shard_range(n, rank, world):
    """contiguous split of n units: -> (start, count)"""
    base, extra = divmod(n, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


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
