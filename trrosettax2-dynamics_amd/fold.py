"""Drop-in for the reference's fold launcher.

`folding_with_pred_npz` keeps the name, signature and file outputs of /root/reference/utils_trX2dy/utils.py:484-505,
but instead of spawning one `python ./folding/folding.py` process per decoy (ThreadPoolExecutor over subprocess.run) it
folds all `repeat` decoys of the distogram as ONE batch on the GPU.  `fold_npz` is the same thing for one output file
and backs the `folding/folding.py` command-line shim.

Differences from the reference, all deliberate:
  D  failed folds raise RuntimeError (FoldError) instead of passing silently (utils.py:498 runs subprocess.run unchecked); the
     decoys of the batch that did fold are written first, the failed ones get no file
  D  start torsions come from an explicit seed (the reference never seeds `random`, utils_ros.py:677)
  D  there is no full-atom model (folding.py:200-268: side chains, ref2015_cart, idealize).  What a backbone sees of that stage
     -- the two restraint re-selections without glycine pairs and the re-weighted score through the FastRelax ramps
     (protocol.relax_runs) -- runs under --fastrelax, which is the DEFAULT as in the reference (arguments.py:24-25;
     run_inference.py:295 never disables it); --no-fastrelax skips it (measured effect: DESIGN.md section 2)
"""
import argparse
import functools
import os
import shlex
import threading

import numpy as np

from . import protocol
from ._lib import Context
from .pdbio import read_fasta, write_pdb

_CTX = {}           # (device, host thread) -> Context: successive calls reuse the stream and buffers; a Context is not
                    # thread-safe, distinct Contexts run concurrently (the NMR and X-ray chains are folded that way)
_SEED = [0x5EED]    # advances per call: distinct decoys across calls unless the caller fixes `seed`
_SEED_LOCK = threading.Lock()
_CTX_LOCK = threading.Lock()
_CTX_OWNER = {}   # (device, thread ident) -> the Thread object that made the cached context


def parse_options(options):
    """the flags of folding/utils_ros/arguments.py:5-25 that can appear in an `options` string or on the CLI"""
    if isinstance(options, str):      # run_inference's loop hands the same string to every one of its hundreds of folds: parsed once
        return argparse.Namespace(**vars(_parse_options_cached(options)))
    return _parse_options(options)


@functools.lru_cache(maxsize=64)
def _parse_options_cached(options):
    return _parse_options(options)


def _parse_options(options):
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("-pd", type=float, dest="pcut", default=0.05)
    ap.add_argument("-m", type=int, dest="mode", default=2, choices=[0, 1, 2, 3])
    ap.add_argument("-r", type=str, dest="rst", default="no-idp", choices=["no-idp", "idp", "gpcr", "af2"])
    ap.add_argument("-w", type=str, dest="wdir", default="/dev/shm")
    ap.add_argument("-n", type=int, dest="steps", default=1000)
    ap.add_argument("--orient", dest="use_orient", action="store_true")
    ap.add_argument("--no-orient", dest="use_orient", action="store_false")
    ap.add_argument("--fastrelax", dest="fastrelax", action="store_true")
    ap.add_argument("--no-fastrelax", dest="fastrelax", action="store_false")
    ap.add_argument("--log", dest="log", default="")
    ap.add_argument("--gpu", dest="gpu", default=-1, type=int)
    ap.add_argument("-KNOWN", type=str, required=False)
    ap.set_defaults(use_orient=True, fastrelax=True)
    args, unknown = ap.parse_known_args(shlex.split(options) if isinstance(options, str) else list(options))
    if args.rst == "af2" and args.use_orient:
        raise RuntimeError("AF2 Not support ")           # utils_ros.py:149-150: gen_rst_af2 refuses --orient
    if args.rst == "gpcr" and not args.KNOWN:
        raise ValueError("-r gpcr needs -KNOWN (npz with the 6-D geometry of the known structures, folding.py:66-67)")
    return args


def relax_lite(args):
    """the backbone-visible part of the full-atom stage follows the reference's flag: on unless --no-fastrelax (arguments.py:24-25)"""
    return bool(args.fastrelax)


def _unquote(p):
    # run_inference.py:51-52 wraps the paths in literal double quotes for the shell
    return p[1:-1] if len(p) >= 2 and p[0] == p[-1] and p[0] in "\"'" else p


SLOTS_PER_LANE = 960  # every decoy in flight up to this many per lane (tools/pool_sweep.py, round 3, L=150: a queue of 5120 decoys on
                      # 2 x 640 / 960 / 1280 / 1920 / 2560 slots -> 1741 / 1803 / 1769 / 1718 / 1646 decoys/s; 192 was round 2's optimum)


def get_context(device=0, lanes=1):
    """The calling thread's context on `device` (created on first use).  Contexts of threads that have ended without
    close_contexts() are closed here, so a long-lived caller with thread churn does not pile up streams and buffers (and a new
    thread that is handed a dead thread's ident never inherits its context)."""
    key = (device, threading.get_ident())
    with _CTX_LOCK:
        alive = {t.ident for t in threading.enumerate()}
        dead = [_CTX.pop(k) for k in list(_CTX) if k[1] not in alive]
        ctx = _CTX.get(key)
        if ctx is None:
            ctx = _CTX[key] = Context(device)
            _CTX_OWNER[key] = threading.current_thread()
        elif _CTX_OWNER.get(key) is not threading.current_thread():   # ident reused by a new thread before the sweep saw the old one gone
            dead.append(ctx)
            ctx = _CTX[key] = Context(device)
            _CTX_OWNER[key] = threading.current_thread()
        for k in [k for k in _CTX_OWNER if k not in _CTX]:
            del _CTX_OWNER[k]
    for c in dead:
        c.close()
    if ctx.lanes != lanes:
        ctx.set_lanes(lanes)
    return ctx


def close_contexts(all_threads=False):
    """Close the calling thread's cached context(s) (all_threads: every cached one -- only when no other thread is folding).
    A context holds a stream, its maps and its batch buffers on the GPU; a worker thread that ends without this leaves them
    allocated until the process ends."""
    me = threading.get_ident()
    with _CTX_LOCK:
        mine = [_CTX.pop(k) for k in list(_CTX) if all_threads or k[1] == me]
        for k in [k for k in _CTX_OWNER if k not in _CTX]:
            del _CTX_OWNER[k]
    for c in mine:
        c.close()


def _on_device(a):
    """a torch tensor living on a GPU (duck-typed: this module does not import torch unless it is handed one)"""
    return hasattr(a, "data_ptr") and bool(getattr(a, "is_cuda", False))


def set_restraints(ctx, npz, seq, args, ang):
    """the `-r` switch of folding/folding.py:60-68 + the idr mask mode 3 needs (folding.py:174)"""
    need_idr = args.rst in ("idp", "gpcr") or args.mode == 3
    if need_idr and "idr" not in npz:
        raise KeyError(f"-r {args.rst} / -m {args.mode} needs an 'idr' pair mask in the npz (folding.py:174, utils_ros.py:198)")
    idr = np.asarray(npz["idr"]) if need_idr else None
    if _on_device(npz["dist"]):
        # In-memory hand-off from the trX2 front-end: the network's softmax outputs are CUDA tensors; the reference
        # writes them to an npz and the fold reads it back (utils_trX2dy/utils.py:783-796, folding.py:56).  Here their device pointers go
        # straight into the table build (trx2_set_map_device): no host copy, no file.
        if args.rst != "no-idp" or need_idr:
            raise ValueError("device-resident distograms are supported for -r no-idp without an idr mask (the reference's default)")
        import torch
        ts = [npz["dist"]] + list(ang)
        for t in ts:
            if not (_on_device(t) and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("device-resident distograms must be contiguous float32 CUDA tensors, all on one device")
            if t.device != ts[0].device or t.device.index != int(ctx.device):
                raise ValueError(f"device-resident distograms must all live on the fold context's device cuda:{int(ctx.device)} "
                                 f"(got {t.device} beside {ts[0].device})")
        # Every stream of the device, not the calling thread's current one: run_single folds the second chain on a worker thread whose
        # current stream is the default stream, and the producer may have written the tensors on a side stream.
        torch.cuda.synchronize(ts[0].device)
        ptrs = [int(t.data_ptr()) for t in ts] + [0] * (4 - len(ts))
        ctx.set_map_device(int(ts[0].shape[0]), *ptrs, seq=seq, pcut=args.pcut)
        return
    if args.rst == "af2":
        ctx.set_map_af2(npz["dist"], npz["bins"], seq=seq, pcut=args.pcut)
    elif args.rst == "idp":
        ctx.set_map(npz["dist"], *ang, seq=seq, idr=idr, kind="idp", pcut=args.pcut)
    else:
        ctx.set_map(npz["dist"], *ang, seq=seq, idr=idr, pcut=args.pcut)
        if args.rst == "gpcr":       # gen_rst tables + the edits of the flagged pairs from the known structures
            from . import restraints
            from ._lib import DEFAULT_PARAMS as P
            prm = dict(MEFF=P["meff"], EBASE=P["ebase"], EREP=list(P["erep"]), DREP=list(P["drep"]), DCUT=P["dcut"], ALPHA=P["alpha"],
                       DSTEP=P["dstep"], ASTEP=P["astep_deg"])
            for ch, (a, b, y) in restraints.gpcr_rows(npz, np.load(args.KNOWN), prm, use_orient=args.use_orient).items():
                ctx.override_rows(ch, a, b, y)


class FoldError(RuntimeError):
    """some decoys of a batch failed; `.result` holds the whole batch, `.bad` the failed indices (the good decoys' files were written)"""

    def __init__(self, msg, result, bad):
        super().__init__(msg)
        self.result, self.bad = result, bad


def _failed(r):
    return np.nonzero((r["status"] != 0) | ~np.isfinite(r["xyz"]).all(axis=(1, 2, 3)))[0]


def _write_decoys(r, seq, base_out, names, seed, verbose=False):
    """Writes the PDB of every decoy that folded; THEN raises FoldError if any did not (never a partial PDB for a
    failed decoy -- but one diverged decoy no longer discards the other B - 1)."""
    bad = set(_failed(r).tolist())
    for k, name in enumerate(names):
        if k in bad:
            continue
        write_pdb(os.path.join(base_out, name), seq, r["xyz"][k], remarks=[f"trx2fold decoy {k} seed {seed} evals {int(r['n_evals'][k])}"])
        if verbose:
            print(f"Folded: {os.path.join(base_out, name)}")
    if bad:
        b = sorted(bad)
        raise FoldError(f"fold failed for decoys {b} (status {r['status'][b].tolist()}); the other {len(names) - len(b)} were written", r, b)


def fold_arrays(npz, seq, n_decoys, options="", device=0, seed=None, decoy0=0, lanes=2, allow_partial=False):
    """fold n_decoys of one distogram -> dict(xyz[B,L,5,3], status, e_terms, ...); raises on any failed decoy unless
    allow_partial (the callers that write files pass it and raise AFTER the good decoys are on disk: _write_decoys).
    lanes=2 (default): 32 or more decoys are folded as two halves on two streams (+24..32 % decoys/s, include/trx2fold.h);
    callers that already fold several chains concurrently (pipeline.run_single with two models) pass lanes=1."""
    args = parse_options(options)
    L = len(seq)
    if npz["dist"].shape[0] != L:
        raise ValueError(f"sequence length {L} does not match the distogram {npz['dist'].shape}")
    ctx = get_context(device, lanes)
    ang = [npz[k] for k in ("omega", "theta", "phi")] if args.use_orient else []
    set_restraints(ctx, npz, seq, args, ang)
    if seed is None:
        with _SEED_LOCK:  # chains are folded from concurrent host threads
            seed = _SEED[0]
            _SEED[0] += 1
    # decoy slots: every decoy in flight up to SLOTS_PER_LANE per lane, beyond that a queue that refills them on the device (a
    # launch over more slots takes proportionally longer: profiles/README.md, "How many slots")
    per_lane = (n_decoys + ctx.lanes - 1) // ctx.lanes if n_decoys >= 32 else n_decoys
    ctx.set_pool(SLOTS_PER_LANE if per_lane > SLOTS_PER_LANE else 0)
    r = ctx.fold_batch(n_decoys, protocol.build_runs(L, args.mode, fastrelax=relax_lite(args)), seed=seed, decoy0=decoy0)
    r["seed"] = seed
    bad = _failed(r)
    if len(bad) and not allow_partial:
        raise FoldError(f"fold failed for decoys {bad.tolist()} (status {r['status'][bad].tolist()})", r, bad.tolist())
    return r


def folding_with_pred_npz(base_npz, base_fasta, base_out, out_name, options="-m 2 -r no-idp --orient", repeat=0,
                          start_id=0, device=0, seed=None):
    """Writes {base_out}/{out_name}{i}.pdb for i in [start_id, start_id+repeat), or {out_name}.pdb when repeat == 0."""
    npz = np.load(_unquote(base_npz))
    seq = read_fasta(_unquote(base_fasta))
    os.makedirs(base_out, exist_ok=True)
    n = repeat if repeat else 1
    r = fold_arrays(npz, seq, n, options, device=device, seed=seed, decoy0=start_id, allow_partial=True)
    names = [f"{out_name}{i}.pdb" for i in range(start_id, start_id + repeat)] if repeat else [f"{out_name}.pdb"]
    _write_decoys(r, seq, base_out, names, r["seed"], verbose=True)
    return r


def fold_resident_to_pdb(ctx, seq, base_out, names, options="", seed=None, decoy0=0):
    """fold_arrays_to_pdb for the map already resident in `ctx` (after Context.feedback_step): no upload, no table build"""
    args = parse_options(options)
    if ctx.L != len(seq) or bool(ctx.use_orient) != bool(args.use_orient):
        raise ValueError("the resident map does not match the sequence / the --orient option")
    if seed is None:
        with _SEED_LOCK:
            seed = _SEED[0]
            _SEED[0] += 1
    os.makedirs(base_out, exist_ok=True)
    r = ctx.fold_batch(len(names), protocol.build_runs(len(seq), args.mode, fastrelax=relax_lite(args)), seed=seed, decoy0=decoy0)
    _write_decoys(r, seq, base_out, names, seed)
    return r


def fold_arrays_to_pdb(arrays, seq, base_out, names, options="", device=0, seed=None, decoy0=0, lanes=2):
    """folding_with_pred_npz for distograms already in memory: writes base_out/name for every name"""
    os.makedirs(base_out, exist_ok=True)
    r = fold_arrays(arrays, seq, len(names), options, device=device, seed=seed, decoy0=decoy0, lanes=lanes, allow_partial=True)
    _write_decoys(r, seq, base_out, names, r["seed"])
    return r


def fold_npz(npz_path, fasta_path, out_path, options="", device=0, seed=None):
    """one decoy to one file: what `python folding/folding.py -NPZ .. -FASTA .. -OUT ..` does"""
    npz = np.load(npz_path)
    seq = read_fasta(fasta_path)
    r = fold_arrays(npz, seq, 1, options, device=device, seed=seed)
    d = os.path.dirname(os.path.abspath(out_path))
    os.makedirs(d, exist_ok=True)
    write_pdb(out_path, seq, r["xyz"][0])
    return r
