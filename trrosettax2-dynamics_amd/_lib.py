"""ctypes binding of libtrx2fold.so (C ABI: include/trx2fold.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is usable, every call raises.  The oracle
under oracle/ is test infrastructure and is never imported from here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRX2FOLD_LIB selects an alternative build of the SAME library (A/B timing of kernel variants); never a CPU path
LIB_PATH = os.environ.get("TRX2FOLD_LIB") or os.path.join(_HERE, "libtrx2fold.so")
NTERMS, NW = 9, 8
ABI_VERSION = 2      # include/trx2fold.h: trx2_abi_version
K = (35, 28, 28, 16)
TERM_NAMES = ("dist", "omega", "theta", "phi", "vdw", "rama", "omega_bb", "cart", "hb")


class Params(C.Structure):
    """trx2_params: folding/data/params.json + -pd (folding/utils_ros/arguments.py:11)"""
    _fields_ = [("ebase", C.c_double), ("erep", C.c_double * 3), ("drep", C.c_double * 3), ("meff", C.c_double),
                ("dcut", C.c_double), ("alpha", C.c_double), ("dstep", C.c_double), ("astep_deg", C.c_double),
                ("pcut", C.c_double)]


class Run(C.Structure):
    """trx2_run (include/trx2_model.h)"""
    _fields_ = [("w", C.c_float * NW), ("max_iter", C.c_int), ("sep_lo", C.c_int), ("sep_hi", C.c_int),
                ("precheck", C.c_int), ("skip_to", C.c_int), ("cartesian", C.c_int), ("pair_filter", C.c_int), ("tol", C.c_float)]


DEFAULT_PARAMS = dict(ebase=-0.5, erep=(10.0, 3.0, 0.5), drep=(0.0, 2.0, 3.5), meff=1e-4, dcut=19.5, alpha=1.57,
                      dstep=0.5, astep_deg=15.0, pcut=0.05)


def make_params(**kw):
    d = {**DEFAULT_PARAMS, **kw}
    p = Params()
    p.ebase, p.meff, p.dcut, p.alpha, p.dstep, p.astep_deg, p.pcut = (d[k] for k in ("ebase", "meff", "dcut", "alpha", "dstep", "astep_deg", "pcut"))
    for i in range(3):
        p.erep[i] = d["erep"][i]
        p.drep[i] = d["drep"][i]
    return p


def make_runs(runs):
    arr = (Run * len(runs))()
    for i, r in enumerate(runs):
        for k in range(NW):
            arr[i].w[k] = float(r["w"][k])
        arr[i].max_iter, arr[i].sep_lo, arr[i].sep_hi = int(r["max_iter"]), int(r["sep_lo"]), int(r["sep_hi"])
        arr[i].precheck, arr[i].skip_to, arr[i].cartesian = int(r.get("precheck", 0)) | (2 if r.get("warm") else 0), int(r.get("skip_to", 0)), int(r.get("cartesian", 0))
        arr[i].pair_filter = int(r.get("pair_filter", 0))
        arr[i].tol = float(r.get("tol", 0.0))
    return arr


_lib = None


def load():
    """Load the HIP library or raise -- never degrade to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C {_HERE}/csrc` (or __graft_entry__.build()); "
                           "there is no CPU fallback for the fold path")
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.trx2_abi_version.restype = C.c_int
    if L.trx2_abi_version() != ABI_VERSION:
        # ABI 2 made trx2_run.precheck a bit field (bit 1 = warm start): an older library would read every warm run as a guarded one
        # and loop to max_evals without an error
        raise RuntimeError(f"{LIB_PATH} has ABI version {L.trx2_abi_version()}, this package needs {ABI_VERSION}: rebuild it with "
                           f"`make -C {_HERE}/csrc` (or __graft_entry__.build())")
    L.trx2_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.trx2_ctx_destroy.argtypes = [vp]
    L.trx2_ctx_destroy.restype = None
    L.trx2_ctx_set_lanes.argtypes = [vp, C.c_int]
    L.trx2_last_error.argtypes = [vp]
    L.trx2_last_error.restype = C.c_char_p
    L.trx2_set_map.argtypes = [vp, C.c_int, C.c_char_p, vp, vp, vp, vp, C.POINTER(Params)]
    L.trx2_set_map_device.argtypes = [vp, C.c_int, C.c_char_p, vp, vp, vp, vp, C.POINTER(Params)]
    L.trx2_set_map_ex.argtypes = [vp, C.c_int, C.c_char_p, vp, vp, vp, vp, C.POINTER(Params), vp, C.c_int]
    L.trx2_set_map_af2.argtypes = [vp, C.c_int, C.c_char_p, vp, vp, C.POINTER(Params)]
    L.trx2_override_table_rows.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.trx2_get_tables.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.trx2_eval_batch.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.trx2_fold_batch.argtypes = [vp, C.c_int, vp, C.c_int, C.c_uint64, C.c_uint32, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.trx2_feedback_bins.argtypes = [vp, C.c_int, C.c_char_p, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_double, vp, vp, vp, vp]
    L.trx2_feedback_step.argtypes = [vp, C.c_char_p, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_double, vp, C.c_int, C.POINTER(C.c_float)]
    L.trx2_get_map.argtypes = [vp, C.c_int, vp]
    L.trx2_glocon_matrix.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, vp, C.c_double, vp]
    L.trx2_reliability_scores.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.trx2_superpose_matrix.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_double, vp, vp]
    L.trx2_feedback_process.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp]
    L.trx2_time_pair_kernel.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, dp, dp]
    L.trx2_last_fold_stats.argtypes = [vp, dp, ip]
    L.trx2_ctx_set_pool.argtypes = [vp, C.c_int]
    L.trx2_ctx_set_tail_compaction.argtypes = [vp, C.c_int]
    L.trx2_ctx_set_single_decoy_waves.argtypes = [vp, C.c_int]
    L.trx2_set_shared_launches.argtypes = [C.c_int]
    L.trx2_set_shared_launch_halves.argtypes = [C.c_int]
    L.trx2_shared_launch_stats.argtypes = [C.c_int, dp]
    L.trx2_set_shared_launch_profiling.argtypes = [C.c_int]
    L.trx2_last_fold_slot_efficiency.argtypes = [vp, dp]
    L.trx2_ctx_set_profiling.argtypes = [vp, C.c_int]
    L.trx2_last_fold_kernel_times.argtypes = [vp, dp, dp, ip]
    L.trx2_ctx_info.argtypes = [vp, C.c_int, dp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def shared_launch_stats(device=0):
    """-> dict(chunks, folds_per_launch, folds, enqueue_s, wait_s) of the device's launch engines (trx2_shared_launch_stats)"""
    v = (C.c_double * 9)()
    load().trx2_shared_launch_stats(int(device), v)
    return dict(chunks=v[0], folds_per_launch=(v[1] / v[0] if v[0] else 0.0), folds=v[2], enqueue_s=v[3], wait_s=v[4],
                pair_ms_sum=v[5], step_ms_sum=v[6], samples=v[7], sampled_folds=v[8])


def set_shared_launch_profiling(on):
    """bracket one launch pair per chunk of the launch engines by HIP events (trx2_set_shared_launch_profiling); read with shared_launch_stats"""
    load().trx2_set_shared_launch_profiling(int(bool(on)))


def set_shared_launches(mode):
    """trx2_set_shared_launches: 1 = single-decoy folds of all contexts share launch pairs, 0 = every fold launches for itself,
    -1 = the library's rule (TRX2_SHARED_LAUNCH if set; otherwise shared from the fifth live context on: include/trx2fold.h)"""
    if load().trx2_set_shared_launches(int(mode)) != 0:
        raise ValueError("mode must be -1, 0 or 1")


def set_shared_launch_halves(mode):
    """trx2_set_shared_launch_halves: 1 = shared launches in half-evaluation form (one kernel steps one half of an engine's folds beside the
    pair terms of the other half), 0 = a pair launch and a step launch per evaluation, -1 = the library's rule (TRX2_ENGINE_HALF if set;
    otherwise from twelve live contexts on: include/trx2fold.h).  Results are bit-identical either way."""
    if load().trx2_set_shared_launch_halves(int(mode)) != 0:
        raise ValueError("mode must be -1, 0 or 1")


class Context:
    """One GPU stream + one distogram.  Mirrors what one folding.py process holds (folding/folding.py:48-63).
    lanes=2: batches of 32 or more decoys are folded as two halves on two streams (trx2_ctx_set_lanes, include/trx2fold.h)."""

    def __init__(self, device=0, lanes=1, pool=0):
        self._l = load()
        h = C.c_void_p()
        rc = self._l.trx2_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"trx2_ctx_create(device={device}) failed with code {rc}: no usable GPU (no CPU fallback)")
        self._h = h
        self.device = int(device)
        self.L = 0
        self.kd = 35
        self.use_orient = False
        self.lanes = 1
        self.pool = 0
        if lanes != 1:
            self.set_lanes(lanes)
        if pool:
            self.set_pool(pool)

    def set_lanes(self, lanes):
        self._chk(self._l.trx2_ctx_set_lanes(self._h, int(lanes)), "trx2_ctx_set_lanes")
        self.lanes = int(lanes)

    def set_pool(self, slots):
        """fold batches on `slots` decoy slots that refill from the batch's queue on the device (trx2_ctx_set_pool); 0 = one per decoy"""
        self._chk(self._l.trx2_ctx_set_pool(self._h, int(slots)), "trx2_ctx_set_pool")
        self.pool = int(slots)

    def close(self):
        if getattr(self, "_h", None):
            self._l.trx2_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self._l.trx2_last_error(self._h).decode()}")

    def set_map(self, dist, omega=None, theta=None, phi=None, seq=None, idr=None, kind="no-idp", **params):
        """kind: "no-idp" (gen_rst) or "idp" (gen_idp_rst, needs idr[L,L]); an idr mask also enables runs with pair_filter=1 (mode 3)"""
        arrs = [np.ascontiguousarray(a, np.float32) if a is not None else None for a in (dist, omega, theta, phi)]
        L = int(arrs[0].shape[0])
        for a, nb in zip(arrs, (37, 25, 25, 13)):
            if a is not None and a.shape != (L, L, nb):
                raise ValueError(f"expected shape {(L, L, nb)}, got {a.shape}")
        s = (seq or "A" * L).encode()
        if idr is None and kind == "no-idp":
            self._chk(self._l.trx2_set_map(self._h, L, s, *[_p(a) for a in arrs], C.byref(make_params(**params))), "trx2_set_map")
        else:
            fl = np.ascontiguousarray(idr, np.uint8) if idr is not None else None
            if fl is not None and fl.shape != (L, L):
                raise ValueError(f"idr mask must have shape {(L, L)}")
            self._chk(self._l.trx2_set_map_ex(self._h, L, s, *[_p(a) for a in arrs], C.byref(make_params(**params)), _p(fl),
                                              {"no-idp": 0, "idp": 1}[kind]), "trx2_set_map_ex")
        self.L, self.use_orient, self.kd = L, all(a is not None for a in arrs), 35

    def set_map_af2(self, dist64, edges63, seq=None, **params):
        """gen_rst_af2: AlphaFold-style distogram dist64[L,L,64] with its 63 bin edges -> C-alpha restraints (60 knots)"""
        d = np.ascontiguousarray(dist64, np.float32); e = np.ascontiguousarray(edges63, np.float64)
        L = int(d.shape[0])
        if d.shape != (L, L, 64) or e.shape != (63,):
            raise ValueError("need dist64[L,L,64] and 63 bin edges")
        self._chk(self._l.trx2_set_map_af2(self._h, L, (seq or "A" * L).encode(), _p(d), _p(e), C.byref(make_params(**params))), "trx2_set_map_af2")
        self.L, self.use_orient, self.kd = L, False, 60

    def override_rows(self, ch, a, b, y):
        """replace table rows (gen_gpcr_rst's edits): y[n, K] as printed; the device recomputes their splines"""
        i = ("dist", "omega", "theta", "phi").index(ch)
        a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32); y = np.ascontiguousarray(y, np.float64)
        self._chk(self._l.trx2_override_table_rows(self._h, i, len(a), _p(a), _p(b), _p(y)), "trx2_override_table_rows")

    def set_map_device(self, L, dist_ptr, omega_ptr=0, theta_ptr=0, phi_ptr=0, seq=None, **params):
        """Same as set_map with DEVICE pointers (float32, row-major [L][L][37/25/25/13]) -- the in-memory hand-off from the
        trX2 front-end, whose softmax outputs are already on the GPU (utils_trX2dy/utils.py:783-796 writes them to an npz
        instead).  Pass e.g. tensor.data_ptr() of contiguous CUDA tensors; the caller keeps them alive during the call."""
        ptrs = [int(dist_ptr), int(omega_ptr), int(theta_ptr), int(phi_ptr)]
        if not ptrs[0] or (any(ptrs[1:]) and not all(ptrs[1:])):
            raise ValueError("dist is required; omega/theta/phi must be given together or not at all")
        s = (seq or "A" * L).encode()
        self._chk(self._l.trx2_set_map_device(self._h, int(L), s, *[C.c_void_p(p) if p else None for p in ptrs],
                                              C.byref(make_params(**params))), "trx2_set_map_device")
        self.L, self.use_orient = int(L), all(ptrs)

    def get_tables(self, ch):
        i = ("dist", "omega", "theta", "phi").index(ch)
        L = self.L
        Kc = getattr(self, "kd", 35) if i == 0 else K[i]
        yy = np.zeros((L, L, Kc, 2), np.float32); kn = np.zeros(Kc, np.float32); pr = np.zeros((L, L), np.float32)
        gen = np.zeros((L, L), np.uint8); sel = np.zeros((L, L), np.uint8)
        self._chk(self._l.trx2_get_tables(self._h, i, _p(yy), _p(kn), _p(pr), _p(gen), _p(sel)), "trx2_get_tables")
        return dict(y=yy[..., 0], y2=yy[..., 1], knots=kn, prob=pr, gen=gen, sel=sel)

    def eval_batch(self, tors, w, sep_lo=1, sep_hi=None):
        if not self.L:
            raise RuntimeError("trx2_eval_batch: no map set")
        tors = np.ascontiguousarray(tors, np.float32)
        B, L = tors.shape[0], self.L
        if tors.shape != (B, L, 3):
            raise ValueError(f"torsions must have shape (B, {L}, 3), got {tors.shape}")
        w = np.ascontiguousarray(w, np.float32)
        e = np.zeros((B, NTERMS)); f = np.zeros(B); g = np.zeros((B, L, 3), np.float32); xyz = np.zeros((B, L, 5, 3), np.float32)
        self._chk(self._l.trx2_eval_batch(self._h, B, _p(tors), _p(w), int(sep_lo), int(L if sep_hi is None else sep_hi),
                                          _p(e), _p(f), _p(g), _p(xyz)), "trx2_eval_batch")
        return f, e, g, xyz

    def fold_batch(self, B, runs, seed=0, decoy0=0, tors0=None, max_evals=0):
        L = self.L
        t0 = np.ascontiguousarray(tors0, np.float32) if tors0 is not None else None
        if t0 is not None and t0.shape != (B, L, 3):
            raise ValueError(f"start torsions must have shape ({B}, {L}, 3), got {t0.shape}")
        arr = make_runs(runs)
        tors = np.zeros((B, L, 3), np.float32); xyz = np.zeros((B, L, 5, 3), np.float32)
        e = np.zeros((B, NTERMS)); f = np.zeros(B)
        st = np.zeros(B, np.int32); ne = np.zeros(B, np.int32); ni = np.zeros(B, np.int32)
        self._chk(self._l.trx2_fold_batch(self._h, B, arr, len(runs), int(seed), int(decoy0), _p(t0), int(max_evals), _p(tors),
                                          _p(xyz), _p(e), _p(f), _p(st), _p(ne), _p(ni)), "trx2_fold_batch")
        sec, nl, eff = C.c_double(), C.c_int(), C.c_double()
        self._l.trx2_last_fold_stats(self._h, C.byref(sec), C.byref(nl))
        self._l.trx2_last_fold_slot_efficiency(self._h, C.byref(eff))
        return dict(tors=tors, xyz=xyz, e_terms=e, f=f, status=st, n_evals=ne, n_iters=ni, seconds=sec.value, launches=nl.value,
                    slot_efficiency=eff.value)

    # ---- feedback step on the device; mirrors feedback.py, which is bit-identical to the reference
    _D_EDGES = np.arange(2, 20.5, 0.5)
    _A_EDGES = np.arange(-np.pi, np.pi, np.pi / 12)
    _P_EDGES = np.arange(0, np.pi, np.pi / 12)

    def feedback_bins(self, xyz, seq, dmax=20.0):
        """xyz[L,5,3] float32 (as read from the decoy's PDB), seq -> realised bins (jd, jt, jo, jp), the return order of
        get_distribution_from_pdb (utils_trX2dy/utils.py:294-316); int8 [L,L]"""
        xyz = np.ascontiguousarray(xyz, np.float32)
        L = xyz.shape[0]
        if xyz.shape != (L, 5, 3) or len(seq) != L:
            raise ValueError("need xyz[L,5,3] and a sequence of the same length")
        out = [np.empty((L, L), np.int8) for _ in range(4)]
        self._chk(self._l.trx2_feedback_bins(self._h, L, seq.encode(), _p(xyz), _p(self._D_EDGES), len(self._D_EDGES), _p(self._A_EDGES),
                                             len(self._A_EDGES), _p(self._P_EDGES), len(self._P_EDGES), float(dmax), *[_p(o) for o in out]),
                  "trx2_feedback_bins")
        jd, jo, jt, jp = out
        return jd, jt, jo, jp

    @staticmethod
    def gaussian_weights(sigma, truncate=4.0):
        """the kernel scipy.ndimage.gaussian_filter1d builds (_gaussian_kernel1d, order 0); radius 4 for sigma = 1"""
        radius = int(truncate * float(sigma) + 0.5)
        if radius != 4:
            raise ValueError("the device filter is built for radius 4 (sigma = 1, the reference's value)")
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
        return np.ascontiguousarray((phi / phi.sum())[::-1], np.float64)

    def feedback_process(self, unprocessed, bins, norm=True, smooth=True, sigma=1.0):
        """process_distribution_with_pred_distribution (utils_trX2dy/utils.py:379-403) for one channel"""
        a = np.ascontiguousarray(unprocessed, np.float32)
        L, K = a.shape[0], a.shape[2]
        b = np.ascontiguousarray(bins, np.int8)
        if a.shape != (L, L, K) or b.shape != (L, L):
            raise ValueError("need unprocessed[L,L,K] and bins[L,L]")
        out = np.empty_like(a)
        self._chk(self._l.trx2_feedback_process(self._h, L, K, _p(a), _p(b), _p(self.gaussian_weights(sigma)), int(bool(norm)),
                                                int(bool(smooth)), _p(out)), "trx2_feedback_process")
        return out

    def feedback_labels(self, arrays, xyz, seq, sigma=1.0, angle=True):
        """feedback.feedback_labels on the device: arrays (dist[/theta/omega/phi][/tmp]) + the decoy's backbone ->
        dict dist[/theta/omega/phi]/tmp for the next fold"""
        jd, jt, jo, jp = self.feedback_bins(xyz, seq)
        labels = {"dist": self.feedback_process(arrays["dist"], jd, True, True, sigma)}
        if angle:
            labels["theta"] = self.feedback_process(arrays["theta"], jt, True, True, sigma)
            labels["omega"] = self.feedback_process(arrays["omega"], jo, True, True, sigma)
            labels["phi"] = self.feedback_process(arrays["phi"], jp, True, True, sigma)
        base = arrays["tmp"] if "tmp" in arrays else arrays["dist"]
        labels["tmp"] = self.feedback_process(base, jd, norm=False)
        return labels

    def feedback_step(self, xyz, seq, sigma=1.0, angle=True, dmax=20.0):
        """One feedback iteration on the distograms resident in this context (trx2_feedback_step): re-weights them with the
        decoy's realised geometry, updates the cumulative tmp array and rebuilds the restraint tables; the next fold_batch
        uses them.  -> max |tmp_new - tmp_old| (the reference stops below 0.01)"""
        xyz = np.ascontiguousarray(xyz, np.float32)
        if xyz.shape != (self.L, 5, 3) or len(seq) != self.L:
            raise ValueError("need xyz[L,5,3] and a sequence of the length of the current map")
        d = C.c_float()
        self._chk(self._l.trx2_feedback_step(self._h, seq.encode(), _p(xyz), _p(self._D_EDGES), len(self._D_EDGES), _p(self._A_EDGES),
                                             len(self._A_EDGES), _p(self._P_EDGES), len(self._P_EDGES), float(dmax),
                                             _p(self.gaussian_weights(sigma)), int(bool(angle)), C.byref(d)), "trx2_feedback_step")
        return float(d.value)

    def get_map(self, channel):
        """the resident distogram: "dist" / "omega" / "theta" / "phi", or "tmp" after a feedback step"""
        k = ("dist", "omega", "theta", "phi", "tmp").index(channel)
        out = np.empty((self.L, self.L, (37, 25, 25, 13, 37)[k]), np.float32)
        self._chk(self._l.trx2_get_map(self._h, k, _p(out)), "trx2_get_map")
        return out

    def glocon_matrix(self, xyz, seqs, dmax=20.0):
        """GloCon matrix of n structures of one length (get_glocon_matrix, utils_trX2dy/utils.py:543-569): xyz[n,L,5,3] as read
        from the PDB files, seqs = n sequences -> float64 [n,n]"""
        xyz = np.ascontiguousarray(xyz, np.float32)
        n, L = xyz.shape[0], xyz.shape[1]
        if xyz.shape != (n, L, 5, 3) or len(seqs) != n or any(len(s) != L for s in seqs):
            raise ValueError("need xyz[n,L,5,3] and n sequences of length L")
        out = np.empty((n, n), np.float64)
        self._chk(self._l.trx2_glocon_matrix(self._h, n, L, "".join(seqs).encode(), _p(xyz), float(dmax), _p(out)), "trx2_glocon_matrix")
        return out

    def set_tail_compaction(self, mode):
        """0 off, 1 (default) the last <= 64 live decoys of a wider batch move into one decoy group, 2 the same with the pair kernel's
        split kept (bitwise equal to 0): trx2_ctx_set_tail_compaction"""
        self._chk(self._l.trx2_ctx_set_tail_compaction(self._h, int(mode)), "trx2_ctx_set_tail_compaction")

    def set_single_decoy_waves(self, waves):
        """pair-kernel shape of this context's single-decoy folds: 4 (default, shortest evaluation) or 1 (one wave per row: the shape
        for many chains sharing launches) -- trx2_ctx_set_single_decoy_waves"""
        self._chk(self._l.trx2_ctx_set_single_decoy_waves(self._h, int(waves)), "trx2_ctx_set_single_decoy_waves")

    def set_profiling(self, every):
        """every > 0: bracket every `every`-th evaluation of the following folds by HIP events (trx2_ctx_set_profiling)"""
        self._chk(self._l.trx2_ctx_set_profiling(self._h, int(every)), "trx2_ctx_set_profiling")

    def last_fold_kernel_times(self):
        """-> (pair kernel ms, step kernel ms, samples): live averages over the sampled evaluations of the last fold"""
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        self._chk(self._l.trx2_last_fold_kernel_times(self._h, C.byref(a), C.byref(b), C.byref(n)), "trx2_last_fold_kernel_times")
        return a.value, b.value, n.value

    def info(self, key):
        """layout facts for the roofline arithmetic: 0 decoys per wave of the pair kernel, 1 bytes of pair-kernel records summed
        per residue by the step kernel, 2 L-BFGS pairs, 3 L, 4 workgroups per pair-kernel launch, 5 stored pairs the Cartesian role stages in LDS"""
        v = C.c_double()
        if self._l.trx2_ctx_info(self._h, int(key), C.byref(v)) != 0:
            raise RuntimeError(f"trx2_ctx_info: unknown key {key}")
        return v.value

    def reliability_scores(self, xyz):
        """calculate_reliability_score (utils_trX2dy/utils.py:352-372) of n decoys at once: xyz[n,L,5,3] as read from their PDB files
        (pdbio.as_read_from_pdb gives that without the files) -> float scores [n]"""
        xyz = np.ascontiguousarray(xyz, np.float32)
        n, L = xyz.shape[0], xyz.shape[1]
        if xyz.shape != (n, L, 5, 3):
            raise ValueError("need xyz[n,L,5,3]")
        c = np.zeros((n, 2), np.int32)
        self._chk(self._l.trx2_reliability_scores(self._h, n, L, _p(xyz), _p(c)), "trx2_reliability_scores")
        return np.array([int(b) / int(a) if a else 0 for a, b in c])

    def superpose_matrix(self, xa, xb=None, l_norm=None, rmsd=True, tm=True):
        """C-alpha RMSD and TM-score of every pair: xa[n,L,3] against xb[m,L,3] (None: xa against itself) -> (rmsd[n,m], tm[n,m]),
        the same numbers as evaluate.rmsd_common / evaluate.tm_score (trx2_superpose_matrix)"""
        xa = np.ascontiguousarray(xa, np.float32)
        n, L = xa.shape[0], xa.shape[1]
        if xb is not None:
            xb = np.ascontiguousarray(xb, np.float32)
            if xb.shape[1:] != (L, 3):
                raise ValueError("both sets need the same number of aligned residues")
        m = n if xb is None else xb.shape[0]
        if xa.shape != (n, L, 3) or not np.isfinite(xa).all() or (xb is not None and not np.isfinite(xb).all()):
            raise ValueError("need finite C-alpha coordinates [n,L,3]")
        r = np.zeros((n, m)) if rmsd else None
        t = np.zeros((n, m)) if tm else None
        self._chk(self._l.trx2_superpose_matrix(self._h, n, m, L, _p(xa), _p(xb), float(l_norm or 0), _p(r), _p(t)), "trx2_superpose_matrix")
        return r, t

    def time_pair_kernel(self, B, w, sep_lo=1, sep_hi=None, n_rep=50):
        w = np.ascontiguousarray(w, np.float32)
        ms, n = C.c_double(), C.c_double()
        self._chk(self._l.trx2_time_pair_kernel(self._h, B, _p(w), int(sep_lo), int(self.L if sep_hi is None else sep_hi), int(n_rep),
                                                C.byref(ms), C.byref(n)), "trx2_time_pair_kernel")
        return ms.value, n.value
