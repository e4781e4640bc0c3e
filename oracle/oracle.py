"""ctypes front-end of the CPU oracle (oracle/trx2_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product
package.  The library is compiled on demand with -march=native for the host it runs on (the .so name carries
a hash of the CPU flags so a build made in one container is not reused on a different CPU).
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NTERMS = 9
NW = 8
KD, KO, KP = 35, 28, 16


class Run(C.Structure):
    """mirror of trx2_run (include/trx2_model.h)"""
    _fields_ = [("w", C.c_float * NW), ("max_iter", C.c_int), ("sep_lo", C.c_int), ("sep_hi", C.c_int),
                ("precheck", C.c_int), ("skip_to", C.c_int), ("cartesian", C.c_int), ("pair_filter", C.c_int),
                ("tol", C.c_float)]


class FoldStats(C.Structure):
    _fields_ = [("n_evals", C.c_int), ("n_iters", C.c_int), ("runs_done", C.c_int), ("status", C.c_int),
                ("e_final", C.c_double * NTERMS), ("f_final", C.c_double)]


def _cpu_tag():
    try:
        txt = [l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags"))][:2]
    except OSError:
        txt = []
    return hashlib.sha1("".join(txt).encode()).hexdigest()[:10]


def build(force=False):
    out = os.path.join(_HERE, "_build", f"libtrx2oracle-{_cpu_tag()}.so")
    src = os.path.join(_HERE, "trx2_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "trx2_model.h")
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = out + f".{os.getpid()}.tmp"
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-shared", "-o", tmp, src, "-lm"])
        os.replace(tmp, out)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        dp, fp, vp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_void_p
        L.orc_build_tables.restype = vp
        L.orc_build_tables.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_double]
        L.orc_tables_free.argtypes = [vp]
        L.orc_tables_set_seq.argtypes = [vp, C.c_char_p]
        L.orc_build_tables_ex.restype = vp
        L.orc_build_tables_ex.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_double, vp, C.c_int]
        L.orc_build_tables_af2.restype = vp
        L.orc_build_tables_af2.argtypes = [C.c_int, vp, vp, vp, C.c_double]
        L.orc_tables_override_rows.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
        L.orc_tables_filter_pairs.argtypes = [vp, vp, C.c_int]
        L.orc_tables_set_idr.argtypes = [vp, vp]
        L.orc_select_relax.argtypes = [vp]
        L.orc_get_relax_selection.argtypes = [vp, C.c_int, vp]
        L.orc_tables_kd.restype = C.c_int
        L.orc_tables_kd.argtypes = [vp]
        L.orc_place_h.argtypes = [vp, vp, vp, vp]
        L.orc_hbond_term.restype = C.c_double
        L.orc_hbond_term.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_tables_ptr.restype = dp
        L.orc_tables_ptr.argtypes = [vp, C.c_int]
        L.orc_tables_prob.restype = fp
        L.orc_tables_prob.argtypes = [vp, C.c_int]
        L.orc_tables_mask.restype = C.POINTER(C.c_ubyte)
        L.orc_tables_mask.argtypes = [vp, C.c_int]
        L.orc_spline_y2.argtypes = [C.c_int, vp, vp, vp]
        L.orc_spline_eval.argtypes = [C.c_int, vp, vp, vp, C.c_double, dp, dp]
        L.orc_dihedral.restype = C.c_double
        L.orc_dihedral.argtypes = [vp, vp, vp, vp, vp]
        L.orc_angle.restype = C.c_double
        L.orc_angle.argtypes = [vp, vp, vp, vp]
        L.orc_nerf.argtypes = [C.c_int, vp, vp]
        L.orc_eval.restype = C.c_double
        L.orc_eval.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
        L.orc_energy_cart.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
        L.orc_eval_cart.restype = C.c_double
        L.orc_eval_cart.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
        L.orc_extract_internal.argtypes = [C.c_int, vp, vp, vp]
        L.orc_per_residue_terms.argtypes = [vp, vp, vp]
        L.orc_nerf_geom.argtypes = [C.c_int, vp, vp, vp]
        L.orc_uniform.restype = C.c_double
        L.orc_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_random_torsions.argtypes = [C.c_int, C.c_uint64, C.c_uint32, vp]
        L.orc_fold.restype = C.c_int
        L.orc_fold.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
        L.orc_fold_batch.restype = C.c_int
        L.orc_fold_batch.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


DEFAULT_PARAMS = dict(EBASE=-0.5, EREP=[10.0, 3.0, 0.5], DREP=[0.0, 2.0, 3.5], MEFF=1e-4, DCUT=19.5, ALPHA=1.57,
                      DSTEP=0.5, ASTEP=15.0)


def params_vec(p=None):
    p = {**DEFAULT_PARAMS, **(p or {})}
    return np.array([p["EBASE"], *p["EREP"], *p["DREP"], p["MEFF"], p["DCUT"], p["ALPHA"], p["DSTEP"], p["ASTEP"]],
                    dtype=np.float64)


class Tables:
    """orc_build_tables: gen_rst + add_rst selection, dense."""

    def __init__(self, dist, omega=None, theta=None, phi=None, params=None, pcut=0.05, seq=None, idr=None, kind="no-idp", af2_bins=None):
        """kind: "no-idp" gen_rst, "idp" gen_idp_rst (needs idr[L,L]), "af2" gen_rst_af2 (dist[L,L,64] + af2_bins[63])"""
        self.L = int(dist.shape[0])
        arrs = [np.ascontiguousarray(a, dtype=np.float32) if a is not None else None for a in (dist, omega, theta, phi)]
        self._keep = arrs
        self.use_orient = all(a is not None for a in arrs)
        if kind == "af2":
            edges = np.ascontiguousarray(af2_bins, np.float64)
            assert arrs[0].shape == (self.L, self.L, 64) and edges.shape == (63,) and not self.use_orient
            self.h = lib().orc_build_tables_af2(self.L, _p(arrs[0]), _p(edges), _p(params_vec(params)), float(pcut))
        else:
            fl = np.ascontiguousarray(idr, np.uint8) if idr is not None else None
            self.h = lib().orc_build_tables_ex(self.L, *[_p(a) for a in arrs], _p(params_vec(params)), float(pcut), _p(fl),
                                               {"no-idp": 0, "idp": 1}[kind])
            if fl is not None:   # runs with pair_filter = 1 (mode 3, first stage) see the unflagged pairs only
                lib().orc_tables_set_idr(self.h, _p(fl))
        self.kd = lib().orc_tables_kd(self.h)
        if seq is not None:   # prolines donate no backbone hydrogen bond
            if len(seq) != self.L:
                raise ValueError("sequence length does not match the map")
            lib().orc_tables_set_seq(self.h, seq.encode())
        # the relax stage's re-selections (PCUT 0.15 / 0.30, no glycine pairs: folding.py:230-237); they need the sequence
        lib().orc_select_relax(self.h)

    def relax_selection(self, k):
        """selection bits [L,L] of relax round k = 0 / 1 (runs with pair_filter 2 / 3)"""
        out = np.zeros((self.L, self.L), np.uint8)
        lib().orc_get_relax_selection(self.h, int(k), _p(out))
        return out

    def __del__(self):
        try:
            lib().orc_tables_free(self.h)
        except Exception:
            pass

    def _arr(self, which, shape):
        ptr = lib().orc_tables_ptr(self.h, which)
        return np.ctypeslib.as_array(ptr, shape=shape).copy()

    def knots(self):
        return dict(dist=self._arr(8, (self.kd,)), omega=self._arr(9, (KO,)), theta=self._arr(10, (KO,)), phi=self._arr(11, (KP,)))

    def y(self, ch):
        k = {"dist": (0, self.kd), "omega": (2, KO), "theta": (4, KO), "phi": (6, KP)}[ch]
        return self._arr(k[0], (self.L, self.L, k[1]))

    def y2(self, ch):
        k = {"dist": (1, self.kd), "omega": (3, KO), "theta": (5, KO), "phi": (7, KP)}[ch]
        return self._arr(k[0], (self.L, self.L, k[1]))

    def override_rows(self, ch, a, b, y):
        """replace table rows (gen_gpcr_rst's edits): y[n, K] as printed; second derivatives are recomputed"""
        i = ["dist", "omega", "theta", "phi"].index(ch)
        a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32); y = np.ascontiguousarray(y, np.float64)
        lib().orc_tables_override_rows(self.h, i, len(a), _p(a), _p(b), _p(y))

    def filter_pairs(self, flag, keep):
        """add_idr_rst: keep the selected restraints only where flag[a,b] == keep"""
        lib().orc_tables_filter_pairs(self.h, _p(np.ascontiguousarray(flag, np.uint8)), int(bool(keep)))

    def prob(self, ch):
        i = ["dist", "omega", "theta", "phi"].index(ch)
        return np.ctypeslib.as_array(lib().orc_tables_prob(self.h, i), shape=(self.L, self.L)).copy()

    def mask(self, selected=True):
        return np.ctypeslib.as_array(lib().orc_tables_mask(self.h, 1 if selected else 0), shape=(self.L, self.L)).copy()


def spline_y2(x, y):
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    y2 = np.zeros_like(y)
    lib().orc_spline_y2(len(x), _p(x), _p(y), _p(y2))
    return y2


def spline_eval(x, y, y2, xq):
    e, de = C.c_double(), C.c_double()
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64); y2 = np.ascontiguousarray(y2, np.float64)
    lib().orc_spline_eval(len(x), _p(x), _p(y), _p(y2), float(xq), C.byref(e), C.byref(de))
    return e.value, de.value


def dihedral(p1, p2, p3, p4, grad=False):
    ps = [np.ascontiguousarray(p, np.float64) for p in (p1, p2, p3, p4)]
    g = np.zeros(12) if grad else None
    v = lib().orc_dihedral(*[_p(p) for p in ps], _p(g))
    return (v, g.reshape(4, 3)) if grad else v


def angle(p1, p2, p3, grad=False):
    ps = [np.ascontiguousarray(p, np.float64) for p in (p1, p2, p3)]
    g = np.zeros(9) if grad else None
    v = lib().orc_angle(*[_p(p) for p in ps], _p(g))
    return (v, g.reshape(3, 3)) if grad else v


def nerf(tors):
    tors = np.ascontiguousarray(tors, np.float64)
    L = tors.shape[0]
    xyz = np.zeros((L, 5, 3))
    lib().orc_nerf(L, _p(tors), _p(xyz))
    return xyz


def evaluate(tab, tors, w, sep_lo=1, sep_hi=None, grad=True):
    """returns (total, terms[8], grad[L,3] or None, xyz[L,5,3])"""
    tors = np.ascontiguousarray(tors, np.float64)
    w = np.ascontiguousarray(w, np.float64)
    L = tab.L
    e = np.zeros(NTERMS); g = np.zeros((L, 3)) if grad else None; xyz = np.zeros((L, 5, 3))
    f = lib().orc_eval(tab.h, _p(tors), _p(w), int(sep_lo), int(sep_hi if sep_hi is not None else L), _p(e), _p(g), _p(xyz))
    return f, e, g, xyz


def energy_cart(tab, xyz, w, sep_lo=1, sep_hi=None):
    xyz = np.ascontiguousarray(xyz, np.float64); w = np.ascontiguousarray(w, np.float64)
    e = np.zeros(NTERMS); gx = np.zeros_like(xyz)
    lib().orc_energy_cart(tab.h, _p(xyz), _p(w), int(sep_lo), int(sep_hi if sep_hi is not None else tab.L), _p(e), _p(gx))
    return e, gx


def eval_cart(tab, xyz, w, sep_lo=1, sep_hi=None, grad=True):
    """Cartesian-space evaluation (sf_cart incl. the bonded term): -> (total, terms[8], gx[L,5,3] or None)"""
    xyz = np.ascontiguousarray(xyz, np.float64); w = np.ascontiguousarray(w, np.float64)
    e = np.zeros(NTERMS); gx = np.zeros_like(xyz) if grad else None
    f = lib().orc_eval_cart(tab.h, _p(xyz), _p(w), int(sep_lo), int(sep_hi if sep_hi is not None else tab.L), _p(e), _p(gx))
    return f, e, gx


def per_residue_terms(tab, xyz):
    """raw surrogate backbone terms split per residue on given coordinates -> [L, 5]: omega_bb, rama, hydrogen bonds, bonded, repulsion"""
    xyz = np.ascontiguousarray(xyz, np.float64)
    out = np.zeros((tab.L, 5))
    lib().orc_per_residue_terms(tab.h, _p(xyz), _p(out))
    return out


def extract_internal(xyz):
    """coordinates -> (torsions[L,3], internal geometry[L,12]) such that nerf_geom reproduces them up to a rigid motion"""
    xyz = np.ascontiguousarray(xyz, np.float64); L = xyz.shape[0]
    t = np.zeros((L, 3)); g = np.zeros((L, 12))
    lib().orc_extract_internal(L, _p(xyz), _p(t), _p(g))
    return t, g


def nerf_geom(tors, geom):
    tors = np.ascontiguousarray(tors, np.float64); geom = np.ascontiguousarray(geom, np.float64)
    xyz = np.zeros((tors.shape[0], 5, 3))
    lib().orc_nerf_geom(tors.shape[0], _p(tors), _p(geom), _p(xyz))
    return xyz


def random_torsions(L, seed, decoy):
    t = np.zeros((L, 3))
    lib().orc_random_torsions(L, int(seed), int(decoy), _p(t))
    return t


def make_runs(runs):
    """runs: list of dicts(w=[8], max_iter, sep_lo, sep_hi, precheck, skip_to, cartesian)"""
    arr = (Run * len(runs))()
    for i, r in enumerate(runs):
        for k in range(NW):
            arr[i].w[k] = float(r["w"][k])
        arr[i].max_iter = int(r["max_iter"]); arr[i].sep_lo = int(r["sep_lo"]); arr[i].sep_hi = int(r["sep_hi"])
        arr[i].precheck = int(r.get("precheck", 0)) | (2 if r.get("warm") else 0); arr[i].skip_to = int(r.get("skip_to", 0))   # bit 1: TRX2_RUN_WARM
        arr[i].cartesian = int(r.get("cartesian", 0)); arr[i].pair_filter = int(r.get("pair_filter", 0))
        arr[i].tol = float(r.get("tol", 0.0))
    return arr


def fold(tab, tors0, runs, max_evals=200000):
    """fold one decoy; returns (torsions, xyz, stats dict)"""
    tors = np.ascontiguousarray(tors0, np.float64).copy()
    arr = make_runs(runs)
    st = FoldStats()
    xyz = np.zeros((tab.L, 5, 3))
    lib().orc_fold(tab.h, _p(tors), arr, len(runs), int(max_evals), C.byref(st), _p(xyz))
    return tors, xyz, dict(n_evals=st.n_evals, n_iters=st.n_iters, runs_done=st.runs_done, status=st.status,
                           e_final=np.array(st.e_final[:]), f_final=st.f_final)


def usable_cores():
    """cores this process may actually use: the scheduler affinity mask, capped by a cgroup CPU quota if one is set"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def fold_batch(tab, tors0, runs, max_evals=200000, nthreads=0):
    """fold B decoys, OpenMP over decoys (orc_fold_batch); tors0[B,L,3] -> (torsions[B,L,3], xyz[B,L,5,3], stats list, threads used)"""
    tors = np.ascontiguousarray(tors0, np.float64).copy()
    B = tors.shape[0]
    arr = make_runs(runs)
    st = (FoldStats * B)()
    xyz = np.zeros((B, tab.L, 5, 3))
    used = lib().orc_fold_batch(tab.h, B, _p(tors), arr, len(runs), int(max_evals), st, _p(xyz), int(nthreads))
    stats = [dict(n_evals=s.n_evals, n_iters=s.n_iters, runs_done=s.runs_done, status=s.status, e_final=np.array(s.e_final[:]),
                  f_final=s.f_final) for s in st]
    return tors, xyz, stats, used

