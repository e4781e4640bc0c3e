"""Host side of the other restraint builders of the reference's fold script (`-r`, /root/reference/folding/folding.py:60-68).

  no-idp  gen_rst        utils_ros.py:6-146     tables built on the device (kernel K2), Context.set_map
  idp     gen_idp_rst    utils_ros.py:196-373   same kernel, pairs flagged in npz['idr'] normalised by their most probable bin
  af2     gen_rst_af2    utils_ros.py:148-194   device kernel k_build_tables_af2, Context.set_map_af2
  gpcr    gen_gpcr_rst   utils_ros.py:484-654   gen_rst on the device + the edits below on the flagged pairs, uploaded with
                                                Context.override_rows (the device recomputes the splines of those rows)

gen_gpcr_rst flattens, on every flagged pair, the five bins that a set of KNOWN structures favours most: it builds a smoothed
histogram of the known structures' realised bins (its `pros` + `get_sample`), turns it into a potential the same way as the
prediction, and linearly interpolates the prediction's potential across the span of that histogram's five lowest-energy bins
(`ling_sumlt`).  This is one-off table preparation from a handful of structures, outside the fold's hot path; it is done here in
numpy, in the reference's dtypes and ORDER of floating-point additions (its outputs go through '%.3f' / '%.5f'), and pinned to
vectors captured from the reference (tests/golden/gen_rst_gpcr_NMR.npz).
"""
import numpy as np

D_EDGES = np.arange(2, 20.5, 0.5)
A_EDGES = np.arange(-np.pi, np.pi, np.pi / 12)
P_EDGES = np.arange(0, np.pi, np.pi / 12)


def realised_bins(known):
    """known: dict dist / omega / theta_asym / phi_asym, each [n][L][L] (6-D geometry of n structures) -> bins jd, jt, jo, jp.
    As in the feedback step: the angle bins are zeroed where the distance bin is, and phi's bins come from THETA on phi's edges
    (utils_ros.py:420, the reference's quirk R1)."""
    d = np.asarray(known["dist"])
    jd = (D_EDGES[None, None, None, :] < d[..., None]).sum(-1)
    jd = np.where(jd >= 37, 0, jd)
    out = [jd]
    for key, edges, src in (("theta_asym", A_EDGES, "theta_asym"), ("omega", A_EDGES, "omega"), ("phi_asym", P_EDGES, "theta_asym")):
        if src not in known:
            break
        j = (edges[None, None, None, :] < np.asarray(known[src])[..., None]).sum(-1)
        out.append(np.where(jd == 0, 0, j))
    return out


def smoothed_histogram(bins, nb):
    """get_sample (utils_ros.py:451-483): per pair, every realised bin k contributes count_k normal densities centred on k, with
    sigma 1.5 / 1.0 / 0.5 by how many of the n structures agree on it; divided by n.  Additions in the reference's order."""
    n = bins.shape[0]
    counts = (bins[..., None] == np.arange(nb)).sum(0)                      # [L][L][nb]
    x = np.arange(nb)
    acc = np.zeros(counts.shape, float)
    for k in range(nb):
        c = counts[..., k]
        if not c.any():
            continue
        std = np.where(c < n / 3, 1.5, np.where(c > 2 * n / 3, 0.5, 1.0))[..., None]
        pdf = (1 / (np.sqrt(2 * np.pi * std ** 2)) * np.exp(-((x - k) ** 2) / (2 * std ** 2)))
        for r in range(n):                                                 # the reference adds the density once per structure
            m = c > r
            acc[m] += pdf[m]
    return acc / n


def flatten_favoured_bins(t, cate, x, rows, rg=5):
    """ling_sumlt (utils_ros.py:375-395) on the listed rows: the rg lowest-energy bins of `cate` are replaced by the straight line
    between the neighbours just outside their span."""
    t = t.copy()
    for i in rows:
        order = np.argsort(cate[i])[:rg]
        lo, hi = order.min() - 1, order.max() + 1
        if lo < 0:
            lo += 1
        if hi >= len(x):
            hi -= 1
        t[i][order] = (x[order] - x[hi]) / (x[lo] - x[hi]) * (t[i][lo] - t[i][hi]) + t[i][hi]
    return t


def _wrap(v):      # omega / theta: two pad knots on each side, periodic (utils_ros.py:87)
    return np.concatenate([v[..., -2:], v[..., 1:], v[..., 1:3]], axis=-1)


def _mirror(v):    # phi: mirrored pads (utils_ros.py:130)
    return np.concatenate([np.flip(v[..., 1:3], axis=-1), v[..., 1:], np.flip(v[..., -2:], axis=-1)], axis=-1)


def gpcr_rows(npz, known, params, use_orient=True):
    """-> {channel: (a[n], b[n], y[n, K])}: the table rows gen_gpcr_rst writes for the pairs flagged in npz['idr'] (values as
    printed), for Context.override_rows on top of the gen_rst tables of the same map."""
    idr = np.asarray(npz["idr"]).astype(bool)
    rnd = lambda v, dp: np.array([[float(("%%.%df" % dp) % q) for q in row] for row in v])
    jb = realised_bins(known)
    MEFF, EBASE, EREP, DCUT, ALPHA = params["MEFF"], params["EBASE"], params["EREP"], params["DCUT"], params["ALPHA"]
    astep = np.deg2rad(params["ASTEP"])
    out = {}
    # ---- distance (utils_ros.py:539-551)
    dist = npz["dist"]
    bins = np.array([4.25 + params["DSTEP"] * i for i in range(32)])
    bkgr = np.array((bins / DCUT) ** ALPHA)
    prob = np.sum(dist[:, :, 5:], axis=-1)
    a, b = np.where((prob > 0.05) & idr)
    keep = b > a
    a, b = a[keep], b[keep]

    def potential(p):   # p[n, 37] -> [n, 35]
        attr = -np.log((p[:, 5:] + MEFF) / (p[:, -1][:, None] * bkgr[None, :] + 1e-6)) + EBASE
        rep = np.maximum(attr[:, 0], np.zeros(len(attr)))[:, None] + np.array(EREP)[None, :]
        return np.concatenate([rep, attr], axis=-1)

    x = np.concatenate([params["DREP"], bins])
    cate = smoothed_histogram(jb[0], 37)
    y = flatten_favoured_bins(potential(dist[a, b]), potential(cate[a, b]), x, range(len(a)))
    out["dist"] = (a, b, rnd(y, 3))
    if not use_orient:
        return out
    # ---- omega, theta (utils_ros.py:571-601), phi (:617-627)
    xs = np.linspace(-np.pi - 1.5 * astep, np.pi + 1.5 * astep, 28)
    xp = np.linspace(-1.5 * astep, np.pi + 1.5 * astep, 16)
    for ch, jk, nb, pad, xk, sym, dp in (("omega", jb[2], 25, _wrap, xs, True, 5), ("theta", jb[1], 25, _wrap, xs, False, 3),
                                         ("phi", jb[3], 13, _mirror, xp, False, 3)):
        arr = npz[ch]
        prob = np.sum(arr[:, :, 1:], axis=-1)
        a, b = np.where((prob > 0.05) & idr)
        keep = (b > a) if sym else (b != a)
        a, b = a[keep], b[keep]
        pot = lambda p: pad(-np.log((p + MEFF) / (p[:, -1] + MEFF)[:, None]))
        cate = smoothed_histogram(jk, nb)
        y = flatten_favoured_bins(pot(arr[a, b]), pot(cate[a, b]), xk, range(len(a)))
        out[ch] = (a, b, rnd(y, dp))
    return out
