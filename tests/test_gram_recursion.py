"""The algebra behind the step kernel's direction (kernel_step.h: gram_recursion): the L-BFGS two-loop recursion written on
the Gram scalars s_i.y_j, y_i.y_j, s_i.g, y_i.g gives the same direction and the same g.d as the recursion on the vectors, for a
full and for a partly filled history.  (The kernel itself is pinned by the GPU tracking tests; this pins the formulas it uses.)"""
import numpy as np
import pytest


def two_loop(S, Y, g, gamma):
    """pairs in age order, 0 = newest (oracle/trx2_oracle.c: lbfgs_direction)"""
    n = len(S)
    q = g.copy()
    al = np.zeros(n)
    for k in range(n):
        al[k] = (S[k] @ q) / (S[k] @ Y[k])
        q -= al[k] * Y[k]
    r = gamma * q
    for k in range(n - 1, -1, -1):
        be = (Y[k] @ r) / (S[k] @ Y[k])
        r += (al[k] - be) * S[k]
    return -r


def gram_direction(S, Y, g, gamma, m=8):
    n = len(S)
    SY = np.zeros((m, m)); YY = np.zeros((m, m)); sg = np.zeros(m); yg = np.zeros(m)
    SY[:n, :n] = S @ Y.T; YY[:n, :n] = Y @ Y.T; sg[:n] = S @ g; yg[:n] = Y @ g
    al = np.zeros(m); c = np.zeros(m); t = np.zeros(m); rho = np.zeros(m)
    for k in range(m):
        rho[k] = 1.0 / SY[k, k] if k < n else 0.0
        a = sg[k] - sum(al[j] * SY[k, j] for j in range(k))
        al[k] = rho[k] * a if k < n else 0.0
    for k in range(m):
        t[k] = yg[k] - sum(al[j] * YY[k, j] for j in range(m))
    for k in range(m - 1, -1, -1):
        yr = gamma * t[k] + sum(c[j] * SY[j, k] for j in range(k + 1, m))
        c[k] = al[k] - rho[k] * yr if k < n else 0.0
    r = gamma * g
    for k in range(n):
        r = r - gamma * al[k] * Y[k] + c[k] * S[k]
    g_r = gamma * (g @ g) + sum(c[k] * sg[k] - gamma * al[k] * yg[k] for k in range(m))
    return -r, -g_r


@pytest.mark.parametrize("n", [1, 3, 8])
def test_gram_form_equals_two_loop(n):
    rng = np.random.default_rng(n)
    dim = 450
    A = rng.normal(size=(dim, dim)); H = A @ A.T / dim + np.eye(dim)       # a positive definite "Hessian": s.y > 0
    S = rng.normal(size=(n, dim)) * 0.01; Y = S @ H
    g = rng.normal(size=dim)
    gamma = (S[0] @ Y[0]) / (Y[0] @ Y[0])
    d1 = two_loop(S, Y, g, gamma)
    d2, gd = gram_direction(S, Y, g, gamma)
    assert np.allclose(d1, d2, rtol=1e-9, atol=1e-12)
    assert abs(gd - g @ d1) <= 1e-9 * abs(g @ d1)
    assert g @ d1 < 0
