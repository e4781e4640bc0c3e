"""Kabsch superposition RMSD (numpy).  Test infrastructure: the parity meter replacing bin/TMscore's RMSD
column (evaluate_utils.py:33-100 parses it); SURVEY.md section 4 reproduced summary.txt with it."""
import numpy as np


def kabsch_rmsd(P, Q):
    P = np.asarray(P, np.float64); Q = np.asarray(Q, np.float64)
    P = P - P.mean(0); Q = Q - Q.mean(0)
    H = P.T @ Q
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    S[-1] *= d
    e0 = (P * P).sum() + (Q * Q).sum()
    return float(np.sqrt(max(e0 - 2 * S.sum(), 0.0) / len(P)))
