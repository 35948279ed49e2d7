"""Bounded experiment of the round-5 review (item 3b): is there a GUARD BAND in sigma_i / sigma_max of J^T J that separates the ICP iterations whose step differs between
float32-sequential sums (the reference's, ICP.cpp:121-136) and double sums from those whose step does not?

Host only (numpy + scipy's exact kd-tree; the oracle supplies frames, normals and the float-sum pose chain).  For every iteration of the bench's pairs: the eigenvalues of
J^T J from fp64 sums and from sequential float32 sums (np.cumsum in float32 = the reference's running sum), JacobiSVD's rank decision (lambda > 6 eps_f lambda_max) for both,
and |x_f32 - x_f64| of the two steps.  Prints one line per iteration and the verdict.

    python tests/tools/icp_sigma_probe.py [pairs=4] [iters=10]
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # (a tool of the test side: the product never imports it)
from onepiece_amd import synthetic as S

EPS_F = float(np.finfo(np.float32).eps)
THR = 6.0 * EPS_F


def solve(JTJ, JTr):
    w, V = np.linalg.eigh(0.5 * (JTJ + JTJ.T))
    keep = np.abs(w) > THR * np.abs(w).max()
    y = np.where(keep, (V.T @ (-JTr)) / np.where(keep, w, 1.0), 0.0)
    return (V @ y).astype(np.float32), w, keep


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cam = O.make_camera()
    rows_out = []
    for k in range(n_pairs):
        d0, _c0, _p0 = S.room_frame(2 * k)
        d1, _c1, _p1 = S.room_frame(2 * k + 1)
        tgt = O.load_from_depth(cam, d0); src = O.load_from_depth(cam, d1)
        nrm = O.estimate_normals(tgt, 0.1, 30)
        ref = O.icp(src, tgt, nrm, None, iters, 0.01, True)       # float32 sums: the reference's chain of poses
        tree = cKDTree(tgt.astype(np.float64))
        T = np.eye(4, dtype=np.float32)
        for it in range(iters):
            sp = (src @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
            dist, idx = tree.query(sp.astype(np.float64), k=1)
            inl = dist * dist < 0.01 * 0.01
            s, t, n = sp[inl], tgt[idx[inl]], nrm[idx[inl]]
            J = np.concatenate([n, np.cross(s, n)], 1).astype(np.float32)                    # row = [n ; s x n]
            r = (np.einsum("ij,ij->i", n.astype(np.float64), s.astype(np.float64)) - np.einsum("ij,ij->i", n.astype(np.float64), t.astype(np.float64)))
            JJ64 = J.astype(np.float64).T @ J.astype(np.float64)
            Jr64 = J.astype(np.float64).T @ r
            outer = (J[:, :, None] * J[:, None, :]).reshape(len(J), 36)                      # float32 products, as the reference forms row * row^T
            JJ32 = np.cumsum(outer, axis=0, dtype=np.float32)[-1].reshape(6, 6).astype(np.float64)
            Jr32 = np.cumsum((J * r[:, None].astype(np.float32)), axis=0, dtype=np.float32)[-1].astype(np.float64)
            x64, w64, k64 = solve(JJ64, Jr64)
            x32, w32, k32 = solve(JJ32, Jr32)
            ratio = np.sort(np.abs(w64) / np.abs(w64).max())
            ratio32 = np.sort(np.abs(w32) / np.abs(w32).max())
            dx = float(np.linalg.norm(x32.astype(np.float64) - x64.astype(np.float64)))
            nearest = float(np.min(np.abs(np.log10(ratio / THR))))                         # decades between the threshold and the closest eigenvalue ratio (fp64 sums)
            rows_out.append((k, it, int(inl.sum()), ratio, ratio32, int(k64.sum()), int(k32.sum()), dx, nearest))
            print("pair %d it %2d inl %6d  lambda/lmax(fp64) %s  rank64 %d rank32 %d  |dx| %.3e  |x| %.3e  nearest-to-threshold %.2f decades  (f32: %s)" % (
                k, it, inl.sum(), " ".join("%.1e" % v for v in ratio[:3]), k64.sum(), k32.sum(), dx, float(np.linalg.norm(x64)), nearest, " ".join("%.1e" % v for v in ratio32[:3])), flush=True)
            T = ref["per_iter_T"][it]                                                          # follow the reference's chain
    differ = [r for r in rows_out if r[7] > 1e-5]
    same = [r for r in rows_out if r[7] <= 1e-5]
    print("threshold 6 eps_f = %.3e" % THR)
    if differ:
        print("iterations whose step differs (|dx| > 1e-5): %d, nearest-to-threshold decades: min %.2f max %.2f" % (len(differ), min(r[8] for r in differ), max(r[8] for r in differ)))
    if same:
        print("iterations whose step agrees: %d, nearest-to-threshold decades: min %.2f max %.2f" % (len(same), min(r[8] for r in same), max(r[8] for r in same)))


if __name__ == "__main__":
    main()
