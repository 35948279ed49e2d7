"""Wide ICP fuzz (one-off confidence run, not part of the suites): random cloud shapes (box, noisy plane, lattice = every query tied,
duplicated targets, clusters, a line = rank-deficient normal equations), sizes from one point to 60 k, thresholds, iteration counts,
both estimators; the HIP path in its reference-order mode (sequential float32 sums, the reference's tie rule, the reference's Kabsch
finish) against the oracle: per-iteration inlier counts, the final correspondence set, RegistrationResult::T must be IDENTICAL, rmse equal to 1e-12 (a double sum in another order).
The opt-in fp64 mode (order-free device sums) runs next to it and its pose difference is reported ("default mode" in the log lines, the name of rounds 4-5).  usage: fuzz_icp_wide.py [seeds=60] [first_seed=0]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import oracle
from onepiece_amd import registration as R
oracle.build()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def cloud(rng, kind, n):
    if kind == "box":
        p = rng.uniform(-1, 1, (n, 3)) * [1.0, 0.6, 0.8]
    elif kind == "plane":
        p = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), 0.02 * rng.standard_normal(n) + 0.3 * np.sin(np.linspace(0, 9, n))], 1)
    elif kind == "lattice":
        m = max(int(round(n ** (1 / 3))), 1)
        g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
        p = g * 0.0625          # exactly representable: exact distance ties everywhere
    elif kind == "duplicates":
        base = rng.uniform(-1, 1, (max(n // 3, 1), 3))
        p = base[rng.integers(0, len(base), n)]
    elif kind == "clusters":
        c = rng.uniform(-1, 1, (5, 3))
        p = c[rng.integers(0, 5, n)] + 0.03 * rng.standard_normal((n, 3))
    else:  # line
        t = rng.uniform(-1, 1, n)
        p = np.stack([t, 0.5 * t, -0.25 * t], 1) + (1e-3 * rng.standard_normal((n, 3)) if rng.random() < 0.5 else 0)
    return np.ascontiguousarray(p, np.float32)


bad = 0
worst_default = 0.0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(91000 + seed)
    kind = str(rng.choice(["box", "plane", "lattice", "duplicates", "clusters", "line"]))
    nt = int(rng.choice([1, 2, 7, 100, 3000, 20000, 60000]))
    ns = int(rng.choice([1, 3, 50, 2000, 15000, 40000]))
    tgt = cloud(rng, kind, nt)
    nt = len(tgt)
    x = np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.05, 0.05, 3)]).astype(np.float32)
    M = oracle.se3_exp(x).astype(np.float32)
    pick = rng.integers(0, nt, ns)
    src = tgt[pick] @ M[:3, :3].T + M[:3, 3]
    if kind != "lattice" or rng.random() < 0.5:
        src = src + float(rng.choice([0.0, 1e-3, 1e-2])) * rng.standard_normal(src.shape)
    else:
        src = tgt[pick] + np.float32(0.03125) * rng.integers(-1, 2, (ns, 3))  # half-way between lattice points: equidistant targets
    out = rng.random(ns) < 0.1
    src[out] += rng.uniform(-2, 2, (int(out.sum()), 3))
    src = np.ascontiguousarray(src, np.float32)
    nrm = rng.standard_normal((nt, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-12
    if kind == "plane":
        nrm = np.tile(np.array([0.0, 0.0, 1.0]), (nt, 1)) + 0.05 * rng.standard_normal((nt, 3))
    nrm = np.ascontiguousarray(nrm, np.float32)
    plane = bool(rng.random() < 0.6)
    thr = float(rng.choice([0.01, 0.05, 0.2, 1.0]))
    iters = int(rng.choice([0, 1, 5, 30]))
    T0 = oracle.se3_exp((0.02 * rng.standard_normal(6)).astype(np.float32)).astype(np.float32) if rng.random() < 0.5 else np.eye(4, dtype=np.float32)
    ref = oracle.icp(src, tgt, nrm if plane else None, init_T=T0, max_iter=iters, threshold=thr, point_to_plane=plane)
    para = R.ICPParameter(max_iteration=iters, threshold=thr)
    s_pc, t_pc = R.PointCloud(src), R.PointCloud(tgt, nrm if plane else None)
    fn = R.PointToPlane if plane else R.PointToPoint
    try:
        got = fn(s_pc, t_pc, T0, para, sums="reference_f32")
        dflt = fn(s_pc, t_pc, T0, para, sums="fp64")   # the opt-in fp64 reduction (the library's default is the reference-order mode since round 6)
        err = None
    except Exception as e:  # noqa
        got, dflt, err = None, None, str(e)
    tag = "seed %d: %s %s nt %d ns %d thr %.2f iters %d" % (seed, kind, "plane" if plane else "point", nt, ns, thr, iters)
    if ref is None or got is None:
        ok = ref is None and got is None
        print("%s: oracle %s, hip %s -> %s" % (tag, "refuses" if ref is None else "runs", "refuses (%s)" % err if got is None else "runs", "same" if ok else "DIFFERENT"), flush=True)
        bad += not ok
        continue
    same_n = np.array_equal(ref["per_iter_inliers"], got.per_iter_inliers)
    same_pairs = np.array_equal(ref["pairs"], got.correspondence_set_index)
    same_T = np.array_equal(ref["T"].view(np.uint32), got.T.view(np.uint32)) or (np.isnan(ref["T"]).any() and np.isnan(got.T).any())
    # (rmse: the reference adds the squared errors one by one in double, the kernels in a fixed tree of doubles: equal to ~1e-16, not bit for bit)
    same_rmse = abs(ref["rmse"] - got.rmse) <= 1e-12 * abs(ref["rmse"]) or (np.isnan(ref["rmse"]) and np.isnan(got.rmse))
    ok = same_n and same_pairs and same_T and same_rmse
    dn = float(np.linalg.norm(dflt.T - ref["T"]) / max(np.linalg.norm(ref["T"]), 1e-30)) if np.isfinite(ref["T"]).all() and np.isfinite(dflt.T).all() else float("nan")
    if np.isfinite(dn): worst_default = max(worst_default, dn)
    print("%s: inliers %d, tied %s -> %s (default mode: T rel diff %.2e)" % (tag, len(ref["pairs"]), got.tie_stats, "identical" if ok else
          "DIFFERENT (counts %s pairs %s T %s rmse %s; T rel %.2e)" % (same_n, same_pairs, same_T, same_rmse, float(np.linalg.norm(got.T - ref["T"]) / max(np.linalg.norm(ref["T"]), 1e-30))), dn), flush=True)
    bad += not ok
print("%d of %d seeds differ; default-mode worst T rel diff %.2e" % (bad, n_seeds, worst_default))
sys.exit(1 if bad else 0)
