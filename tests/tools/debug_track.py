"""Diagnostic: GPU tracker vs oracle (float sums) vs oracle (double sums) on the multi-scale loop."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from onepiece_amd import odometry as O, integration as I
from oracle import oracle
from helpers import track_levels, rel_err

odo = O.Odometry(I.PinholeCamera("OPEN3D_DATASET"))
for (i, j, scale, term) in [(300, 302, 2, 1), (300, 302, 2, 2), (300, 301, 2, 1), (300, 301, 2, 2), (300, 301, 1, 1), (300, 301, 1, 2), (500, 503, 1, 1), (500, 503, 1, 2)]:
    levels, T_true = track_levels(i, j, holes=True, scale=scale)
    cam = I.PinholeCamera("OPEN3D_DATASET"); cam.width, cam.height = levels[0]["width"], levels[0]["height"]
    odo.SetCamera(cam); odo.SetMultiScale(3); odo.iter_count_per_level = [4, 8, 16]
    got = odo.MultiScaleComputing(levels, None, term, want_log=True)
    ref = oracle.dense_track(levels, (4, 8, 16), term=term)
    oracle.lib().orc_set_accumulate_double(1)
    refd = oracle.dense_track(levels, (4, 8, 16), term=term)
    oracle.lib().orc_set_accumulate_double(0)
    print("case", i, j, scale, term, "iters", got.iterations, ref["iterations"], refd["iterations"])
    print("  T: gpu-vs-float %.2e  gpu-vs-double %.2e  float-vs-double %.2e   |T-true| gpu %.2e float %.2e double %.2e" % (
        rel_err(got.T, ref["T"]), rel_err(got.T, refd["T"]), rel_err(ref["T"], refd["T"]),
        np.abs(got.T - T_true).max(), np.abs(ref["T"] - T_true).max(), np.abs(refd["T"] - T_true).max()))
    n = min(len(got.per_iter_count), len(ref["per_iter_count"]), len(refd["per_iter_count"]))
    print("  count diff gpu-float ", np.abs(got.per_iter_count[:n].astype(int) - ref["per_iter_count"][:n]).tolist())
    print("  count diff gpu-double", np.abs(got.per_iter_count[:n].astype(int) - refd["per_iter_count"][:n]).tolist())
    print("  count diff float-dbl ", np.abs(ref["per_iter_count"][:n].astype(int) - refd["per_iter_count"][:n]).tolist())
    k = 0
    for a, b in zip(got.per_iter_T, refd["per_iter_T"]):
        if k in (0, 3, 7, 15, 16, 23, 24, n - 1):
            print("   it %d  relT gpu-double %.2e  gpu-float %.2e" % (k, rel_err(a, b), rel_err(a, ref["per_iter_T"][k])))
        k += 1
    print("  rmse", got.rmse, ref["rmse"], refd["rmse"], "n", got.n_correspondences, len(ref["pixel_correspondences"]), len(refd["pixel_correspondences"]))
