import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from onepiece_amd import registration as R, synthetic as S, integration as I
from oracle import oracle as O
from helpers import rel_err
cam = O.make_camera()
d0, _, _ = S.room_frame(0); d1, _, _ = S.room_frame(1)
tgt = O.load_from_depth(cam, d0); src = O.load_from_depth(cam, d1)
nrm = O.estimate_normals(tgt, 0.1, 30)
def kab64(pairs):
    s = src[pairs[:, 0]].astype(np.float64); t = tgt[pairs[:, 1]].astype(np.float64)
    ms, mt = s.mean(0), t.mean(0)
    W = (s - ms).T @ (t - mt)
    U, Sg, Vt = np.linalg.svd(W)
    Rm = Vt.T @ U.T
    if np.linalg.det(Rm) < 0:
        Vt[2] *= -1; Rm = Vt.T @ U.T
    T = np.eye(4); T[:3, :3] = Rm; T[:3, 3] = mt - Rm @ ms
    return T
for iters in (1, 10):
    got = R.PointToPlane(R.PointCloud(src), R.PointCloud(tgt, nrm), None, R.ICPParameter(iters, 0.01))
    ref = O.icp(src, tgt, nrm, None, iters, 0.01, True)
    print(iters, "inliers", got.per_iter_inliers[-2:], ref["per_iter_inliers"][-2:], len(got.correspondence_set_index), len(ref["pairs"]))
    print("  last_T rel", rel_err(got.last_T, ref["last_T"]), " T rel", rel_err(got.T, ref["T"]))
    kg, kr = kab64(got.correspondence_set_index), kab64(ref["pairs"])
    print("  vs float64 Kabsch on own pairs: gpu", rel_err(got.T, kg), " oracle", rel_err(ref["T"], kr), " f64(gpu pairs) vs f64(oracle pairs)", rel_err(kg, kr))
    print("  pairs equal", np.array_equal(got.correspondence_set_index, ref["pairs"]))
