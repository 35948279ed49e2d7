import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from onepiece_amd import registration as R, synthetic as S
from oracle import oracle as O
s = 4
camt = (S.FX / s, S.FY / s, S.CX / s, S.CY / s, S.W // s, S.H // s, 1000.0)
cam = O.make_camera(*camt); cl = []; dp = []
for i in (0, 3):
    d, c = S.room_render(S.room_pose(i), width=camt[4], height=camt[5], fx=camt[0], fy=camt[1], cx=camt[2], cy=camt[3])
    cl.append(O.load_from_depth(cam, d)); dp.append(d)
tgt, src = cl
nrm = S.image_normals(dp[0], *camt[:4])
for iters in (1, 2, 5):
    got = R.PointToPlane(R.PointCloud(src), R.PointCloud(tgt, nrm), None, R.ICPParameter(iters, 0.05))
    ref = O.icp(src, tgt, nrm, None, iters, 0.05, True)
    print(iters, got.per_iter_inliers, ref["per_iter_inliers"], len(got.correspondence_set_index), len(ref["pairs"]))
    print(" last_T diff", np.abs(got.last_T - ref["last_T"]).max(), " T diff", np.abs(got.T - ref["T"]).max(), got.rmse, ref["rmse"])
    if len(got.correspondence_set_index) == len(ref["pairs"]):
        print(" pairs equal", np.array_equal(got.correspondence_set_index, ref["pairs"]))
    print(got.T); print(ref["T"])
