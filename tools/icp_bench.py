"""ICP-only timing (used under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from onepiece_amd import registration as R, synthetic as S, integration as I, _lib as L
lib = L.load()
cam = I.PinholeCamera()
d0, _, _ = S.room_frame(0); d1, _, _ = S.room_frame(1)
tgt_pc = R.PointCloud.LoadFromDepth(d0, cam); src = R.PointCloud.LoadFromDepth(d1, cam).points
tgt_pc.EstimateNormals(0.1, 30)
tgt, nrm = tgt_pc.points, tgt_pc.normals
h = C.c_void_p()
t = time.perf_counter()
L.check(lib.op_icp_create(C.c_void_p(tgt.ctypes.data), C.c_void_p(nrm.ctypes.data), len(tgt), 0.01, 0, 0, C.byref(h)))
print("create (grid build) ms", (time.perf_counter() - t) * 1e3)
L.check(lib.op_icp_set_source(h, C.c_void_p(src.ctypes.data), len(src), 0))
res = L.IcpResult(); T0 = np.eye(4, dtype=np.float32).reshape(16)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
L.check(lib.op_icp_run(h, 1, fp(T0), 5, C.byref(res), None, 0, None, None))
for mode in (1, 0):
    t = time.perf_counter()
    L.check(lib.op_icp_run(h, mode, fp(T0), 100, C.byref(res), None, 0, None, None))
    dt = time.perf_counter() - t
    print("mode", mode, "iters/s", 100 / dt, "us/iter", dt / 100 * 1e6, "inliers", res.n_inliers)
lib.op_icp_destroy(h)
