"""ICP loop rate against the density of the target cloud (same source, target subsampled 1:k): how much of an iteration is candidate scanning."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from onepiece_amd import _lib as L, registration as R, integration as I, synthetic as S
lib = L.load()
cam = I.PinholeCamera()
d0, _c, _p = S.room_frame(0); d1, _c, _p = S.room_frame(1)
tgt_pc = R.PointCloud.LoadFromDepth(d0, cam, device=0); tgt_pc.EstimateNormals(0.1, 30, device=0)
src = R.PointCloud.LoadFromDepth(d1, cam, device=0).points
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
T0 = np.eye(4, dtype=np.float32).reshape(16)
for k in (1, 2, 4, 8, 16):
    tgt, nrm = np.ascontiguousarray(tgt_pc.points[::k]), np.ascontiguousarray(tgt_pc.normals[::k])
    h = C.c_void_p()
    L.check(lib.op_icp_create(C.c_void_p(tgt.ctypes.data), C.c_void_p(nrm.ctypes.data), len(tgt), 0.01, L.OP_MEM_HOST, 0, C.byref(h)))
    L.check(lib.op_icp_set_source(h, C.c_void_p(src.ctypes.data), len(src), L.OP_MEM_HOST))
    L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_FINISH, L.OP_ICP_FINISH_FP64))
    res = L.IcpResult()
    L.check(lib.op_icp_run(h, 1, fp(T0), 5, C.byref(res), None, 0, None, None))
    best = {}
    for its in (10, 70):
        b = 1e9
        for _ in range(5):
            t = time.perf_counter(); L.check(lib.op_icp_run(h, 1, fp(T0), its, C.byref(res), None, 0, None, None)); b = min(b, time.perf_counter() - t)
        best[its] = b
    print("target 1:%d (%d points): %.2f us per iteration, inliers %d" % (k, len(tgt), (best[70] - best[10]) / 60 * 1e6, res.n_inliers))
    lib.op_icp_destroy(h)
