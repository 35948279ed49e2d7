"""ICP replicas in the reference-order mode: K contexts through op_icp_run_many (their sequential sums in ONE k_seq_sums_many launch per round) against K independent
runs (op_icp_run_enqueue: K one-workgroup k_seq_sums launches on K streams).  Under `rocprofv3 --kernel-trace --stats` the per-kernel table shows the mechanism: the same
number of sum workgroups, 1/K as many launches, every launch as long as one k_seq_sums.      python tools/icp_many_probe.py [K=16] [iterations=20] [reference|fp64]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from onepiece_amd import integration as I, registration as R, synthetic as S, _lib as L

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sums = {"reference": L.OP_ICP_SUMS_REFERENCE_F32, "fp64": L.OP_ICP_SUMS_FP64}[sys.argv[3] if len(sys.argv) > 3 else "reference"]
lib = L.load()
if len(sys.argv) > 4:
    L.check(lib.op_runtime_set_option(L.OP_RUNTIME_OPT_ICP_MANY_IN_FLIGHT, int(sys.argv[4])))
cam = I.PinholeCamera("OPEN3D_DATASET")
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
ctxs = []
for k in range(K):
    d0, _c, _p = S.room_frame(2 * k); d1, _c, _p = S.room_frame(2 * k + 1)
    tp = R.PointCloud.LoadFromDepth(d0, cam); tp.EstimateNormals(0.1, 30)
    sp = R.PointCloud.LoadFromDepth(d1, cam).points
    h = C.c_void_p()
    L.check(lib.op_icp_create(C.c_void_p(tp.points.ctypes.data), C.c_void_p(tp.normals.ctypes.data), len(tp.points), 0.01, L.OP_MEM_HOST, 0, C.byref(h)))
    L.check(lib.op_icp_set_source(h, C.c_void_p(sp.ctypes.data), len(sp), L.OP_MEM_HOST))
    L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_SUMS, sums))
    ctxs.append(h)
T0 = np.eye(4, dtype=np.float32).reshape(16)
arr = (C.c_void_p * K)(*[c.value for c in ctxs])
many = (L.IcpResult * K)()
dt_many = 1e9
for rep in range(5):
    t = time.perf_counter()
    L.check(lib.op_icp_run_many(arr, K, 1, None, iters, C.cast(many, C.c_void_p)))
    dt_many = min(dt_many, time.perf_counter() - t)
alone = [L.IcpResult() for _ in range(K)]
dt_alone = 1e9
for rep in range(5):
    t = time.perf_counter()
    for k in range(K):
        L.check(lib.op_icp_run_enqueue(ctxs[k], 1, fp(T0), iters, C.byref(alone[k]), None, 0))
    for k in range(K):
        L.check(lib.op_icp_wait(ctxs[k]))
    dt_alone = min(dt_alone, time.perf_counter() - t)
same = all(bytes(many[k].T) == bytes(alone[k].T) and many[k].n_inliers == alone[k].n_inliers for k in range(K))
print("K = %d contexts x %d iterations (%s sums): op_icp_run_many %.0f iterations/s, independent runs %.0f iterations/s, results identical: %s"
      % (K, iters, sys.argv[3] if len(sys.argv) > 3 else "reference", K * iters / dt_many, K * iters / dt_alone, same))
for h in ctxs:
    lib.op_icp_destroy(h)
