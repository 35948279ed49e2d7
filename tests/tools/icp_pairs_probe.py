import sys, time, ctypes as C, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from onepiece_amd import registration as R, integration as I, synthetic as S, _lib as L
from oracle import oracle as O
lib = L.load(); cam = I.PinholeCamera("OPEN3D_DATASET")
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)); T0 = np.eye(4, dtype=np.float32).reshape(16)
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
for a in (0, 2, 4, 6, 100, 300):
    d0, _c, _p = S.room_frame(a); d1, _c, _p = S.room_frame(a + 1)
    tp = R.PointCloud.LoadFromDepth(d0, cam, device=0); tp.EstimateNormals(0.1, 30, device=0)
    sp = R.PointCloud.LoadFromDepth(d1, cam, device=0).points
    h = C.c_void_p(); L.check(lib.op_icp_create(C.c_void_p(tp.points.ctypes.data), C.c_void_p(tp.normals.ctypes.data), len(tp.points), 0.01, L.OP_MEM_HOST, 0, C.byref(h)))
    L.check(lib.op_icp_set_source(h, C.c_void_p(sp.ctypes.data), len(sp), L.OP_MEM_HOST))
    r = L.IcpResult(); L.check(lib.op_icp_run(h, 1, fp(T0), 10, C.byref(r), None, 0, None, None))
    L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_SUMS, L.OP_ICP_SUMS_REFERENCE_F32))
    rs = L.IcpResult(); L.check(lib.op_icp_run(h, 1, fp(T0), 10, C.byref(rs), None, 0, None, None))
    ref = O.icp(sp, tp.points, tp.normals, None, 10, 0.01, True)
    O.lib().orc_set_accumulate_double(1)
    refd = O.icp(sp, tp.points, tp.normals, None, 10, 0.01, True)
    O.lib().orc_set_accumulate_double(0)
    g = np.array(r.T).reshape(4, 4); gs = np.array(rs.T).reshape(4, 4); gl = np.array(r.last_T).reshape(4, 4); gsl = np.array(rs.last_T).reshape(4,4)
    print("pair", a, "default vs cpu: T %.2e last_T %.2e inl %d/%d | strict vs cpu: T %.2e last_T %.2e inl %d | cpu float vs cpu double sums: T %.2e last_T %.2e inl %d | default vs cpu-double: T %.2e last_T %.2e" % (
        rel(g, ref["T"]), rel(gl, ref["last_T"]), r.n_inliers, len(ref["pairs"]), rel(gs, ref["T"]), rel(gsl, ref["last_T"]), rs.n_inliers,
        rel(ref["T"], refd["T"]), rel(ref["last_T"], refd["last_T"]), len(refd["pairs"]), rel(g, refd["T"]), rel(gl, refd["last_T"])))
    print("   per-iter inliers cpu ", ref["per_iter_inliers"].tolist()); print("   per-iter inliers cpu-double", refd["per_iter_inliers"].tolist())
    lib.op_icp_destroy(h)

# rate of the reference-order mode (device sums), one context and four in flight
ctx = []
for k in range(4):
    d0, _c, _p = S.room_frame(2 * k); d1, _c, _p = S.room_frame(2 * k + 1)
    tp = R.PointCloud.LoadFromDepth(d0, cam, device=0); tp.EstimateNormals(0.1, 30, device=0)
    sp = R.PointCloud.LoadFromDepth(d1, cam, device=0).points
    h = C.c_void_p(); L.check(lib.op_icp_create(C.c_void_p(tp.points.ctypes.data), C.c_void_p(tp.normals.ctypes.data), len(tp.points), 0.01, L.OP_MEM_HOST, 0, C.byref(h)))
    L.check(lib.op_icp_set_source(h, C.c_void_p(sp.ctypes.data), len(sp), L.OP_MEM_HOST))
    L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_SUMS, L.OP_ICP_SUMS_REFERENCE_F32)); ctx.append(h)
for K in (1, 2, 4):
    best = 1e9
    for rep in range(3):
        rs = [L.IcpResult() for _ in range(K)]
        t = time.perf_counter()
        for k in range(K): L.check(lib.op_icp_run_enqueue(ctx[k], 1, fp(T0), 20, C.byref(rs[k]), None, 0))
        for k in range(K): L.check(lib.op_icp_wait(ctx[k]))
        best = min(best, time.perf_counter() - t)
    print("reference-order sums, K", K, "aggregate it/s %.0f" % (K * 20 / best))
