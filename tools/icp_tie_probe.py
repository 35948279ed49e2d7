"""Runs ON THE GPU BOX: how many queries of bench.py's ICP pair (frames 0 and 1 of the room sequence) have exactly equidistant nearest targets, per run length."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from onepiece_amd import registration as R, integration as I, synthetic as S, _lib as L
dev = torch.device("cuda:0")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
depth, rgb, poses = S.room_sequence_torch(first, 2, dev)
cam = I.CubeHandler(device=0).camera
tgt_pc = R.PointCloud.LoadFromDepth(depth[0].cpu().numpy(), cam)
src = R.PointCloud.LoadFromDepth(depth[1].cpu().numpy(), cam).points
tgt_pc.EstimateNormals(0.1, 30)
tgt, nrm = tgt_pc.points, tgt_pc.normals
lib = L.load()
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
T0 = np.eye(4, dtype=np.float32).reshape(16)
for ties in (L.OP_ICP_TIES_LOWEST_INDEX, L.OP_ICP_TIES_REFERENCE):
    h = C.c_void_p()
    L.check(lib.op_icp_create(C.c_void_p(tgt.ctypes.data), C.c_void_p(nrm.ctypes.data), len(tgt), C.c_double(0.01), L.OP_MEM_HOST, 0, C.byref(h)))
    L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_TIES, ties))
    L.check(lib.op_icp_set_source(h, C.c_void_p(src.ctypes.data), len(src), L.OP_MEM_HOST))
    res = L.IcpResult()
    L.check(lib.op_icp_run(h, 1, fp(T0), 5, C.byref(res), None, 0, None, None))
    for iters in (1, 2, 5, 20, 60):
        a, b = C.c_uint64(), C.c_uint64()
        L.check(lib.op_icp_tie_stats(h, C.byref(a), C.byref(b))); a0, b0 = a.value, b.value
        t = time.perf_counter()
        L.check(lib.op_icp_run(h, 1, fp(T0), iters, C.byref(res), None, 0, None, None))
        dt = time.perf_counter() - t
        L.check(lib.op_icp_tie_stats(h, C.byref(a), C.byref(b)))
        print("ties=%d iters=%2d: %.3f ms (%.0f it/s), tied queries %d, changed %d, inliers %d" % (ties, iters, dt * 1e3, iters / dt, a.value - a0, b.value - b0, res.n_inliers))
    lib.op_icp_destroy(h)
