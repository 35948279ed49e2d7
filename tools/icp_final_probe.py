"""Runs ON THE GPU BOX: what the final CountInliers' re-decision (op_icp_final_stats: correspondences the grid search cannot vouch for under the
last pose, looked up again in the tree the reference would search) costs a registration of bench.py's ICP pair, per mode and run length.
usage: icp_final_probe.py [first_frame=0]"""
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
for mode in (1, 0):
    h = C.c_void_p()
    L.check(lib.op_icp_create(C.c_void_p(tgt.ctypes.data), C.c_void_p(nrm.ctypes.data), len(tgt), C.c_double(0.01), L.OP_MEM_HOST, 0, C.byref(h)))
    L.check(lib.op_icp_set_source(h, C.c_void_p(src.ctypes.data), len(src), L.OP_MEM_HOST))
    res = L.IcpResult()
    L.check(lib.op_icp_run(h, mode, fp(T0), 60, C.byref(res), None, 0, None, None))  # warm (also builds whatever of the tie tree the pair needs)
    for iters in (1, 2, 5, 10, 20, 30, 60):
        best, red = 1e9, 0
        for _ in range(3):
            r0, r = C.c_uint64(), C.c_uint64()
            L.check(lib.op_icp_final_stats(h, C.byref(r0)))   # (the counter runs over the context's lifetime)
            t = time.perf_counter()
            L.check(lib.op_icp_run(h, mode, fp(T0), iters, C.byref(res), None, 0, None, None))
            best = min(best, time.perf_counter() - t)
            L.check(lib.op_icp_final_stats(h, C.byref(r))); red = r.value - r0.value
        print("%s iters=%2d: %.3f ms per call, final correspondences re-decided in the tree %d, inliers %d" % ("point-to-plane" if mode else "point-to-point", iters, best * 1e3, red, res.n_inliers), flush=True)
    lib.op_icp_destroy(h)
