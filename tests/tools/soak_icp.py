"""Soak run (not part of the suite): the sums of the ICP iteration kernel -- folded across workgroups by whichever
workgroup arrives last -- and whole op_icp_run calls must be bit-identical launch after launch."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
from onepiece_amd import _lib as L
sys.path.insert(0, '/root/repo/tests')
from helpers import room_cloud
lib = L.load()
_, src, _ = room_cloud(301, scale=1)
_, tgt, nrm = room_cloud(300, scale=1)
h = C.c_void_p()
L.check(lib.op_icp_create(C.c_void_p(tgt.ctypes.data), C.c_void_p(nrm.ctypes.data), len(tgt), C.c_double(0.01), L.OP_MEM_HOST, 0, C.byref(h)))
L.check(lib.op_icp_set_source(h, C.c_void_p(src.ctypes.data), len(src), L.OP_MEM_HOST))
T = np.eye(4, dtype=np.float32)
ref = None; bad = 0
for mode in (1, 0):
    ref = None
    for k in range(3000):
        out = np.zeros(42, np.float64); cnt = C.c_uint64(); err = C.c_double()
        L.check(lib.op_icp_iterate(h, T.ctypes.data_as(C.POINTER(C.c_float)), mode, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt), C.byref(err)))
        cur = (out.tobytes(), cnt.value, err.value)
        if ref is None: ref = cur
        elif cur != ref: bad += 1
    print("mode", mode, "launches 3000 mismatches", bad, "inliers", ref[1])
# whole runs: identical results call after call (host-solve loop with published rows)
res = L.IcpResult()
first = None; bad = 0
for k in range(200):
    L.check(lib.op_icp_run(h, 1, T.ctypes.data_as(C.POINTER(C.c_float)), 30, C.byref(res), None, 0, None, None))
    cur = (bytes(res.T), bytes(res.last_T), res.n_inliers, res.rmse)
    if first is None: first = cur
    elif cur != first: bad += 1
print("op_icp_run x200 mismatches", bad, "inliers", first[2])
lib.op_icp_destroy(h)
