import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, ctypes as C
from onepiece_amd import registration as R, synthetic as S, _lib as L
from oracle import oracle as O
s = 4
camt = (S.FX / s, S.FY / s, S.CX / s, S.CY / s, S.W // s, S.H // s, 1000.0)
cam = O.make_camera(*camt); cl = []; dp = []
for i in (0, 3):
    d, c = S.room_render(S.room_pose(i), width=camt[4], height=camt[5], fx=camt[0], fy=camt[1], cx=camt[2], cy=camt[3])
    cl.append(O.load_from_depth(cam, d)); dp.append(d)
tgt, src = cl
nrm = S.image_normals(dp[0], *camt[:4])
nn = np.empty(len(src), np.int64)
for a in range(0, len(src), 512):
    q = src[a:a+512]
    dx = q[:, None, 0]-tgt[None, :, 0]; dy = q[:, None, 1]-tgt[None, :, 1]; dz = q[:, None, 2]-tgt[None, :, 2]
    nn[a:a+512] = ((dx*dx+dy*dy)+dz*dz).argmin(1)
lib = L.load()
h = C.c_void_p()
L.check(lib.op_icp_create(C.c_void_p(tgt.ctypes.data), C.c_void_p(nrm.ctypes.data), len(tgt), 0.05, 0, 0, C.byref(h)))
L.check(lib.op_icp_set_source(h, C.c_void_p(src.ctypes.data), len(src), 0))
T0 = np.eye(4, dtype=np.float32).reshape(16)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
sums = (C.c_double * 42)(); ninl = C.c_uint64(0); err = C.c_double(0)
L.check(lib.op_icp_iterate(h, fp(T0), 0, sums, C.byref(ninl), C.byref(err)))
sp = np.array(sums[:15])
print("point mode: n_inl", ninl.value, "sum t gpu", sp[3:6], "cpu", tgt[nn].astype(np.float64).sum(0), "sum s gpu", sp[0:3], "cpu", src.astype(np.float64).sum(0))
L.check(lib.op_icp_iterate(h, fp(T0), 1, sums, C.byref(ninl), C.byref(err)))
g = np.array(sums[:42])
inl = np.stack([np.arange(len(src)), nn], 1).astype(np.int32)
T = np.empty(16, np.float32); JTJ = np.empty(36, np.float32); JTr = np.empty(6, np.float32)
O.lib().orc_p2plane_step(fp(src), fp(tgt), fp(nrm), inl.ctypes.data_as(C.POINTER(C.c_int32)), len(src), fp(T), fp(JTJ), fp(JTr))
print("JTJ rel diff", np.abs(g[:36]-JTJ).max()/np.abs(JTJ).max(), "JTr gpu", g[36:], "cpu", JTr)
res = L.IcpResult(); pairs = np.empty((len(src), 2), np.int32)
L.check(lib.op_icp_run(h, 1, fp(T0), 1, C.byref(res), pairs.ctypes.data_as(C.POINTER(C.c_int32)), len(pairs), None, None))
print("gpu last_T", np.array(res.last_T).reshape(4, 4)); print("cpu T", T.reshape(4, 4))
x = O.solve6(g[:36].astype(np.float32), g[36:].astype(np.float32)); print("x from gpu sums via oracle solve", x, "->", O.se3_exp(x)[:3, 3])
x2 = O.solve6(JTJ, JTr); print("x from cpu sums", x2)
lib.op_icp_destroy(h)
