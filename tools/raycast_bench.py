"""Raycast timing: 640x480 rays through a 5 mm volume fused from N room frames; outputs stay in HBM."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
depth, rgb, poses = S.room_sequence_torch(0, n, dev)
vol = I.CubeHandler(I.PinholeCamera("OPEN3D_DATASET")); vol.SetVoxelResolution(0.005)
vol.IntegrateSequence(depth, rgb, poses)
print("blocks", vol.BlockCount())
lib = L.load()
d = torch.zeros((480, 640), dtype=torch.float32, device=dev)
nr = torch.zeros((480, 640, 3), dtype=torch.float32, device=dev)
cl = torch.zeros((480, 640, 3), dtype=torch.float32, device=dev)
fp = lambda a: a.ctypes.data_as(L._fp)
vp = lambda t: C.cast(C.c_void_p(t.data_ptr()), L._fp)
for want in (False, True):
    def run(k):
        p = np.ascontiguousarray(poses[(n // 2 + k) % n], np.float32).reshape(16)
        L.check(lib.op_volume_raycast(vol._h, C.byref(vol.camera), fp(p), vp(d), vp(nr) if want else None, vp(cl) if want else None, L.OP_MEM_DEVICE))
    run(0)
    t = time.perf_counter()
    for k in range(50):
        run(k)
    dt = (time.perf_counter() - t) / 50
    print("normals+colours" if want else "depth only    ", "%.3f ms per 640x480 raycast, hit fraction %.3f" % (dt * 1e3, float((d > 0).float().mean())))
