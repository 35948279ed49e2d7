"""Full-size parity run: N frames of the bench workload at 640x480 / 5 mm through the HIP sequence path and through
the CPU oracle (all host cores; results do not depend on the thread count), compared bit for bit.  The driver-run
version of this is tests/test_full_size_gpu.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
threads = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
depth, rgb, poses = S.room_sequence_torch(0, n, dev)
torch.cuda.synchronize()
hv = I.CubeHandler(max_blocks=1 << 18); hv.SetVoxelResolution(0.005)
t = time.perf_counter(); hv.IntegrateSequence(depth, rgb, poses); hv.Synchronize(); tg = time.perf_counter() - t
dn, cn = depth.cpu().numpy(), rgb.cpu().numpy()
O.set_fusion_threads(threads)
ov = O.Volume(voxel_res=0.005)
t = time.perf_counter()
upd = 0
for k in range(n):
    upd += ov.integrate(dn[k], cn[k], poses[k])[2]
tc = time.perf_counter() - t
hk, hx = hv.GetCubeMap(); ok, ox = ov.export()
st = hv.Stats()
print({"frames": n, "gpu_s": tg, "cpu_s": tc, "cpu_threads": threads or os.cpu_count(), "blocks": int(len(ok)), "keys_equal": bool(np.array_equal(hk, ok)),
       "voxels_bit_equal": bool(np.array_equal(hx.view(np.uint32), ox.view(np.uint32))), "updates_equal": st["voxels_updated"] == upd,
       "voxel_frames_updated": upd})
