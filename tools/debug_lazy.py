import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S
dev = torch.device("cuda:0")
n = 300
depth, rgb, poses = S.room_sequence_torch(0, n, dev)
torch.cuda.synchronize()
hv = I.CubeHandler(max_blocks=1 << 19); hv.SetVoxelResolution(0.005)
for rep in range(5):
    hv.Clear()
    t = time.perf_counter()
    for k in range(n):
        hv.IntegrateImage(depth[k], rgb[k], poses[k])
    t1 = time.perf_counter()
    hv.Synchronize()
    dt = time.perf_counter() - t
    print("rep", rep, "enqueue %.2f ms, total %.2f ms -> %.0f fps" % ((t1 - t) * 1e3, dt * 1e3, n / dt), hv.Stats()["frames"], hv.BlockCount())
for rep in range(3):
    hv.Clear()
    t = time.perf_counter()
    hv.IntegrateSequence(depth, rgb, poses); hv.Synchronize()
    dt = time.perf_counter() - t
    print("sequence rep", rep, "%.2f ms -> %.0f fps" % (dt * 1e3, n / dt))
