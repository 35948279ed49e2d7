import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S
dev = torch.device("cuda:0")
n = 1000
seqs = [S.room_sequence_torch(0, n, dev), S.room_sequence_torch(1000, n, dev)]
torch.cuda.synchronize()
def run(nvol, prof=False):
    vols = [I.CubeHandler(max_blocks=1 << 19) for _ in range(nvol)]
    for v in vols: v.SetVoxelResolution(0.005)
    best = 1e9
    for rep in range(3):
        for v in vols: v.Clear(); v.Synchronize(); v.ProfileEnable(1 if prof else 0)
        t = time.perf_counter()
        for s in range(0, n, 100):
            for k, v in enumerate(vols):
                d, c, p = seqs[k]
                v.IntegrateSequence(d[s:s+100], c[s:s+100], p[s:s+100])
        for v in vols: v.Synchronize()
        best = min(best, time.perf_counter() - t)
    if prof:
        for v in vols: print("   ", v.ProfileRead())
    return nvol * n / best
print("1 volume : %.0f frames/s" % run(1, True))
print("2 volumes: %.0f frames/s aggregate" % run(2, True))
