"""One-off measurements quoted in DESIGN.md: PCIe-inclusive fusion rate (host buffers through
op_volume_integrate) and the cost of the multi-GPU merge steps on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S, distributed as D
dev = torch.device("cuda:0")
n = 300
depth, rgb, poses = S.room_sequence_torch(0, n, dev)
torch.cuda.synchronize()
dn, cn = depth.cpu().numpy(), rgb.cpu().numpy()
hv = I.CubeHandler(max_blocks=1 << 19); hv.SetVoxelResolution(0.005)
for rep in range(2):
    hv.Clear()
    t = time.perf_counter()
    for k in range(n):
        hv.IntegrateImage(dn[k], cn[k], poses[k])      # host buffers: staged H2D copy + batch of one, sync per call
    hv.Synchronize()
    dt = time.perf_counter() - t
    print("host-buffer IntegrateImage (PCIe + sync per frame): %.1f frames/s" % (n / dt))
for rep in range(2):
    hv.Clear()
    t = time.perf_counter()
    for k in range(n):
        hv.IntegrateImage(depth[k], rgb[k], poses[k])  # device buffers, one frame per call, no sync
    hv.Synchronize()
    dt = time.perf_counter() - t
    print("device-buffer IntegrateImage, one frame per launch group: %.1f frames/s" % (n / dt))
hv.Clear(); hv.IntegrateSequence(depth, rgb, poses); hv.Synchronize()
ops = D.HipVolumeOps(hv, dev)
def tm(f, reps=3):
    torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best, r
t_keys, keys = tm(ops.keys)
allk = torch.cat([keys + torch.tensor([0, 0, r % 3], dtype=torch.int32, device=dev) for r in range(8)])  # 8 overlapping rank key sets
def uni():
    off = 1 << 20; k64 = allk.to(torch.int64) + off
    pk = torch.unique((k64[:, 0] << 42) | (k64[:, 1] << 21) | k64[:, 2])
    return torch.stack([(pk >> 42) - off, ((pk >> 21) & 0x1FFFFF) - off, (pk & 0x1FFFFF) - off], 1).to(torch.int32).contiguous()
t_uni, union = tm(uni)
t_pack, packed = tm(lambda: ops.pack_sum(union))
hv2 = I.CubeHandler(max_blocks=1 << 19); hv2.SetVoxelResolution(0.005)
ops2 = D.HipVolumeOps(hv2, dev)
t_unpack, _ = tm(lambda: ops2.unpack_sum(union, packed))
print("merge steps for %d local / %d union blocks: keys %.2f ms, packed-int64 unique %.2f ms, pack_sum %.2f ms (%.1f GB), unpack_sum %.2f ms"
      % (keys.shape[0], union.shape[0], t_keys * 1e3, t_uni * 1e3, t_pack * 1e3, packed.numel() * 4 / 1e9, t_unpack * 1e3))
