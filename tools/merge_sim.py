"""The N-rank merge of bench.py emulated on ONE GPU: rank r fuses frames [r*F, (r+1)*F) of the room sequence
into its own volume, then the steps of distributed.merge_volumes run with the RCCL reduce replaced by a
torch sum of the packed buffers.  Prints the union size (does it fit the root's pool?), the bytes one rank
hands to the reduce, the device time of the local steps, and checks the result against a sequential
CubeHandler::Merge chain.  usage: merge_sim.py [ranks=8] [frames_per_rank=1000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S, distributed as D

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
MAXB = 1 << 19
vols, ops = [], []
for r in range(R):
    depth, rgb, poses = S.room_sequence_torch(r * F, F, dev)
    torch.cuda.synchronize()
    hv = I.CubeHandler(max_blocks=MAXB); hv.SetVoxelResolution(0.005)
    t = time.perf_counter()
    hv.IntegrateSequence(depth, rgb, poses); hv.Synchronize()
    dt = time.perf_counter() - t
    print("rank %d: %d frames -> %d blocks (%.1f frames/s)" % (r, F, hv.BlockCount(), F / dt), flush=True)
    vols.append(hv); ops.append(D.HipVolumeOps(hv, dev))
    del depth, rgb


def tm(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return time.perf_counter() - t, r


t_keys, keys = zip(*[tm(o.keys) for o in ops])
allk = torch.cat(list(keys))


def uni():
    off = 1 << 20; k64 = allk.to(torch.int64) + off
    pk = torch.unique((k64[:, 0] << 42) | (k64[:, 1] << 21) | k64[:, 2])
    return torch.stack([(pk >> 42) - off, ((pk >> 21) & 0x1FFFFF) - off, (pk & 0x1FFFFF) - off], 1).to(torch.int32).contiguous()


t_uni0, union = tm(uni)
t_uni, union = tm(uni)
n_union = union.shape[0]
print("union of %d ranks: %d blocks (sum of local %d; root pool holds %d) -> %.2f GB per rank into the reduce"
      % (R, n_union, allk.shape[0], MAXB, n_union * 5 * 512 * 4 / 1e9))
acc = None
t_pack = []
t_pack0, p = tm(lambda: ops[0].pack_sum(union))   # first call: fresh hipMalloc of the exchange buffer
del p
for o in ops:
    t, p = tm(lambda: o.pack_sum(union))
    t_pack.append(t)
    acc = p if acc is None else acc.add_(p)
    del p
root = I.CubeHandler(max_blocks=MAXB); root.SetVoxelResolution(0.005)
t_unpack, _ = tm(lambda: D.HipVolumeOps(root, dev).unpack_sum(union, acc))
print("local steps per rank: keys %.2f ms, union %.2f ms (first call %.1f), pack_sum %.2f ms (first call, buffer not yet allocated: %.1f), "
      "unpack_sum (root) %.2f ms" % (max(t_keys) * 1e3, t_uni * 1e3, t_uni0 * 1e3, max(t_pack) * 1e3, t_pack0 * 1e3, t_unpack * 1e3))

# reference semantics: a sequential Merge chain (CubeHandler.h:145-167)
seq = vols[0]
t_seq, _ = tm(lambda: [seq.Merge(v) for v in vols[1:]])
ks, vs = seq.GetCubeMap()
kr, vr = root.GetCubeMap()
o1, o2 = np.lexsort(ks.T[::-1]), np.lexsort(kr.T[::-1])
assert np.array_equal(ks[o1], kr[o2]), "key sets differ"
a, b = vs[o1], vr[o2]
obs = a[..., 1] > 0
assert np.array_equal(a[..., 1], b[..., 1]), "weights differ"
err = np.abs(a[obs] - b[obs]).max(axis=0)
print("vs sequential Merge chain (%.1f ms): keys equal, weights equal, max |diff| sdf %.2e colour %.2e (fp32 summation order)"
      % (t_seq * 1e3, err[0], err[2:].max()))
