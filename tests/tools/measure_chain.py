"""Measurement behind tests/test_full_size_gpu.py's config-4 bars: a tracked pose chain over N frames on the GPU vs the
CPU oracle (per pair and chained), the oracle's own float-vs-double self-distance per pair, drift of both chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S, dense_slam as DS
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
depth, rgb, poses = S.room_sequence_torch(first, n, dev)
torch.cuda.synchronize()
slam = DS.DenseSlam(I.PinholeCamera("OPEN3D_DATASET"))
pairT = []
for i in range(n):
    slam.UpdateFrame(rgb[i], depth[i])
hd, hc = depth.cpu().numpy(), rgb.cpu().numpy()
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
ocam = O.make_camera()
ref = [np.eye(4, dtype=np.float32)]
t = time.perf_counter()
per_pair, noise = [], []
for i in range(1, n):
    r = O.dense_tracking(ocam, hc[i - 1], hc[i], hd[i - 1], hd[i], (4, 8, 16), 0)
    O.lib().orc_set_accumulate_double(1)
    rd = O.dense_tracking(ocam, hc[i - 1], hc[i], hd[i - 1], hd[i], (4, 8, 16), 0)
    O.lib().orc_set_accumulate_double(0)
    ref.append(DS._mat4_mul_f32(ref[-1], O.mat4_inverse(r["T"])))
    # the GPU's pair transform, recovered from its chain: T^-1 = global[i-1]^-1 * global[i]
    gp = np.linalg.inv(np.asarray(slam.global_poses[i - 1], np.float64)) @ np.asarray(slam.global_poses[i], np.float64)
    per_pair.append(rel(gp, np.linalg.inv(r["T"].astype(np.float64))))
    noise.append(rel(r["T"], rd["T"]))
print("oracle s/pair", (time.perf_counter() - t) / (n - 1) / 2)
chain = [rel(slam.global_poses[i], ref[i]) for i in range(n)]
g0 = np.linalg.inv(poses[0].astype(np.float64))
dg = [float(np.abs(np.asarray(slam.global_poses[i], np.float64) - g0 @ poses[i].astype(np.float64))[:3, 3].max()) for i in range(n)]
dr = [float(np.abs(np.asarray(ref[i], np.float64) - g0 @ poses[i].astype(np.float64))[:3, 3].max()) for i in range(n)]
pp, nz = np.array(per_pair), np.array(noise)
print("per-pair rel err: max %.2e median %.2e  <=1e-4: %d/%d" % (pp.max(), np.median(pp), (pp <= 1e-4).sum(), len(pp)))
print("oracle float-vs-double per pair: max %.2e median %.2e" % (nz.max(), np.median(nz)))
print("pairs where gpu err > 3x oracle self-noise and > 1e-4:", [(i + 1, "%.1e" % pp[i], "%.1e" % nz[i]) for i in range(len(pp)) if pp[i] > 1e-4 and pp[i] > 3 * nz[i]])
print("chain rel err: max %.2e at %d, final %.2e" % (max(chain), int(np.argmax(chain)), chain[-1]))
print("drift vs ground truth (m): gpu max %.4f final %.4f | oracle max %.4f final %.4f" % (max(dg), dg[-1], max(dr), dr[-1]))
print("tracked", sum(bool(x) for x in slam.tracking_success), "/", n)
