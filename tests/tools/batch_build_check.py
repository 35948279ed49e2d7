"""Fuses K frames of W x H pixels (the analytic room, rendered on the GPU) with the library ONEPIECE_HIP_LIBRARY names and prints a JSON line:
block count, statistics and SHA-256 digests of the sorted keys / voxels.  Used to compare two BUILDS of the library (default 32-frame batches,
-DOP_MAX_BATCH=64) on images at the pixel limit.  usage: batch_build_check.py W H K voxel [probe_W probe_H]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from onepiece_amd import integration as I, synthetic as S, _lib as L

W, H, K, voxel = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
dev = torch.device("cuda:0")
sx, sy = W / S.W, H / S.H
cam = I.PinholeCamera()
cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height = S.FX * sx, S.FY * sy, S.CX * sx, S.CY * sy, W, H
out = {}
if len(sys.argv) > 6:   # is a camera of this size admitted at all?
    big = I.PinholeCamera(); big.width, big.height = int(sys.argv[5]), int(sys.argv[6])
    try:
        I.CubeHandler(big); out["probe_admitted"] = True
    except L.OnePieceHipError as e:
        out["probe_admitted"] = False; out["probe_error"] = str(e)
depth = torch.empty((K, H, W), dtype=torch.float32, device=dev)
rgb = torch.empty((K, H, W, 3), dtype=torch.uint8, device=dev)
poses = np.empty((K, 4, 4), np.float32)
for k in range(K):
    poses[k] = S.room_pose(300 + 2 * k)
    depth[k], rgb[k] = S.room_render(poses[k], xp=torch, device=dev, width=W, height=H, fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy)
torch.cuda.synchronize()
hv = I.CubeHandler(cam)
hv.SetVoxelResolution(voxel)
hv.IntegrateSequence(depth, rgb, poses)
hv.Synchronize()
st = hv.Stats()
k, v = hv.GetCubeMap()
out.update(blocks=int(len(k)), frames=int(st["frames"]), launches=int(st["launches"]), voxels_updated=int(st["voxels_updated"]),
           keys_sha=hashlib.sha256(np.ascontiguousarray(k).tobytes()).hexdigest(), voxels_sha=hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest(),
           library=os.path.basename(L.SO_PATH))
print(json.dumps(out))
