"""example/DenseFusion/DenseFusion.cpp on the MI355X path: frame-to-frame dense tracking of every frame,
then TSDF fusion of every `--every`-th tracked frame with the tracked poses, trajectory.txt out.

    python examples/dense_fusion.py <basepath> <voxel_resolution> [--every 8]     # a TUM-format folder
    python examples/dense_fusion.py - 0.01 --synthetic 200                         # the analytic room

Differences from the reference example, stated: no submap registration / FastBA (out
of scope) so poses are pure odometry; the bilateral depth filter (DenseFusion.cpp:92-93) follows
cv::bilateralFilter's documented definition (OpenCV is unpinned); no mesh written.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepiece_amd import integration as I, dense_slam as DS, synthetic as S, sequence as Q, tool  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("basepath")
    ap.add_argument("voxel_resolution", type=float)
    ap.add_argument("--synthetic", type=int, default=0, help="ignore basepath, render N frames of the analytic room")
    ap.add_argument("--every", type=int, default=8, help="fuse every n-th tracked frame (DenseFusion.cpp:87)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    cam = I.PinholeCamera("OPEN3D_DATASET")
    if args.synthetic:
        frames = [S.room_frame(i) for i in range(args.synthetic)]
        rgbs, depths, gt = [f[1] for f in frames], [f[0] for f in frames], [f[2] for f in frames]
    else:
        rgb_files, depth_files = Q.ReadImageSequence(args.basepath)
        rgbs = [Q.imread(f) for f in rgb_files]
        depths = [Q.imread(f, True) for f in depth_files]
        gt = None
    slam = DS.DenseSlam(cam)
    t0 = time.perf_counter()
    for rgb, depth in zip(rgbs, depths):
        slam.UpdateFrame(rgb, depth)
    t1 = time.perf_counter()
    vol = I.CubeHandler(cam)
    vol.SetVoxelResolution(args.voxel_resolution)
    fused = 0
    for i, (rgb, depth) in enumerate(zip(rgbs, depths)):
        if not slam.tracking_success[i] or i % args.every:
            continue
        filtered_depth = tool.BilateralFilter(depth, depth_scale=cam.depth_scale)   # ConvertDepthTo32F + BilateralFilter (:92-93)
        vol.IntegrateImage(filtered_depth, rgb, slam.global_poses[i])
        fused += 1
    n_blocks = vol.BlockCount()
    t2 = time.perf_counter()
    out = args.out or (os.path.join(args.basepath, "trajectory.txt") if not args.synthetic else None)
    if out:
        with open(out, "w") as f:
            for i, T in enumerate(slam.global_poses):
                if slam.tracking_success[i]:
                    f.write(" ".join("%g" % v for v in np.asarray(T).reshape(16)) + "\n")
    print("tracked %d/%d frames in %.3f s (%.1f frames/s); fused %d frames into %d blocks in %.3f s"
          % (sum(slam.tracking_success), len(rgbs), t1 - t0, len(rgbs) / (t1 - t0), fused, n_blocks, t2 - t1))
    if gt is not None:
        # poses are relative to frame 0: compare with inv(gt0) * gt_i
        g0 = np.linalg.inv(gt[0].astype(np.float64))
        err = max(np.abs(np.asarray(slam.global_poses[i], np.float64) - g0 @ gt[i].astype(np.float64))[:3, 3].max() for i in range(len(gt)))
        print("max translation drift vs ground truth: %.4f m over %d frames" % (err, len(gt)))


if __name__ == "__main__":
    main()
