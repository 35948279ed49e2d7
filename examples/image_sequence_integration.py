#!/usr/bin/env python3
"""Mirror of example/ImageSequenceIntegration.cpp:15-45 on the HIP path: read a sequence directory
(associate.txt + trajectory.txt), fuse every 10th frame into a 6.25 mm TSDF, report the volume.
The OpenCV bilateral filter of the original (un-vendored, unpinned) is not applied.

    python examples/image_sequence_integration.py <sequence_dir> [--write-synthetic N]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from onepiece_amd import integration as I, sequence as Q, synthetic as S


def main():
    path = sys.argv[1]
    if "--write-synthetic" in sys.argv:
        n = int(sys.argv[sys.argv.index("--write-synthetic") + 1])
        frames = [S.room_frame(i) for i in range(n)]
        Q.WriteImageSequence(path, [f[0] for f in frames], [f[1] for f in frames], [f[2] for f in frames])
    camera = I.PinholeCamera()                      # camera::PinholeCamera camera;  (OPEN3D preset)
    cube_handler = I.CubeHandler(camera)
    cube_handler.SetVoxelResolution(0.00625)        # ImageSequenceIntegration.cpp:21
    rgb_files, depth_files, poses = Q.ReadImageSequenceWithPose(path)
    t = time.perf_counter()
    used = 0
    # every 10th frame (:29); PNG pairs are decoded ahead on host threads so IO overlaps the fusion
    for i, rgb, depth in Q.FramePrefetcher(rgb_files, depth_files, indices=range(0, len(poses), 10)):
        refined_depth = Q.ConvertDepthTo32F(depth, camera.depth_scale)
        cube_handler.IntegrateImage(refined_depth, rgb, poses[i])
        used += 1
    cube_handler.Synchronize()
    dt = time.perf_counter() - t
    pts, _ = cube_handler.GetPointCloud()
    print("fused %d of %d frames in %.3f s (incl. PNG decode); %d blocks, %d surface-band voxels"
          % (used, len(poses), dt, cube_handler.BlockCount(), len(pts)))


if __name__ == "__main__":
    main()
