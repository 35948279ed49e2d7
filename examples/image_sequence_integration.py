#!/usr/bin/env python3
"""Mirror of example/ImageSequenceIntegration.cpp:15-45 on the HIP path: read a sequence directory
(associate.txt + trajectory.txt), fuse every 10th frame into a 6.25 mm TSDF, report the volume.
tool::ConvertDepthTo32F + tool::BilateralFilter run on the GPU, on the CubeHandler's stream, from the raw 16-bit
depth (the filter follows cv::bilateralFilter's documented definition; OpenCV itself is unpinned).

    python examples/image_sequence_integration.py <sequence_dir> [--write-synthetic N]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from onepiece_amd import integration as I, sequence as Q, synthetic as S, tool as T


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
    # every 10th frame (:29); PNG pairs are decoded ahead on host threads so IO overlaps the fusion.  Frames go to
    # HBM 16 at a time: raw CV_16UC1 depth -> ConvertDepthTo32F + BilateralFilter (:36-37) -> IntegrateImage (:38),
    # all on the CubeHandler's stream
    batch = []

    def flush():
        if not batch:
            return
        idx = [b[0] for b in batch]
        d16 = torch.from_numpy(np.stack([b[2] for b in batch]).view(np.int16)).cuda().view(torch.uint16)
        c8 = torch.from_numpy(np.stack([b[1] for b in batch])).cuda()
        torch.cuda.current_stream().synchronize()            # uploads final before the library's stream reads them
        filtered_depth = T.BilateralFilter(d16, depth_scale=camera.depth_scale, stream=cube_handler.Stream())
        cube_handler.IntegrateSequence(filtered_depth, c8, poses[idx])
        cube_handler.Synchronize()                           # the batch's tensors may be released after this
        batch.clear()

    for i, rgb, depth in Q.FramePrefetcher(rgb_files, depth_files, indices=range(0, len(poses), 10)):
        batch.append((i, rgb, depth))
        used += 1
        if len(batch) == 16:
            flush()
    flush()
    cube_handler.Synchronize()
    dt = time.perf_counter() - t
    pts, _ = cube_handler.GetPointCloud()
    print("fused %d of %d frames in %.3f s (incl. PNG decode); %d blocks, %d surface-band voxels"
          % (used, len(poses), dt, cube_handler.BlockCount(), len(pts)))


if __name__ == "__main__":
    main()
