"""Host-side mirror of the TRACKING half of example/DenseFusion's DenseSlam
(/root/reference/example/DenseFusion/DenseSlam.{h,cpp}): per frame `Odometry::DenseTracking(last
tracked frame, current frame, Identity)` and the pose chaining of DenseSlam.cpp:21-33.

Not mirrored (out of scope): submap models, FPFH + RANSAC registration and FastBA
(`RegisterSubmap`, `Optimize`) -- they only touch the poses every `step` frames; without them the
trajectory is pure frame-to-frame odometry, which is what this module reports.
The tracking itself runs inside libonepiece_hip.so (op_tracker_dense_tracking).
"""
import numpy as np

from . import integration as I
from . import odometry as O


def _mat4_mul_f32(A, B):
    """Matrix4f * Matrix4f in float32 with Eigen's accumulation order ((a0 b0 + a1 b1) + a2 b2) + a3 b3."""
    A, B = np.asarray(A, np.float32), np.asarray(B, np.float32)
    out = np.empty((4, 4), np.float32)
    for r in range(4):
        out[r] = ((A[r, 0] * B[0] + A[r, 1] * B[1]) + A[r, 2] * B[2]) + A[r, 3] * B[3]
    return out


class DenseSlam:
    """one_piece::DenseSlam (DenseSlam.h:41-128), tracking part."""

    def __init__(self, camera=None, device=0):
        self.camera = camera if camera is not None else I.PinholeCamera()
        self.rgbd_odometry = O.Odometry(self.camera, device=device)    # DenseSlam.h:56
        self.global_poses = []
        self.tracking_success = []
        self.rmse = []
        self.last_tracking_frame_id = -1                               # DenseSlam.h:118
        self.max_reprojection_error_3d = 1.5                           # DenseSlam.h:122
        self._last = None                                              # (rgb, depth) of the last tracked frame

    def UpdateFrame(self, rgb, depth):
        """DenseSlam::UpdateFrame (DenseSlam.cpp:8-36).  rgb (h,w,3) uint8, depth (h,w) f32 metres or
        u16 raw; numpy or CUDA torch tensors (device-resident frames are used in place)."""
        frame_id = len(self.global_poses)
        self.global_poses.append(np.eye(4, dtype=np.float32))
        ok = True
        rmse = 0.0
        if frame_id > 0:
            lrgb, ldepth = self._last
            res = self.rgbd_odometry.DenseTracking(lrgb, rgb, ldepth, depth, np.eye(4, dtype=np.float32), 0,
                                                   want_correspondences=False)
            rmse = float(np.float32(res.rmse))                         # `float rmse = tracking_result->rmse`
            ok = bool(res.tracking_success and rmse < self.max_reprojection_error_3d)
            if ok:                                                     # global = global[last] * T.inverse()
                self.global_poses[frame_id] = _mat4_mul_f32(self.global_poses[self.last_tracking_frame_id], I.mat4_inverse(res.T))
        self.tracking_success.append(ok)
        self.rmse.append(rmse)
        if ok:
            self.last_tracking_frame_id = frame_id
            self._last = (rgb, depth)
        return ok
