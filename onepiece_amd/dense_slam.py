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
    """one_piece::DenseSlam (DenseSlam.h:41-128), tracking part.

    `pipeline` > 1 keeps that many frame pairs in flight on separate trackers (each owns a HIP stream):
    the pair (last tracked, current) only depends on earlier frames through the success flag, so pairs are
    enqueued speculatively assuming success and resolved in order; results, poses and the frame each pair
    was tracked against are exactly those of the sequential loop (a failed frame makes the pairs enqueued
    after it re-run against the right source).  `UpdateFrame` then returns None until `Finish()` /
    a later call has resolved the frame; `on_tracked(frame_id, rgb, depth, pose)` is called, in frame order,
    for every successfully tracked frame as soon as its pose is known."""

    def __init__(self, camera=None, device=0, pipeline=1, on_tracked=None):
        self.camera = camera if camera is not None else I.PinholeCamera()
        self.pipeline = max(1, int(pipeline))
        self._trackers = [O.Odometry(self.camera, device=device) for _ in range(self.pipeline)]
        self.rgbd_odometry = self._trackers[0]                         # DenseSlam.h:56
        self.global_poses = []
        self.tracking_success = []
        self.rmse = []
        self.last_tracking_frame_id = -1                               # DenseSlam.h:118
        self.max_reprojection_error_3d = 1.5                           # DenseSlam.h:122
        self.on_tracked = on_tracked
        self._frames = {}                                              # frame_id -> (rgb, depth), kept while needed
        self._inflight = []                                            # [(frame_id, source_id, tracker)] in frame order
        self._spec_last = -1                                           # speculative "last tracked frame"

    def SetSums(self, sums):
        """How every tracker sums an iteration's normal equations (Odometry.SetSums): "reference_f32" (default) -- the reference's own sequential float32
        order, the mode whose poses follow the CPU path step for step (north_star's 1e-4 bar) -- or "fp64", the fast order-free reduction."""
        for t in self._trackers:
            t.SetSums(sums)

    # -- resolution of one finished pair: DenseSlam.cpp:21-36
    def _resolve(self, frame_id, source_id, res):
        rmse = float(np.float32(res.rmse))                             # `float rmse = tracking_result->rmse`
        ok = bool(res.tracking_success and rmse < self.max_reprojection_error_3d)
        if ok:                                                         # global = global[last] * T.inverse()
            self.global_poses[frame_id] = _mat4_mul_f32(self.global_poses[source_id], I.mat4_inverse(res.T))
        self.tracking_success[frame_id] = ok
        self.rmse[frame_id] = rmse
        return ok

    def _commit(self, frame_id, ok):
        if ok:
            self.last_tracking_frame_id = frame_id
            if self.on_tracked is not None:
                rgb, depth = self._frames[frame_id]
                self.on_tracked(frame_id, rgb, depth, self.global_poses[frame_id])
        # frames older than the last tracked one can never be a source again
        for k in [k for k in self._frames if k < self.last_tracking_frame_id]:
            del self._frames[k]

    def _drain(self, keep):
        """Resolve in-flight pairs in order until at most `keep` remain."""
        while len(self._inflight) > keep:
            frame_id, source_id, trk = self._inflight.pop(0)
            ok = self._resolve(frame_id, source_id, trk.Wait(False))
            self._commit(frame_id, ok)
            if not ok:
                self._spec_last = self.last_tracking_frame_id          # the next pair tracks against the last GOOD frame
            if not ok and self._inflight:
                # the pairs enqueued after this frame speculated on its success: redo them sequentially against
                # the frame the reference would have used
                redo = [(f, t) for f, _s, t in self._inflight]
                for _f, _s, t in self._inflight:
                    t.Wait(False)
                self._inflight = []
                for f, t in redo:
                    src = self.last_tracking_frame_id
                    lrgb, ldepth = self._frames[src]
                    rgb, depth = self._frames[f]
                    r = t.DenseTracking(lrgb, rgb, ldepth, depth, np.eye(4, dtype=np.float32), 0, want_correspondences=False)
                    self._commit(f, self._resolve(f, src, r))
                self._spec_last = self.last_tracking_frame_id

    def UpdateFrame(self, rgb, depth):
        """DenseSlam::UpdateFrame (DenseSlam.cpp:8-36).  rgb (h,w,3) uint8, depth (h,w) f32 metres or
        u16 raw; numpy or CUDA torch tensors (device-resident frames are used in place).  Returns the
        tracking flag (pipeline == 1) or None when the frame is still in flight."""
        frame_id = len(self.global_poses)
        self.global_poses.append(np.eye(4, dtype=np.float32))
        self.tracking_success.append(None)
        self.rmse.append(0.0)
        self._frames[frame_id] = (rgb, depth)
        if frame_id == 0:
            self.tracking_success[0] = True
            self._commit(0, True)
            self._spec_last = 0
            return True
        self._drain(self.pipeline - 1)                                 # frees the tracker this pair will use
        src = self._spec_last
        lrgb, ldepth = self._frames[src]
        trk = self._trackers[frame_id % self.pipeline]
        trk.DenseTrackingEnqueue(lrgb, rgb, ldepth, depth, None, 0)
        self._inflight.append((frame_id, src, trk))
        self._spec_last = frame_id
        if self.pipeline == 1:
            self._drain(0)
            return self.tracking_success[frame_id]
        return None

    def Finish(self):
        """Resolve everything still in flight."""
        self._drain(0)
