"""Host-side mirror of one_piece::odometry::Odometry's DENSE tracker over the C-ABI.

Same names, argument meaning and result fields as /root/reference/src/Odometry/Odometry.h:29-37
(DenseTrackingResult) and :82-88,100-120,168-170 (DenseTracking, SetMultiScale,
CreatePyramidCameras, multi_scale_level, iter_count_per_level).  Nothing is computed here: the image
preparation, the coarse-to-fine Gauss-Newton loop -- projective association with the reference's
source-indexed "z-buffer", the hybrid / photo / depth Jacobians, the 6x6 LDL^T solve and the pose
update -- all run inside libonepiece_hip.so (csrc/odometry.hip, odometry_prep.hip, odometry_emit.hip).

Two entry points, as in the reference:
  * `DenseTracking(source_color, target_color, source_depth, target_depth, T0, term)`: from raw frames,
    end to end on the GPU.  The preparation stage (cvtColor / GaussianBlur / pyrDown / Sobel) is OpenCV
    in the reference and not vendored by it, so the library implements OpenCV's published definitions;
    that stage is outside the pinned parity boundary.
  * `MultiScaleComputing(levels, T0, term)`: pyramids supplied by the caller (e.g. the reference's own
    OpenCV pyramids from an RGBDFrame) -- the pinned boundary.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .integration import PinholeCamera

TRACK_IMAGES = ("source_color", "source_depth", "target_color", "target_depth", "target_color_dx",
                "target_color_dy", "target_depth_dx", "target_depth_dy")

class DenseTrackingResult:
    """odometry::DenseTrackingResult (Odometry.h:29-37)."""

    def __init__(self):
        self.T = np.eye(4, dtype=np.float32)
        self.pixel_correspondence_set = np.zeros((0, 4), np.int32)   # {v_s, u_s, v_t, u_t}
        self.correspondence_set = np.zeros((0, 2, 3), np.float32)    # (source xyz, target xyz)
        self.rmse = 1e6
        self.tracking_success = False
        self.iterations = 0
        self.per_iter_count = None
        self.per_iter_T = None


def make_level(cam, source_color, source_depth, target_color, target_depth, target_color_dx, target_color_dy,
               target_depth_dx, target_depth_dy):
    lv = {"width": int(cam.width), "height": int(cam.height), "fx": float(cam.fx), "fy": float(cam.fy),
          "cx": float(cam.cx), "cy": float(cam.cy)}
    for k, v in zip(TRACK_IMAGES, (source_color, source_depth, target_color, target_depth, target_color_dx,
                                   target_color_dy, target_depth_dx, target_depth_dy)):
        lv[k] = v
    return lv


def _image_ptr(a, h, w, keep):
    """numpy float32 (h,w) -> host pointer; torch CUDA float32 tensor -> device pointer."""
    if isinstance(a, np.ndarray) or not hasattr(a, "data_ptr"):
        a = np.ascontiguousarray(a, np.float32)
        if a.shape != (h, w):
            raise ValueError("image shape %s != (%d, %d)" % (a.shape, h, w))
        keep.append(a)
        return a.ctypes.data, L.OP_MEM_HOST
    import torch
    if a.dtype != torch.float32 or not a.is_contiguous() or tuple(a.shape) != (h, w):
        raise ValueError("device images must be contiguous float32 (h, w) tensors")
    keep.append(a)
    L.torch_ready(a)
    return a.data_ptr(), (L.OP_MEM_DEVICE if a.is_cuda else L.OP_MEM_HOST)


def _levels_arg(levels):
    arr = (L.TrackLevel * len(levels))()
    keep, mems = [], set()
    for k, lv in enumerate(levels):
        arr[k].width, arr[k].height = int(lv["width"]), int(lv["height"])
        arr[k].fx, arr[k].fy, arr[k].cx, arr[k].cy = (float(lv[c]) for c in ("fx", "fy", "cx", "cy"))
        for name in TRACK_IMAGES:
            ptr, mem = _image_ptr(lv[name], arr[k].height, arr[k].width, keep)
            setattr(arr[k], name, ptr)
            mems.add(mem)
    if len(mems) != 1:
        raise ValueError("all pyramid images must live in the same memory space")
    return arr, mems.pop(), keep


class Odometry:
    """odometry::Odometry, dense part (Odometry.h:38-175)."""

    def __init__(self, camera=None, device=0):
        self.camera = camera if camera is not None else PinholeCamera()
        self.multi_scale_level = 3                     # Odometry.h:168
        self.iter_count_per_level = [4, 8, 16]         # Odometry.h:170
        self.device = device
        self._h = C.c_void_p()
        L.check(L.load().op_tracker_create(device, C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and L is not None and getattr(L, "_lib", None) is not None:
            L._lib.op_tracker_destroy(h)
            self._h = None

    def SetCamera(self, camera):
        self.camera = camera

    def SetSums(self, sums="reference_f32"):
        """op_tracker_set_option(OP_TRACK_OPT_SUMS): "reference_f32" (the default of new trackers since round 6) -- every iteration's
        Jacobian rows summed sequentially in float32 in raster order as the reference does, by one wave on the device: every pair within 1e-4 of the CPU
        path --, "fp64" (device reduction, no host round trip: ~30 x faster, 20 of 23 pairs of the bench's chain within 1e-4) or
        "reference_f32_host", the same sums on one host thread (the slow cross-check of the device sums)."""
        L.check(L.load().op_tracker_set_option(self._h, L.OP_TRACK_OPT_SUMS,
                                               {"fp64": L.OP_TRACK_SUMS_FP64, "reference_f32": L.OP_TRACK_SUMS_REFERENCE_F32,
                                                "reference_f32_host": L.OP_TRACK_SUMS_REFERENCE_F32_HOST}[sums]))

    def SetMultiScale(self, layer_count):
        """Odometry.h:100-104: resize(layer_count, 4) keeps existing entries, pads with 4."""
        self.multi_scale_level = int(layer_count)
        cur = list(self.iter_count_per_level)[:layer_count]
        self.iter_count_per_level = cur + [4] * (layer_count - len(cur))

    def CreatePyramidCameras(self):
        """Odometry.h:110-120 / Camera.h:38-42: fx,fy,cx,cy halved (float), width/height halved (int)."""
        cams = []
        for i in range(self.multi_scale_level):
            if i == 0:
                c = self.camera
                cams.append(L.Camera(c.fx, c.fy, c.cx, c.cy, c.width, c.height, c.depth_scale))
            else:
                p = cams[-1]
                cams.append(L.Camera(p.fx / 2.0, p.fy / 2.0, p.cx / 2.0, p.cy / 2.0,   # halving is exact in float
                                     p.width // 2, p.height // 2, p.depth_scale))
        return cams

    # -- the GPU path --
    def ComputeCorrespondencePixelWise(self, level, T=None):
        """DenseOdometryFunction.cpp:72-128 on the GPU: (n,4) int32 {v_s,u_s,v_t,u_t}, raster order."""
        arr, mem, keep = _levels_arg([level])
        T = np.ascontiguousarray(np.eye(4) if T is None else T, np.float32).reshape(16)
        cap = arr[0].width * arr[0].height
        out = np.empty((cap, 4), np.int32)
        n = C.c_size_t(0)
        L.check(L.load().op_tracker_correspondences(self._h, arr, T.ctypes.data_as(L._fp), mem, C.c_void_p(out.ctypes.data),
                                                    cap, C.byref(n)))
        return out[:n.value].copy()

    def MultiScaleComputing(self, levels, T=None, term_type=0, want_correspondences=True, want_log=False):
        """Odometry::MultiScaleComputing + result assembly (Odometry.cpp:621-687, :600-607)."""
        arr, mem, keep = _levels_arg(levels)
        if len(levels) != len(self.iter_count_per_level):
            raise ValueError("levels and iter_count_per_level differ in length")
        T0 = np.ascontiguousarray(np.eye(4) if T is None else T, np.float32).reshape(16)
        iters = np.asarray(self.iter_count_per_level, np.int32)
        total = int(iters.sum())
        res = L.TrackResult()
        cap = max(int(lv["width"]) * int(lv["height"]) for lv in levels) if want_correspondences else 0
        pix = np.empty((cap, 4), np.int32) if want_correspondences else None
        pts = np.empty((cap, 6), np.float32) if want_correspondences else None
        per_n = np.zeros(max(total, 1), np.int32) if want_log else None
        per_T = np.zeros((max(total, 1), 16), np.float32) if want_log else None
        vp = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None
        L.check(L.load().op_tracker_track(self._h, arr, len(levels), iters.ctypes.data_as(L._ip), int(self.camera.width),
                                          int(self.camera.height), int(term_type), T0.ctypes.data_as(L._fp), mem, C.byref(res),
                                          vp(pix), vp(pts), cap, vp(per_n), vp(per_T)))
        out = DenseTrackingResult()
        out.T = np.array(res.T, np.float32).reshape(4, 4)
        out.rmse = float(res.rmse)
        out.tracking_success = bool(res.tracking_success)
        out.iterations = int(res.iterations)
        out.n_correspondences = int(res.n_correspondences)
        if want_correspondences:
            n = out.n_correspondences
            out.pixel_correspondence_set = pix[:n].copy()
            out.correspondence_set = pts[:n].reshape(-1, 2, 3).copy()
        if want_log:
            out.per_iter_count = per_n[:out.iterations].copy()
            out.per_iter_T = per_T[:out.iterations].reshape(-1, 4, 4).copy()
        return out

    def DenseTrackingEnqueue(self, source_color, target_color, source_depth, target_depth, initial_T=None, term_type=0,
                             want_point_correspondences=False):
        """First half of DenseTracking: puts the whole computation on this tracker's stream and returns.
        Several Odometry objects can thus work on independent frame pairs at once; `Wait()` returns the result.
        Device-resident images must stay alive until then (a reference is kept here)."""
        from .integration import _image_arg
        ps, fs, ms, k0 = _image_arg(source_depth, "depth")
        pt, ft, mt, k1 = _image_arg(target_depth, "depth")
        cs, _, mcs, k2 = _image_arg(source_color, "rgb")
        ct, _, mct, k3 = _image_arg(target_color, "rgb")
        if len({ms, mt, mcs, mct}) != 1 or fs != ft:
            raise ValueError("all four images must share memory space and depth format")
        T0 = np.ascontiguousarray(np.eye(4) if initial_T is None else initial_T, np.float32).reshape(16)
        iters = np.asarray(self.iter_count_per_level, np.int32)
        if len(iters) != self.multi_scale_level:
            raise ValueError("iter_count_per_level must have multi_scale_level entries")
        self._inflight = (k0, k1, k2, k3)
        L.check(L.load().op_tracker_dense_tracking_enqueue(self._h, C.byref(self.camera), self.multi_scale_level,
                                                           iters.ctypes.data_as(L._ip), cs, ct, ps, pt, fs, T0.ctypes.data_as(L._fp),
                                                           int(term_type), ms, 1 if want_point_correspondences else 0))

    def Wait(self, want_correspondences=False):
        """Second half: synchronise this tracker's stream and return the DenseTrackingResult."""
        res = L.TrackResult()
        cap = int(self.camera.width) * int(self.camera.height) if want_correspondences else 0
        pix = np.empty((cap, 4), np.int32) if want_correspondences else None
        pts = np.empty((cap, 6), np.float32) if want_correspondences else None
        vp = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None
        L.check(L.load().op_tracker_wait(self._h, C.byref(res), vp(pix), vp(pts), cap))
        self._inflight = None
        out = DenseTrackingResult()
        out.T = np.array(res.T, np.float32).reshape(4, 4)
        out.rmse = float(res.rmse)
        out.tracking_success = bool(res.tracking_success)
        out.iterations = int(res.iterations)
        out.n_correspondences = int(res.n_correspondences)
        if want_correspondences:
            out.pixel_correspondence_set = pix[:out.n_correspondences].copy()
            out.correspondence_set = pts[:out.n_correspondences].reshape(-1, 2, 3).copy()
        return out

    def DenseTracking(self, source_color, target_color, source_depth, target_depth, initial_T=None, term_type=0,
                      want_correspondences=True):
        """Odometry::DenseTracking, cv::Mat overload (Odometry.cpp:463-524), end to end on the GPU
        (op_tracker_dense_tracking): image preparation, NormalizeIntensity, pyramids, MultiScaleComputing.
        colour: (h,w,3) uint8; depth: (h,w) float32 metres or uint16 raw; numpy or CUDA torch tensors."""
        self.DenseTrackingEnqueue(source_color, target_color, source_depth, target_depth, initial_T, term_type,
                                  want_point_correspondences=want_correspondences)
        return self.Wait(want_correspondences)

    def ReadPyramid(self, frame, kind, level):
        """Image prepared by the last DenseTracking call (frame 0 source / 1 target; kind index into
        (colour, depth, colour_dx, colour_dy, depth_dx, depth_dy))."""
        w, h = int(self.camera.width) >> level, int(self.camera.height) >> level
        out = np.empty((h, w), np.float32)
        L.check(L.load().op_tracker_read_pyramid(self._h, int(frame), int(kind), int(level), out.ctypes.data_as(L._fp), out.size))
        return out

    def PreparedLevels(self):
        """The pyramids the last DenseTracking call built on the GPU, as `levels` for MultiScaleComputing
        (downloaded copies)."""
        cams = self.CreatePyramidCameras()
        levels = []
        for l in range(self.multi_scale_level):
            levels.append(make_level(cams[l], self.ReadPyramid(0, 0, l), self.ReadPyramid(0, 1, l), self.ReadPyramid(1, 0, l),
                                     self.ReadPyramid(1, 1, l), self.ReadPyramid(1, 2, l), self.ReadPyramid(1, 3, l),
                                     self.ReadPyramid(1, 4, l), self.ReadPyramid(1, 5, l)))
        return levels
