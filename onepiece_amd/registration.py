"""Host-side mirror of one_piece::registration (ICP) over the C-ABI (include/onepiece_hip.h).

Same names, argument meaning and error behaviour as /root/reference/src/Registration/ICP.h:13-26
and RegistrationResult.h:9-16.  No arithmetic happens here: the loop, the 6x6 solve and the Kabsch
finish run inside libonepiece_hip.so (kernels in csrc/icp_grid.hip / icp_iter.hip / icp.hip, host solve in csrc/host_math.hpp).
"""
import ctypes as C

import numpy as np

from . import _lib as L


class ICPParameter:
    """registration::ICPParameter (ICP.h:13-19)."""

    def __init__(self, max_iteration=30, threshold=0.2, scaling=1.0):
        self.max_iteration = max_iteration
        self.threshold = threshold
        self.scaling = scaling


class PointCloud:
    """The part of geometry::PointCloud the ICP path reads: points, normals, HasNormals()."""

    def __init__(self, points, normals=None):
        self.points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        self.normals = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)

    def HasNormals(self):
        return self.normals is not None and len(self.normals) == len(self.points) and len(self.points) > 0

    def EstimateNormals(self, radius=0.1, knn=30, device=0):
        """PointCloud::EstimateNormals (Geometry/PointCloud.cpp:102-144) on the GPU; fills self.normals."""
        out = np.zeros_like(self.points)
        L.check(L.load().op_estimate_normals(C.c_void_p(self.points.ctypes.data), len(self.points), float(radius), int(knn),
                                             L.OP_MEM_HOST, device, C.c_void_p(out.ctypes.data)))
        self.normals = out

    @staticmethod
    def LoadFromDepth(depth, camera, device=0):
        """PointCloud::LoadFromDepth (Geometry/PointCloud.cpp:72-100), computed on the GPU."""
        from .integration import _image_arg
        p, fmt, mem, _keep = _image_arg(depth, "depth")
        if mem != L.OP_MEM_HOST:
            raise ValueError("LoadFromDepth mirror takes host images")
        xyz = np.empty((camera.width * camera.height, 3), np.float32)
        n = C.c_size_t(0)
        L.check(L.load().op_points_from_depth(C.byref(camera), p, fmt, mem, device, C.c_void_p(xyz.ctypes.data), C.byref(n)))
        return PointCloud(xyz[:n.value].copy())


def LoadFromRGBD(rgb, depth, camera, device=0):
    """PointCloud::LoadFromRGBD (Geometry/PointCloud.cpp:17-48) on the GPU -> (PointCloud, colors [n,3])."""
    from .integration import _image_arg
    p, fmt, mem, _k0 = _image_arg(depth, "depth")
    c, _, memc, _k1 = _image_arg(rgb, "rgb")
    if mem != L.OP_MEM_HOST or memc != L.OP_MEM_HOST:
        raise ValueError("LoadFromRGBD mirror takes host images")
    npix = camera.width * camera.height
    xyz, col = np.empty((npix, 3), np.float32), np.empty((npix, 3), np.float32)
    n = C.c_size_t(0)
    L.check(L.load().op_points_from_rgbd(C.byref(camera), p, fmt, c, mem, device, C.c_void_p(xyz.ctypes.data), C.c_void_p(col.ctypes.data), C.byref(n)))
    return PointCloud(xyz[:n.value].copy()), col[:n.value].copy()


class RegistrationResult:
    """registration::RegistrationResult (RegistrationResult.h:9-16)."""

    def __init__(self):
        self.T = np.zeros((4, 4), np.float32)  # uninitialised in the reference when ICP refuses to run
        self.correspondence_set_index = np.zeros((0, 2), np.int32)
        self.correspondence_set = np.zeros((0, 2, 3), np.float32)
        self.rmse = float("nan")
        # extras (not in the reference struct): accumulated start_T and per-iteration trace
        self.last_T = None
        self.per_iter_inliers = None
        self.per_iter_T = None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _run(mode, source, target, init_T, icp_para, device, finish="reference", sums="reference_f32", ties="reference"):
    lib = L.load()
    res = RegistrationResult()
    # ICP.cpp:150-163: scaling != 1 or missing normals -> error line, default result
    if mode == L.OP_ICP_POINT_TO_PLANE and (not target.HasNormals() or icp_para.scaling != 1):
        print("[ERROR]::[ICPPointToPlane]::target point cloud need to have normals.")
        return res
    src = source.points
    tgt = target.points
    if icp_para.scaling != 1:  # ICP.cpp:37-43 (PointToPoint only)
        src = (src * np.float32(icp_para.scaling)).astype(np.float32)
        tgt = (tgt * np.float32(icp_para.scaling)).astype(np.float32)
    T0 = np.ascontiguousarray(np.eye(4) if init_T is None else init_T, np.float32).reshape(16)
    h = C.c_void_p()
    nrm = target.normals if mode == L.OP_ICP_POINT_TO_PLANE else None
    L.check(lib.op_icp_create(C.c_void_p(tgt.ctypes.data), C.c_void_p(nrm.ctypes.data) if nrm is not None else None,
                              len(tgt), float(icp_para.threshold), L.OP_MEM_HOST, device, C.byref(h)))
    try:
        L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_FINISH, {"reference": L.OP_ICP_FINISH_REFERENCE, "fp64": L.OP_ICP_FINISH_FP64}[finish]))
        L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_SUMS, {"fp64": L.OP_ICP_SUMS_FP64, "reference_f32": L.OP_ICP_SUMS_REFERENCE_F32}[sums]))
        L.check(lib.op_icp_set_option(h, L.OP_ICP_OPT_TIES, {"lowest_index": L.OP_ICP_TIES_LOWEST_INDEX, "reference": L.OP_ICP_TIES_REFERENCE}[ties]))
        L.check(lib.op_icp_set_source(h, C.c_void_p(src.ctypes.data), len(src), L.OP_MEM_HOST))
        out = L.IcpResult()
        iters = max(int(icp_para.max_iteration), 0)
        pairs = np.empty((max(len(src), 1), 2), np.int32)
        per_n = np.zeros(max(iters, 1), np.int32)
        per_T = np.zeros((max(iters, 1), 16), np.float32)
        L.check(lib.op_icp_run(h, mode, _fp(T0), iters, C.byref(out), pairs.ctypes.data_as(C.POINTER(C.c_int32)),
                               len(pairs), per_n.ctypes.data_as(C.POINTER(C.c_int32)), _fp(per_T)))
        tied, changed = C.c_uint64(0), C.c_uint64(0)
        L.check(lib.op_icp_tie_stats(h, C.byref(tied), C.byref(changed)))
        res.tie_stats = (int(tied.value), int(changed.value))   # ties="reference": tied queries over all passes, and how many got another partner
        redone = C.c_uint64(0)
        L.check(lib.op_icp_final_stats(h, C.byref(redone)))
        res.final_redecided = int(redone.value)                 # correspondences the final CountInliers had re-decided in the reference's tree (op_icp_final_stats)
    finally:
        lib.op_icp_destroy(h)
    n = int(out.n_inliers)
    res.T = np.array(out.T, np.float32).reshape(4, 4)
    if icp_para.scaling != 1:
        # ICP.cpp:207-221 un-scales the clouds before the final Kabsch: same R, translation / scaling
        res.T[:3, 3] = res.T[:3, 3] / np.float32(icp_para.scaling)
    res.last_T = np.array(out.last_T, np.float32).reshape(4, 4)
    res.rmse = float(out.rmse)
    res.correspondence_set_index = pairs[:n].copy()
    # ICP.cpp:215-219: pairs of (source point, target point) in the caller's (unscaled) units
    res.correspondence_set = np.stack([source.points[pairs[:n, 0]], target.points[pairs[:n, 1]]], axis=1)
    res.per_iter_inliers = per_n[:iters].copy()
    res.per_iter_T = per_T[:iters].reshape(-1, 4, 4).copy()
    return res


def PointToPlane(source, target, init_T=None, icp_para=None, device=0, finish="reference", sums="reference_f32", ties="reference"):
    """registration::PointToPlane (ICP.cpp:146-224).  finish / sums / ties: op_icp_set_option (include/onepiece_hip.h) --
    "reference" finish (default) forms RegistrationResult::T with the reference's sequential float32 sums;
    sums="reference_f32" (default since round 6) also sums every iteration's JTJ/JTr that way -- the mode within 1e-4 of the CPU path on EVERY
    pair (~0.6 k iterations/s); sums="fp64" is the fast order-free reduction (~24 k iterations/s; equals the CPU path with double sums);
    ties="reference" (default) pairs a source point whose nearest targets are exactly equidistant with the one the reference's kd-tree
    returns, ties="lowest_index" with the smallest index (no marking in the search: 2 % faster)."""
    return _run(L.OP_ICP_POINT_TO_PLANE, source, target, init_T, icp_para or ICPParameter(), device, finish, sums, ties)


def PointToPoint(source, target, init_T=None, icp_para=None, device=0, finish="reference", sums="reference_f32", ties="reference"):
    """registration::PointToPoint (ICP.cpp:31-107)."""
    return _run(L.OP_ICP_POINT_TO_POINT, source, target, init_T, icp_para or ICPParameter(), device, finish, sums, ties)


def Se3ToSE3(x):
    """geometry::Se3ToSE3 (Geometry/Geometry.cpp:9-13)."""
    x = np.ascontiguousarray(x, np.float32).reshape(6)
    T = np.empty(16, np.float32)
    L.check(L.load().op_se3_exp(_fp(x), _fp(T)))
    return T.reshape(4, 4)


def EstimateRigidTransformationPointToPlane(source, target, target_normal, inliers, device=0, sums="fp64"):
    """registration::EstimateRigidTransformationPointToPlane (ICP.h:24-26): one point-to-plane step over the
    given inliers (n x 2: source id, target id); `source` = the already transformed points."""
    src = np.ascontiguousarray(source, np.float32).reshape(-1, 3)
    tgt = np.ascontiguousarray(target, np.float32).reshape(-1, 3)
    nrm = np.ascontiguousarray(target_normal, np.float32).reshape(-1, 3)
    inl = np.ascontiguousarray(inliers, np.int32).reshape(-1, 2)
    T = np.empty(16, np.float32)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    L.check(L.load().op_estimate_rigid_point_to_plane_ex(vp(src), len(src), vp(tgt), vp(nrm), len(tgt), vp(inl), len(inl), L.OP_MEM_HOST, device,
                                                         {"fp64": L.OP_ICP_SUMS_FP64, "reference_f32": L.OP_ICP_SUMS_REFERENCE_F32}[sums], _fp(T)))
    return T.reshape(4, 4)


def EstimateRigidTransformation(correspondence_set, device=0, finish="reference"):
    """geometry::EstimateRigidTransformation (Geometry.cpp:107-151): Kabsch over (n, 2, 3) point pairs."""
    pairs = np.ascontiguousarray(correspondence_set, np.float32).reshape(-1, 6)
    T = np.empty(16, np.float32)
    L.check(L.load().op_estimate_rigid_transformation_ex(C.c_void_p(pairs.ctypes.data), len(pairs), L.OP_MEM_HOST, device,
                                                         {"reference": L.OP_ICP_FINISH_REFERENCE, "fp64": L.OP_ICP_FINISH_FP64}[finish], _fp(T)))
    return T.reshape(4, 4)
