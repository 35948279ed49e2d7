"""ctypes wrapper around oracle/libonepiece_oracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from onepiece_amd/.  Parity status: see onepiece_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libonepiece_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("onepiece_oracle.c", "onepiece_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libonepiece_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int), ("height", C.c_int), ("depth_scale", C.c_float)]


class IcpResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("last_T", C.c_float * 16), ("rmse", C.c_double),
                ("n_inliers", C.c_size_t), ("iterations", C.c_int)]


class TrackLevel(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float)] + [
        (k, C.POINTER(C.c_float)) for k in ("source_color", "source_depth", "target_color", "target_depth",
                                            "target_color_dx", "target_color_dy", "target_depth_dx",
                                            "target_depth_dy")]


class TrackResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("rmse", C.c_double), ("n_correspondences", C.c_size_t),
                ("tracking_success", C.c_int), ("iterations", C.c_int)]


_lib = None
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_hash.restype = C.c_uint64
        L.orc_hash.argtypes = [C.c_int] * 3
        L.orc_mat4_inverse.argtypes = [_fp, _fp]
        L.orc_frustum_planes.argtypes = [C.POINTER(Camera), _fp, C.c_float, C.c_float, _fp]
        L.orc_compute_bounding.restype = C.c_size_t
        L.orc_compute_bounding.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_int, _fp, C.c_float,
                                           C.c_float, _fp, _fp]
        L.orc_get_sdf.restype = C.c_float
        L.orc_get_sdf.argtypes = [C.POINTER(Camera), _fp, _fp, C.c_void_p, C.c_int]
        L.orc_cube_id.argtypes = [C.c_float, _fp, C.POINTER(C.c_int)]
        L.orc_volume_create.restype = C.c_void_p
        L.orc_volume_create.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_volume_destroy.argtypes = [C.c_void_p]
        L.orc_volume_clear.argtypes = [C.c_void_p]
        L.orc_volume_block_count.restype = C.c_size_t
        L.orc_volume_block_count.argtypes = [C.c_void_p]
        L.orc_volume_prepare_cubes.restype = C.c_size_t
        L.orc_volume_prepare_cubes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _fp, _ip, C.c_size_t,
                                               C.POINTER(C.c_size_t)]
        L.orc_volume_integrate.restype = C.c_size_t
        L.orc_volume_integrate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, _fp,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_volume_export.restype = C.c_size_t
        L.orc_volume_export.argtypes = [C.c_void_p, _ip, _fp, C.c_size_t]
        L.orc_volume_import.argtypes = [C.c_void_p, _ip, _fp, C.c_size_t]
        L.orc_volume_merge.restype = C.c_int
        L.orc_volume_merge.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_volume_transform.restype = C.c_void_p
        L.orc_volume_transform.argtypes = [C.c_void_p, _fp, C.c_int]
        L.orc_volume_resolution.restype = C.c_float
        L.orc_volume_resolution.argtypes = [C.c_void_p]
        L.orc_volume_point_cloud.restype = C.c_size_t
        L.orc_volume_point_cloud.argtypes = [C.c_void_p, _fp, _fp, C.c_size_t]
        L.orc_volume_write_file.restype = C.c_int
        L.orc_volume_write_file.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_volume_read_file.restype = C.c_int
        L.orc_volume_read_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.orc_volume_extract_mesh.restype = C.c_size_t
        L.orc_volume_extract_mesh.argtypes = [C.c_void_p, _ip, _ip, _ip, _fp, _fp, C.c_size_t]
        L.orc_volume_raycast.argtypes = [C.c_void_p, C.POINTER(Camera), _fp, _fp, _fp, _fp]
        L.orc_load_from_depth.restype = C.c_size_t
        L.orc_load_from_depth.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_int, _fp]
        L.orc_estimate_normals.argtypes = [_fp, C.c_size_t, C.c_float, C.c_int, _fp]
        L.orc_knn_search.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, C.c_int, _ip, _fp, _ip]
        L.orc_knn_search.restype = None
        L.orc_se3_exp.argtypes = [_fp, _fp]
        L.orc_kabsch.argtypes = [_fp, C.c_size_t, _fp]
        L.orc_solve6.argtypes = [_fp, _fp, _fp]
        L.orc_p2plane_step.argtypes = [_fp, _fp, _fp, _ip, C.c_size_t, _fp, _fp, _fp]
        L.orc_icp.restype = C.c_int
        L.orc_icp.argtypes = [C.c_int, _fp, C.c_size_t, _fp, C.c_size_t, _fp, _fp, C.c_int,
                              C.c_double, C.POINTER(IcpResult), _ip, _ip, _fp]
        L.orc_mat3_inverse.argtypes = [_fp, _fp]
        L.orc_track_projection.argtypes = [_fp, _fp, _fp, _fp, _fp]
        L.orc_track_project_pixel.argtypes = [_fp, _fp, C.c_float, C.c_int, C.c_int, _fp, C.POINTER(C.c_int)]
        L.orc_pixel_correspondences.restype = C.c_size_t
        L.orc_pixel_correspondences.argtypes = [C.POINTER(TrackLevel), _fp, _ip]
        L.orc_track_normal_equations.argtypes = [C.POINTER(TrackLevel), _fp, _ip, C.c_size_t, C.c_int, _fp, _fp, _fp]
        L.orc_track_accumulate_rows.argtypes = [_fp, _fp, C.c_size_t, _fp, _fp, _fp]
        L.orc_ldlt_solve6.argtypes = [_fp, _fp, _fp]
        L.orc_track_iteration.restype = C.c_size_t
        L.orc_track_iteration.argtypes = [C.POINTER(TrackLevel), C.c_int, _fp, _ip]
        L.orc_dense_track.restype = C.c_int
        L.orc_dense_track.argtypes = [C.POINTER(TrackLevel), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                      _fp, C.POINTER(TrackResult), _ip, _ip, _fp]
        L.orc_prep_intensity.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp]
        L.orc_prep_depth_nan.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, _fp]
        L.orc_prep_blur3.argtypes = [_fp, C.c_int, C.c_int, _fp]
        L.orc_prep_pyrdown.argtypes = [_fp, C.c_int, C.c_int, _fp]
        L.orc_prep_sobel.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp]
        L.orc_bilateral_filter.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp]
        L.orc_normalize_intensity.argtypes = [_fp, _fp, C.c_int, C.c_int, _ip, C.c_size_t]
        L.orc_dense_tracking.restype = C.c_int
        L.orc_dense_tracking.argtypes = [C.POINTER(Camera), C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int, _fp, C.POINTER(TrackResult), _ip,
                                         C.POINTER(C.POINTER(C.c_float))]
        L.orc_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_fp):
    return a.ctypes.data_as(t)


def set_fusion_threads(n=None):
    """Threads for compute_bounding / prepare_cubes / integrate (None = min(32, host cores): measured best on the
    256-core GPU box, tests/tools/oracle_thread_scaling.py -- 73 ms/frame with 1 thread, 6.6 with 32, 31 with 128;
    1 = the reference's serial path and the default).  Results are bit-identical for any value; tests use it to run
    full-size sequences."""
    import os
    L = lib()
    L.orc_set_fusion_threads.restype = None
    L.orc_set_fusion_threads.argtypes = [C.c_int]
    L.orc_set_fusion_threads(int(n if n is not None else min(32, os.cpu_count() or 1)))


def make_camera(fx=514.817, fy=515.375, cx=318.771, cy=238.447, width=640, height=480,
                depth_scale=1000.0):
    """Default = OPEN3D_DATASET preset (Camera/Camera.h:94-104)."""
    return Camera(fx, fy, cx, cy, width, height, depth_scale)


def _depth_arg(depth):
    depth = np.ascontiguousarray(depth)
    if depth.dtype == np.uint16:
        return depth, 1
    return _f32(depth), 0


def hash_key(x, y, z):
    return int(lib().orc_hash(int(x), int(y), int(z)))


def mat4_inverse(m):
    m = _f32(m).reshape(16)
    out = np.empty(16, np.float32)
    lib().orc_mat4_inverse(_p(m), _p(out))
    return out.reshape(4, 4)


def frustum_planes(cam, pose, far=5.0, near=0.5):
    pose = _f32(pose).reshape(16)
    out = np.empty(24, np.float32)
    lib().orc_frustum_planes(C.byref(cam), _p(pose), far, near, _p(out))
    return out.reshape(6, 4)


def compute_bounding(cam, depth, pose, far=5.0, near=0.5):
    d, u16 = _depth_arg(depth)
    pose = _f32(pose).reshape(16)
    mx, mn = np.empty(3, np.float32), np.empty(3, np.float32)
    n = lib().orc_compute_bounding(C.byref(cam), d.ctypes.data, u16, _p(pose), far, near, _p(mx), _p(mn))
    return mx, mn, int(n)


def se3_exp(x):
    x = _f32(x).reshape(6)
    T = np.empty(16, np.float32)
    lib().orc_se3_exp(_p(x), _p(T))
    return T.reshape(4, 4)


def kabsch(src, tgt):
    pairs = _f32(np.concatenate([_f32(src).reshape(-1, 3), _f32(tgt).reshape(-1, 3)], axis=1))
    T = np.empty(16, np.float32)
    lib().orc_kabsch(_p(pairs), pairs.shape[0], _p(T))
    return T.reshape(4, 4)


def solve6(JTJ, JTr):
    a, b = _f32(JTJ).reshape(36), _f32(JTr).reshape(6)
    x = np.empty(6, np.float32)
    lib().orc_solve6(_p(a), _p(b), _p(x))
    return x


def load_from_depth(cam, depth):
    d, u16 = _depth_arg(depth)
    xyz = np.empty((cam.width * cam.height, 3), np.float32)
    n = lib().orc_load_from_depth(C.byref(cam), d.ctypes.data, u16, _p(xyz))
    return xyz[:n].copy()


def estimate_normals(points, radius=0.1, knn=30):
    """PointCloud::EstimateNormals (PointCloud.cpp:102-144); normals are defined up to sign."""
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros_like(pts)
    lib().orc_estimate_normals(_p(pts), len(pts), radius, knn, _p(out))
    return out


def knn_search(target, queries, k=1):
    """The oracle's own kd-tree searches (k = 1: ICP.cpp:69,189; k > 1: PointCloud.cpp:120) -> (index [nq, k], squared distance [nq, k], found [nq])."""
    t, q = _f32(target).reshape(-1, 3), _f32(queries).reshape(-1, 3)
    idx = np.empty((len(q), k), np.int32)
    d2 = np.empty((len(q), k), np.float32)
    found = np.empty(len(q), np.int32)
    lib().orc_knn_search(_p(t), len(t), _p(q), len(q), k, _p(idx, _ip), _p(d2), _p(found, _ip))
    return idx, d2, found


class Volume:
    """Mirror of integration::CubeHandler restricted to the hot path (CubeHandler.h:24-366)."""

    def __init__(self, cam=None, voxel_res=0.01, trunc=0.1, far=5.0, near=0.5):
        self.cam = cam or make_camera()
        self._h = lib().orc_volume_create(C.byref(self.cam), voxel_res, trunc, far, near)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:  # _lib may already be gone at interpreter exit
            _lib.orc_volume_destroy(self._h)
            self._h = None

    def clear(self):
        lib().orc_volume_clear(self._h)

    def block_count(self):
        return int(lib().orc_volume_block_count(self._h))

    def prepare_cubes(self, depth, pose):
        d, u16 = _depth_arg(depth)
        pose = _f32(pose).reshape(16)
        cap = 1 << 17
        while True:
            ids = np.empty((cap, 3), np.int32)
            nc = C.c_size_t(0)
            n = lib().orc_volume_prepare_cubes(self._h, d.ctypes.data, u16, _p(pose), _p(ids, _ip),
                                               cap, C.byref(nc))
            if n <= cap:
                return ids[:n].copy(), int(nc.value)
            cap = int(n)

    def integrate(self, depth, rgb, pose):
        d, u16 = _depth_arg(depth)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        pose = _f32(pose).reshape(16)
        nv, nu = C.c_uint64(0), C.c_uint64(0)
        n = lib().orc_volume_integrate(self._h, d.ctypes.data, u16, rgb.ctypes.data, _p(pose),
                                       C.byref(nv), C.byref(nu))
        return int(n), int(nv.value), int(nu.value)

    def export(self, sort=True):
        n = self.block_count()
        keys = np.empty((n, 3), np.int32)
        vox = np.empty((n, 512, 5), np.float32)
        lib().orc_volume_export(self._h, _p(keys, _ip), _p(vox), n)
        if sort and n:
            order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
            keys, vox = keys[order], vox[order]
        return keys, vox

    def load(self, keys, vox):
        keys = np.ascontiguousarray(keys, np.int32)
        vox = _f32(vox)
        lib().orc_volume_import(self._h, _p(keys, _ip), _p(vox), keys.shape[0])

    def merge(self, other):
        return int(lib().orc_volume_merge(self._h, other._h))

    def transform(self, T, nearest=False):
        """CubeHandler::Transform / TransformNearest -> new Volume."""
        T = _f32(T).reshape(16)
        out = Volume.__new__(Volume)
        out.cam = self.cam
        out._h = lib().orc_volume_transform(self._h, _p(T), 1 if nearest else 0)
        return out

    def resolution(self):
        return float(lib().orc_volume_resolution(self._h))

    def point_cloud(self):
        n = lib().orc_volume_point_cloud(self._h, None, None, 0)
        xyz = np.empty((max(n, 1), 3), np.float32)
        col = np.empty((max(n, 1), 3), np.float32)
        lib().orc_volume_point_cloud(self._h, _p(xyz), _p(col), n)
        return xyz[:n], col[:n]

    def extract_mesh(self, tri_table, edge_pairs, only_block=None):
        """ExtractTriangleMesh / GenerateMeshByCube with caller-supplied tables -> (points [n,3], colors [n,3]);
        triangle k = vertices 3k..3k+2."""
        tt = np.ascontiguousarray(tri_table, np.int32).reshape(256 * 16)
        ep = np.ascontiguousarray(edge_pairs, np.int32).reshape(24)
        ob = None if only_block is None else np.ascontiguousarray(only_block, np.int32).reshape(3)
        obp = None if ob is None else _p(ob, _ip)
        n = lib().orc_volume_extract_mesh(self._h, _p(tt, _ip), _p(ep, _ip), obp, None, None, 0)
        pts, col = np.empty((max(n, 1), 3), np.float32), np.empty((max(n, 1), 3), np.float32)
        lib().orc_volume_extract_mesh(self._h, _p(tt, _ip), _p(ep, _ip), obp, _p(pts), _p(col), n)
        return pts[:n].copy(), col[:n].copy()

    def raycast(self, pose, cam=None):
        """No reference counterpart: CPU restatement of op_volume_raycast's definition."""
        cam = cam or self.cam
        pose = _f32(pose).reshape(16)
        d = np.zeros((cam.height, cam.width), np.float32)
        n = np.zeros((cam.height, cam.width, 3), np.float32)
        c = np.zeros((cam.height, cam.width, 3), np.float32)
        lib().orc_volume_raycast(self._h, C.byref(cam), _p(pose), _p(d), _p(n), _p(c))
        return d, n, c

    def write_file(self, path):
        return int(lib().orc_volume_write_file(self._h, str(path).encode()))

    def read_file(self, path, legacy_float=False):
        return int(lib().orc_volume_read_file(self._h, str(path).encode(), 1 if legacy_float else 0))


def icp(src, tgt, tgt_normals=None, init_T=None, max_iter=30, threshold=0.2, point_to_plane=True):
    """registration::PointToPlane / PointToPoint (ICP.cpp:146-224 / :31-107)."""
    src, tgt = _f32(src).reshape(-1, 3), _f32(tgt).reshape(-1, 3)
    nrm = _f32(tgt_normals).reshape(-1, 3) if tgt_normals is not None else None
    T0 = _f32(np.eye(4) if init_T is None else init_T).reshape(16)
    res = IcpResult()
    pairs = np.empty((max(len(src), 1), 2), np.int32)
    per_n = np.zeros(max(max_iter, 1), np.int32)
    per_T = np.zeros((max(max_iter, 1), 16), np.float32)
    rc = lib().orc_icp(1 if point_to_plane else 0, _p(src), len(src), _p(tgt), len(tgt),
                       _p(nrm) if nrm is not None else None, _p(T0), max_iter, threshold,
                       C.byref(res), _p(pairs, _ip), _p(per_n, _ip), _p(per_T))
    if rc:
        return None
    n = int(res.n_inliers)
    return {"T": np.array(res.T, np.float32).reshape(4, 4),
            "last_T": np.array(res.last_T, np.float32).reshape(4, 4),
            "rmse": float(res.rmse), "pairs": pairs[:n].copy(),
            "per_iter_inliers": per_n[:max_iter].copy(),
            "per_iter_T": per_T[:max_iter].reshape(-1, 4, 4).copy()}


# ---- dense RGB-D tracker (Odometry/) ---------------------------------------------------------
TRACK_IMAGES = ("source_color", "source_depth", "target_color", "target_depth", "target_color_dx",
                "target_color_dy", "target_depth_dx", "target_depth_dy")


def _track_levels(levels):
    """levels: list of dicts {width,height,fx,fy,cx,cy + the 8 TRACK_IMAGES as (h,w) float32}."""
    arr = (TrackLevel * len(levels))()
    keep = []
    for k, lv in enumerate(levels):
        arr[k].width, arr[k].height = int(lv["width"]), int(lv["height"])
        arr[k].fx, arr[k].fy, arr[k].cx, arr[k].cy = (float(lv[c]) for c in ("fx", "fy", "cx", "cy"))
        for name in TRACK_IMAGES:
            a = _f32(lv[name])
            assert a.shape == (arr[k].height, arr[k].width), (name, a.shape)
            keep.append(a)
            setattr(arr[k], name, _p(a))
    return arr, keep


def mat3_inverse(m):
    m = _f32(m).reshape(9)
    out = np.empty(9, np.float32)
    lib().orc_mat3_inverse(_p(m), _p(out))
    return out.reshape(3, 3)


def track_projection(cam4, T):
    cam4, T = _f32(cam4).reshape(4), _f32(T).reshape(16)
    ki, krk, kt = np.empty(9, np.float32), np.empty(9, np.float32), np.empty(3, np.float32)
    lib().orc_track_projection(_p(cam4), _p(T), _p(ki), _p(krk), _p(kt))
    return ki.reshape(3, 3), krk.reshape(3, 3), kt


def track_project_pixel(krk, kt, d, j, i):
    krk, kt = _f32(krk).reshape(9), _f32(kt).reshape(3)
    uv = np.empty(3, np.float32)
    ut = (C.c_int * 2)()
    lib().orc_track_project_pixel(_p(krk), _p(kt), float(d), int(j), int(i), _p(uv), ut)
    return uv, (int(ut[0]), int(ut[1]))


def pixel_correspondences(level, T):
    """ComputeCorrespondencePixelWise (DenseOdometryFunction.cpp:72-128): (n,4) {v_s,u_s,v_t,u_t}."""
    arr, keep = _track_levels([level])
    T = _f32(T).reshape(16)
    corr = np.empty((arr[0].width * arr[0].height, 4), np.int32)
    n = lib().orc_pixel_correspondences(arr, _p(T), _p(corr, _ip))
    return corr[:n].copy()


def track_normal_equations(level, T, corr, term=0):
    arr, keep = _track_levels([level])
    T = _f32(T).reshape(16)
    corr = np.ascontiguousarray(corr, np.int32).reshape(-1, 4)
    JTJ, JTr, r2 = np.empty(36, np.float32), np.empty(6, np.float32), np.empty(1, np.float32)
    lib().orc_track_normal_equations(arr, _p(T), _p(corr, _ip), len(corr), int(term), _p(JTJ), _p(JTr), _p(r2))
    return JTJ.reshape(6, 6), JTr, float(r2[0])


def track_accumulate_rows(J, r):
    J, r = _f32(J).reshape(-1, 6), _f32(r).reshape(-1)
    JTJ, JTr, r2 = np.empty(36, np.float32), np.empty(6, np.float32), np.empty(1, np.float32)
    lib().orc_track_accumulate_rows(_p(J), _p(r), len(r), _p(JTJ), _p(JTr), _p(r2))
    return JTJ.reshape(6, 6), JTr, r2[0]


def ldlt_solve6(JTJ, JTr):
    JTJ, JTr = _f32(JTJ).reshape(36), _f32(JTr).reshape(6)
    x = np.empty(6, np.float32)
    lib().orc_ldlt_solve6(_p(JTJ), _p(JTr), _p(x))
    return x


def dense_track(levels, iters=(4, 8, 16), full_w=None, full_h=None, term=0, init_T=None):
    """Odometry::MultiScaleComputing (Odometry.cpp:621-687); levels[0] = full resolution."""
    arr, keep = _track_levels(levels)
    full_w = int(levels[0]["width"] if full_w is None else full_w)
    full_h = int(levels[0]["height"] if full_h is None else full_h)
    T0 = _f32(np.eye(4) if init_T is None else init_T).reshape(16)
    it = (C.c_int * len(levels))(*[int(v) for v in iters])
    total = int(sum(iters))
    res = TrackResult()
    cap = max(int(lv["width"]) * int(lv["height"]) for lv in levels)
    corr = np.empty((cap, 4), np.int32)
    per_n = np.zeros(max(total, 1), np.int32)
    per_T = np.zeros((max(total, 1), 16), np.float32)
    lib().orc_dense_track(arr, len(levels), it, full_w, full_h, int(term), _p(T0), C.byref(res),
                          _p(corr, _ip), _p(per_n, _ip), _p(per_T))
    n = int(res.n_correspondences)
    return {"T": np.array(res.T, np.float32).reshape(4, 4), "rmse": float(res.rmse),
            "tracking_success": bool(res.tracking_success), "iterations": int(res.iterations),
            "pixel_correspondences": corr[:n].copy(),
            "per_iter_count": per_n[:res.iterations].copy(),
            "per_iter_T": per_T[:res.iterations].reshape(-1, 4, 4).copy()}


# ---- tracker image preparation (own definitions, NOT pinned to OpenCV; see onepiece_oracle.h) ----
PYRAMID_KINDS = ("color", "depth", "color_dx", "color_dy", "depth_dx", "depth_dy")


def prep_intensity(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    out = np.empty((h, w), np.float32)
    lib().orc_prep_intensity(C.c_void_p(rgb.ctypes.data), w, h, _p(out))
    return out


def prep_depth_nan(depth, depth_scale=1000.0):
    d, is16 = _depth_arg(depth)
    h, w = d.shape
    out = np.empty((h, w), np.float32)
    lib().orc_prep_depth_nan(C.c_void_p(d.ctypes.data), is16, float(depth_scale), w, h, _p(out))
    return out


def _img_op(fn, img, *extra, half=False):
    img = _f32(img)
    h, w = img.shape
    out = np.empty((h // 2, w // 2) if half else (h, w), np.float32)
    fn(_p(img), w, h, *extra, _p(out))
    return out


def bilateral_filter(depth, d=7, sigma_color=0.03, sigma_space=4.5, depth_scale=1000.0):
    """tool::BilateralFilter after tool::ConvertDepthTo32F (float32 metres or uint16 / depth_scale in)."""
    a = np.ascontiguousarray(depth)
    assert a.dtype in (np.float32, np.uint16) and a.ndim == 2
    out = np.empty(a.shape, np.float32)
    lib().orc_bilateral_filter(C.c_void_p(a.ctypes.data), int(a.dtype == np.uint16), float(depth_scale), a.shape[1], a.shape[0],
                               int(d), float(sigma_color), float(sigma_space), out.ctypes.data_as(_fp))
    return out


def prep_blur3(img):
    return _img_op(lib().orc_prep_blur3, img)


def prep_pyrdown(img):
    return _img_op(lib().orc_prep_pyrdown, img, half=True)


def prep_sobel(img, axis):
    return _img_op(lib().orc_prep_sobel, img, int(axis))


def dense_tracking(cam, src_rgb, tgt_rgb, src_depth, tgt_depth, iters=(4, 8, 16), term=0, init_T=None, want_pyramids=False):
    """Odometry::DenseTracking, cv::Mat overload (Odometry.cpp:463-524), incl. the image preparation."""
    sr, tr = np.ascontiguousarray(src_rgb, np.uint8), np.ascontiguousarray(tgt_rgb, np.uint8)
    sd, is16 = _depth_arg(src_depth)
    td, is16b = _depth_arg(tgt_depth)
    assert is16 == is16b
    n = len(iters)
    it = (C.c_int * n)(*[int(v) for v in iters])
    T0 = _f32(np.eye(4) if init_T is None else init_T).reshape(16)
    res = TrackResult()
    corr = np.empty((cam.width * cam.height, 4), np.int32)
    pyr = (C.POINTER(C.c_float) * (12 * n))() if want_pyramids else None
    lib().orc_dense_tracking(C.byref(cam), n, it, C.c_void_p(sr.ctypes.data), C.c_void_p(tr.ctypes.data),
                             C.c_void_p(sd.ctypes.data), C.c_void_p(td.ctypes.data), is16, int(term), _p(T0),
                             C.byref(res), _p(corr, _ip), pyr)
    out = {"T": np.array(res.T, np.float32).reshape(4, 4), "rmse": float(res.rmse),
           "tracking_success": bool(res.tracking_success), "iterations": int(res.iterations),
           "pixel_correspondences": corr[:int(res.n_correspondences)].copy()}
    if want_pyramids:
        P = {}
        for f, fname in enumerate(("source", "target")):
            for k, kname in enumerate(PYRAMID_KINDS):
                for l in range(n):
                    w, h = cam.width >> l, cam.height >> l
                    ptr = pyr[(f * 6 + k) * n + l]
                    P[(fname, kname, l)] = np.ctypeslib.as_array(ptr, shape=(h, w)).copy()
                    lib().orc_free(ptr)
        out["pyramids"] = P
    return out
