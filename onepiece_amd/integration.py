"""Host-side mirror of one_piece::integration::CubeHandler over the C-ABI (include/onepiece_hip.h).

Same method names, argument meaning and error behaviour as the reference class
(/root/reference/src/Integration/CubeHandler.h:24-366) for the hot path, so the parity tests read
like the reference's own usage (example/ImageSequenceIntegration.cpp:27-42).  This module contains
no arithmetic of its own: every method forwards to libonepiece_hip.so, which fails loudly when no
GPU / no built library is present.

Images are numpy arrays (host, copied by the call) or torch CUDA tensors (used in place).  Poses
are 4x4 camera-to-world matrices (row-major numpy, i.e. `pose[r, c]` == Eigen `pose(r, c)`).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import Camera


def PinholeCamera(camera_type="OPEN3D_DATASET"):
    """camera::PinholeCamera presets (Camera/Camera.h:76-104); default ctor = OPEN3D_DATASET."""
    cam = Camera()
    L.check(L.load().op_camera_preset({"TUM_DATASET": 0, "OPEN3D_DATASET": 1}[camera_type], C.byref(cam)))
    return cam


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _image_arg(img, kind):
    """-> (pointer, fmt, mem, keepalive).  kind: 'depth' | 'rgb'.

    CUDA torch tensors are used in place by kernels on the LIBRARY's streams (each volume / tracker / ICP object
    owns one), not on torch's: the tensor's contents must be final when the kernels run, so torch's current stream is
    drained first if it still has work queued (L.torch_ready; nothing when it is idle).  A tensor produced on ANOTHER
    torch stream is the caller's to synchronise."""
    if hasattr(img, "data_ptr"):  # torch tensor
        if not img.is_cuda or not img.is_contiguous():
            raise ValueError("torch images must be contiguous CUDA tensors")
        import torch
        if kind == "depth":
            if img.dtype == torch.float32:
                fmt = L.OP_DEPTH_F32
            elif img.dtype in (torch.uint16, torch.int16):
                fmt = L.OP_DEPTH_U16
            else:
                raise ValueError("depth must be float32 (CV_32FC1) or uint16 (CV_16UC1)")
        else:
            if img.dtype != torch.uint8:
                raise ValueError("rgb must be uint8 (CV_8UC3)")
            fmt = 0
        L.torch_ready(img)
        return C.c_void_p(img.data_ptr()), fmt, L.OP_MEM_DEVICE, img
    a = np.ascontiguousarray(img)
    if kind == "depth":
        if a.dtype == np.uint16:
            fmt = L.OP_DEPTH_U16
        else:
            a = _f32(a)
            fmt = L.OP_DEPTH_F32
    else:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        fmt = 0
    return C.c_void_p(a.ctypes.data), fmt, L.OP_MEM_HOST, a


class CubeHandler:
    """integration::CubeHandler: the voxel-block-hashed TSDF volume, resident on one MI355X."""

    def __init__(self, camera=None, device=0, max_blocks=0):
        self._lib = L.load()
        self.camera = camera if camera is not None else PinholeCamera()
        self.device = device
        self._h = C.c_void_p()
        # defaults: VoxelResolution 0.01 (VoxelCube.h:27), truncation 0.1 (Integrator.h:23),
        # far 5.0 / near 0.5 (CubeHandler.h:363-364)
        self._res, self._trunc, self._far, self._near = 0.01, 0.1, 5.0, 0.5
        self._alive = []
        L.check(self._lib.op_volume_create(C.byref(self.camera), self._res, self._trunc, self._far,
                                           self._near, device, max_blocks, C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.op_volume_destroy(h)
            self._h = None

    # -- configuration (CubeHandler.h:36-39,137-144,339-346)
    def SetVoxelResolution(self, resolution):
        self._res = float(resolution)
        L.check(self._lib.op_volume_set_resolution(self._h, self._res))

    def SetTruncation(self, trunc):
        self._trunc = float(trunc)
        L.check(self._lib.op_volume_set_truncation(self._h, self._trunc))

    def SetCamera(self, camera):
        self.camera = camera
        L.check(self._lib.op_volume_set_camera(self._h, C.byref(camera)))

    def SetFarPlane(self, far):
        self._far = float(far)
        L.check(self._lib.op_volume_set_near_far(self._h, self._near, self._far))

    def SetNearPlane(self, near):
        self._near = float(near)
        L.check(self._lib.op_volume_set_near_far(self._h, self._near, self._far))

    def Clear(self):
        L.check(self._lib.op_volume_clear(self._h))

    def SetUpdateMode(self, mode):
        """Extension (op_volume_set_option / OP_VOLUME_OPT_UPDATE): "exact" (default; the reference's frame-by-frame running mean, voxels
        bit-identical to the CPU path) or "sum_form" (one weighted mean per batch of frames: same blocks and weights, sdf / colour to
        float rounding; a faster integrate kernel)."""
        value = {"exact": L.OP_VOLUME_UPDATE_EXACT, "sum_form": L.OP_VOLUME_UPDATE_SUM_FORM}[mode]
        L.check(self._lib.op_volume_set_option(self._h, L.OP_VOLUME_OPT_UPDATE, value))

    def SetSelectMode(self, mode):
        """Extension (op_volume_set_option / OP_VOLUME_OPT_SELECT): which form of the selection step batches take -- "auto" (default: in batches of
        >= 20 frames the frames record their selections and one pass claims every block once), "direct" (every frame claims its blocks itself) or an
        int n >= 1 (every batch of >= 2 frames records, except frames whose range has more than n super-blocks).  The selected set is the same in every mode."""
        value = {"auto": L.OP_VOLUME_SELECT_AUTO, "direct": L.OP_VOLUME_SELECT_DIRECT}.get(mode, mode)
        L.check(self._lib.op_volume_set_option(self._h, L.OP_VOLUME_OPT_SELECT, int(value)))

    # -- the hot path
    def ComputeBounding(self, depth, pose):
        """CubeHandler.cpp:116-145 -> (max_pos, min_pos, n_points_inside_frustum)."""
        p, fmt, mem, _keep = _image_arg(depth, "depth")
        pose = _f32(pose).reshape(16)
        mx, mn, n = np.empty(3, np.float32), np.empty(3, np.float32), C.c_size_t(0)
        L.check(self._lib.op_volume_compute_bounding(self._h, p, fmt, mem, _fp(pose), _fp(mx), _fp(mn), C.byref(n)))
        return mx, mn, int(n.value)

    def PrepareCubes(self, depth, pose, pose_inv=None, return_candidates=False):
        """CubeHandler.cpp:147-196 -> cube_id_list (n x 3 int32, reference loop order)."""
        p, fmt, mem, _keep = _image_arg(depth, "depth")
        pose = _f32(pose).reshape(16)
        pinv = _f32(pose_inv).reshape(16) if pose_inv is not None else None
        cap = 1 << 16
        while True:
            ids = np.empty((cap, 3), np.int32)
            n, nc = C.c_size_t(0), C.c_size_t(0)
            L.check(self._lib.op_volume_prepare_cubes(self._h, p, fmt, mem, _fp(pose), _fp(pinv) if pinv is not None else None,
                                                      _ip(ids), cap, C.byref(n), C.byref(nc)))
            if n.value <= cap:
                out = ids[:n.value].copy()
                return (out, int(nc.value)) if return_candidates else out
            cap = int(n.value)

    def IntegrateImage(self, depth, rgb, pose, pose_inv=None):
        """CubeHandler.cpp:197-210.  Asynchronous; any accessor below synchronises."""
        pd, fmt, mem, _k1 = _image_arg(depth, "depth")
        pr, _f, mem2, _k2 = _image_arg(rgb, "rgb")
        if mem != mem2:
            raise ValueError("depth and rgb must both be host arrays or both be device tensors")
        pose = _f32(pose).reshape(16)
        pinv = _f32(pose_inv).reshape(16) if pose_inv is not None else None
        # host buffers are borrowed for the call only (the library stages them before returning); device
        # tensors must stay alive until the next synchronising call -- keep a reference until then
        L.check(self._lib.op_volume_integrate(self._h, pd, fmt, pr, mem, _fp(pose), _fp(pinv) if pinv is not None else None))
        if mem == L.OP_MEM_DEVICE:
            self._alive.append((_k1, _k2))
            if len(self._alive) > 4096:
                self.Synchronize()

    def IntegrateCubes(self, depth, rgb, pose, cube_ids, pose_inv=None):
        """Integrator::IntegrateImage (Integrator.cpp:36-94) for a caller-chosen cube list (n x 3 int32): the frame is fused
        into exactly those cubes, allocated when absent, without PrepareCubes' selection.  Synchronous."""
        pd, fmt, mem, _k1 = _image_arg(depth, "depth")
        pr, _f, mem2, _k2 = _image_arg(rgb, "rgb")
        if mem != mem2:
            raise ValueError("depth and rgb must both be host arrays or both be device tensors")
        pose = _f32(pose).reshape(16)
        pinv = _f32(pose_inv).reshape(16) if pose_inv is not None else None
        ids = np.ascontiguousarray(cube_ids, np.int32).reshape(-1, 3)
        L.check(self._lib.op_volume_integrate_cubes(self._h, pd, fmt, pr, mem, _fp(pose), _fp(pinv) if pinv is not None else None, _ip(ids), len(ids)))

    def IntegrateFrame(self, rgbd, pose, pose_inv=None):
        """CubeHandler::IntegrateImage(const geometry::RGBDFrame&, pose) (CubeHandler.cpp:211-214)."""
        return self.IntegrateImage(rgbd.depth, rgbd.rgb, pose, pose_inv)

    def IntegrateSequence(self, depth, rgb, poses):
        """n frames resident on the device (torch tensors [n,h,w] / [n,h,w,3]); identical to n
        IntegrateImage calls in order (example/ImageSequenceIntegration.cpp:27-42)."""
        n = depth.shape[0]
        pd, fmt, mem, _k1 = _image_arg(depth, "depth")
        pr, _f, mem2, _k2 = _image_arg(rgb, "rgb")
        if mem != L.OP_MEM_DEVICE or mem2 != L.OP_MEM_DEVICE:
            raise ValueError("IntegrateSequence needs device-resident frames")
        poses = _f32(poses).reshape(n, 16)
        npx = self.camera.width * self.camera.height
        dstride = npx * (2 if fmt == L.OP_DEPTH_U16 else 4)
        L.check(self._lib.op_volume_integrate_sequence(self._h, pd, dstride, fmt, pr, npx * 3, _fp(poses), n))
        # a remainder of the call waits in the library's queue for the next frames: the tensors stay referenced until the next Synchronize
        self._alive.append((_k1, _k2))
        if len(self._alive) > 4096:
            self.Synchronize()

    def Flush(self):
        """launch the queued frames now (a batch smaller than 32) without waiting for them"""
        L.check(self._lib.op_volume_flush(self._h))

    def Synchronize(self):
        L.check(self._lib.op_volume_sync(self._h))
        self._alive = []

    def Stream(self):
        s = C.c_void_p()
        L.check(self._lib.op_volume_stream(self._h, C.byref(s)))
        return s.value

    def Stats(self):
        f, b, vis, upd = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        L.check(self._lib.op_volume_stats(self._h, C.byref(f), C.byref(b), C.byref(vis), C.byref(upd)))
        ln, br, vw, sc = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        L.check(self._lib.op_volume_stats_launches(self._h, C.byref(ln), C.byref(br), C.byref(vw), C.byref(sc)))
        return {"frames": f.value, "blocks_selected": b.value, "voxels_visited": vis.value, "voxels_updated": upd.value,
                "launches": ln.value, "blocks_read": br.value, "voxels_written": vw.value, "integrate_shader_cycles": sc.value}

    def GrowthStats(self):
        """capacity in blocks, pool growths, batches launched again after a growth (no synchronisation)"""
        m, g, r = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        L.check(self._lib.op_volume_growth_stats(self._h, C.byref(m), C.byref(g), C.byref(r)))
        return {"max_blocks": m.value, "grows": g.value, "replayed_batches": r.value}

    def ProfileEnable(self, sample_every=1):
        """HIP-event timing of K1/K2/K3 on the volume's own stream (measurement hook)."""
        L.check(self._lib.op_volume_profile_enable(self._h, int(sample_every)))

    def ProfileRead(self):
        """-> per-LAUNCH average ms of the three kernels + launches/frames covered by the samples."""
        ms = (C.c_double * 3)()
        n, nf = C.c_uint64(0), C.c_uint64(0)
        L.check(self._lib.op_volume_profile_read(self._h, ms, C.byref(n), C.byref(nf)))
        n, nf = int(n.value), int(nf.value)
        names = ("prepare_ms", "select_ms", "integrate_ms")
        return {"launches": n, "frames": nf, **{k: (ms[i] / n if n else float("nan")) for i, k in enumerate(names)}}

    # -- accessors (all synchronise)
    def BlockCount(self):
        n = C.c_size_t(0)
        L.check(self._lib.op_volume_block_count(self._h, C.byref(n)))
        return int(n.value)

    def HasCube(self, cube_id):
        r = C.c_int(0)
        L.check(self._lib.op_volume_has_cube(self._h, int(cube_id[0]), int(cube_id[1]), int(cube_id[2]), C.byref(r)))
        return bool(r.value)

    def GetCubeMap(self, sort=True):
        """CubeHandler.h:347-350 -> (keys n x 3 int32, voxels n x 512 x 5 float32 {sdf,w,c0,c1,c2}).
        The reference map's iteration order is unspecified; keys come back sorted (x, y, z)."""
        n = self.BlockCount()
        keys = np.empty((n, 3), np.int32)
        vox = np.empty((n, 512, 5), np.float32)
        got = C.c_size_t(0)
        L.check(self._lib.op_volume_download(self._h, _ip(keys), _fp(vox), n, C.byref(got)))
        if sort and n:
            order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
            keys, vox = keys[order], vox[order]
        return keys, vox

    def SetCubeMap(self, keys, voxels):
        """CubeHandler.h:351-356."""
        self.Clear()
        self.AddCubes(keys, voxels)

    def AddCubes(self, keys, voxels):
        keys = np.ascontiguousarray(keys, np.int32).reshape(-1, 3)
        voxels = _f32(voxels).reshape(-1, 512, 5)
        L.check(self._lib.op_volume_upload(self._h, _ip(keys), _fp(voxels), keys.shape[0]))

    def AddCube(self, cube_id):
        """CubeHandler::AddCube (CubeHandler.h:191-197): a default block (sdf 999, weight 0, colour -1) unless present."""
        if self.HasCube(cube_id):
            return
        vox = np.empty((1, 512, 5), np.float32)
        vox[..., 0], vox[..., 1], vox[..., 2:] = 999.0, 0.0, -1.0
        self.AddCubes(np.asarray(cube_id, np.int32).reshape(1, 3), vox)

    def Merge(self, another, trans=None):
        """CubeHandler::Merge (CubeHandler.h:145-177): mismatching resolution -> warning line, no change.
        With `trans`: Merge(*another.Transform(trans)) (:168-177)."""
        if trans is not None:
            if another.GetVoxelResolution() != self.GetVoxelResolution():
                print("[Warning]::[MergeVoxelHash]::Voxel resolution is not identical.")
                return
            another = another.Transform(trans)
        rc = self._lib.op_volume_merge(self._h, another._h)
        if rc == L.OP_ERR_MISMATCH:
            print(self._lib.op_last_error().decode())
            return
        L.check(rc)

    # -- resampling, point cloud, .map files (CubeHandler.h:40-128,242-338; CubeHandler.cpp:45-69)
    @classmethod
    def _wrap(cls, handle, camera, device):
        out = cls.__new__(cls)
        out._lib = L.load()
        out.camera, out.device, out._h = camera, device, handle
        out._alive = []
        res = C.c_float(0)
        L.check(out._lib.op_volume_resolution(handle, C.byref(res)))
        out._res, out._trunc, out._far, out._near = float(res.value), None, None, None
        return out

    def _transform(self, trans, nearest, max_blocks):
        T = _f32(trans).reshape(16)
        h = C.c_void_p()
        L.check(self._lib.op_volume_transform(self._h, _fp(T), None, 1 if nearest else 0, max_blocks, C.byref(h)))
        out = CubeHandler._wrap(h, self.camera, self.device)
        out._trunc, out._far, out._near = self._trunc, self._far, self._near
        return out

    def Transform(self, trans, max_blocks=0):
        """CubeHandler::Transform (CubeHandler.h:242-298): trilinear resampling -> new CubeHandler."""
        return self._transform(trans, False, max_blocks)

    def TransformNearest(self, trans, max_blocks=0):
        """CubeHandler::TransformNearest (CubeHandler.h:299-338).  Like the reference, the result keeps
        the DEFAULT voxel resolution 0.01 (c_para is not copied there)."""
        return self._transform(trans, True, max_blocks)

    def GetVoxelResolution(self):
        return self._res

    def GetPointCloud(self):
        """CubeHandler::GetPointCloud (CubeHandler.cpp:45-69) -> (points [n,3], colors [n,3])."""
        n = C.c_size_t(0)
        L.check(self._lib.op_volume_point_cloud(self._h, None, None, 0, C.byref(n)))
        xyz = np.empty((max(n.value, 1), 3), np.float32)
        col = np.empty((max(n.value, 1), 3), np.float32)
        L.check(self._lib.op_volume_point_cloud(self._h, _fp(xyz), _fp(col), n.value, C.byref(n)))
        return xyz[:n.value], col[:n.value]

    def ExtractTriangleMesh(self, tri_table, edge_pairs, only_block=None):
        """CubeHandler::ExtractTriangleMesh (CubeHandler.cpp:9-44) on the GPU -> (points [n,3], colors [n,3]); triangle k
        is vertices 3k..3k+2 (MarchingCube() pushes unshared vertices).  tri_table (256 x 16) / edge_pairs (12 x 2) are
        the caller's marching-cubes tables (the reference's MCLookTable / EdgeIndexPairs)."""
        tt = np.ascontiguousarray(tri_table, np.int32).reshape(256 * 16)
        ep = np.ascontiguousarray(edge_pairs, np.int32).reshape(24)
        ob = None if only_block is None else np.ascontiguousarray(only_block, np.int32).reshape(3)
        obp = None if ob is None else _ip(ob)
        n = C.c_size_t(0)
        L.check(self._lib.op_volume_extract_mesh(self._h, _ip(tt), _ip(ep), obp, None, None, 0, C.byref(n)))
        pts, col = np.empty((max(n.value, 1), 3), np.float32), np.empty((max(n.value, 1), 3), np.float32)
        if n.value:
            L.check(self._lib.op_volume_extract_mesh(self._h, _ip(tt), _ip(ep), obp, _fp(pts), _fp(col), n.value, C.byref(n)))
        return pts[:n.value].copy(), col[:n.value].copy()

    def GenerateMeshByCube(self, cube_id, tri_table, edge_pairs):
        """CubeHandler::GenerateMeshByCube (CubeHandler.cpp:70-114) for one block."""
        return self.ExtractTriangleMesh(tri_table, edge_pairs, only_block=cube_id)

    def Raycast(self, pose, camera=None):
        """Ray casting (north_star; no reference counterpart) -> (depth [h,w], normals [h,w,3],
        colors [h,w,3]); depth 0 = no hit."""
        cam = camera if camera is not None else self.camera
        pose = _f32(pose).reshape(16)
        d = np.zeros((cam.height, cam.width), np.float32)
        n = np.zeros((cam.height, cam.width, 3), np.float32)
        c = np.zeros((cam.height, cam.width, 3), np.float32)
        L.check(self._lib.op_volume_raycast(self._h, C.byref(cam), _fp(pose), _fp(d), _fp(n), _fp(c), L.OP_MEM_HOST))
        return d, n, c

    def RaycastDevice(self, pose, depth_ptr, normals_ptr=0, colors_ptr=0, camera=None):
        """The same with outputs that stay in HBM: device addresses (e.g. torch tensor.data_ptr()) of W*H / W*H*3 float32 buffers on the
        volume's device; 0 = not wanted.  Returns when the images are complete."""
        cam = camera if camera is not None else self.camera
        pose = _f32(pose).reshape(16)
        as_fp = lambda a: C.cast(C.c_void_p(int(a)), L._fp) if a else None
        L.check(self._lib.op_volume_raycast(self._h, C.byref(cam), _fp(pose), as_fp(depth_ptr), as_fp(normals_ptr), as_fp(colors_ptr), L.OP_MEM_DEVICE))

    def RaycastStats(self):
        """Measurement hook: what the last Raycast call did (op_volume_raycast_stats)."""
        v = [C.c_uint64(0) for _ in range(4)]
        L.check(self._lib.op_volume_raycast_stats(self._h, *[C.byref(x) for x in v]))
        return {"visible_blocks": v[0].value, "dropped_unloaded": v[1].value, "loaded_blocks": v[2].value, "marched_blocks": v[3].value}

    def SetRaycastPrune(self, on):
        """Extension (OP_VOLUME_OPT_RAYCAST_PRUNE): whether later views of the unchanged volume may drop blocks by what earlier views learnt
        about them (default on; the images are identical either way)."""
        L.check(self._lib.op_volume_set_option(self._h, L.OP_VOLUME_OPT_RAYCAST_PRUNE, 1 if on else 0))

    def WriteToFile(self, filename):
        L.check(self._lib.op_volume_write_file(self._h, str(filename).encode()))
        return True

    def ReadFromFile(self, filename):
        L.check(self._lib.op_volume_read_file(self._h, str(filename).encode(), 0))
        return True

    def ReadFromFileFloat(self, filename):
        L.check(self._lib.op_volume_read_file(self._h, str(filename).encode(), 1))
        return True

    def GetCubeID(self, point):
        """CubeHandler.h:185-189 / VoxelCube.h:63-74 (float floor, then floor-div by 8)."""
        p = _f32(point)
        pb = np.floor(p / np.float32(self._res)).astype(np.int64)
        return (pb >> 3).astype(np.int32)


def hash_key(x, y, z):
    """geometry::VoxelGridHasher (Geometry/Geometry.h:101-112)."""
    return int(L.load().op_hash_key(int(x), int(y), int(z)))


def mat4_inverse(m):
    m = _f32(m).reshape(16)
    out = np.empty(16, np.float32)
    L.check(L.load().op_mat4_inverse(_fp(m), _fp(out)))
    return out.reshape(4, 4)


def frustum_planes(camera, pose, far=5.0, near=0.5):
    pose = _f32(pose).reshape(16)
    out = np.empty(24, np.float32)
    L.check(L.load().op_frustum_planes(C.byref(camera), _fp(pose), far, near, _fp(out)))
    return out.reshape(6, 4)
