"""The C-ABI library loads and exports every symbol include/onepiece_hip.h declares; its host-side
geometry matches the oracle bit for bit; every compute entry point fails loudly without a GPU.
No GPU compute is attempted here (-m "not gpu")."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "onepiece_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(op_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(hip):
    lib = hip.load()
    names = _declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libonepiece_hip.so does not export %s" % n
    assert set(names) == set(hip.SIGNATURES), "python binding table and header disagree: %s" % (set(names) ^ set(hip.SIGNATURES))
    assert lib.op_abi_version() == 1


def test_library_is_self_contained_hip_code():
    """The product .so must not link the oracle or torch; it contains gfx950 code objects."""
    import subprocess
    so = os.path.join(ROOT, "onepiece_amd", "libonepiece_hip.so")
    ldd = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "torch" not in ldd and "libamdhip64" in ldd
    blob = open(so, "rb").read()
    assert b"gfx950" in blob and b"k_integrate" in blob and b"k_icp_iter" in blob


def test_host_geometry_matches_oracle_bitwise(hip, oracle):
    from onepiece_amd import integration as I, registration as R
    rng = np.random.default_rng(7)
    cam_h, cam_o = I.PinholeCamera(), oracle.make_camera()
    for k in range(300):
        T = np.eye(4, dtype=np.float32)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        T[:3, :3] = q.astype(np.float32)
        T[:3, 3] = (rng.normal(size=3) * 2).astype(np.float32)
        if k % 4 == 3:
            T = rng.normal(size=(4, 4)).astype(np.float32)
        assert np.array_equal(I.mat4_inverse(T).view(np.uint32), oracle.mat4_inverse(T).view(np.uint32))
        if k % 4 != 3:
            assert np.array_equal(I.frustum_planes(cam_h, T).view(np.uint32), oracle.frustum_planes(cam_o, T).view(np.uint32))
        x = (rng.normal(size=6) * (1e-6 if k % 5 == 0 else 0.05)).astype(np.float32)
        assert np.abs(R.Se3ToSE3(x) - oracle.se3_exp(x)).max() <= 1e-7
        key = rng.integers(-5000, 5000, size=3)
        assert I.hash_key(*key) == oracle.hash_key(*key)


def test_host_inverse_matches_eigen_golden(hip):
    from onepiece_amd import integration as I
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "eigen_golden.json")))
    for case in g["inverse"]:
        m = np.array(case["m"], np.uint32).view(np.float32).reshape(4, 4)
        assert np.array_equal(I.mat4_inverse(m).reshape(16).view(np.uint32), np.array(case["inv"], np.uint32))


def test_camera_presets(hip):
    from onepiece_amd import integration as I
    o3d, tum = I.PinholeCamera(), I.PinholeCamera("TUM_DATASET")
    assert (o3d.width, o3d.height, o3d.depth_scale) == (640, 480, 1000.0)
    assert abs(o3d.fx - 514.817) < 1e-3 and abs(tum.cy - 255.3) < 1e-3 and tum.depth_scale == 5000.0


def test_no_gpu_means_loud_failure_not_fallback(hip):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from onepiece_amd import integration as I, registration as R
    with pytest.raises(hip.OnePieceHipError) as e:
        I.CubeHandler()
    assert e.value.code == hip.OP_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    pts = np.zeros((10, 3), np.float32)
    with pytest.raises(hip.OnePieceHipError):
        R.PointToPoint(R.PointCloud(pts), R.PointCloud(pts))
    with pytest.raises(hip.OnePieceHipError):
        R.PointCloud.LoadFromDepth(np.ones((480, 640), np.float32), I.PinholeCamera())


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/: the package, the C++ class
    surface (host/), the C-ABI header and the drivers (examples/, tools/) never include, import, link or load it.  The one
    run-time binding in the product is RCCL (csrc/merge_rccl.hip; distributed.py's RcclCommunicator makes the ncclComm_t from the same
    library), and its dlopen names nothing else."""
    for top in ("onepiece_amd", "host", "include", "examples", "tools"):
        for dirpath, _d, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".sh")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', text), f
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                    assert "libonepiece_oracle" not in text and "onepiece_oracle.h" not in text, f
                    code = re.sub(r"/\*.*?\*/|//[^\n]*", "", text, flags=re.S) if not f.endswith(".py") else text
                    if "dlopen(" in code or "CDLL(" in code:
                        assert f in ("merge_rccl.hip", "_lib.py", "distributed.py"), f
                        for name in re.findall(r'"([^"]*\.so[^"]*)"', text):
                            assert "rccl" in name or "onepiece_hip" in name, (f, name)


def test_pixel_rounding_fast_path_equals_double_formula(hip):
    """csrc/px_round.hpp: the fp32/integer pixel rounding the kernels use must give the same int as
    the reference's double formula (Integrator.cpp:20-21) wherever the bounds test can pass, and
    must reject exactly the same values elsewhere.  Host build of the same inline functions."""
    import subprocess, tempfile, textwrap
    src = textwrap.dedent(r'''
        #include <cstdio>
        #include <cstdint>
        #include <cstring>
        #include "px_round.hpp"
        int main() {
            float cs[12] = {318.771f, 238.447f, 318.6f, 255.3f, 79.69275f, 59.61175f, 0.0f, 756.24762f, 530.00418f, -3.25f, 1.0f, 99999.25f};
            uint32_t st = 12345u; long bad = 0, n = 0;
            auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st; };
            if (px_split(99999.5f).exact || px_split(0.3f).exact) { printf("inexact splits must be refused\n"); return 2; }
            for (int rep = 0; rep < 2; ++rep)
            for (float& c : cs) {
                if (rep == 1) c = 1.0f + (float)(rnd() >> 8) * (2000.0f / 16777216.0f); // random principal points
                const PxSplit sp = px_split(c);
                if (!sp.exact) { if (rep == 1) { n += 3500000 / 2; continue; } printf("split of %g not exact\n", c); return 2; }
                for (long it = 0; it < 3500000 / 2; ++it) {
                    float a; const int mode = it % 8;
                    if (mode < 3) { uint32_t b = rnd(); memcpy(&a, &b, 4); }
                    else if (mode < 6) { a = ((int)(rnd() >> 8) % 4000000 - 2000000) / 1000.0f; a += (float)(rnd() >> 9) * 1.1920929e-10f; }
                    else { int k = (int)(rnd() >> 20) % 1400 - 700; float base = (float)((double)k - (double)c - 0.5 + (mode == 7 ? 1.0 : 0.0));
                           int32_t bi; memcpy(&bi, &base, 4); bi += (int)(rnd() >> 28) - 8; memcpy(&a, &bi, 4); }
                    const int r0 = px_round_dp(a, c), r1 = px_round_sp(a, sp); ++n;
                    const bool rej0 = r0 < 0 || r0 >= 100000000, rej1 = r1 < 0 || r1 >= 100000000;
                    if (!(r0 == r1 || (rej0 && rej1))) { if (bad < 5) printf("c=%g a=%.9g dp=%d sp=%d\n", c, a, r0, r1); ++bad; }
                }
            }
            printf("%ld %ld\n", n, bad);
            return bad != 0;
        }
    ''')
    with tempfile.TemporaryDirectory() as td:
        cpp = os.path.join(td, "t.cpp")
        open(cpp, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "onepiece_amd", "csrc"), cpp, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout
        n, bad = map(int, out.stdout.split()[-2:])
        assert n == 2 * 12 * (3500000 // 2) and bad == 0
    # and the library's own hook agrees with numpy's double evaluation on pixel-scale values
    lib = hip.load()
    rng = np.random.default_rng(3)
    for c in (318.771, 238.447):
        a = (rng.uniform(-700, 1400, 20000)).astype(np.float32)
        ref = np.trunc(a.astype(np.float64) + 0.5 + np.float64(np.float32(c))).astype(np.int64)
        got = np.array([lib.op_debug_project_px(float(x), float(np.float32(c)), 1) for x in a[:4000]])
        assert np.array_equal(got, ref[:4000])
    assert lib.op_debug_project_px(float("nan"), 318.771, 1) == -2**31 == lib.op_debug_project_px(float("nan"), 318.771, 0)
    assert lib.op_debug_project_px(-0.6, 0.0, 1) == 0 == lib.op_debug_project_px(-0.6, 0.0, 0)   # (-1,0) truncates to 0


def test_in_image_pixel_rounding_equals_double_formula(hip):
    """csrc/px_round.hpp px_pixel_sp -- what k_select / k_integrate use: "is the pixel inside the image, and which one" from
    two range compares on a = f*X/Z and two threshold compares on its fraction.  Against the reference's double formula
    (Integrator.cpp:20-21 + the bounds test of :63) on 5e7 values: every bit pattern class, pixel-scale values, values within
    a few ulp of every rounding threshold and of both image borders, for the presets' principal points, random ones and
    ragged image sizes; principal points whose thresholds are not exact in fp32 must be refused (the kernels then take the
    double formula)."""
    import subprocess, tempfile, textwrap
    src = textwrap.dedent(r'''
        #include <cstdio>
        #include <cstdint>
        #include <cstring>
        #include "px_round.hpp"
        int main() {
            const float cs[10] = {318.771f, 238.447f, 318.6f, 255.3f, 79.69275f, 59.61175f, 756.24762f, 530.00418f, 1.25f, 0.0f};
            const int ext[10] = {640, 480, 640, 480, 160, 120, 1440, 1080, 3, 1};
            uint32_t st = 2463534242u; long bad = 0, n = 0, refused = 0;
            auto rnd = [&]() { st ^= st << 13; st ^= st >> 17; st ^= st << 5; return st; };
            if (px_axis(318.5f, 640).exact || px_axis(0.3f, 640).exact || px_axis(3.0e6f, 640).exact) { printf("inexact thresholds must be refused\n"); return 2; }
            for (int rep = 0; rep < 3; ++rep)
            for (int k = 0; k < 10; ++k) {
                float c = cs[k]; int extent = ext[k];
                if (rep == 1) { c = 1.0f + (float)(rnd() >> 8) * (2000.0f / 16777216.0f); extent = 1 + (int)(rnd() % 4000u); }
                if (rep == 2) { c = (float)(int)(rnd() % 2000u) + 0.25f * (float)(rnd() % 4u) + 0.125f; extent = 1 + (int)(rnd() % 4000u); }
                const PxAxis ax = px_axis(c, extent);
                if (!ax.exact) { ++refused; continue; }
                for (long it = 0; it < 1700000; ++it) {
                    float a; const int mode = it % 8;
                    if (mode < 2) { uint32_t b = rnd(); memcpy(&a, &b, 4); }                                   // any bit pattern
                    else if (mode < 5) { a = ((int)(rnd() >> 8) % 6000000 - 3000000) / 1000.0f; a += (float)(rnd() >> 9) * 1.1920929e-10f; }
                    else {                                                                                      // around thresholds / borders
                        const int kk = (int)(rnd() >> 18) % (extent + 40) - 20;
                        const double K = (double)c + 0.5;
                        double tgt = mode == 5 ? (double)kk - K : (mode == 6 ? -1.0 - K + (rnd() & 1) * (double)(extent + 1) : (double)kk - K + 1.0e-6 * ((int)(rnd() % 5u) - 2));
                        float base = (float)tgt; int32_t bi; memcpy(&bi, &base, 4); bi += (int)(rnd() >> 28) - 8; memcpy(&a, &bi, 4);
                    }
                    int u = -7;
                    const bool in_sp = px_pixel_sp(a, ax, u);
                    const int r = px_round_dp(a, c);
                    const bool in_dp = r >= 0 && r < extent;
                    ++n;
                    if (in_sp != in_dp || (in_dp && u != r)) { if (bad < 5) printf("c=%.9g extent=%d a=%.9g dp=%d sp=%d/%d\n", c, extent, a, r, (int)in_sp, u); ++bad; }
                }
            }
            printf("%ld %ld %ld\n", n, refused, bad);
            return bad != 0;
        }
    ''')
    with tempfile.TemporaryDirectory() as td:
        cpp = os.path.join(td, "t.cpp")
        open(cpp, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "onepiece_amd", "csrc"), cpp, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout
        n, refused, bad = map(int, out.stdout.split()[-3:])
        assert bad == 0 and n >= 30 * 1700000 - refused * 1700000 and n > 3.0e7


def test_tracker_host_math_matches_eigen_golden(hip):
    """op_track_projection (K*R*K^-1, K*t) bit-exact and op_ldlt_solve6 within float rounding of Eigen's
    ldlt().solve() -- the host/device-shared arithmetic of csrc/odometry.hip, checked without a GPU."""
    lib = hip.load()
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "odometry_golden.json")))
    f = lambda b: np.array(b, np.uint32).view(np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for c in g["projective"]:
        cam, T = f(c["cam"]).copy(), f(c["T"]).copy()
        krk, kt = np.empty(9, np.float32), np.empty(3, np.float32)
        assert lib.op_track_projection(fp(cam), fp(T), fp(krk), fp(kt)) == 0
        assert np.array_equal(krk.view(np.uint32), np.array(c["KRK_inv"], np.uint32))
        assert np.array_equal(kt.view(np.uint32), np.array(c["Kt"], np.uint32))
    for c in g["gauss_newton"]:
        JTJ, JTr = f(c["JTJ"]).astype(np.float64), f(c["JTr"]).astype(np.float64)
        x, xr = np.empty(6, np.float32), f(c["delta"])
        assert lib.op_ldlt_solve6(dp(JTJ), dp(JTr), fp(x)) == 0
        assert np.linalg.norm(x - xr) <= 1e-5 * np.linalg.norm(xr) + 1e-7
    # singular system: zero pivots -> zero components (Eigen's pseudo-inverse of D)
    x = np.ones(6, np.float32)
    assert lib.op_ldlt_solve6(dp(np.zeros(36)), dp(np.zeros(6)), fp(x)) == 0 and not x.any()


def test_tracker_refuses_without_gpu(hip):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert hip.load().op_tracker_create(0, C.byref(h)) == hip.OP_ERR_NO_DEVICE  # no CPU fallback
    assert b"no CPU fallback" in hip.load().op_last_error()


def test_only_test_infrastructure_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/: the developer
    tools under tools/ and the examples stay oracle-free (validation scripts that need it live in tests/tools/)."""
    for sub in ("tools", "examples", "host", "include"):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".sh", ".cpp", ".hpp", ".h")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert not re.search(r"\b(from|import)\s+oracle\b|oracle/|onepiece_oracle|from helpers import track_levels", text), os.path.join(dirpath, f)


def test_frustum_entry_points_and_argument_checks(hip, oracle):
    """op_frustum_from_camera / _from_vectors (integration::Frustum's arithmetic, host side, no device needed): planes bit-equal to the
    oracle's restatement of Frustum.cpp, the two entry points agree with each other, corners optional, null arguments and an oversized
    camera are refused with a message."""
    import ctypes as C
    lib = hip.load()
    fp = C.POINTER(C.c_float)
    cam_o = oracle.make_camera()
    cam_h = hip.Camera(cam_o.fx, cam_o.fy, cam_o.cx, cam_o.cy, cam_o.width, cam_o.height, cam_o.depth_scale)
    rng = np.random.default_rng(11)
    for _ in range(50):
        T = np.eye(4, dtype=np.float32)
        q, _r = np.linalg.qr(rng.normal(size=(3, 3)))
        T[:3, :3] = q.astype(np.float32); T[:3, 3] = (rng.normal(size=3) * 2).astype(np.float32)
        planes, corners = np.zeros(24, np.float32), np.zeros(24, np.float32)
        hip.check(lib.op_frustum_from_camera(C.byref(cam_h), T.ctypes.data_as(fp), 5.0, 0.5, planes.ctypes.data_as(fp), corners.ctypes.data_as(fp)))
        assert np.array_equal(planes.view(np.uint32), oracle.frustum_planes(cam_o, T).reshape(24).view(np.uint32))
        # the same frustum from its vectors (Frustum.cpp:14-23: right = R col 0, up = -R col 1, forward = R col 2, position = t)
        fwd, pos, right, up = (np.ascontiguousarray(v, np.float32) for v in (T[:3, 2], T[:3, 3], T[:3, 0], -T[:3, 1]))
        aspect = np.float32(np.float32(cam_o.fy * np.float32(cam_o.width)) / np.float32(cam_o.fx * np.float32(cam_o.height)))
        fov = np.float32(np.arctan2(np.float64(cam_o.cy), np.float64(cam_o.fy)) + np.arctan2(np.float64(np.float32(cam_o.height) - np.float32(cam_o.cy)), np.float64(cam_o.fy)))
        p2, c2 = np.zeros(24, np.float32), np.zeros(24, np.float32)
        hip.check(lib.op_frustum_from_vectors(fwd.ctypes.data_as(fp), pos.ctypes.data_as(fp), right.ctypes.data_as(fp), up.ctypes.data_as(fp), 5.0, 0.5,
                                              float(fov), float(aspect), p2.ctypes.data_as(fp), c2.ctypes.data_as(fp)))
        assert np.array_equal(p2.view(np.uint32), planes.view(np.uint32)) and np.array_equal(c2.view(np.uint32), corners.view(np.uint32))
        p3 = np.zeros(24, np.float32)
        hip.check(lib.op_frustum_from_camera(C.byref(cam_h), T.ctypes.data_as(fp), 5.0, 0.5, p3.ctypes.data_as(fp), None))   # corners are optional
        assert np.array_equal(p3, planes)
    assert lib.op_frustum_from_camera(C.byref(cam_h), None, 5.0, 0.5, planes.ctypes.data_as(fp), None) == hip.OP_ERR_INVALID
    assert lib.op_frustum_from_vectors(None, pos.ctypes.data_as(fp), right.ctypes.data_as(fp), up.ctypes.data_as(fp), 5.0, 0.5, 1.0, 1.0, planes.ctypes.data_as(fp), None) == hip.OP_ERR_INVALID
    big = hip.Camera(500.0, 500.0, 4096.0, 4096.0, 8192, 8192, 1000.0)      # 2^26 pixels: beyond what the fusion kernels address
    assert lib.op_frustum_from_camera(C.byref(big), T.ctypes.data_as(fp), 5.0, 0.5, planes.ctypes.data_as(fp), None) == hip.OP_ERR_INVALID
    assert b"16777216 pixels" in lib.op_last_error()     # 2^24 with batches of up to 32 frames (2^23 in the 64-frame build)
    got = C.c_float(0)
    assert lib.op_get_sdf(C.byref(cam_h), None, T.ctypes.data_as(fp), None, C.c_void_p(planes.ctypes.data), hip.OP_DEPTH_F32, C.byref(got)) == hip.OP_ERR_INVALID


def test_new_volume_entry_points_refuse_without_a_volume(hip):
    """The entry points added in round 3 check their handle like the others (no device work without a valid volume)."""
    import ctypes as C
    lib = hip.load()
    u = C.c_uint64(0)
    assert lib.op_volume_stats_launches(None, C.byref(u), C.byref(u), C.byref(u), C.byref(u)) != 0
    assert lib.op_volume_growth_stats(None, C.byref(u), C.byref(u), C.byref(u)) != 0
    assert lib.op_volume_integrate_cubes(None, None, 0, None, 0, None, None, None, 0) != 0
    assert lib.op_volume_flush(None) != 0
    n = C.c_size_t(0)
    st = hip.MergeStats()
    assert lib.op_volume_merge_rccl_stats(None, None, 0, C.byref(n), C.byref(st)) == hip.OP_ERR_INVALID and st.ranks == 0


def test_select_option_constants_match_the_header():
    """OP_VOLUME_OPT_SELECT and its values: _lib.py mirrors include/onepiece_hip.h."""
    import re
    from onepiece_amd import _lib as L
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "onepiece_hip.h")).read()
    for name in ("OP_VOLUME_OPT_SELECT", "OP_VOLUME_SELECT_AUTO", "OP_VOLUME_SELECT_DIRECT"):
        m = re.search(r"#define\s+%s\s+(-?\d+)" % name, text)
        assert m and int(m.group(1)) == getattr(L, name), name


def test_loading_the_library_changes_nothing_in_the_process_environment():
    """A drop-in libone_piece replacement must not edit its host's environment or bind libraries by environment variable: rounds 1-4 set
    GPU_MAX_HW_QUEUES from a load-time constructor.  Now that is op_runtime_configure's job -- an explicit call -- and op_runtime_set_option /
    op_runtime_set_rccl_library replace the ONEPIECE_* variables the library used to read."""
    import subprocess, sys
    lib = os.path.join(ROOT, "onepiece_amd", "libonepiece_hip.so")
    code = (
        "import ctypes, os\n"
        "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
        "assert libc.getenv(b'GPU_MAX_HW_QUEUES') is None\n"
        "env_before = dict(os.environ)\n"
        "L = ctypes.CDLL(%r)\n"
        "assert libc.getenv(b'GPU_MAX_HW_QUEUES') is None, 'loading the library set GPU_MAX_HW_QUEUES'\n"
        "q = ctypes.c_int(0); assert L.op_runtime_hw_queues(ctypes.byref(q)) == 0 and q.value == 4\n"
        "assert L.op_runtime_configure(8) == 0 and libc.getenv(b'GPU_MAX_HW_QUEUES') == b'8'\n"
        "assert L.op_runtime_hw_queues(ctypes.byref(q)) == 0 and q.value == 8\n"
        "L.op_runtime_set_option.argtypes = [ctypes.c_int, ctypes.c_longlong]\n"
        "assert L.op_runtime_set_option(0, 1) == 0 and L.op_runtime_set_option(0, 7) != 0 and L.op_runtime_set_option(99, 0) != 0\n"
        "assert L.op_runtime_set_option(5, 0) == 0 and L.op_runtime_set_option(5, 64 << 30) == 0 and L.op_runtime_set_option(5, -1) != 0\n"
        "assert L.op_runtime_set_rccl_library(b'/nonexistent/librccl.so') == 0 and L.op_runtime_set_rccl_library(None) == 0\n"
        "print('ok')\n" % lib)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES" and not k.startswith("ONEPIECE_")}
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stdout + run.stderr
    # and no source of the product reads an ONEPIECE_* variable or calls setenv outside op_runtime_configure
    import re
    hits = []
    for d in ("onepiece_amd/csrc",):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if not f.endswith((".hip", ".hpp")):
                continue
            text = open(os.path.join(ROOT, d, f)).read()
            code_only = re.sub(r"//[^\n]*", "", text)
            hits += ["%s: %s" % (f, m) for m in re.findall(r'getenv\("ONEPIECE_[A-Z_]*"\)|__attribute__\(\(constructor\)\)', code_only)]
            if f != "volume.hip":
                hits += ["%s: setenv" % f for _ in re.findall(r"\bsetenv\(", code_only)]
    assert not hits, hits


@pytest.mark.parametrize("name", ["uniform_1nn", "surface_1nn", "lattice_ties_1nn", "quantised_1nn"])
def test_host_tie_tree_equals_the_real_nanoflann(name, tmp_path):
    """onepiece_amd/csrc/nn_tree.hpp -- the host-side search the ICP path falls back to for queries with exactly equidistant nearest
    candidates (OP_ICP_TIES_REFERENCE) -- returns the real nanoflann's index for every query of the fixture, ties included."""
    import subprocess
    from helpers import nanoflann_case
    c = nanoflann_case(name)
    exe = tmp_path / "nn_tree_check"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-msse4.2", "-ffp-contract=off", "-I", os.path.join(ROOT, "onepiece_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "nn_tree_check.cpp"), "-o", str(exe)])
    blob = tmp_path / "case.bin"
    with open(blob, "wb") as f:
        f.write(np.array([len(c["target"]), len(c["query"])], np.int32).tobytes())
        f.write(np.ascontiguousarray(c["target"], np.float32).tobytes())
        f.write(np.ascontiguousarray(c["query"], np.float32).tobytes())
        f.write(np.ascontiguousarray(c["index"][:, 0], np.int32).tobytes())
    run = subprocess.run([str(exe), str(blob)], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr


KD_CASES = ["uniform_1nn", "lattice_ties_1nn", "quantised_1nn", "quantised_knn30", "uniform_knn30", "tiny_knn30", "lattice_knn30", "hist33_knn5",
            "radius_sorted", "radius_unsorted", "radius_capped"]


@pytest.mark.parametrize("eigen", [False, True])
def test_class_surface_kdtree_equals_the_reference_wrapper_over_nanoflann(eigen, tmp_path):
    """geometry::KDTree<T> (host/one_piece/Geometry/KDTree.h; reference: src/Geometry/KDTree.h:62-259) -- KnnSearch and RadiusSearch return what
    the reference's wrapper over the real nanoflann returned for the fixture: indices and squared distances bit for bit, in its order, with
    exactly equidistant points (lattice, duplicates, quantised coordinates, integer histograms in 33 dimensions), sorted / unsorted and
    capped radius searches.  With the look-alike matrix types and, where the reference tree is present, with the real Eigen."""
    import subprocess
    from helpers import nanoflann_case
    eig = "/root/reference/3rdparty/Eigen"
    if eigen and not os.path.isdir(eig):
        pytest.skip("the reference's vendored Eigen is not on this machine")
    exe = tmp_path / "kdtree_check"
    flags = ["-DONEPIECE_HAVE_EIGEN", "-I", eig, "-w"] if eigen else ["-Wall"]
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-msse4.2", "-ffp-contract=off"] + flags + ["-I", os.path.join(ROOT, "host", "one_piece"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "kdtree_check.cpp"), "-o", str(exe)])
    for name in KD_CASES:
        c = nanoflann_case(name)
        blob = tmp_path / (name + ".bin")
        with open(blob, "wb") as f:
            f.write(np.array([c["dim"], len(c["target"]), len(c["query"]), c["k"], 0 if c["radius"] is None else 1, c["sorted"]], np.int32).tobytes())
            f.write(np.array([c["radius"] or 0.0], np.float32).tobytes())
            for key, dt in (("target", np.float32), ("query", np.float32), ("found", np.int32), ("index", np.int32), ("dist2", np.float32)):
                f.write(np.ascontiguousarray(c[key], dt).tobytes())
        run = subprocess.run([str(exe), str(blob)], capture_output=True, text=True)
        assert run.returncode == 0, name + ": " + run.stdout + run.stderr
