"""The reference's OWN example sources against this repository's class surface, and the surface against the reference's headers.

BASELINE.json north_star: "keeping the existing Integration/Registration C++ class and operator surface so example/...
ImageSequenceIntegration link against it unchanged".  oracle/tools/build_ref_examples.sh compiles
example/{ImageIntegration,ImageSequenceIntegration,ICPTest}.cpp where they lie under /root/reference -- unedited, nothing
copied -- against host/one_piece and links them with libone_piece_hip_host.so; the only stand-in on the include path is the
headless viewer (the OpenGL GUI is out of scope).  The binaries land in oracle/_ref/examples/ (git-ignored, they travel to
the GPU box as built artefacts), where the `-m gpu` tests RUN them on synthetic inputs and check what they write against
the CPU oracle.  CPU tests: the examples compile and link, every public member function the reference's headers declare
for the surface classes is declared here with the same signature, and the host-side members (Frustum, mesh / cloud
helpers, GetSDF) compute what their definitions say."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from onepiece_amd import synthetic as S, sequence as Q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = os.path.join(ROOT, "host", "one_piece")
EXDIR = os.path.join(ROOT, "oracle", "_ref", "examples")
EXAMPLES = ("ImageIntegration", "ImageSequenceIntegration", "ICPTest", "MergeMultipleSubmaps", "MCGenerateMesh", "EstimateNormals", "ReadRGBD",
            "ConvertImageSequenceToPCD", "ReadPLYPointCloud", "ReadPLYMesh", "DenseOdometry", "SimplifyMeshClustering",
            "PruneMesh", "EigenTest", "DenseFusion")
have_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "example")), reason="reference tree not present on this machine")


def _host_lib():
    subprocess.check_call(["make", "-s"], cwd=HOST)
    return os.path.join(HOST, "libone_piece_hip_host.so")


# --------------------------------------------------------------------------------------------------------------------
# compile + link, unchanged
# --------------------------------------------------------------------------------------------------------------------
@have_ref
def test_reference_examples_compile_and_link_unchanged(hip, tmp_path):
    """g++ -std=c++11 on the reference's example sources IN PLACE; -I host/one_piece, -I include and the headless viewer."""
    _host_lib()
    for ex in EXAMPLES:
        p = os.path.join(EXDIR, ex + ".bin")
        if os.path.exists(p):
            os.unlink(p)
    out = subprocess.run([os.path.join(ROOT, "oracle", "tools", "build_ref_examples.sh")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    for ex in EXAMPLES:
        exe = os.path.join(EXDIR, ex + ".bin")
        assert os.path.exists(exe), ex
        # the binary starts, resolves every symbol (lazy binding off) and prints the example's own usage line
        run = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, LD_BIND_NOW="1"), cwd=str(tmp_path), timeout=120)
        # (EstimateNormals.cpp takes no arguments: it goes straight to work and ends in the viewer)
        assert run.returncode in (0, 1) and ("usage" in run.stdout.lower() or "[headless viewer]" in run.stdout), (ex, run.stdout, run.stderr)
    # nothing of the reference was copied next to the binaries
    assert sorted(os.listdir(EXDIR)) == sorted(e + ".bin" for e in EXAMPLES)


# --------------------------------------------------------------------------------------------------------------------
# declarations: the reference's headers vs host/one_piece
# --------------------------------------------------------------------------------------------------------------------
def class_members(path, cls):
    """{(access, return type, name, (parameter types...), is_const)} of the member functions `cls` declares in `path`:
    comments, parameter names, default values, inline bodies and constructor initialiser lists are dropped, whitespace
    is removed from the types."""
    s = open(path).read()
    s = re.sub(r"//[^\n]*", "", s)
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    m = re.search(r"\b(class|struct)\s+" + cls + r"\b[^;{]*\{", s)
    assert m, (path, cls)
    access = "private" if m.group(1) == "class" else "public"
    k, depth, body = m.end() - 1, 0, ""
    while True:
        c = s[k]
        if c == "{":
            depth += 1
            if depth == 2:
                body += ";"          # an inline body ends the declaration
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        elif depth == 1:
            body += c
        k += 1
    out = set()
    for stmt in body.split(";"):
        st = " ".join(stmt.split())
        while True:
            mm = re.match(r"^(public|protected|private)\s*:\s*(.*)$", st)
            if not mm:
                break
            access, st = mm.group(1), mm.group(2)
        mm = re.match(r"^([^()]*?)(operator\s*[^\s\w()]+|~?\w+)\s*\(", st)   # the name is the identifier (or operator) before the FIRST parenthesis
        if not mm or st.startswith(("typedef", "using", "static_assert", "friend", "#")):
            continue
        ret, name = mm.group(1).strip(), mm.group(2).replace(" ", "")
        if "=" in ret:
            continue                  # a data member with an initialiser, not a function
        i, d = mm.end(), 1            # the balanced parameter list
        while i < len(st) and d:
            d += st[i] == "("
            d -= st[i] == ")"
            i += 1
        params, tail = st[mm.end():i - 1], st[i:].strip()
        types, cur, da = [], "", 0
        for ch in params + ",":
            da += ch in "<("
            da -= ch in ">)"
            if ch == "," and da == 0:
                p, cur = cur.split("=")[0].strip(), ""
                if not p:
                    continue
                toks = re.findall(r"[\w:]+(?:<[^>]*>)?|[&\*]", p)
                if len(toks) >= 2 and re.match(r"^[A-Za-z_]\w*$", toks[-1]) and toks[-1] not in ("int", "float", "double", "bool", "char", "long", "unsigned", "size_t"):
                    toks = toks[:-1]  # the parameter's name
                types.append("".join(toks).replace(" ", ""))
            else:
                cur += ch
        for q in ("static ", "inline ", "virtual ", "explicit "):
            ret = ret.replace(q, "")
        out.add((access, ret.replace(" ", ""), name, tuple(types), tail.startswith("const")))
    return out


# what the reference declares and this surface deliberately does not -- each with its reason
EXEMPT = {
    ("CubeHandler", "CollectGarbage"): "declared at CubeHandler.h:357 and defined nowhere in the reference: there is no behaviour to provide",
    ("TriangleMesh", "QuadricSimplify"): "quadric edge-collapse simplifier (Geometry/MeshSimplification.cpp): out of scope, SURVEY section 2",
    ("RGBDFrame", "PrepareDownSamplePointCloud"): "submap / global-registration bookkeeping of example/DenseFusion: out of scope",
    ("RGBDFrame", "IsPreprocessedDense"): "sparse / dense odometry caches of the reference's Odometry class: the tracker here keeps its pyramids on the device",
    ("RGBDFrame", "IsPreprocessedSparse"): "as above",
}
_SPARSE = "sparse odometry (ORB features, brute-force / MILD matching, RANSAC on OpenCV): out of scope, SURVEY section 2"
_ON_DEVICE = "the pyramids live on the GPU here: exposed as op_tracker_track / op_tracker_read_pyramid in the C-ABI"
for _name in ("Find2DMathes", "SparseTracking", "SparseTrackingMILD", "ComputeTransformation", "GetLocalPointsFromKeyPoints", "GetCorrespondencesFromMatches",
              "SetFeatureNumber", "GetFeatureNumber"):
    EXEMPT[("Odometry", _name)] = _SPARSE
for _name in ("CreateImagePyramid", "CreateImageXYZPyramid", "MultiScaleComputing", "InitializeRGBDDenseTracking"):
    EXEMPT[("Odometry", _name)] = _ON_DEVICE
SURFACE = [("Integration/CubeHandler.h", "CubeHandler"), ("Integration/Frustum.h", "Frustum"), ("Integration/Integrator.h", "Integrator"),
           ("Integration/VoxelCube.h", "VoxelCube"), ("Integration/VoxelCube.h", "CubePara"), ("Integration/TSDFVoxel.h", "TSDFVoxel"),
           ("Geometry/PointCloud.h", "PointCloud"), ("Geometry/TriangleMesh.h", "TriangleMesh"), ("Geometry/RGBDFrame.h", "RGBDFrame"),
           ("Camera/Camera.h", "PinholeCamera"), ("Registration/RegistrationResult.h", "RegistrationResult"), ("Odometry/Odometry.h", "Odometry"),
           ("Geometry/KDTree.h", "KDTree"), ("Geometry/KDTree.h", "SearchParameter")]


@have_ref
@pytest.mark.parametrize("header,cls", SURFACE)
def test_public_members_match_the_reference_headers(header, cls):
    """Every public member function of the reference's class is declared here with the same return type, parameter types
    and constness (read from the reference's header in place).  Extra members are allowed (documented extensions)."""
    pub = lambda members: {m[1:] for m in members if m[0] == "public"}
    theirs, ours = pub(class_members(os.path.join(REF, "src", header), cls)), pub(class_members(os.path.join(HOST, header), cls))
    implicit = {("", cls, ("const" + cls + "&",), False), ("", "~" + cls, (), False)}   # copy constructor / destructor: implicitly declared when not spelled out
    missing = sorted(m for m in theirs - ours - implicit if (cls, m[1]) not in EXEMPT)
    assert not missing, "%s: declared by the reference, not by host/one_piece: %s" % (cls, missing)


@have_ref
def test_free_functions_of_the_path_match_the_reference_headers():
    norm = lambda s: re.sub(r"\s+", "", re.sub(r"//[^\n]*", "", s))
    ours = norm(open(os.path.join(HOST, "Registration", "ICP.h")).read())
    theirs = norm(open(os.path.join(REF, "src", "Registration", "ICP.h")).read())
    for fn in ("PointToPlane", "PointToPoint", "EstimateRigidTransformationPointToPlane"):
        decl = re.search(r"[\w:<>]+%s\([^;{]*\)" % fn, theirs).group(0)
        # same types in the same order; parameter names are the reference's in both
        assert decl in ours, fn
    for sig in ("intmax_iteration=30;", "doublethreshold=0.2;", "doublescaling=1.0;"):
        assert sig in theirs and sig in ours
    geo_t, geo_o = [norm(open(os.path.join(d, "Geometry", "Geometry.h")).read()) for d in (os.path.join(REF, "src"), HOST)]
    for decl in ("Matrix4Se3ToSE3(constVector6&input);", "voidTransformPoints(constMatrix4&T,Point3List&points);", "Point3TransformPoint(constMatrix4&T,constPoint3&point);",
                 "voidTransformNormals(constMatrix4&T,Point3List&normals);", "PlaneGetPlane(constPoint3&p1,constPoint3&p2,constPoint3&p3);",
                 "TransformationMatrixEstimateRigidTransformation(constPointCorrespondenceSet&correspondence_set);", "typedefVector4Plane;"):
        assert decl in geo_t and decl in geo_o, decl


# --------------------------------------------------------------------------------------------------------------------
# host-side members
# --------------------------------------------------------------------------------------------------------------------
def _probe(tmp_path):
    _host_lib()
    lib = os.path.join(ROOT, "onepiece_amd")
    exe = os.path.join(ROOT, "tests", "cpp", "host_only_check.bin")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I", HOST, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_only_check.cpp"),
                           "-L", HOST, "-lone_piece_hip_host", "-L", lib, "-lonepiece_hip", "-Wl,-rpath," + HOST, "-Wl,-rpath," + lib, "-o", exe])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, check=True).stdout
    rows = {}
    for line in out.splitlines():
        tok = line.split()
        if tok and re.match(r"^[a-z_0-9]+$", tok[0]):
            rows[tok[0]] = tok[1:]
    return rows


def test_frustum_and_host_geometry_members(hip, oracle, tmp_path):
    """integration::Frustum from C++ (no GPU): planes bit-equal to the oracle's restatement of Frustum.cpp, corners and
    lines consistent with them in the reference's order; the mesh / cloud helpers do what their declarations say."""
    r = _probe(tmp_path)
    f32 = lambda toks: np.array([np.float32(t) for t in toks], np.float32)
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = (0.25, -0.5, 1.0); T[0, 0] = 0.8; T[0, 2] = 0.6; T[2, 0] = -0.6; T[2, 2] = 0.8
    planes = np.stack([f32(r["plane%d" % k]) for k in range(6)])
    assert np.array_equal(planes.view(np.uint32), oracle.frustum_planes(oracle.make_camera(), T, far=4.0, near=0.5).view(np.uint32))
    corners = np.stack([f32(r["corner%d" % k]) for k in range(8)]).astype(np.float64)
    # reference order (Frustum.cpp:49-56): ftl ftr fbl fbr nbr ntl ntr nbl; each plane contains its face's four corners
    faces = {0: (0, 1, 5, 6), 1: (0, 2, 5, 7), 2: (1, 3, 6, 4), 3: (2, 3, 7, 4), 4: (4, 5, 6, 7), 5: (0, 1, 2, 3)}  # top left right bottom near far
    for k, idx in faces.items():
        assert np.abs(corners[list(idx)] @ planes[k, :3].astype(np.float64) + planes[k, 3]).max() < 2e-5, k
    cam_pos = T[:3, 3].astype(np.float64)
    depth_along = (corners - cam_pos) @ T[:3, 2].astype(np.float64)
    assert np.allclose(depth_along[[0, 1, 2, 3]], 4.0, atol=1e-5) and np.allclose(depth_along[[4, 5, 6, 7]], 0.5, atol=1e-5)
    edge = [(0, 1), (3, 2), (1, 3), (2, 0), (4, 7), (6, 5), (5, 7), (6, 4), (0, 5), (1, 6), (2, 7), (3, 4)]   # Frustum.cpp:58-93
    for k, (a, b) in enumerate(edge):
        assert np.array_equal(f32(r["line%d" % k]), np.concatenate([corners[a], corners[b]]).astype(np.float32)), k
    assert r["contain"] == ["1", "0", "0"] and r["frustum_cloud"] == ["12000", "12000"]
    assert np.array_equal(f32(r["frustum_cloud_first"]), corners[0].astype(np.float32))
    assert np.linalg.norm(f32(r["frustum_cloud_last_of_edge0"]) - corners[1]) < 0.01          # 999/1000 of the way along edge 0
    assert np.allclose(f32(r["getplane"]), [3 ** -0.5] * 3 + [-3 ** -0.5], atol=1e-6)
    tn = np.tan(0.5)
    assert np.allclose(f32(r["vec_corner0"]), [-2 * tn * 1.5, 2 * tn, 2], atol=1e-5) and np.allclose(f32(r["vec_corner4"]), [tn * 1.5, -tn, 1], atol=1e-5)
    # ---- mesh: 10 x 10 grid of vertices on the plane z = 0.1 x, plus one far lone triangle
    assert r["mesh"] == ["103", "163", "has_normals", "1"]
    n = np.array([-0.01, 0.0, 0.1]); n /= np.linalg.norm(n)
    assert np.allclose(f32(r["normal_grid"]), n, atol=1e-5) and np.allclose(f32(r["normal_lone"]), [0, 0, 1], atol=1e-6)
    assert r["pruned"] == ["100", "162"]                                                        # the 3-vertex component is gone
    # vertex clustering at 0.25: 4 x 4 cells over [0, 0.9]^2; the lone triangle collapses into one cell and disappears
    assert r["clustered"][:2] == ["16", "18"] and r["clustered"][3] == "16" and r["clustered"][5] == "16"
    pts = np.stack([f32(r["clustered_p%d" % k]) for k in range(16)])
    cells = np.floor(pts[:, :2] / np.float32(0.25)).astype(int)
    assert len({tuple(c) for c in cells}) == 16 and cells.min() == 0 and cells.max() == 3 and np.allclose(pts[:, 2], 0.1 * pts[:, 0], atol=1e-6)
    assert r["clustered_zero_grid"] == ["103", "163"]                                            # refused, mesh unchanged
    assert r["malformed_ply"] == ["0", "0", "1", "3", "1", "0"]      # absurd vertex count refused; the face naming vertex 99 dropped, the good one kept; negative list count refused
    rt = r["roundtrip"]
    assert rt[:4] == ["1", "1", "1", "0"] and rt[5] == "1" and float(rt[7]) == 0 and float(rt[9]) == 0 and float(rt[11]) < 3 ** 0.5 / 255 + 1e-6 and float(rt[13]) == 0
    assert rt[15] == "1" and rt[17] == "1"
    assert r["joined"] == ["206", "326", "last", "203", "204", "205"]
    # ---- cloud
    assert r["downsample"] == ["27", "27"] and np.allclose(f32(r["downsample_p0"]), [0.045] * 3, atol=1e-6)
    assert r["merge_refused"] == ["1"] and r["merge_ok"] == ["1001", "1001"] and r["from_xyz"] == ["4"] and r["cloud_roundtrip"] == ["1", "1001", "1"]
    assert r["cube_float"] == ["ptr", "14", "v5", "0.25", "2", "0.5", "0.25", "0", "v7", "-0.5", "1"] and r["timer"] == ["1"]


def test_get_sdf_matches_the_oracle_bitwise(hip, oracle):
    """Integrator::GetSDF (Integrator.cpp:8-35) through op_get_sdf: random world points around the room frame, float and
    uint16 depth -- identical bits to the oracle, including the 999 cases (off-image, no depth)."""
    lib = hip.load()
    cam_t = (S.FX / 2, S.FY / 2, S.CX / 2, S.CY / 2, S.W // 2, S.H // 2, 1000.0)
    ocam = oracle.make_camera(*cam_t)
    hcam = hip.Camera(*cam_t)
    pose = S.room_pose(40)
    d, _ = S.room_render(pose, width=cam_t[4], height=cam_t[5], fx=cam_t[0], fy=cam_t[1], cx=cam_t[2], cy=cam_t[3])
    d = d.copy(); d[40:60, 50:90] = 0.0
    d16 = np.clip(np.round(d * 1000), 0, 65535).astype(np.uint16)
    inv = oracle.mat4_inverse(pose)
    rng = np.random.default_rng(5)
    # points on rays through the image (some beyond the borders), at depths around the surface
    u, v = rng.uniform(-20, cam_t[4] + 20, 4000), rng.uniform(-20, cam_t[5] + 20, 4000)
    z = rng.uniform(0.3, 5.0, 4000)
    pc = np.stack([(u - cam_t[2]) * z / cam_t[0], (v - cam_t[3]) * z / cam_t[1], z, np.ones_like(z)], 1)
    pw = (pose.astype(np.float64) @ pc.T).T[:, :3].astype(np.float32)
    fp = C.POINTER(C.c_float)
    n999 = 0
    for img, fmt, u16 in ((d, hip.OP_DEPTH_F32, 0), (d16, hip.OP_DEPTH_U16, 1)):
        for p in pw:
            got = C.c_float(0)
            p = np.ascontiguousarray(p)
            hip.check(lib.op_get_sdf(C.byref(hcam), p.ctypes.data_as(fp), np.ascontiguousarray(pose, np.float32).ctypes.data_as(fp), None, C.c_void_p(img.ctypes.data), fmt, C.byref(got)))
            want = oracle.lib().orc_get_sdf(C.byref(ocam), p.ctypes.data_as(fp), np.ascontiguousarray(inv, np.float32).ctypes.data_as(fp), C.c_void_p(img.ctypes.data), u16)
            assert np.float32(got.value).view(np.uint32) == np.float32(want).view(np.uint32)
            n999 += got.value == 999.0
    assert 200 < n999 < 7000        # both outcomes are exercised


# --------------------------------------------------------------------------------------------------------------------
# GPU: run the reference's examples (built in the build container) and check what they write
# --------------------------------------------------------------------------------------------------------------------
def _example(name):
    exe = os.path.join(EXDIR, name + ".bin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/examples/%s.bin was not built (needs the reference tree at build time)" % name)
    return exe


def _read_ply(path):
    """Binary little-endian PLY as written by the surface: -> (points [n,3] f32, normals or None, triangles [m,3] u32)."""
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    nv = int([l for l in lines if l.startswith("element vertex")][0].split()[2])
    nf = [l for l in lines if l.startswith("element face")]
    props = [l.split()[1:] for l in lines if l.startswith("property") and "list" not in l]
    dt = np.dtype([(name, "<f4" if t == "float" else "u1") for t, name in props])
    verts = np.frombuffer(body, dt, nv)
    pts = np.stack([verts["x"], verts["y"], verts["z"]], 1)
    nrm = np.stack([verts["nx"], verts["ny"], verts["nz"]], 1) if "nx" in dt.names else None
    tris = np.zeros((0, 3), np.uint32)
    if nf:
        m = int(nf[0].split()[2])
        fdt = np.dtype([("n", "u1"), ("v", "<u4", 3)])
        faces = np.frombuffer(body, fdt, m, offset=nv * dt.itemsize)
        assert np.all(faces["n"] == 3)
        tris = faces["v"]
    return pts, nrm, tris


def _default_mc_tables():
    lib = C.CDLL(_host_lib())
    tri = np.zeros((256, 16), np.int32); edges = np.zeros((12, 2), np.int32)
    lib.op_host_generate_mc_tables(tri.ctypes.data_as(C.POINTER(C.c_int)), edges.ctypes.data_as(C.POINTER(C.c_int)))
    return tri, edges


def _surfaces_agree(got, want, tol):
    """Two samplings of one surface: every point of each lies within tol of the other."""
    from scipy.spatial import cKDTree
    a, b = cKDTree(got), cKDTree(want)
    d_gw = b.query(got)[0]
    d_wg = a.query(want)[0]
    return float(d_gw.max()), float(d_wg.max()), float(np.quantile(d_wg, 0.99))


@pytest.mark.gpu
def test_reference_image_sequence_integration_example_runs_on_the_gpu(hip, oracle, tmp_path):
    """example/ImageSequenceIntegration.cpp, compiled unedited: reads a TUM-format sequence (associate.txt, trajectory.txt,
    16-bit depth PNGs), fuses every 10th frame at 6.25 mm after ConvertDepthTo32F + BilateralFilter, TransformNearest by the
    middle pose, ExtractTriangleMesh, ClusteringSimplify(0.00625), writes ./image_integration.ply.  The surface it writes is
    the surface the CPU oracle extracts from the same frames (every vertex of either within 2 clustering cells of the other)."""
    exe = _example("ImageSequenceIntegration")
    seq = str(tmp_path / "seq")
    n = 21
    frames = [S.room_frame(3 * i) for i in range(n)]            # OPEN3D camera 640 x 480 = the example's default PinholeCamera
    Q.WriteImageSequence(seq, [f[0] for f in frames], [f[1] for f in frames], [f[2] for f in frames], 1000.0)
    run = subprocess.run([exe, seq], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    for i in (0, 10, 20):
        assert "Processing on %dth image" % i in run.stdout
    assert "[headless viewer] mesh with" in run.stdout
    pts, nrm, tris = _read_ply(str(tmp_path / "image_integration.ply"))
    assert len(pts) > 20000 and len(tris) > 40000 and tris.max() < len(pts)
    assert np.all((tris[:, 0] != tris[:, 1]) & (tris[:, 0] != tris[:, 2]) & (tris[:, 1] != tris[:, 2]))
    # the same pipeline on the CPU oracle (decoded frames, its bilateral filter, 6.25 mm, nearest-neighbour resampling)
    rgb_files, depth_files, poses = Q.ReadImageSequenceWithPose(seq)
    ov = oracle.Volume(oracle.make_camera(), voxel_res=0.00625)
    for i in (0, 10, 20):
        d = oracle.bilateral_filter(Q.imread(depth_files[i], unchanged=True), 7, 0.03, 4.5, 1000.0)
        ov.integrate(d, Q.imread(rgb_files[i]), poses[i])
    moved = ov.transform(poses[n // 2], nearest=True)
    tab, edges = _default_mc_tables()
    want, _ = moved.extract_mesh(tab, edges)
    cell = 0.00625
    d_gw, d_wg, q99 = _surfaces_agree(pts, want[:: 7], 2 * cell * 3 ** 0.5)
    assert d_gw < 3 * cell * 3 ** 0.5 and q99 < 2 * cell * 3 ** 0.5, (d_gw, d_wg, q99)


@pytest.mark.gpu
def test_reference_image_integration_example_runs_on_the_gpu(hip, oracle, tmp_path):
    """example/ImageIntegration.cpp, unedited: one RGB-D pair, identity pose, Frustum + LoadFromRGBD + ExtractTriangleMesh +
    ComputeNormals, writes ./image_integration.ply with normals.  Same triangle count as the oracle's mesh of the same frame
    (within the handful of cells the two bilateral filters round differently) and the same surface."""
    from PIL import Image
    exe = _example("ImageIntegration")
    d, c, _ = S.room_frame(60)
    d16 = np.clip(np.round(d * 1000), 0, 65535).astype(np.uint16)
    Image.fromarray(d16).save(str(tmp_path / "d.png"))
    Image.fromarray(np.ascontiguousarray(c[:, :, ::-1])).save(str(tmp_path / "c.png"))
    run = subprocess.run([exe, str(tmp_path / "c.png"), str(tmp_path / "d.png")], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    pts, nrm, tris = _read_ply(str(tmp_path / "image_integration.ply"))
    assert nrm is not None and len(tris) * 3 == len(pts) > 30000                 # unshared vertices, as ExtractTriangleMesh emits them
    ln = np.linalg.norm(nrm, axis=1)
    assert np.all((np.abs(ln - 1) < 1e-4) | (ln == 0))
    ov = oracle.Volume(oracle.make_camera())                                       # default 1 cm voxels, like the example's CubeHandler
    ov.integrate(oracle.bilateral_filter(d16, 7, 0.03, 4.5, 1000.0), c, np.eye(4, dtype=np.float32))
    tab, edges = _default_mc_tables()
    want, _ = ov.extract_mesh(tab, edges)
    assert abs(len(want) - len(pts)) <= 0.002 * len(want)
    d_gw, d_wg, q99 = _surfaces_agree(pts[::3], want[::3], 0.01)
    assert d_gw < 0.02 and q99 < 0.005, (d_gw, d_wg, q99)


@pytest.mark.gpu
def test_reference_icp_example_runs_on_the_gpu(hip, oracle, tmp_path):
    """example/ICPTest.cpp, unedited: two PLY clouds, EstimateNormals() with its defaults on both, PointToPlane with
    threshold 0.01, prints result->T.  The printed pose is the CPU oracle's (1e-4 relative)."""
    exe = _example("ICPTest")
    cam = (S.FX / 2, S.FY / 2, S.CX / 2, S.CY / 2, S.W // 2, S.H // 2, 1000.0)
    ocam = oracle.make_camera(*cam)
    clouds = []
    for k, i in enumerate((100, 101)):
        d, _ = S.room_render(S.room_pose(i), width=cam[4], height=cam[5], fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3])
        p = oracle.load_from_depth(ocam, d)
        clouds.append(p)
        with open(str(tmp_path / ("%d.ply" % k)), "wb") as f:
            f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(p)).encode())
            f.write(np.ascontiguousarray(p, "<f4").tobytes())
    tgt, src = clouds
    run = subprocess.run([exe, str(tmp_path / "1.ply"), str(tmp_path / "0.ply")], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    rows = []
    for line in run.stdout.splitlines():
        tok = line.split()
        try:
            vals = [float(t) for t in tok]
        except ValueError:
            rows = []
            continue
        rows = rows + [vals] if len(vals) == 4 else []
        if len(rows) == 4:
            break
    assert len(rows) == 4, run.stdout[-1500:]
    T = np.array(rows)
    nrm = oracle.estimate_normals(tgt, 0.1, 30)
    ref = oracle.icp(src, tgt, nrm, None, 30, 0.01, True)
    err = np.linalg.norm(T - ref["T"].astype(np.float64)) / np.linalg.norm(ref["T"].astype(np.float64))
    assert err <= 1e-4, (err, T, ref["T"])


# --------------------------------------------------------------------------------------------------------------------
# more of the reference's examples on this path: submap merging, marching cubes from a .map, normals, point clouds
# --------------------------------------------------------------------------------------------------------------------
def _write_legacy_map(path, keys, vox):
    """A volume in the all-float stream ReadFromFileFloat reads (CubeHandler.h:73-109, VoxelCube.h:168-193): per cube the id and a size slot,
    (index, sdf, weight)* of its observed voxels, -2, then the number of coloured voxels and (index, r, g, b, colour weight)* with the
    colours as weighted sums of 0..255 values."""
    buf = [0.0, float(len(keys))]
    for k, v in zip(keys, vox):
        buf += [float(k[0]), float(k[1]), float(k[2]), 0.0]
        obs = np.nonzero((v[:, 1] > 0) & (v[:, 0] < 1))[0]
        for i in obs:
            buf += [float(i), float(v[i, 0]), float(v[i, 1])]
        buf.append(-2.0)
        col = [i for i in obs if v[i, 2] >= 0]
        buf.append(float(len(col)))
        for i in col:
            buf += [float(i), float(v[i, 2]) * 255.0, float(v[i, 3]) * 255.0, float(v[i, 4]) * 255.0, 1.0]
    np.array(buf, np.float32).tofile(str(path))


def _submaps(oracle, tmp_path, n_sub=3, per_sub=4, res=0.02):
    """n_sub small volumes, each fused in the frame of its own first camera, as legacy .map files + the pose file of
    example/MergeMultipleSubmaps.cpp (count, then per submap: id, R row-major, t).  -> list of submap poses"""
    poses = []
    d = tmp_path / "maps"; d.mkdir()
    with open(str(tmp_path / "poses.txt"), "w") as pf:
        pf.write("%d\n" % n_sub)
        for s_ in range(n_sub):
            frames = [S.room_frame(40 * s_ + 5 * k) for k in range(per_sub)]
            base = frames[0][2].astype(np.float64)
            ov = oracle.Volume(oracle.make_camera(), voxel_res=res)
            for dd, cc, pp in frames:
                ov.integrate(dd, cc, (np.linalg.inv(base) @ pp.astype(np.float64)).astype(np.float32))
            k, v = ov.export()
            _write_legacy_map(d / ("m%d.map" % s_), k, v)
            T = base.astype(np.float32)
            poses.append(T)
            pf.write("%d %s %s\n" % (s_, " ".join(repr(float(x)) for x in T[:3, :3].reshape(-1)), " ".join(repr(float(x)) for x in T[:3, 3])))
    return poses


@pytest.mark.gpu
def test_reference_merge_multiple_submaps_example_runs_on_the_gpu(hip, oracle, tmp_path):
    """example/MergeMultipleSubmaps.cpp, unedited: ReadFromFileFloat of every submap, Transform by its pose (trilinear resampling),
    Merge, ExtractTriangleMesh, ClusteringSimplify(0.01), ./merged_mesh.ply -- rows I6, I8, I9 and N2 behind the reference's own driver.
    The surface it writes is the one the CPU oracle gets from the same files with the same calls."""
    exe = _example("MergeMultipleSubmaps")
    res = 0.02
    poses = _submaps(oracle, tmp_path, res=res)
    run = subprocess.run([exe, str(tmp_path / "maps"), str(tmp_path / "poses.txt")], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    for i in range(len(poses)):
        assert "merge %dth submap" % i in run.stdout
    pts, nrm, tris = _read_ply(str(tmp_path / "merged_mesh.ply"))
    assert len(pts) > 5000 and len(tris) > 10000 and tris.max() < len(pts) and nrm is not None
    total = oracle.Volume(voxel_res=0.01)        # the example's CubeHandler is default-constructed: Transform keeps the SOURCE resolution...
    first = True
    for i, T in enumerate(poses):
        sub = oracle.Volume(voxel_res=0.01)
        assert sub.read_file(tmp_path / "maps" / ("m%d.map" % i), legacy_float=True) == 0
        moved = sub.transform(T, nearest=False)
        if first:
            total, first = moved, False          # ... and Merge into an EMPTY handler of another resolution is refused by the reference
        else:
            total.merge(moved)
    tab, edges = _default_mc_tables()
    want, _ = total.extract_mesh(tab, edges)
    if len(want) == 0:
        pytest.skip("oracle produced no surface for this configuration")
    d_gw, d_wg, q99 = _surfaces_agree(pts, want[::5], 0.03)
    assert d_gw < 0.05 and q99 < 0.03, (d_gw, d_wg, q99)


@pytest.mark.gpu
def test_reference_mc_generate_mesh_example_runs_on_the_gpu(hip, oracle, tmp_path):
    """example/MCGenerateMesh.cpp, unedited: ReadFromFileFloat + ExtractTriangleMesh + ComputeNormals -> ./mc_mesh.ply.  Same triangle
    soup as the oracle's marching cubes of the same file (vertex for vertex after sorting)."""
    exe = _example("MCGenerateMesh")
    ov = oracle.Volume(oracle.make_camera(), voxel_res=0.01)
    for k in range(3):
        ov.integrate(*S.room_frame(7 * k))
    keys, vox = ov.export()
    _write_legacy_map(tmp_path / "room.map", keys, vox)
    run = subprocess.run([exe, str(tmp_path / "room.map")], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    pts, nrm, tris = _read_ply(str(tmp_path / "mc_mesh.ply"))
    ref = oracle.Volume(voxel_res=0.01)
    assert ref.read_file(tmp_path / "room.map", legacy_float=True) == 0
    tab, edges = _default_mc_tables()
    want, _ = ref.extract_mesh(tab, edges)
    assert len(pts) == len(want) > 30000 and len(tris) * 3 == len(pts) and nrm is not None
    order = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    assert np.array_equal(order(np.ascontiguousarray(pts)), order(np.ascontiguousarray(want, np.float32)))


@pytest.mark.gpu
def test_reference_estimate_normals_example_runs_on_the_gpu(hip, tmp_path):
    """example/EstimateNormals.cpp, unedited: 100 x 100 points of the plane x + 2y + 3z + 4 = 0, PointCloud::EstimateNormals() with its
    defaults, ./bunny_n.ply.  Every normal is +-(1, 2, 3)/sqrt(14)."""
    exe = _example("EstimateNormals")
    run = subprocess.run([exe], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    pts, nrm, _ = _read_ply(str(tmp_path / "bunny_n.ply"))
    assert len(pts) == 10000 and nrm is not None
    n0 = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    assert np.abs(np.abs(nrm @ n0) - 1.0).max() < 1e-3
    assert np.abs(pts @ np.array([1.0, 2.0, 3.0]) + 4.0).max() < 1e-3


@pytest.mark.gpu
def test_reference_point_cloud_examples_run_on_the_gpu(hip, oracle, tmp_path):
    """example/ReadRGBD.cpp and example/ConvertImageSequenceToPCD.cpp, unedited: PointCloud::LoadFromRGBD of 16-bit depth + colour PNGs
    (the first with its own camera literal, the second with the default camera over a TUM-format sequence), WriteToPLY.  The points are
    the oracle's LoadFromDepth of the same images, in order."""
    from PIL import Image
    d, c, _ = S.room_frame(21)
    d16 = np.clip(np.round(d * 1000), 0, 65535).astype(np.uint16)
    d16[100:140, 200:260] = 0
    Image.fromarray(d16).save(str(tmp_path / "d.png"))
    Image.fromarray(np.ascontiguousarray(c[:, :, ::-1])).save(str(tmp_path / "c.png"))
    run = subprocess.run([_example("ReadRGBD"), str(tmp_path / "c.png"), str(tmp_path / "d.png")], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    pts, _n, _t = _read_ply(str(tmp_path / "0.ply"))
    cam = oracle.make_camera(914.494141, 914.377991, 958.065430 / 3, 548.986206 * 4 / 9, 640, 480, 1000.0)   # example/ReadRGBD.cpp:13
    want = oracle.load_from_depth(cam, d16)
    assert pts.shape == want.shape and np.array_equal(pts.view(np.uint32), want.view(np.uint32))
    # the sequence converter
    seq = str(tmp_path / "seq")
    frames = [S.room_frame(9 * i) for i in range(3)]
    Q.WriteImageSequence(seq, [f[0] for f in frames], [f[1] for f in frames], [f[2] for f in frames], 1000.0)
    (tmp_path / "pcd").mkdir()
    run = subprocess.run([_example("ConvertImageSequenceToPCD"), seq], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    rgb_files, depth_files = Q.ReadImageSequence(seq)
    for i in range(3):
        pts, _n, _t = _read_ply(str(tmp_path / "pcd" / ("%d.ply" % i)))
        want = oracle.load_from_depth(oracle.make_camera(), Q.imread(depth_files[i], unchanged=True))
        assert pts.shape == want.shape and np.array_equal(pts.view(np.uint32), want.view(np.uint32)), i


def test_reference_ply_examples_run_on_the_host(hip, tmp_path):
    """example/ReadPLYPointCloud.cpp and example/ReadPLYMesh.cpp, unedited (host-only members: LoadFromPLY, WriteToPLY, DownSample,
    TriangleMesh::LoadFromFile, ComputeNormals): the cloud comes back as written, the down-sampled one is smaller."""
    exe = _example("ReadPLYPointCloud")
    rng = np.random.default_rng(5)
    p = rng.uniform(-1, 1, (5000, 3)).astype("<f4")
    with open(str(tmp_path / "in.ply"), "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(p)).encode())
        f.write(p.tobytes())
    run = subprocess.run([exe, str(tmp_path / "in.ply"), "0.25"], capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    pts, _n, _t = _read_ply(str(tmp_path / "transformed_pcd.ply"))
    assert np.array_equal(pts, p)
    m = [l for l in run.stdout.splitlines() if l.startswith("down sample: from 5000 to ")]
    assert m and 100 < int(m[0].split()[-1]) <= 512          # 8^3 cells of 0.25 over [-1, 1]^3
    # a mesh: the unit square as two triangles
    with open(str(tmp_path / "mesh.ply"), "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                b"element face 2\nproperty list uchar int vertex_indices\nend_header\n")
        f.write(np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], "<f4").tobytes())
        f.write(np.array([(3, (0, 1, 2)), (3, (0, 2, 3))], np.dtype([("n", "u1"), ("v", "<i4", 3)])).tobytes())
    run = subprocess.run([_example("ReadPLYMesh"), str(tmp_path / "mesh.ply")], capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
    assert run.returncode == 0 and "[headless viewer] mesh with 4 vertices, 2 triangles" in run.stdout, run.stdout + run.stderr


@pytest.mark.gpu
def test_reference_dense_odometry_example_runs_on_the_gpu(hip, oracle, tmp_path):
    """example/DenseOdometry.cpp, unedited: odometry::Odometry(camera).DenseTracking(source_frame, target_frame, identity, 0) on two
    RGB-D pairs read from PNGs, prints success and T.  The printed pose is the oracle's DenseTracking of the same images (to the
    agreement the default summation mode has with the reference's float32 sums, tests/test_odometry_gpu.py)."""
    from PIL import Image
    exe = _example("DenseOdometry")
    files = []
    frames = []
    for k, i in enumerate((301, 300)):
        d, c, _ = S.room_frame(i)
        d16 = np.clip(np.round(d * 1000), 0, 65535).astype(np.uint16)
        Image.fromarray(d16).save(str(tmp_path / ("d%d.png" % k)))
        Image.fromarray(np.ascontiguousarray(c[:, :, ::-1])).save(str(tmp_path / ("c%d.png" % k)))
        frames.append((Q.imread(str(tmp_path / ("c%d.png" % k))), Q.imread(str(tmp_path / ("d%d.png" % k)), unchanged=True)))
        files += [str(tmp_path / ("c%d.png" % k)), str(tmp_path / ("d%d.png" % k))]
    run = subprocess.run([exe] + files, capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "Successful Matching!" in run.stdout
    rows = []
    for line in run.stdout.splitlines():
        try:
            vals = [float(t) for t in line.split()]
        except ValueError:
            rows = []
            continue
        rows = rows + [vals] if len(vals) == 4 else []
        if len(rows) == 4:
            break
    assert len(rows) == 4, run.stdout[-1500:]
    T = np.array(rows, np.float64)
    ref = oracle.dense_tracking(oracle.make_camera(), frames[0][0], frames[1][0], frames[0][1], frames[1][1], (4, 8, 16), 0)
    assert ref["tracking_success"]
    err = np.linalg.norm(T - ref["T"].astype(np.float64)) / np.linalg.norm(ref["T"].astype(np.float64))
    assert err <= 1e-3, (err, T, ref["T"])          # printed with 6 significant digits; per-pair agreement of the default mode is 1e-4..1e-3


def test_reference_mesh_tool_examples_run_on_the_host(hip, tmp_path):
    """example/SimplifyMeshClustering.cpp and example/PruneMesh.cpp, unedited, on a 40 x 40 grid mesh plus a stray triangle: the clustered
    mesh keeps the surface with fewer vertices (one per occupied cell), pruning drops the small component."""
    n = 40
    gx, gy = np.meshgrid(np.arange(n, dtype=np.float32) * 0.01, np.arange(n, dtype=np.float32) * 0.01)
    v = np.stack([gx.ravel(), gy.ravel(), np.zeros(n * n, np.float32)], 1)
    idx = lambda i, j: i * n + j
    t = []
    for i in range(n - 1):
        for j in range(n - 1):
            t += [(idx(i, j), idx(i, j + 1), idx(i + 1, j + 1)), (idx(i, j), idx(i + 1, j + 1), idx(i + 1, j))]
    v = np.concatenate([v, np.array([[5, 5, 5], [5.01, 5, 5], [5, 5.01, 5]], np.float32)])   # a component of 3 points far away
    t.append((n * n, n * n + 1, n * n + 2))
    with open(str(tmp_path / "grid.ply"), "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(v), len(t))).encode())
        f.write(np.ascontiguousarray(v, "<f4").tobytes())
        f.write(np.array([(3, tri) for tri in t], np.dtype([("n", "u1"), ("v", "<i4", 3)])).tobytes())
    run = subprocess.run([_example("SimplifyMeshClustering"), str(tmp_path / "grid.ply"), "0.04", str(tmp_path / "clustered.ply")],
                         capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    pts, _n, tris = _read_ply(str(tmp_path / "clustered.ply"))
    assert 50 < len(pts) < 200 and len(tris) > 100 and tris.max() < len(pts)           # ~10 x 10 cells of 4 cm over 0.39 m (+ the stray one)
    on_sheet = pts[pts[:, 2] < 1]
    assert np.abs(on_sheet[:, 2]).max() < 1e-6 and on_sheet[:, :2].min() >= 0 and on_sheet[:, :2].max() <= 0.39 + 1e-6
    run = subprocess.run([_example("PruneMesh"), str(tmp_path / "grid.ply"), "10"], capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    pts, _n, tris = _read_ply(str(tmp_path / "grid.ply_pruned.ply"))
    assert len(pts) == n * n and len(tris) == 2 * (n - 1) * (n - 1) and pts.max() < 1


@pytest.mark.gpu
def test_reference_dense_fusion_example_runs_on_the_gpu(hip, tmp_path):
    """example/DenseFusion/{DenseFusion,DenseSlam}.cpp -- the link target BASELINE.json's north_star names -- compiled unedited against
    host/one_piece and run headless on a 118-frame synthetic sequence: every frame is tracked against the last tracked one
    (odometry::Odometry::DenseTracking on the GPU), submaps of 49 frames are registered (submap 1 against 0 by one ICP iteration on the GPU,
    submap 2 against 0 by FPFH + RANSAC on the host, registration::RansacRegistration), the submap poses go through optimization::Optimizer::
    FastBA, every 8th frame is fused with its optimised pose (ConvertDepthTo32F + BilateralFilter + CubeHandler::IntegrateImage on the GPU),
    the mesh is extracted, simplified and written.  Checked: the example's own progress lines, one trajectory line per frame with rigid poses
    that stay within centimetres of the synthetic ground truth, a mesh of the room.  (FPFH / RANSAC / FastBA are host code with no pinned
    parity: the reference seeds its RANSAC from std::random_device.)"""
    exe = _example("DenseFusion")
    seq = str(tmp_path / "seq")
    n = 118
    frames = [S.room_frame(600 + i) for i in range(n)]
    Q.WriteImageSequence(seq, [f[0] for f in frames], [f[1] for f in frames], [f[2] for f in frames], 1000.0)
    run = subprocess.run([exe, seq, "0.01"], capture_output=True, text=True, cwd=str(tmp_path), timeout=1500)
    assert run.returncode == 1, run.stdout[-3000:] + run.stderr[-3000:]        # the example's main ends with `return 1;` (DenseFusion.cpp:110)
    out = run.stdout
    assert "Process on %dth image" % (n - 1) in out and out.count("tracking successful!") == n
    assert "Matching 0 ..." in out and "Matching 1 ..." in out                   # submap 1 vs 0 (ICP), submap 2 vs 0 (RANSAC) and vs 1 (ICP)
    assert "[ERROR]" not in out and "There are unconnected components" not in out
    for i in range(8, n, 8):   # (frame 0 is never marked tracked -- DenseSlam.cpp:22-30 only sets the flag for frame_id > 0, RGBDFrame.h:44 defaults it to false -- so
        assert "Processing on %dth image" % i in out   # the reference's own fusion loop skips it, DenseFusion.cpp:83)
    assert "Processing on 0th image" not in out
    got = np.loadtxt(os.path.join(seq, "trajectory.txt")).reshape(-1, 4, 4)
    assert len(got) == n - 1                                                       # frames 1 .. n-1: the loop that fuses also writes the poses, and skips frame 0
    g0 = np.linalg.inv(frames[0][2].astype(np.float64))
    terr, rerr = [], []
    for i in range(1, n):
        P = got[i - 1]
        R = P[:3, :3]
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-3 and abs(np.linalg.det(R) - 1) < 1e-3 and np.array_equal(P[3], [0, 0, 0, 1]), i
        want = g0 @ frames[i][2].astype(np.float64)
        terr.append(np.abs(P[:3, 3] - want[:3, 3]).max()); rerr.append(np.abs(R - want[:3, :3]).max())
    step = np.linalg.norm(np.diff(got[:, :3, 3], axis=0), axis=1)
    print("reference DenseFusion on %d frames: translation error vs ground truth max %.3f m, rotation entries max %.3f, largest step %.4f m" % (n, max(terr), max(rerr), step.max()))
    # frame-to-frame dense odometry drifts (the CPU path's own chain drifts alike: dense_fusion_parity in the bench line); this is a sanity bound on the
    # whole pipeline -- tracking, submap registration, pose-graph optimisation -- not an accuracy claim
    # (measured: 0.064 m, 0.12, 0.032 m -- the pose graph is pulled by a RANSAC registration whose transform is the winning draw's 8-pair fit, as in the reference)
    assert max(terr) < 0.25 and max(rerr) < 0.25, (max(terr), max(rerr))
    assert step.max() < 0.10                                                       # a continuous camera path, also across submap borders after FastBA
    pts, nrm, tris = _read_ply(str(tmp_path / "densefusion_generated_mesh.ply"))
    assert len(pts) > 20000 and len(tris) > 40000 and tris.max() < len(pts)
    assert "[headless viewer] mesh with" in out
