"""Shared test helpers (inputs only; no reference arithmetic lives here)."""
import numpy as np

from onepiece_amd import synthetic as S


def small_camera(scale=4):
    """OPEN3D intrinsics scaled down by `scale` (tuple for oracle.make_camera / op_camera)."""
    return (S.FX / scale, S.FY / scale, S.CX / scale, S.CY / scale, S.W // scale, S.H // scale, 1000.0)


def room_cloud(i, scale=1):
    """Back-projected room frame i with image-space normals (test input): points [n,3], normals [n,3]."""
    fx, fy, cx, cy, w, h, _ = small_camera(scale)
    depth, _rgb = S.room_render(S.room_pose(i), width=w, height=h, fx=fx, fy=fy, cx=cx, cy=cy)
    u = np.arange(w, dtype=np.float32)[None, :]
    v = np.arange(h, dtype=np.float32)[:, None]
    P = np.stack([(u - np.float32(cx)) * depth / np.float32(fx), (v - np.float32(cy)) * depth / np.float32(fy), depth], axis=-1)
    du = np.zeros_like(P); dv = np.zeros_like(P)
    du[:, 1:-1] = P[:, 2:] - P[:, :-2]; du[:, 0] = P[:, 1] - P[:, 0]; du[:, -1] = P[:, -1] - P[:, -2]
    dv[1:-1] = P[2:] - P[:-2]; dv[0] = P[1] - P[0]; dv[-1] = P[-1] - P[-2]
    n = np.cross(du, dv)
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)
    return depth, P.reshape(-1, 3).astype(np.float32), n.reshape(-1, 3).astype(np.float32)


def rel_err(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))


def track_levels(i, j, n_levels=3, holes=True, scale=1):
    """Pyramid inputs of Odometry::MultiScaleComputing for synthetic room frames i (source) -> j (target),
    prepared on the CPU with the oracle's image preparation (test INPUT; the tracker's pinned parity
    boundary starts at these arrays).  holes: punch NaN regions / out-of-range depth into both frames.
    Returns (levels, T_true) with T_true = pose_j^-1 * pose_i (source camera -> target camera)."""
    from oracle import oracle as ORC
    fx, fy, cx, cy, w, h, _ = small_camera(scale)

    def prep(k):
        pose = S.room_pose(k)
        d, rgb = S.room_render(pose, width=w, height=h, fx=fx, fy=fy, cx=cx, cy=cy)
        d = d.copy()
        if holes:
            d[h // 5:h // 5 + h // 12, w // 3:w // 3 + w // 6] = 0.0          # sensor dropout
            d[(3 * h) // 5:(3 * h) // 5 + h // 10, w // 8:w // 8 + w // 10] = 7.5  # beyond MAX_DEPTH
            d[::37, ::29] = 0.2                                                # below MIN_DEPTH speckle
        g = ORC.prep_blur3(ORC.prep_intensity(rgb))
        dd = ORC.prep_blur3(ORC.prep_depth_nan(d))
        return g, dd, pose

    def pyramid(img):
        out = [img]
        for _ in range(1, n_levels):
            out.append(ORC.prep_pyrdown(out[-1]))
        return out

    sg, sd, ps = prep(i)
    tg, td, pt = prep(j)
    scp, sdp, tcp, tdp = pyramid(sg), pyramid(sd), pyramid(tg), pyramid(td)
    levels = []
    for l in range(n_levels):
        levels.append({"width": w, "height": h, "fx": fx, "fy": fy, "cx": cx, "cy": cy,
                       "source_color": scp[l], "source_depth": sdp[l], "target_color": tcp[l], "target_depth": tdp[l],
                       "target_color_dx": ORC.prep_sobel(tcp[l], 0), "target_color_dy": ORC.prep_sobel(tcp[l], 1),
                       "target_depth_dx": ORC.prep_sobel(tdp[l], 0), "target_depth_dy": ORC.prep_sobel(tdp[l], 1)})
        fx, fy, cx, cy, w, h = fx / 2, fy / 2, cx / 2, cy / 2, w // 2, h // 2
    T_true = np.linalg.inv(pt.astype(np.float64)) @ ps.astype(np.float64)
    return levels, T_true.astype(np.float32)


# canonical cube edges for the corner numbering of VoxelCube.h:45-47 (bottom ring, top ring, verticals)
MC_EDGE_PAIRS = np.array([[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4], [0, 4], [1, 5], [2, 6], [3, 7]], np.int32)


def procedural_mc_table():
    """A 256 x 16 marching-cubes style table for TESTS: for every sign configuration, a triangle fan over the
    edges whose end points differ in sign (so every interpolation has a non-zero denominator), at most five
    triangles, rows terminated by -1.  It is NOT the reference's table (that one is data of the reference's
    header and is passed in by the caller in a real build); the extraction code is table-agnostic, so any
    well-formed table exercises the same path on both sides."""
    tab = np.full((256, 16), -1, np.int32)
    for case in range(256):
        active = [e for e, (a, b) in enumerate(MC_EDGE_PAIRS) if ((case >> a) & 1) != ((case >> b) & 1)]
        tris = [(active[0], active[k], active[k + 1]) for k in range(1, len(active) - 1)][:5] if len(active) >= 3 else []
        flat = [e for t in tris for e in t]
        tab[case, :len(flat)] = flat
    return tab


def triangle_soup(points, colors):
    """Order-independent form of an unshared-vertex mesh: rows = triangles (18 floats), sorted lexicographically."""
    t = np.concatenate([points.reshape(-1, 9), colors.reshape(-1, 9)], 1)
    return t[np.lexsort(t.T[::-1])]


def nanoflann_case(name):
    """One case of the fixture oracle/tools/gen_nanoflann_golden.cpp wrote: inputs and the real nanoflann's answers."""
    import base64
    import json, os
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "nanoflann_golden.json")))
    c = cases[name]
    a = lambda key, dt, case=c: np.frombuffer(base64.b64decode(case[key]), dtype=dt)
    k, dim = c["k"], c.get("dim", 3)
    return {"k": k, "dim": dim, "radius": c.get("radius"), "sorted": c.get("sorted", 1),
            "target": a("target_f32", np.float32, cases[c.get("target_of", name)]).reshape(-1, dim), "query": a("query_f32", np.float32).reshape(-1, dim),
            "found": a("found_i32", np.int32), "index": a("index_i32", np.int32).reshape(-1, k), "dist2": a("dist2_f32", np.float32).reshape(-1, k)}


def squared_distances(target, q):
    """nanoflann's L2_Simple_Adaptor: ((0 + dx*dx) + dy*dy) + dz*dz in float32."""
    d = q[None, :] - target
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
