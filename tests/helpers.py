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
