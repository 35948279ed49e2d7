"""Seeded / analytic synthetic RGB-D input for tests and bench.py (there is no dataset access).

Two generators, both pure functions of the frame index (no RNG state), both emitting what the
reference's examples hand to CubeHandler::IntegrateImage after their own preprocessing
(example/ImageSequenceIntegration.cpp:31-41): a CV_32FC1 depth image in metres, a CV_8UC3 colour
image, and a 4x4 camera-to-world pose.

  * wall_frame(i)  -- the SURVEY.md section 6 / Appendix B scene: depth
    d(u,v) = 2.0 + 0.3 sin(u/40) cos(v/50) evaluated in float32, pose = identity + i cm along x.
    Its reference-run statistics are the committed anchor tests/golden/survey_wall_anchor.json.
  * room_frame(i)  -- an analytic room (axis-aligned box + two spheres) ray-cast per pixel from a
    camera orbiting near the room centre and looking outward; geometrically consistent across
    frames, depth within [0.5, 5] m everywhere.  This is the BASELINE.json workload
    ("synthetic TUM-format 640x480 sequence").

`xp` may be numpy or torch (torch lets bench.py generate frames directly in HBM); the math is
float32 in both.  Generated frames are inputs only -- nothing here is on the measured path.
"""
import math

import numpy as np

W, H = 640, 480
# OPEN3D_DATASET preset (Camera/Camera.h:94-104)
FX, FY, CX, CY = 514.817, 515.375, 318.771, 238.447

ROOM_HALF = (2.6, 1.4, 2.6)
SPHERES = ((1.2, 0.7, 1.6, 0.55), (-1.4, 0.5, -1.1, 0.7))
LOOP = 1000  # frames per orbit


def wall_depth():
    u = np.arange(W, dtype=np.float32)[None, :]
    v = np.arange(H, dtype=np.float32)[:, None]
    d = np.float32(2.0) + np.float32(0.3) * np.sin((u / np.float32(40)).astype(np.float32)) * \
        np.cos((v / np.float32(50)).astype(np.float32))
    return d.astype(np.float32)


def wall_frame(i):
    """-> depth [H,W] f32, rgb [H,W,3] u8, pose [4,4] f32 (identity + i cm in x)."""
    d = wall_depth()
    u = np.arange(W)[None, :]
    v = np.arange(H)[:, None]
    rgb = np.stack([(u * 255 // (W - 1)) + 0 * v, (v * 255 // (H - 1)) + 0 * u, ((u + v + 7 * i) % 256)], axis=-1).astype(np.uint8)
    pose = np.eye(4, dtype=np.float32)
    pose[0, 3] = np.float32(0.01 * i)
    return d, rgb, pose


def room_pose(i):
    """Camera-to-world pose of global frame i: orbit of radius 0.4..0.75 m, looking outward."""
    th = 2.0 * math.pi * (i % LOOP) / LOOP
    loop = i // LOOP
    radius = 0.4 + 0.05 * (loop % 8)
    pitch = 0.12 * math.sin(2.0 * th + 0.3 * loop)
    cy_, sy_ = math.cos(th), math.sin(th)
    cp, sp = math.cos(pitch), math.sin(pitch)
    Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    T = np.eye(4)
    T[:3, :3] = Ry @ Rx
    T[:3, 3] = [radius * sy_, 0.15 * math.sin(3.0 * th), radius * cy_]
    return T.astype(np.float32)


def _ops(xp):
    if xp is np:
        return dict(where=np.where, sqrt=np.sqrt, minimum=np.minimum, sin=np.sin, abs=np.abs,
                    full=lambda shape, v, like: np.full(shape, v, np.float32),
                    f32=lambda a: np.asarray(a, dtype=np.float32),
                    u8=lambda a: a.astype(np.uint8), stack=lambda xs: np.stack(xs, axis=-1),
                    clip=lambda a, lo, hi: np.clip(a, lo, hi))
    import torch
    return dict(where=torch.where, sqrt=torch.sqrt, minimum=torch.minimum, sin=torch.sin, abs=torch.abs,
                full=lambda shape, v, like: torch.full(shape, v, dtype=torch.float32, device=like.device),
                f32=None, u8=lambda a: a.to(torch.uint8), stack=lambda xs: torch.stack(xs, dim=-1),
                clip=lambda a, lo, hi: torch.clamp(a, lo, hi))


def room_render(pose, xp=np, device=None, width=W, height=H, fx=FX, fy=FY, cx=CX, cy=CY):
    """Ray-cast the analytic room from `pose` -> (depth [H,W] f32 metres (z-depth), rgb [H,W,3] u8)."""
    o = _ops(xp)
    if xp is np:
        u = np.arange(width, dtype=np.float32)[None, :]
        v = np.arange(height, dtype=np.float32)[:, None]
        P = np.asarray(pose, dtype=np.float32)
        f = np.float32
    else:
        import torch
        u = torch.arange(width, dtype=torch.float32, device=device)[None, :]
        v = torch.arange(height, dtype=torch.float32, device=device)[:, None]
        P = np.asarray(pose, dtype=np.float32)
        f = float
    dxc = (u - f(cx)) / f(fx) + 0 * v
    dyc = (v - f(cy)) / f(fy) + 0 * u
    # world ray: origin t, direction R * (dxc, dyc, 1); with dz_cam == 1 the ray parameter is z-depth
    dx = f(P[0, 0]) * dxc + f(P[0, 1]) * dyc + f(P[0, 2])
    dy = f(P[1, 0]) * dxc + f(P[1, 1]) * dyc + f(P[1, 2])
    dz = f(P[2, 0]) * dxc + f(P[2, 1]) * dyc + f(P[2, 2])
    ox, oy, oz = f(P[0, 3]), f(P[1, 3]), f(P[2, 3])
    big = f(1e9)
    eps = f(1e-9)

    def slab(orig, d, half):
        t_hi = (f(half) - orig) / o["where"](o["abs"](d) > eps, d, eps + 0 * d)
        t_lo = (f(-half) - orig) / o["where"](o["abs"](d) > eps, d, eps + 0 * d)
        t = o["where"](d > 0, t_hi, t_lo)
        return o["where"](o["abs"](d) > eps, t, big + 0 * d)

    t = o["minimum"](o["minimum"](slab(ox, dx, ROOM_HALF[0]), slab(oy, dy, ROOM_HALF[1])), slab(oz, dz, ROOM_HALF[2]))
    dd = dx * dx + dy * dy + dz * dz
    for (sx, sy, sz, sr) in SPHERES:
        lx, ly, lz = ox - f(sx), oy - f(sy), oz - f(sz)
        b = dx * lx + dy * ly + dz * lz
        c = lx * lx + ly * ly + lz * lz - f(sr * sr)
        disc = b * b - dd * c
        ts = (-b - o["sqrt"](o["clip"](disc, 0.0, 1e30))) / dd
        hit = (disc > 0) & (ts > f(0.05))
        t = o["where"](hit & (ts < t), ts, t)
    hx, hy, hz = ox + t * dx, oy + t * dy, oz + t * dz
    r = o["clip"](f(128.0) + f(110.0) * o["sin"](f(3.1) * hx + f(0.7) * hy), 0.0, 255.0)
    g = o["clip"](f(128.0) + f(110.0) * o["sin"](f(2.3) * hy + f(1.9) * hz), 0.0, 255.0)
    bl = o["clip"](f(128.0) + f(110.0) * o["sin"](f(2.7) * hz - f(1.3) * hx), 0.0, 255.0)
    rgb = o["u8"](o["stack"]([bl, g, r]))  # stored channel order B, G, R like cv::imread
    if xp is np:
        t = t.astype(np.float32)
    return t, rgb


def room_frame(i, xp=np, device=None):
    pose = room_pose(i)
    depth, rgb = room_render(pose, xp=xp, device=device)
    return depth, rgb, pose


def room_sequence_torch(first, count, device):
    """Generate `count` consecutive room frames directly on `device` (torch)."""
    import torch
    depth = torch.empty((count, H, W), dtype=torch.float32, device=device)
    rgb = torch.empty((count, H, W, 3), dtype=torch.uint8, device=device)
    poses = np.empty((count, 4, 4), np.float32)
    for k in range(count):
        d, c, p = room_frame(first + k, xp=torch, device=device)
        depth[k], rgb[k], poses[k] = d, c, p
    return depth, rgb, poses


def image_normals(depth, fx=FX, fy=FY, cx=CX, cy=CY):
    """Per-pixel normals from image-space cross products of the back-projected depth (an INPUT
    generator for ICP tests/bench: any unit normals are valid input to PointToPlane).  Returns the
    normals of the pixels with depth > 0 in row-major order, matching LoadFromDepth's compaction."""
    h, w = depth.shape
    u = np.arange(w, dtype=np.float32)[None, :]
    v = np.arange(h, dtype=np.float32)[:, None]
    P = np.stack([(u - np.float32(cx)) * depth / np.float32(fx), (v - np.float32(cy)) * depth / np.float32(fy), depth], axis=-1)
    du = np.zeros_like(P)
    dv = np.zeros_like(P)
    du[:, 1:-1] = P[:, 2:] - P[:, :-2]; du[:, 0] = P[:, 1] - P[:, 0]; du[:, -1] = P[:, -1] - P[:, -2]
    dv[1:-1] = P[2:] - P[:-2]; dv[0] = P[1] - P[0]; dv[-1] = P[-1] - P[-2]
    n = np.cross(du, dv)
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)
    return np.ascontiguousarray(n.reshape(-1, 3)[depth.reshape(-1) > 0], np.float32)
