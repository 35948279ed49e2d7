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
