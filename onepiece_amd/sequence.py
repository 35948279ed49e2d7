"""The reference's on-disk RGB-D sequence format ("TUM-format" in BASELINE.json):

    <dir>/associate.txt    one line per frame:  t_rgb  rgb_path  t_depth  depth_path
    <dir>/trajectory.txt   one line per frame:  16 floats = row-major 4x4 camera-to-world pose

read by tool::ReadImageSequence / ReadImageSequenceWithPose (/root/reference/src/Tool/IO.cpp:59-108).
Depth images are 16-bit PNGs in 1/depth_scale metres, colour images 8-bit PNGs; the examples load
them with cv::imread (colour arrives as B,G,R) and convert depth with tool::ConvertDepthTo32F
(Tool/ImageProcessing.cpp:68-91).  This module is harness code (file IO + one elementwise
conversion), not part of the measured path; PNG coding is PIL's.
"""
import os

import numpy as np


def WriteImageSequence(path, depths_m, rgbs_bgr, poses, depth_scale=1000.0):
    """Writes the sequence; depth in metres is quantised to uint16 = round(d * depth_scale)."""
    from PIL import Image
    os.makedirs(os.path.join(path, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(path, "depth"), exist_ok=True)
    with open(os.path.join(path, "associate.txt"), "w") as fa, open(os.path.join(path, "trajectory.txt"), "w") as ft:
        for i, (d, c, p) in enumerate(zip(depths_m, rgbs_bgr, poses)):
            t = "%.6f" % (i / 30.0)
            rgb_name, depth_name = "rgb/%06d.png" % i, "depth/%06d.png" % i
            d16 = np.clip(np.round(np.asarray(d, np.float64) * depth_scale), 0, 65535).astype(np.uint16)
            Image.fromarray(d16).save(os.path.join(path, depth_name))                         # 16-bit greyscale PNG
            Image.fromarray(np.ascontiguousarray(np.asarray(c, np.uint8)[:, :, ::-1])).save(os.path.join(path, rgb_name))  # stored as RGB
            fa.write("%s %s %s %s\n" % (t, rgb_name, t, depth_name))
            ft.write(" ".join("%.9g" % float(x) for x in np.asarray(p, np.float32).reshape(16)) + "\n")


def ReadImageSequence(path):
    """tool::ReadImageSequence (IO.cpp:59-80) -> (rgb_files, depth_files)."""
    rgb_files, depth_files = [], []
    with open(os.path.join(path, "associate.txt")) as f:
        for line in f:
            tok = line.split()
            if len(tok) < 4:
                continue
            rgb_files.append(os.path.join(path, tok[1]))
            depth_files.append(os.path.join(path, tok[3]))
    return rgb_files, depth_files


def ReadImageSequenceWithPose(path):
    """tool::ReadImageSequenceWithPose (IO.cpp:81-108) -> (rgb_files, depth_files, poses[n,4,4] f32)."""
    traj = os.path.join(path, "trajectory.txt")
    if not os.path.exists(traj):
        print("[ReadImageSequenceWithPose]::[ERROR]::No file named trajectory.txt.")
        return [], [], np.zeros((0, 4, 4), np.float32)
    rgb_files, depth_files = ReadImageSequence(path)
    poses = []
    with open(traj) as f:
        for line in f:
            tok = line.split()
            if len(tok) >= 16:
                poses.append(np.array([np.float32(t) for t in tok[:16]], np.float32).reshape(4, 4))
    if len(poses) != len(rgb_files):
        print("[ReadImageSequenceWithPose]::[WARNING]:: The number of images and poses do not match.")
    return rgb_files, depth_files, np.stack(poses) if poses else np.zeros((0, 4, 4), np.float32)


def imread(path, unchanged=False):
    """What cv::imread hands the examples: colour as uint8 B,G,R; with unchanged (-1) the stored
    16-bit depth as uint16."""
    from PIL import Image
    img = Image.open(path)
    a = np.array(img)
    if unchanged:
        return a.astype(np.uint16) if a.dtype != np.uint16 else a
    if a.ndim == 2:
        a = np.stack([a, a, a], axis=-1)
    return np.ascontiguousarray(a[:, :, 2::-1].astype(np.uint8))


def ConvertDepthTo32F(depth, depth_scale):
    """tool::ConvertDepthTo32F (ImageProcessing.cpp:68-91): uint16 / depth_scale in float32, negatives -> 0."""
    d = np.asarray(depth)
    if d.dtype == np.float32:
        return d.copy()
    if d.dtype != np.uint16:
        raise SystemExit("[ImageProcessing]::[ERROR]::Unknown depth image type: %s" % d.dtype)
    out = d.astype(np.float32) / np.float32(depth_scale)
    out[out < 0] = 0
    return out


class FramePrefetcher:
    """Decodes the PNG pairs of a sequence ahead of the consumer on a pool of host threads (PIL releases
    the GIL while inflating), so that file IO does not cap the fusion rate (1000 frames/s
    x 2.1 MB = 2.1 GB/s of decoded pixels).  Iterating yields (index, rgb_bgr_u8, depth_u16) in order.

        for i, rgb, depth in FramePrefetcher(rgb_files, depth_files, indices=range(0, n, 10)):
            ...
    """

    def __init__(self, rgb_files, depth_files, indices=None, workers=16, ahead=64):
        from concurrent.futures import ThreadPoolExecutor
        self.rgb_files, self.depth_files = rgb_files, depth_files
        self.indices = list(range(len(rgb_files)) if indices is None else indices)
        self.ahead = max(1, int(ahead))
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))

    def _load(self, i):
        return i, imread(self.rgb_files[i]), imread(self.depth_files[i], unchanged=True)

    def __iter__(self):
        from collections import deque
        pending = deque()
        it = iter(self.indices)
        try:
            for i in it:
                pending.append(self._pool.submit(self._load, i))
                if len(pending) >= self.ahead:
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()
        finally:
            for f in pending:
                f.cancel()

    def close(self):
        self._pool.shutdown(wait=False)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
