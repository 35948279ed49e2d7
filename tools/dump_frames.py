"""Writes n synthetic room frames to a binary file for tools/prof_driver.cpp (numpy only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from onepiece_amd import synthetic as S
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/frames.bin"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # frame first + i * stride (a stride of 4 over 250 frames covers the whole 1000-frame orbit)
with open(out, "wb") as f:
    np.array([n, S.W, S.H], np.int32).tofile(f)
    for i in range(n):
        d, c, p = S.room_frame(first + i * stride)
        p.astype(np.float32).tofile(f); d.astype(np.float32).tofile(f); c.astype(np.uint8).tofile(f)
print("wrote", out, n, "frames")
