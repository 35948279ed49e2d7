"""onepiece_amd -- MI355X (gfx950) native implementation of OnePiece's TSDF-fusion + ICP hot path.

Layout: csrc/ (hand-written HIP kernels + the extern "C" ABI of include/onepiece_hip.h, built to
libonepiece_hip.so), integration.py / registration.py (host mirrors of the reference's
CubeHandler / ICP surface over that ABI), distributed.py (frame-sharded multi-GPU merge),
synthetic.py (input generators for tests and bench).  No CPU fallback exists.
"""
__all__ = ["integration", "registration", "synthetic", "distributed"]

# Several trackers / volumes working concurrently own one HIP stream each; the runtime maps streams onto 4 hardware queues unless told
# otherwise, and streams that share a queue serialise (2.6 k instead of 4.5 k frames/s with four frame pairs in flight).
# The variable is read once, when the HIP runtime initialises -- importing this package before the first GPU call is early enough.
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
