"""onepiece_amd -- MI355X (gfx950) native implementation of OnePiece's TSDF-fusion + ICP hot path.

Layout: csrc/ (hand-written HIP kernels + the extern "C" ABI of include/onepiece_hip.h, built to
libonepiece_hip.so), integration.py / registration.py (host mirrors of the reference's
CubeHandler / ICP surface over that ABI), distributed.py (frame-sharded multi-GPU merge),
synthetic.py (input generators for tests and bench).  No CPU fallback exists.
"""
__all__ = ["integration", "registration", "synthetic", "distributed"]
