"""Host-side mirror of the one_piece::tool image functions the fusion drivers call between reading a frame and
CubeHandler::IntegrateImage (/root/reference/src/Tool/ImageProcessing.h:19-20, ImageProcessing.cpp:64-91;
call sites example/ImageSequenceIntegration.cpp:36-38, DenseFusion/DenseFusion.cpp:92-94).  No arithmetic
here: BilateralFilter forwards to op_bilateral_filter_depth (kernel in csrc/imgproc.hip), which fails loudly
without a GPU.  The sequence-file helpers live in onepiece_amd.sequence and are re-exported for convenience.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .sequence import ConvertDepthTo32F, ReadImageSequence, ReadImageSequenceWithPose, imread  # noqa: F401

SIGMA_COLOR, SIGMA_SPACE = 0.03, 4.5   # ImageProcessing.cpp:66


def BilateralFilter(source, range=7, depth_scale=1000.0, device=0, stream=None, out=None):
    """tool::BilateralFilter(source, target, range = 7) = cv::bilateralFilter(source, target, range, 0.03, 4.5).

    source: [h, w] or [n, h, w]; float32 metres (what ConvertDepthTo32F returns) or uint16, in which case the
    division by `depth_scale` of ConvertDepthTo32F is folded in.  numpy arrays are filtered through a staged
    copy and a numpy array is returned; contiguous CUDA torch tensors are filtered in HBM and a torch tensor
    (or `out`) is returned.  With `stream` (a hipStream_t handle such as CubeHandler.Stream()) the call only
    enqueues on that stream, so the result can feed IntegrateImage/IntegrateSequence of that CubeHandler with no
    host synchronisation; without it the result is final on return.  As for every device input of this
    library, a tensor written by an asynchronous torch op must be made final by the caller first."""
    lib = L.load()
    if hasattr(source, "data_ptr"):
        import torch
        if not source.is_cuda or not source.is_contiguous():
            raise ValueError("torch images must be contiguous CUDA tensors")
        L.torch_ready(source)
        if source.dtype == torch.float32:
            fmt = L.OP_DEPTH_F32
        elif source.dtype in (torch.uint16, torch.int16):
            fmt = L.OP_DEPTH_U16
        else:
            raise ValueError("depth must be float32 (CV_32FC1) or uint16 (CV_16UC1)")
        if source.dim() not in (2, 3):
            raise ValueError("expected [h, w] or [n, h, w]")
        h, w = source.shape[-2:]
        n = source.shape[0] if source.dim() == 3 else 1
        if out is None:
            out = torch.empty(source.shape, dtype=torch.float32, device=source.device)
        elif out.shape != source.shape or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 CUDA tensor of the source's shape")
        dev = source.device.index if source.device.index is not None else device
        L.check(lib.op_bilateral_filter_depth(C.c_void_p(source.data_ptr()), fmt, float(depth_scale), w, h, n, int(range), SIGMA_COLOR,
                                              SIGMA_SPACE, L.OP_MEM_DEVICE, dev, C.c_void_p(stream) if stream else None,
                                              C.c_void_p(out.data_ptr())))
        return out
    if stream is not None:
        raise ValueError("a stream can only be given with device tensors")
    a = np.ascontiguousarray(source)
    if a.dtype == np.uint16:
        fmt = L.OP_DEPTH_U16
    else:
        a = np.ascontiguousarray(a, np.float32)
        fmt = L.OP_DEPTH_F32
    if a.ndim not in (2, 3):
        raise ValueError("expected [h, w] or [n, h, w]")
    h, w = a.shape[-2:]
    n = a.shape[0] if a.ndim == 3 else 1
    res = np.empty(a.shape, np.float32)
    L.check(lib.op_bilateral_filter_depth(C.c_void_p(a.ctypes.data), fmt, float(depth_scale), w, h, n, int(range), SIGMA_COLOR, SIGMA_SPACE,
                                          L.OP_MEM_HOST, device, None, C.c_void_p(res.ctypes.data)))
    return res
