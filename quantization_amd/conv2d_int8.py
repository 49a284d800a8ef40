"""conv2d_int8(input, kernel, output, stride, padding): the reference's native int8 Conv2d entry
point (cuda_kernels/conv2d_int8/conv2d_w8a8.cu:81-119), served by the implicit-GEMM MFMA kernel.
"""
import torch

from . import _native


def conv2d_int8(input, kernel, output, stride, padding):
    """Writes the int32 NCHW convolution of int8 NCHW `input` with int8 OIHW `kernel` into
    `output`, resizing it to [B, Cout, Ho, Wo] first as the reference does (conv2d_w8a8.cu:94).
    Returns None.  Unlike the reference's kernel the stride is applied to the input index
    (F.conv2d semantics, which the reference's own test expects), and the call does not block
    the host.
    """
    _native.require_device(input, "input")
    _native.require_device(kernel, "kernel")
    if output.dtype != torch.int32 or not output.is_cuda:
        raise RuntimeError("output must be an int32 CUDA tensor")
    B, _, H, W = input.shape
    Co, _, KH, KW = kernel.shape
    Ho = (H + 2 * padding - KH) // stride + 1
    Wo = (W + 2 * padding - KW) // stride + 1
    output.resize_((B, Co, Ho, Wo))
    _native.conv2d_i8(input, kernel, stride, padding, out_dtype=torch.int32, out=output)
