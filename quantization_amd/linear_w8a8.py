"""linear_w8a8(input, weights) -> int32: the reference's native int8 Linear entry point
(cuda_kernels/linear_int8/linear_w8a8.cpp:11-24), served by the CDNA4 MFMA GEMM.

    import quantization_amd.linear_w8a8 as linear_w8a8
    out = linear_w8a8.linear_w8a8(input, weights)      # input int8 [M,K], weights int8 [K,N]
"""
import torch

from . import _native


def linear_w8a8(input, weights):
    """C[m,n] = sum_k int32(input[m,k]) * int32(weights[k,n]); a fresh int32 [M,N] tensor.

    Raises RuntimeError for a non-GPU or non-contiguous argument, as CHECK_INPUT does in the
    reference (linear_w8a8.cpp:7-9), and additionally for a dtype or shape mismatch, which the
    reference leaves unchecked.
    """
    return _native.gemm_i8(input, weights, w_layout=_native.MIQ_W_KN, out_dtype=torch.int32)
