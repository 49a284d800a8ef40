"""Dynamic per-tensor int8 quantization of activations, feeding the int8 MFMA GEMM.

The reference configures such a quantizer on every layer (`Quantizer.input_quantizer`: asymmetric,
per tensor, quant/Weight_Only/Quantizer.py:38-44) but never runs it: its call sites are commented
out (Quantizer.py:69-70, 88-90) and `Qop.quantize` raises for affine='tensor'
(Qop.compute_scales_zero_point_dimension indexes `tensor.shape[None]`, Qop.py:110).  This module is
what those lines compute when followed through with the reference's own formulas, for the signed
int8 range the MFMA path consumes:

    min, max            over the whole tensor                         (Qop.py:97-99)
    scale, zero_point   Qop.calculate_params, per-tensor affine branch  (Qop.py:40-41, 62-65)
    q                   round(x / scale + zero_point).clamp(-128, 127)  (Qop.py:157)

and the layer forward `scale_w[n] * (x_hat . Wq^T + b[n])` of the weight-only Quantizer
(Quantizer.py:79-86) with x_hat = scale_x * (q - zero_point) in place of x:

    y[m, n] = (acc[m, n] + (b[n] / scale_x - zero_point * rowsum(Wq)[n])) * (scale_w[n] * scale_x)

which is the GEMM's fused epilogue with a per-call bias and alpha.  Nothing here synchronises with
the host: the two statistics, the parameters and the epilogue vectors stay on the device.

A convolution is the same expression once its padding is made explicit: a padded position holds the
real value 0, whose quantized form is the zero point, not 0, so the quantized image is framed with
zero_point and convolved without padding; every window is then full and the correction term stays
the per-channel constant zero_point * sum(Wq[n]).
"""
import torch

from . import _native

Q_MIN, Q_MAX = -128, 127
EPS = 2.0 ** -23   # torch.finfo(torch.float32).eps, the reference's floor for a scale (Qop.py:22)


@torch.no_grad()
def tensor_params_i8(x):
    """(scale [1] fp32, zero_point [1] int32) of x for the int8 range, on x's device: one launch, no host synchronisation."""
    scale_zp, _, _ = _native.dynamic_quant_params(x.contiguous())
    return scale_zp[0:1], scale_zp[2:3].view(torch.int32)


@torch.no_grad()
def quantize_tensor_i8(x):
    """x (float) -> (q int8 of x's shape, scale [1], zero_point [1]).  Two launches: statistics + parameters, quantize."""
    q, scale_zp, _, _ = _native.dynamic_quantize_i8(x.contiguous())   # one call into the library
    return q, scale_zp[0:1], scale_zp[2:3].view(torch.int32)


@torch.no_grad()
def weight_rowsum(wq):
    """sum_k Wq[n, k] as fp32 (exact: |sum| < 2^24 up to K = 131072); compute once per layer."""
    return wq.reshape(wq.shape[0], -1).sum(dim=1, dtype=torch.int32).to(torch.float32)


@torch.no_grad()
def linear_dynamic_w8a8(x, wq, w_scale, bias=None, w_rowsum=None, out_dtype=torch.float32):
    """Weight-only quantized Linear on a float activation, through the int8 GEMM: x is quantized per
    tensor on the fly and the zero-point correction, both scales and the bias ride the fused epilogue.
    Three launches, nothing on the host: statistics + parameters + epilogue vectors, quantize, GEMM."""
    _native.require_device(x, "input")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    if w_rowsum is None:
        w_rowsum = weight_rowsum(wq)
    y = _native.linear_dynamic_w8a8(x2, wq, w_scale, bias, w_rowsum, out_dtype)   # one call into the library
    return y.view(*lead, y.shape[-1])


@torch.no_grad()
def conv2d_dynamic_w8a8(x, wq, w_scale, bias=None, w_rowsum=None, stride=1, padding=0, out_dtype=torch.float32,
                        dilation=1, groups=1, packed=None):
    """Weight-only quantized Conv2d on a float activation x [B, Cin, H, W] (or one image [Cin, H, W]), through the int8
    convolution: x is quantized per tensor on the fly straight into a frame filled with its zero point, and the zero-point
    correction, both scales and the bias ride the fused epilogue.  Nothing on the host.  stride / padding / dilation /
    groups as nn.Conv2d takes them; anything but one stride, one numeric padding, dilation 1 and groups 1 runs on the
    general convolution entry point.  packed: conv2d_i8_pack_weight(wq), made once per layer."""
    _native.require_device(x, "input")
    if x.dim() == 3:   # F.conv2d also takes one image
        return conv2d_dynamic_w8a8(x.unsqueeze(0), wq, w_scale, bias, w_rowsum, stride, padding, out_dtype, dilation, groups,
                                   packed).squeeze(0)
    if x.dim() != 4:
        raise RuntimeError(f"input must be [B, C, H, W] or [C, H, W] (got {tuple(x.shape)})")
    x = x.contiguous()
    if w_rowsum is None:
        w_rowsum = weight_rowsum(wq)
    pad = _native.conv_padding(padding, wq.shape[2], wq.shape[3], stride, dilation)
    xq, _, alpha, corr = _native.dynamic_quantize_i8(x, pad, w_scale.reshape(-1), bias, w_rowsum)
    sh, sw = _native._pair(stride)
    dh, dw = _native._pair(dilation)
    if sh == sw and dh == dw == 1 and groups == 1:
        return _native.conv2d_i8(xq, wq, sh, 0, bias=corr, alpha=alpha, beta=0.0, out_dtype=out_dtype, packed=packed)
    return _native.conv2d_general(xq, wq, (sh, sw), 0, (dh, dw), groups, bias=corr, alpha=alpha, beta=0.0,
                                  out_dtype=out_dtype)
