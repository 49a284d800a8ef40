"""ctypes binding of libmiq.so, the C-ABI library holding the HIP kernels (include/miq.h).

PyTorch is used for device memory and streams only: tensors are allocated with torch, and
their data pointers plus torch's current HIP stream are handed to the C entry points.
There is no CPU implementation behind these functions: without the library or without a
GPU they raise.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmiq.so")

MIQ_F32, MIQ_F16, MIQ_BF16, MIQ_I32, MIQ_I8 = 0, 1, 2, 3, 4
MIQ_POOL_MAX, MIQ_POOL_AVG = 0, 1
MIQ_W_NK, MIQ_W_KN = 0, 1
MIQ_OUT_MN, MIQ_OUT_NM = 0, 1

_FLOAT_CODES = {torch.float32: MIQ_F32, torch.float16: MIQ_F16, torch.bfloat16: MIQ_BF16}
_OUT_CODES = {torch.float32: MIQ_F32, torch.bfloat16: MIQ_BF16, torch.int32: MIQ_I32}
_OUT_CODES_CL = {**_OUT_CODES, torch.int8: MIQ_I8}

_i64 = ctypes.c_int64
_p = ctypes.c_void_p
_int = ctypes.c_int
_size = ctypes.c_size_t

SYMBOLS = {
    # name: (restype, argtypes)
    "miq_version": (_int, []),
    "miq_last_error": (ctypes.c_char_p, []),
    "miq_quantize_rows_sym_i8": (_int, [_p, _int, _i64, _i64, _p, _p, _p, _p, _p]),
    "miq_row_max": (_int, [_p, _int, _i64, _i64, _p, _p]),
    "miq_quantize_rows_given_i8": (_int, [_p, _int, _i64, _i64, _p, _p, _int, _p, _p]),
    "miq_dequantize_channels_f32": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p]),
    "miq_dequantize_rows_i8": (_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "miq_epilogue_channels_f32": (_int, [_p, _p, _p, ctypes.c_float, _i64, _i64, _i64, _p, _p]),
    "miq_gemm_i8_workspace": (_size, [_int, _i64, _i64, _i64]),
    "miq_gemm_i8": (_int, [_p, _p, _int, _i64, _i64, _i64, _p, _p, ctypes.c_float, _p, _int, _p, _size, _p]),
    "miq_gemm_i8_ex_workspace": (_size, [_int, _int, _i64, _i64, _i64]),
    "miq_gemm_i8_ex": (_int, [_p, _p, _int, _i64, _i64, _i64, _p, _p, ctypes.c_float, _p, _int, _int, _i64, ctypes.c_float,
                              _p, _size, _p]),
    "miq_gemm_i8_noalloc_max_workspace": (_size, []),
    "miq_gemm_i8_noalloc": (_int, [_p, _p, _int, _i64, _i64, _i64, _p, _p, ctypes.c_float, _p, _int, _p]),
    "miq_gemm_w8_workspace": (_size, [_int, _i64, _i64, _i64]),
    "miq_gemm_w8": (_int, [_p, _int, _p, _i64, _i64, _i64, _p, _p, ctypes.c_float, _p, _int, _p, _size, _p]),
    "miq_conv2d_w8_workspace": (_size, [_int] + [_i64] * 9),
    "miq_conv2d_w8": (_int, [_p, _int, _p] + [_i64] * 9 + [_p, _p, ctypes.c_float, _p, _int, _p, _size, _p]),
    "miq_transpose_i8": (_int, [_p, _i64, _i64, _p, _p]),
    "miq_conv2d_i8_workspace": (_size, [_i64] * 9),
    "miq_conv2d_i8": (_int, [_p, _p] + [_i64] * 9 + [_p, _p, ctypes.c_float, _p, _int, _p, _size, _p]),
    "miq_conv2d_i8_prepacked_workspace": (_size, [_i64] * 9),
    "miq_conv2d_i8_prepacked": (_int, [_p, _p] + [_i64] * 9 + [_p, _p, ctypes.c_float, _p, _int, _p, _size, _p]),
    "miq_conv2d_w8_packed_weight_bytes": (_size, [_int] + [_i64] * 4),
    "miq_conv2d_w8_pack_weight": (_int, [_p, _int] + [_i64] * 4 + [_p, _p]),
    "miq_conv2d_w8_prepacked_workspace": (_size, [_int] + [_i64] * 9),
    "miq_conv2d_w8_prepacked": (_int, [_p, _int, _p] + [_i64] * 9 + [_p, _p, ctypes.c_float, _p, _int, _p, _size, _p]),
    "miq_conv2d_i8_packed_weight_bytes": (_size, [_i64] * 4),
    "miq_conv2d_i8_pack_weight": (_int, [_p] + [_i64] * 4 + [_p, _p]),
    "miq_conv2d_i8_nhwc": (_int, [_p, _p] + [_i64] * 9 + [_p, _p, ctypes.c_float, _p, _int, _i64, _p]),
    "miq_conv2d_general_workspace": (_size, [_int] + [_i64] * 16),
    "miq_conv2d_general": (_int, [_p, _int, _p] + [_i64] * 16 + [_p, _p, ctypes.c_float, _p, _int, _p, _size, _p]),
    "miq_pool2d_to_i8": (_int, [_p, _int] + [_i64] * 4 + [_int] * 7 + [_p, _p]),
    "miq_tensor_minmax_workspace": (_size, [_i64]),
    "miq_tensor_minmax": (_int, [_p, _int, _i64, _p, _p, _size, _p]),
    "miq_dynamic_epilogue_i8": (_int, [_p, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "miq_dynamic_quant_workspace": (_size, [_i64]),
    "miq_dynamic_quant_params_i8": (_int, [_p, _int, _i64, _p, _p, _p, _i64, _p, _p, _p, _p, _size, _p, _p]),
    "miq_quantize_tensor_i8": (_int, [_p, _int] + [_i64] * 7 + [_p, _p, _p]),
    "miq_dynamic_quantize_i8": (_int, [_p, _int] + [_i64] * 7 + [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _size, _p, _p]),
    "miq_linear_dynamic_w8a8_workspace": (_size, [_i64] * 3),
    "miq_linear_dynamic_w8a8": (_int, [_p, _int, _p] + [_i64] * 3 + [_p, _p, _p, _p, _int, _p, _size, _p, _p]),
}

_lib = None


def load(path):
    """A library built from this package's sources, with the argument types of include/miq.h set."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m quantization_amd.build` "
            "(the quantization hot path has no CPU fallback)")
    L = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def lib():
    """The loaded library; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"miq error {rc}: {lib().miq_last_error().decode()}")


def _stream():
    return _p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return _p(t.data_ptr()) if t is not None else _p(0)


def require_device(t, name):
    """Same checks, same wording, as CHECK_INPUT in the reference's linear_w8a8.cpp:7-9."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


class _on:
    """Makes `device` current for the duration of a call; free when it already is (the usual case:
    torch.cuda.device() costs several microseconds per entry, a decode-size kernel ~15)."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx, self.prev = device.index, None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if self.idx is not None and self.idx != cur:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def _f32_vec(t, n, name, device):
    if t is None:
        return None
    if t.dtype is torch.float32 and t.device == device and t.is_contiguous() and t.numel() == n:
        return t   # only its address is used: any dense shape with n elements will do
    t = t.detach().reshape(-1)
    if t.numel() != n:
        raise RuntimeError(f"{name} must have {n} elements, got {t.numel()}")
    return t.to(device=device, dtype=torch.float32).contiguous()


def quantize_rows_sym_i8(w, want_row_max=False):
    """w: [N, ...] float tensor on the GPU -> (q int8 like w, scale fp32 [N], zp int64 [N][, row_max])."""
    require_device(w, "tensor")
    if w.dtype not in _FLOAT_CODES:
        raise RuntimeError(f"unsupported weight dtype {w.dtype}")
    if w.dim() < 1:
        raise RuntimeError("tensor must have at least one dimension")
    rows = w.shape[0]
    cols = w.numel() // rows if rows else 0
    q = torch.empty(w.shape, dtype=torch.int8, device=w.device)
    scale = torch.empty(rows, dtype=torch.float32, device=w.device)
    zp = torch.empty(rows, dtype=torch.int64, device=w.device)
    row_max = torch.empty(rows, dtype=torch.float32, device=w.device) if want_row_max else None
    with _on(w.device):
        _check(lib().miq_quantize_rows_sym_i8(_ptr(w), _FLOAT_CODES[w.dtype], rows, cols, _ptr(q), _ptr(scale),
                                              _ptr(zp), _ptr(row_max), _stream()))
    return (q, scale, zp, row_max) if want_row_max else (q, scale, zp)


def row_max(w):
    """Signed maximum of every w[n, ...] as fp32 [N]."""
    require_device(w, "tensor")
    if w.dtype not in _FLOAT_CODES:
        raise RuntimeError(f"unsupported weight dtype {w.dtype}")
    rows = w.shape[0]
    cols = w.numel() // rows if rows else 0
    out = torch.empty(rows, dtype=torch.float32, device=w.device)
    with _on(w.device):
        _check(lib().miq_row_max(_ptr(w), _FLOAT_CODES[w.dtype], rows, cols, _ptr(out), _stream()))
    return out


def quantize_rows_given_i8(w, scale, zp=None, out=None):
    """q = int8(clamp(rint(w / scale + zp))) with scale/zp of `rows` entries or one."""
    require_device(w, "tensor")
    if w.dtype not in _FLOAT_CODES:
        raise RuntimeError(f"unsupported weight dtype {w.dtype}")
    rows = w.shape[0]
    cols = w.numel() // rows if rows else 0
    scale = scale.detach().reshape(-1).to(device=w.device, dtype=torch.float32).contiguous()
    if scale.numel() not in (1, rows):
        raise RuntimeError(f"scales must have 1 or {rows} entries, got {scale.numel()}")
    per_row = int(scale.numel() == rows and rows != 1)
    if zp is not None:
        zp = zp.detach().reshape(-1).to(device=w.device, dtype=torch.float32).contiguous()
        if zp.numel() != scale.numel():
            raise RuntimeError("zero_point and scales must have the same number of entries")
    if out is None:
        q = torch.empty(w.shape, dtype=torch.int8, device=w.device)
    else:
        require_device(out, "out")
        if out.numel() != w.numel() or out.dtype != torch.int8 or out.device != w.device:
            raise RuntimeError("out has the wrong size, dtype or device")
        q = out
    with _on(w.device):
        _check(lib().miq_quantize_rows_given_i8(_ptr(w), _FLOAT_CODES[w.dtype], rows, cols, _ptr(scale), _ptr(zp),
                                                per_row, _ptr(q), _stream()))
    return q


def dequantize_channels_f32(y, scale, zp=None):
    """scale[c] * (y - zp[c]) along axis 1 of a float32 tensor y."""
    require_device(y, "quantized_tensor")
    y = y.to(torch.float32)
    outer, channels = y.shape[0], y.shape[1]
    inner = y.numel() // max(outer * channels, 1)
    scale = _f32_vec(scale, channels, "scales", y.device)
    zp = _f32_vec(zp, channels, "zero_point", y.device)
    out = torch.empty_like(y)
    with _on(y.device):
        _check(lib().miq_dequantize_channels_f32(_ptr(y), _ptr(scale), _ptr(zp), outer, channels, inner, _ptr(out),
                                                 _stream()))
    return out


def dequantize_rows_i8(q, scale, zp=None):
    require_device(q, "quantized_tensor")
    rows = q.shape[0]
    cols = q.numel() // rows if rows else 0
    scale = _f32_vec(scale, rows, "scales", q.device)
    if zp is not None:
        zp = zp.detach().reshape(-1).to(device=q.device, dtype=torch.int64).contiguous()
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    with _on(q.device):
        _check(lib().miq_dequantize_rows_i8(_ptr(q), _ptr(scale), _ptr(zp), rows, cols, _ptr(out), _stream()))
    return out


def transpose_i8(src):
    require_device(src, "src")
    rows, cols = src.shape
    dst = torch.empty((cols, rows), dtype=torch.int8, device=src.device)
    with _on(src.device):
        _check(lib().miq_transpose_i8(_ptr(src), rows, cols, _ptr(dst), _stream()))
    return dst


def tensor_minmax(x):
    """[min(x), max(x)] as a 2-element fp32 tensor on x's device (no host synchronisation)."""
    require_device(x, "tensor")
    if x.dtype not in _FLOAT_CODES:
        raise RuntimeError(f"tensor must be float32, float16 or bfloat16 (got {x.dtype})")
    n = x.numel()
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    need = lib().miq_tensor_minmax_workspace(n)
    ws = torch.empty(max(need, 4), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        _check(lib().miq_tensor_minmax(_ptr(x), _FLOAT_CODES[x.dtype], n, _ptr(out), _ptr(ws), ws.numel(), _stream()))
    return out


def dynamic_epilogue_i8(min_max, w_scale, bias, w_rowsum):
    """(scale_zp [2], alpha [N], corr [N]) for a weight-only layer run on a per-tensor quantized activation."""
    require_device(min_max, "min_max")
    n = w_scale.numel()
    dev = min_max.device
    w_scale = w_scale.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
    w_rowsum = w_rowsum.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
    if bias is not None:
        bias = bias.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty(2 + 2 * n, dtype=torch.float32, device=dev)
    scale_zp, alpha, corr = out[:2], out[2:2 + n], out[2 + n:]
    with _on(dev):
        _check(lib().miq_dynamic_epilogue_i8(_ptr(min_max), _ptr(w_scale), _ptr(bias), _ptr(w_rowsum), n, _ptr(scale_zp),
                                             _ptr(alpha), _ptr(corr), _stream()))
    return scale_zp, alpha, corr


_counters = {}   # (device index, stream) -> a zeroed int32 word for miq_dynamic_quant_params_i8


def _counter(device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    c = _counters.get(key)
    if c is None:
        c = _counters[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return c


def dynamic_quant_params(x, w_scale=None, bias=None, w_rowsum=None):
    """One launch: min / max of x, the per-tensor affine int8 parameters and (with w_scale / w_rowsum given) the epilogue
    vectors of a weight-only layer run on the quantized activation.  Returns (scale_zp [3] fp32: scale, zero point, the
    zero point's int32 bits; alpha [N] or None; corr [N] or None).  No host synchronisation."""
    require_device(x, "tensor")
    if x.dtype not in _FLOAT_CODES:
        raise RuntimeError(f"tensor must be float32, float16 or bfloat16 (got {x.dtype})")
    if x.numel() == 0:
        raise RuntimeError("the tensor is empty")
    dev = x.device
    n = 0
    if w_scale is not None:
        n = w_scale.numel()
        w_scale = _f32_vec(w_scale, n, "w_scale", dev)
        w_rowsum = _f32_vec(w_rowsum, n, "w_rowsum", dev)
        bias = _f32_vec(bias, n, "bias", dev)
    out = torch.empty(4 + 2 * n, dtype=torch.float32, device=dev)
    scale_zp, alpha, corr = out[:3], (out[4:4 + n] if n else None), (out[4 + n:] if n else None)
    L = lib()
    need = L.miq_dynamic_quant_workspace(x.numel())
    ws = torch.empty(max(need, 4), dtype=torch.uint8, device=dev)
    with _on(dev):
        _check(L.miq_dynamic_quant_params_i8(_ptr(x), _FLOAT_CODES[x.dtype], x.numel(), _ptr(w_scale), _ptr(bias), _ptr(w_rowsum),
                                             n, _ptr(scale_zp), _ptr(alpha), _ptr(corr), _ptr(ws), ws.numel(), _ptr(_counter(dev)),
                                             _stream()))
    return scale_zp, alpha, corr


def quantize_tensor_i8(x, scale_zp, pad=(0, 0, 0, 0)):
    """q = int8(clamp(rint(x / scale_zp[0] + scale_zp[1]))) with the parameters read on the device.  pad = (top, bottom, left,
    right): x is then [..., H, W] and q gets a frame of that width filled with the zero point."""
    require_device(x, "tensor")
    if x.dtype not in _FLOAT_CODES:
        raise RuntimeError(f"tensor must be float32, float16 or bfloat16 (got {x.dtype})")
    pt, pb, pl, pr = (int(v) for v in pad)
    if pt or pb or pl or pr:
        if x.dim() < 2:
            raise RuntimeError("a framed tensor needs at least two dimensions")
        H, W = x.shape[-2], x.shape[-1]
        planes = x.numel() // max(H * W, 1)
        q = torch.empty((*x.shape[:-2], H + pt + pb, W + pl + pr), dtype=torch.int8, device=x.device)
    else:
        planes, H, W = 1, 1, x.numel()
        q = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    with _on(x.device):
        _check(lib().miq_quantize_tensor_i8(_ptr(x), _FLOAT_CODES[x.dtype], planes, H, W, pt, pb, pl, pr, _ptr(scale_zp), _ptr(q),
                                            _stream()))
    return q


def dynamic_quantize_i8(x, pad=(0, 0, 0, 0), w_scale=None, bias=None, w_rowsum=None):
    """dynamic_quant_params + quantize_tensor_i8 as one call into the library (two launches): returns (q, scale_zp [3],
    alpha, corr); pad frames the last two dimensions of x with the zero point."""
    require_device(x, "tensor")
    if x.dtype not in _FLOAT_CODES:
        raise RuntimeError(f"tensor must be float32, float16 or bfloat16 (got {x.dtype})")
    if x.numel() == 0:
        raise RuntimeError("the tensor is empty")
    dev = x.device
    n = 0
    if w_scale is not None:
        n = w_scale.numel()
        w_scale = _f32_vec(w_scale, n, "w_scale", dev)
        w_rowsum = _f32_vec(w_rowsum, n, "w_rowsum", dev)
        bias = _f32_vec(bias, n, "bias", dev)
    pt, pb, pl, pr = (int(v) for v in pad)
    if pt or pb or pl or pr:
        H, W = x.shape[-2], x.shape[-1]
        planes = x.numel() // (H * W)
        q = torch.empty((*x.shape[:-2], H + pt + pb, W + pl + pr), dtype=torch.int8, device=dev)
    else:
        planes, H, W = 1, 1, x.numel()
        q = torch.empty(x.shape, dtype=torch.int8, device=dev)
    L = lib()
    need = L.miq_dynamic_quant_workspace(x.numel())
    nf = 4 + 2 * n
    buf = torch.empty(nf + (need + 3) // 4, dtype=torch.float32, device=dev)   # parameters, vectors and the workspace in one
    scale_zp, alpha, corr = buf[:3], (buf[4:4 + n] if n else None), (buf[4 + n:nf] if n else None)
    with _on(dev):
        _check(L.miq_dynamic_quantize_i8(_ptr(x), _FLOAT_CODES[x.dtype], planes, H, W, pt, pb, pl, pr, _ptr(w_scale), _ptr(bias),
                                         _ptr(w_rowsum), n, _ptr(scale_zp), _ptr(alpha), _ptr(corr), _ptr(q),
                                         _p(buf.data_ptr() + 4 * nf), need, _ptr(_counter(dev)), _stream()))
    return q, scale_zp, alpha, corr


def linear_dynamic_w8a8(x, wq, w_scale, bias, w_rowsum, out_dtype=torch.float32):
    """Weight-only Linear on a float activation x [M, K] through the int8 product with a per-tensor dynamic activation
    quantizer: one call into the library (three launches), nothing on the host."""
    require_device(x, "input")
    require_device(wq, "weights")
    if x.dtype not in _FLOAT_CODES or wq.dtype != torch.int8 or x.dim() != 2 or wq.dim() != 2 or x.shape[1] != wq.shape[1]:
        raise RuntimeError(f"expected a float [M,K] input and an int8 [N,K] weight, got {tuple(x.shape)} {x.dtype} and "
                           f"{tuple(wq.shape)} {wq.dtype}")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"unsupported output dtype {out_dtype}")
    M, K = x.shape
    n = wq.shape[0]
    dev = x.device
    w_scale = _f32_vec(w_scale, n, "w_scale", dev)
    w_rowsum = _f32_vec(w_rowsum, n, "w_rowsum", dev)
    bias = _f32_vec(bias, n, "bias", dev)
    out = torch.empty((M, n), dtype=out_dtype, device=dev)
    if M == 0 or n == 0:
        return out
    L = lib()
    need = L.miq_linear_dynamic_w8a8_workspace(M, n, K)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    with _on(dev):
        _check(L.miq_linear_dynamic_w8a8(_ptr(x), _FLOAT_CODES[x.dtype], _ptr(wq), M, n, K, _ptr(w_scale), _ptr(bias),
                                         _ptr(w_rowsum), _ptr(out), _OUT_CODES[out_dtype], _ptr(ws), need, _ptr(_counter(dev)),
                                         _stream()))
    return out


def pool2d_to_i8(x, kernel_size, stride=None, padding=0, mode="max"):
    """Max / average pooling of x [B, C, H, W] (float or int8) with an int8 result, as the reference's
    QuantizedMaxPool2d / QuantizedAvgPool2d (pool x.float(), then .to(torch.int8))."""
    require_device(x, "input")
    if x.dim() != 4:
        raise RuntimeError(f"input must be [B, C, H, W] (got {tuple(x.shape)})")
    code = MIQ_I8 if x.dtype == torch.int8 else _FLOAT_CODES.get(x.dtype)
    if code is None:
        raise RuntimeError(f"input must be float32, float16, bfloat16 or int8 (got {x.dtype})")

    def pair(v):
        return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))
    kh, kw = pair(kernel_size)
    sh, sw = pair(kernel_size if stride is None else stride)
    ph, pw = pair(padding)
    B, C, H, W = x.shape
    if H + 2 * ph < kh or W + 2 * pw < kw:
        raise RuntimeError("pooling window larger than the padded input")
    out = torch.empty((B, C, (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1), dtype=torch.int8, device=x.device)
    with _on(x.device):
        _check(lib().miq_pool2d_to_i8(_ptr(x), code, B, C, H, W, kh, kw, sh, sw, ph, pw,
                                      MIQ_POOL_MAX if mode == "max" else MIQ_POOL_AVG, _ptr(out), _stream()))
    return out


def gemm_i8(a, w, w_layout=MIQ_W_NK, bias=None, alpha=None, beta=0.0, out_dtype=torch.float32, out=None,
            transpose_out=False, requant_scale=1.0):
    """out[m,n] = (float(sum_k a[m,k]*w[n,k]) + bias[n]) * alpha[n] + beta, or the raw int32 sums.

    The forms a column-sharded layer uses (miq_gemm_i8_ex): `out` may be a column block of a wider row-major matrix (a 2-D
    view with unit column stride); transpose_out=True writes the transposed result out[n, m] (shape [N, M]);
    out_dtype=torch.int8 requantises, clamp(rint(y / requant_scale)) with the IEEE division of Qop.quantize."""
    require_device(a, "input")
    require_device(w, "weights")
    if a.dtype != torch.int8 or w.dtype != torch.int8:
        raise RuntimeError(f"input and weights must be int8 (got {a.dtype}, {w.dtype})")
    if a.dim() != 2 or w.dim() != 2:
        raise RuntimeError("input and weights must be 2-D")
    M, K = a.shape
    if w_layout == MIQ_W_NK:
        N, Kw = w.shape
    else:
        Kw, N = w.shape
    if K != Kw:
        raise RuntimeError(f"shape mismatch: input is [{M},{K}], weights contract over {Kw}")
    if out_dtype not in _OUT_CODES_CL:
        raise RuntimeError(f"unsupported output dtype {out_dtype}")
    bias = _f32_vec(bias, N, "bias", a.device)
    alpha = _f32_vec(alpha, N, "alpha", a.device)
    shape = (N, M) if transpose_out else (M, N)
    if out is None:
        out = torch.empty(shape, dtype=out_dtype, device=a.device)
    else:
        if not out.is_cuda:
            raise RuntimeError("out must be a CUDA tensor")
        if tuple(out.shape) != shape or out.dtype != out_dtype or out.device != a.device:
            raise RuntimeError("out has the wrong shape, dtype or device")
        if out.dim() != 2 or (out.numel() and (out.stride(1) != 1 or out.stride(0) < shape[1])):
            raise RuntimeError("out must be dense or a column block of a row-major matrix (unit column stride)")
    ldo = out.stride(0) if (out.numel() and shape[0] > 1) else shape[1]
    L = lib()
    if not transpose_out and out_dtype != torch.int8 and ldo == shape[1]:
        need = L.miq_gemm_i8_workspace(w_layout, M, N, K)
        ws = torch.empty(need, dtype=torch.int8, device=a.device) if need else None
        with _on(a.device):
            _check(L.miq_gemm_i8(_ptr(a), _ptr(w), w_layout, M, N, K, _ptr(bias), _ptr(alpha), float(beta), _ptr(out),
                                 _OUT_CODES[out_dtype], _ptr(ws), need, _stream()))
        return out
    layout = MIQ_OUT_NM if transpose_out else MIQ_OUT_MN
    need = L.miq_gemm_i8_ex_workspace(w_layout, layout, M, N, K)
    ws = torch.empty(need, dtype=torch.int8, device=a.device) if need else None
    with _on(a.device):
        _check(L.miq_gemm_i8_ex(_ptr(a), _ptr(w), w_layout, M, N, K, _ptr(bias), _ptr(alpha), float(beta), _ptr(out),
                                _OUT_CODES_CL[out_dtype], layout, ldo, float(requant_scale), _ptr(ws), need, _stream()))
    return out


def gemm_w8(x, w, bias=None, alpha=None, beta=0.0, out_dtype=torch.float32, out=None):
    """Weight-only Linear: out[m,n] = (sum_k x[m,k]*float(w[n,k]) + bias[n]) * alpha[n] + beta for floating-point
    x [M,K] (fp32 / fp16 / bf16) and int8 w [N,K]."""
    require_device(x, "input")
    require_device(w, "weights")
    if x.dtype not in _FLOAT_CODES or w.dtype != torch.int8:
        raise RuntimeError(f"input must be fp32/fp16/bf16 and weights int8 (got {x.dtype}, {w.dtype})")
    if x.dim() != 2 or w.dim() != 2:
        raise RuntimeError("input and weights must be 2-D")
    M, K = x.shape
    N, Kw = w.shape
    if K != Kw:
        raise RuntimeError(f"shape mismatch: input is [{M},{K}], weights contract over {Kw}")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"unsupported output dtype {out_dtype}")
    if alpha is None:
        raise RuntimeError("alpha (the per-channel scale) is required")
    bias = _f32_vec(bias, N, "bias", x.device)
    alpha = _f32_vec(alpha, N, "alpha", x.device)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=x.device)
    else:
        require_device(out, "out")
        if out.shape != (M, N) or out.dtype != out_dtype:
            raise RuntimeError("out has the wrong shape or dtype")
    L = lib()
    code = _FLOAT_CODES[x.dtype]
    need = L.miq_gemm_w8_workspace(code, M, N, K)
    ws = torch.empty(need, dtype=torch.int8, device=x.device) if need else None
    with _on(x.device):
        _check(L.miq_gemm_w8(_ptr(x), code, _ptr(w), M, N, K, _ptr(bias), _ptr(alpha), float(beta), _ptr(out),
                             _OUT_CODES[out_dtype], _ptr(ws), need, _stream()))
    return out


def conv2d_w8_pack_weight(w, x_dtype):
    """OIHW int8 weight -> the packed form miq_conv2d_w8_prepacked stages from for activations of `x_dtype` (once per layer)."""
    require_device(w, "kernel")
    if w.dtype != torch.int8 or w.dim() != 4 or x_dtype not in _FLOAT_CODES:
        raise RuntimeError("kernel must be a 4-D int8 tensor (OIHW) and x_dtype fp32 / fp16 / bf16")
    Co, Ci, KH, KW = w.shape
    L = lib()
    code = _FLOAT_CODES[x_dtype]
    wp = torch.empty(max(L.miq_conv2d_w8_packed_weight_bytes(code, Ci, Co, KH, KW), 256), dtype=torch.int8, device=w.device)
    with _on(w.device):
        _check(L.miq_conv2d_w8_pack_weight(_ptr(w), code, Ci, Co, KH, KW, _ptr(wp), _stream()))
    return wp


def conv2d_w8_prepack(w, x_dtype, x_shape=None, stride=1, padding=0):
    """The `packed=` argument of conv2d_w8 for this filter and activation type (x_shape, stride and padding are not
    needed: the packed weight depends on the filter's shape and the activation type only)."""
    return conv2d_w8_pack_weight(w, x_dtype)


def conv2d_w8(x, w, stride=1, padding=0, bias=None, alpha=None, beta=0.0, out_dtype=torch.float32, out=None, packed=None):
    """Weight-only Conv2d: floating-point NCHW x (fp32 / fp16 / bf16), int8 OIHW w, per-channel (acc + bias) * alpha + beta.
    packed: conv2d_w8_pack_weight(w, x.dtype), made once per layer; the call then skips the filter re-layout."""
    require_device(x, "input")
    require_device(w, "kernel")
    if x.dtype not in _FLOAT_CODES or w.dtype != torch.int8:
        raise RuntimeError(f"input must be fp32/fp16/bf16 and kernel int8 (got {x.dtype}, {w.dtype})")
    if x.dim() != 4 or w.dim() != 4 or x.shape[1] != w.shape[1]:
        raise RuntimeError(f"expected NCHW input and OIHW kernel with equal channels, got {tuple(x.shape)} and {tuple(w.shape)}")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"unsupported output dtype {out_dtype}")
    if alpha is None:
        raise RuntimeError("alpha (the per-channel scale) is required")
    B, C, H, W = x.shape
    Co, _, KH, KW = w.shape
    stride, padding = int(stride), int(padding)
    if H + 2 * padding < KH or W + 2 * padding < KW or stride < 1 or padding < 0:
        raise RuntimeError("kernel larger than the padded input, or bad stride / padding")
    Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    bias = _f32_vec(bias, Co, "bias", x.device)
    alpha = _f32_vec(alpha, Co, "alpha", x.device)
    if out is None:
        out = torch.empty((B, Co, Ho, Wo), dtype=out_dtype, device=x.device)
    else:
        require_device(out, "out")
        if tuple(out.shape) != (B, Co, Ho, Wo) or out.dtype != out_dtype:
            raise RuntimeError("out has the wrong shape or dtype")
    L = lib()
    code = _FLOAT_CODES[x.dtype]
    if packed is not None:
        require_device(packed, "packed kernel")
        if packed.dtype != torch.int8 or packed.numel() < L.miq_conv2d_w8_packed_weight_bytes(code, C, Co, KH, KW):
            raise RuntimeError("packed kernel does not belong to this filter shape / activation type")
        need = L.miq_conv2d_w8_prepacked_workspace(code, B, C, H, W, Co, KH, KW, stride, padding)
        ws = torch.empty(max(need, 1), dtype=torch.int8, device=x.device)
        with _on(x.device):
            _check(L.miq_conv2d_w8_prepacked(_ptr(x), code, _ptr(packed), B, C, H, W, Co, KH, KW, stride, padding, _ptr(bias),
                                             _ptr(alpha), float(beta), _ptr(out), _OUT_CODES[out_dtype], _ptr(ws), need, _stream()))
        return out
    need = L.miq_conv2d_w8_workspace(code, B, C, H, W, Co, KH, KW, stride, padding)
    ws = torch.empty(need, dtype=torch.int8, device=x.device) if need else None
    with _on(x.device):
        _check(L.miq_conv2d_w8(_ptr(x), code, _ptr(w), B, C, H, W, Co, KH, KW, stride, padding, _ptr(bias), _ptr(alpha),
                               float(beta), _ptr(out), _OUT_CODES[out_dtype], _ptr(ws), need, _stream()))
    return out


def conv2d_i8_prepack(w, x_shape=None, stride=1, padding=0):
    """The `packed=` argument of conv2d_i8 for this filter (the packed weight depends on the filter's shape only)."""
    return conv2d_i8_pack_weight(w)


def conv2d_i8(x, w, stride=1, padding=0, bias=None, alpha=None, beta=0.0, out_dtype=torch.float32, out=None, packed=None):
    """NCHW int8 x, OIHW int8 w -> NCHW out with the per-channel epilogue of gemm_i8.
    packed: conv2d_i8_pack_weight(w), made once per layer; the call then skips the filter re-layout."""
    require_device(x, "input")
    require_device(w, "kernel")
    if x.dtype != torch.int8 or w.dtype != torch.int8:
        raise RuntimeError(f"input and kernel must be int8 (got {x.dtype}, {w.dtype})")
    if x.dim() != 4 or w.dim() != 4:
        raise RuntimeError("input and kernel must be 4-D (NCHW / OIHW)")
    B, C, H, W = x.shape
    Co, Ci, KH, KW = w.shape
    if C != Ci:
        raise RuntimeError(f"shape mismatch: input has {C} channels, kernel expects {Ci}")
    if out_dtype not in _OUT_CODES:
        raise RuntimeError(f"unsupported output dtype {out_dtype}")
    stride, padding = int(stride), int(padding)
    if stride < 1 or padding < 0:
        raise RuntimeError(f"bad stride / padding ({stride}, {padding})")
    Ho = (H + 2 * padding - KH) // stride + 1
    Wo = (W + 2 * padding - KW) // stride + 1
    if H + 2 * padding < KH or W + 2 * padding < KW or Ho <= 0 or Wo <= 0:
        raise RuntimeError("kernel larger than the padded input")
    bias = _f32_vec(bias, Co, "bias", x.device)
    alpha = _f32_vec(alpha, Co, "alpha", x.device)
    if out is None:
        out = torch.empty((B, Co, Ho, Wo), dtype=out_dtype, device=x.device)
    else:
        require_device(out, "out")
        if tuple(out.shape) != (B, Co, Ho, Wo) or out.dtype != out_dtype or out.device != x.device:
            raise RuntimeError("out has the wrong shape, dtype or device")
    L = lib()
    if packed is not None:
        require_device(packed, "packed kernel")
        if packed.dtype != torch.int8 or packed.numel() < L.miq_conv2d_i8_packed_weight_bytes(C, Co, KH, KW):
            raise RuntimeError("packed kernel does not belong to this filter shape")
        need = L.miq_conv2d_i8_prepacked_workspace(B, C, H, W, Co, KH, KW, stride, padding)
        ws = torch.empty(max(need, 1), dtype=torch.int8, device=x.device)
        with _on(x.device):
            _check(L.miq_conv2d_i8_prepacked(_ptr(x), _ptr(packed), B, C, H, W, Co, KH, KW, stride, padding, _ptr(bias),
                                             _ptr(alpha), float(beta), _ptr(out), _OUT_CODES[out_dtype], _ptr(ws), need, _stream()))
        return out
    need = L.miq_conv2d_i8_workspace(B, C, H, W, Co, KH, KW, stride, padding)
    ws = torch.empty(max(need, 1), dtype=torch.int8, device=x.device)
    with _on(x.device):
        _check(L.miq_conv2d_i8(_ptr(x), _ptr(w), B, C, H, W, Co, KH, KW, stride, padding, _ptr(bias), _ptr(alpha),
                               float(beta), _ptr(out), _OUT_CODES[out_dtype], _ptr(ws), need, _stream()))
    return out


def conv2d_i8_pack_weight(w):
    """OIHW int8 weight -> the packed form miq_conv2d_i8_nhwc stages from (done once per layer)."""
    require_device(w, "kernel")
    if w.dtype != torch.int8 or w.dim() != 4:
        raise RuntimeError("kernel must be a 4-D int8 tensor (OIHW)")
    Co, Ci, KH, KW = w.shape
    L = lib()
    wp = torch.empty(max(L.miq_conv2d_i8_packed_weight_bytes(Ci, Co, KH, KW), 16), dtype=torch.int8, device=w.device)
    with _on(w.device):
        _check(L.miq_conv2d_i8_pack_weight(_ptr(w), Ci, Co, KH, KW, _ptr(wp), _stream()))
    return wp


def conv2d_i8_nhwc(x, wp, kernel_shape, stride=1, padding=0, bias=None, alpha=None, beta=0.0, out_dtype=torch.int8,
                   out_pad=0, out=None):
    """int8 Conv2d on channels-last activations, no re-layout pass.

    x: int8 [B, H + 2*padding, W + 2*padding, Cin] whose border is present and zero; wp: conv2d_i8_pack_weight(w);
    kernel_shape = w.shape (Cout, Cin, KH, KW).  Returns / fills out [B, Ho + 2*out_pad, Wo + 2*out_pad, Cout]: only the
    interior is written (a fresh buffer is zero-filled here; pass `out` to reuse one whose border is already zero).
    out_dtype int8 requantises: clamp(rint((acc + bias) * alpha + beta), -128, 127)."""
    require_device(x, "input")
    require_device(wp, "packed kernel")
    if x.dtype != torch.int8 or wp.dtype != torch.int8 or x.dim() != 4:
        raise RuntimeError("input must be a 4-D int8 tensor [B, Hp, Wp, Cin] and the packed kernel int8")
    Co, Ci, KH, KW = (int(v) for v in kernel_shape)
    stride, padding, out_pad = int(stride), int(padding), int(out_pad)
    B, Hp, Wp, C = x.shape
    H, W = Hp - 2 * padding, Wp - 2 * padding
    if C != Ci or H <= 0 or W <= 0 or stride < 1 or padding < 0 or out_pad < 0:
        raise RuntimeError(f"input {tuple(x.shape)} does not match the kernel {tuple(kernel_shape)} / padding {padding}")
    if Hp < KH or Wp < KW:
        raise RuntimeError("kernel larger than the padded input")
    if out_dtype not in _OUT_CODES_CL:
        raise RuntimeError(f"unsupported output dtype {out_dtype}")
    Ho, Wo = (Hp - KH) // stride + 1, (Wp - KW) // stride + 1
    bias = _f32_vec(bias, Co, "bias", x.device)
    alpha = _f32_vec(alpha, Co, "alpha", x.device)
    shape = (B, Ho + 2 * out_pad, Wo + 2 * out_pad, Co)
    if out is None:
        out = (torch.zeros if out_pad else torch.empty)(shape, dtype=out_dtype, device=x.device)
    else:
        require_device(out, "out")
        if tuple(out.shape) != shape or out.dtype != out_dtype or out.device != x.device:
            raise RuntimeError("out has the wrong shape, dtype or device")
    with _on(x.device):
        _check(lib().miq_conv2d_i8_nhwc(_ptr(x), _ptr(wp), B, Ci, H, W, Co, KH, KW, stride, padding, _ptr(bias), _ptr(alpha),
                                        float(beta), _ptr(out), _OUT_CODES_CL[out_dtype], out_pad, _stream()))
    return out


def _pair(v):
    if isinstance(v, (tuple, list)):
        if len(v) == 1:
            return int(v[0]), int(v[0])
        if len(v) != 2:
            raise RuntimeError(f"expected one or two values, got {v!r}")
        return int(v[0]), int(v[1])
    return int(v), int(v)


def conv_padding(padding, KH, KW, stride=1, dilation=1):
    """(top, bottom, left, right) zero rows / columns for nn.Conv2d's `padding` argument: an integer, a pair, 'valid', or
    'same' (as F.conv2d: dil * (k - 1) in total per axis, the odd one behind; stride 1 only)."""
    sh, sw = _pair(stride)
    dh, dw = _pair(dilation)
    if isinstance(padding, str):
        if padding == "valid":
            return 0, 0, 0, 0
        if padding == "same":
            if sh != 1 or sw != 1:
                raise RuntimeError("padding='same' is not supported for strided convolutions")
            th, tw = dh * (KH - 1), dw * (KW - 1)
            return th // 2, th - th // 2, tw // 2, tw - tw // 2
        raise RuntimeError(f"unknown padding mode {padding!r}")
    pt, pl = _pair(padding)
    if pt < 0 or pl < 0:
        raise RuntimeError(f"bad padding {padding}")
    return pt, pt, pl, pl


def conv2d_general(x, w, stride=1, padding=0, dilation=1, groups=1, bias=None, alpha=None, beta=0.0,
                   out_dtype=torch.float32, out=None):
    """Conv2d with nn.Conv2d's full argument set (a stride / dilation / padding per axis, groups, padding 'valid' or
    'same'): NCHW x (int8, fp32, fp16 or bf16), int8 w [Cout, Cin / groups, KH, KW], per-channel (acc + bias) * alpha + beta
    (out_dtype int32, int8 x only: the raw sums)."""
    require_device(x, "input")
    require_device(w, "kernel")
    codes = {**_FLOAT_CODES, torch.int8: MIQ_I8}
    if x.dtype not in codes or w.dtype != torch.int8:
        raise RuntimeError(f"input must be int8/fp32/fp16/bf16 and kernel int8 (got {x.dtype}, {w.dtype})")
    if x.dim() != 4 or w.dim() != 4:
        raise RuntimeError("input and kernel must be 4-D (NCHW / OIHW)")
    groups = int(groups)
    B, C, H, W = x.shape
    Co, Cg, KH, KW = w.shape
    if groups < 1 or C % groups or Co % groups or Cg != C // groups:
        raise RuntimeError(f"shape mismatch: input {tuple(x.shape)}, kernel {tuple(w.shape)}, groups {groups}")
    if out_dtype not in _OUT_CODES or (out_dtype == torch.int32 and x.dtype != torch.int8):
        raise RuntimeError(f"unsupported output dtype {out_dtype} for {x.dtype} input")
    if out_dtype != torch.int32 and alpha is None:
        raise RuntimeError("alpha (the per-channel scale) is required")
    sh, sw = _pair(stride)
    dh, dw = _pair(dilation)
    if sh < 1 or sw < 1 or dh < 1 or dw < 1:
        raise RuntimeError(f"bad stride / dilation ({stride}, {dilation})")
    pt, pb, pl, pr = conv_padding(padding, KH, KW, (sh, sw), (dh, dw))
    eh, ew = dh * (KH - 1) + 1, dw * (KW - 1) + 1
    if H + pt + pb < eh or W + pl + pr < ew:
        raise RuntimeError("kernel larger than the padded input")
    Ho, Wo = (H + pt + pb - eh) // sh + 1, (W + pl + pr - ew) // sw + 1
    bias = _f32_vec(bias, Co, "bias", x.device)
    alpha = _f32_vec(alpha, Co, "alpha", x.device)
    if out is None:
        out = torch.empty((B, Co, Ho, Wo), dtype=out_dtype, device=x.device)
    else:
        require_device(out, "out")
        if tuple(out.shape) != (B, Co, Ho, Wo) or out.dtype != out_dtype or out.device != x.device:
            raise RuntimeError("out has the wrong shape, dtype or device")
    L = lib()
    need = L.miq_conv2d_general_workspace(codes[x.dtype], B, C, H, W, Co, KH, KW, sh, sw, pt, pb, pl, pr, dh, dw, groups)
    ws = torch.empty(need, dtype=torch.int8, device=x.device) if need else None
    with _on(x.device):
        _check(L.miq_conv2d_general(_ptr(x), codes[x.dtype], _ptr(w), B, C, H, W, Co, KH, KW, sh, sw, pt, pb, pl, pr, dh, dw,
                                    groups, _ptr(bias), _ptr(alpha), float(beta), _ptr(out), _OUT_CODES[out_dtype], _ptr(ws),
                                    need, _stream()))
    return out


def epilogue_channels_f32(y, bias, alpha, beta=0.0):
    """(y + bias[c]) * alpha[c] + beta along axis 1 of a float tensor y (any float dtype in, fp32 out)."""
    require_device(y, "y")
    y = y.to(torch.float32)
    outer, channels = y.shape[0], y.shape[1]
    inner = y.numel() // max(outer * channels, 1)
    bias = _f32_vec(bias, channels, "bias", y.device)
    alpha = _f32_vec(alpha, channels, "alpha", y.device)
    out = torch.empty_like(y)
    with _on(y.device):
        _check(lib().miq_epilogue_channels_f32(_ptr(y), _ptr(bias), _ptr(alpha), float(beta), outer, channels,
                                               inner, _ptr(out), _stream()))
    return out
