"""CPU restatement (numpy) of the reference's int8 quantization hot path.

TEST INFRASTRUCTURE ONLY.  This module is the checker for the HIP kernels: it may be
imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and by
nothing else.  The shipped package (quantization_amd/) never imports it and has no CPU
fallback.

Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py)
against fixtures in tests/golden/ that were produced by running the reference's own
Python (quant/Weight_Only, quant/Weight_Activation) in the build container with
oracle/gen_golden.py.  The two native CUDA kernels of the reference cannot run here (no
CUDA); their restatements follow the kernel source line by line and are pinned by the
reference's own acceptance check, i.e. equality with a float matmul / F.conv2d on integer
inputs (cuda_kernels/linear_int8/test.py:13-15, cuda_kernels/conv2d_int8/test.py:80-97).

All paths cited are relative to the reference checkout.
"""
import numpy as np

# torch.finfo(torch.float32).eps, the scale floor (quant/Weight_Only/Qop.py:22).
EPS_F32 = np.float32(2.0 ** -23)

INT8_MIN = -128
INT8_MAX = 127


# --------------------------------------------------------------------------------------
# reduced-precision helpers (the reference's fp16 / bf16 weights keep their dtype through
# the scale computation: quant/Weight_Only/Qop.py:89, :50-51)
# --------------------------------------------------------------------------------------
def round_f32_to_bf16(x):
    """Round float32 -> bfloat16 (round-to-nearest-even), result returned as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> np.uint64(16)) & np.uint64(1)
    r = ((u + np.uint64(0x7FFF) + lsb) >> np.uint64(16)) << np.uint64(16)
    out = r.astype(np.uint32).view(np.float32)
    nan = np.isnan(x)
    if nan.any():
        out = out.copy()
        out[nan] = np.float32(np.nan)
    return out.reshape(x.shape)


def _round_to(x_f32, kind):
    """Round an fp32 array to the storage precision `kind` and return it as fp32."""
    if kind == "f32":
        return np.asarray(x_f32, dtype=np.float32)
    if kind == "f16":
        return np.asarray(x_f32, dtype=np.float32).astype(np.float16).astype(np.float32)
    if kind == "bf16":
        return round_f32_to_bf16(x_f32)
    raise ValueError(kind)


# --------------------------------------------------------------------------------------
# quantizer
# --------------------------------------------------------------------------------------
def row_signed_max(w):
    """Per-output-channel *signed* maximum over all trailing dims.

    quant/Weight_Only/Qop.py:78-92 (channel branch, symmetric): max_val[n] =
    torch.amax(tensor[n]); min_val = zeros.  It is not an abs-max.
    """
    w = np.asarray(w)
    return w.reshape(w.shape[0], -1).max(axis=1)


def calculate_params_symmetric(max_val, min_val=None, q_min=INT8_MIN, q_max=INT8_MAX,
                               kind="f32"):
    """Symmetric branch of Qop.calculate_params (quant/Weight_Only/Qop.py:37-54).

    max_val/min_val: fp32 arrays holding values representable in `kind`.
    Returns (scale as fp32 array holding `kind` values, zero_point int64 zeros).
    """
    max_val = np.asarray(max_val, dtype=np.float32)
    if min_val is None:
        min_val = np.zeros_like(max_val)
    min_val = np.asarray(min_val, dtype=np.float32)
    min_neg = np.minimum(min_val, np.float32(0))             # Qop.py:40
    max_pos = np.maximum(max_val, np.float32(0))             # Qop.py:41
    max_pos = np.maximum(-min_neg, max_pos)                  # Qop.py:49
    half_range = np.float32(float(q_max - q_min) / 2)        # 127.5, Qop.py:50
    scale = _round_to(max_pos / half_range, kind)            # true division, Qop.py:50
    eps = _round_to(np.array(EPS_F32), kind)                 # 0-dim eps does not promote
    scale = np.maximum(scale, eps)                           # Qop.py:51
    zero_point = np.zeros(max_val.shape, dtype=np.int64)     # Qop.py:45
    return scale.astype(np.float32), zero_point


def calculate_params_affine_tensor(min_val, max_val, q_min=INT8_MIN, q_max=INT8_MAX):
    """Per-tensor affine branch of Qop.calculate_params (quant/Weight_Only/Qop.py:60-65).

    Scalars in, ([scale] fp32, [zero_point] int32) out.
    """
    min_neg = np.minimum(np.float32(min_val), np.float32(0))
    max_pos = np.maximum(np.float32(max_val), np.float32(0))
    scale = np.float32(max_pos - min_neg) / np.float32(float(q_max - q_min))
    scale = np.maximum(scale, EPS_F32)
    zp = np.float32(q_min) - np.rint(np.float32(min_neg / scale))
    zp = np.int32(zp)
    zp = np.clip(zp, q_min, q_max).astype(np.int32)
    return np.array([scale], dtype=np.float32), np.array([zp], dtype=np.int32)


def quantize_channel_sym_int8(w, kind="f32"):
    """Qop.quantize for affine='channel', affine_dim=0, symmetric=True, dtype=int8.

    quant/Weight_Only/Qop.py:139-162 (and the identical quant/Weight_Activation/Qop.py:137-160):
      scales = max(max(amax(W[n]), 0) / 127.5, eps); zp = 0
      q = int8(clamp(round_half_even(W / scales), -128, 127))   with a true fp division
    `w` is an fp32 array whose values are representable in `kind` ('f32' | 'f16' | 'bf16');
    for the 16-bit kinds the quotient is rounded to that precision before round().
    Returns (q int8 like w, scales fp32 [N,1,...], zero_point int64 [N,1,...]).
    """
    w = np.ascontiguousarray(w, dtype=np.float32)
    n = w.shape[0]
    w2 = w.reshape(n, -1)
    scale, zp = calculate_params_symmetric(row_signed_max(w2), kind=kind)
    quot = _round_to(w2 / scale[:, None], kind)               # Qop.py:155
    q = np.clip(np.rint(quot), INT8_MIN, INT8_MAX).astype(np.int8)
    bshape = (n,) + (1,) * (w.ndim - 1)                        # Qop.py:108-113
    return q.reshape(w.shape), scale.reshape(bshape), zp.reshape(bshape)


# --------------------------------------------------------------------------------------
# native kernels of the reference
# --------------------------------------------------------------------------------------
def linear_w8a8(a, b):
    """C[m,n] = sum_k int32(A[m,k]) * int32(B[k,n]).

    cuda_kernels/linear_int8/linear_w8a8_kernel.cu:15-21.  A int8 [M,K], B int8 [K,N]
    (K-major rows), C int32 [M,N].  Accumulation is done in int64 and narrowed, which is
    identical while |acc| < 2^31 (always true for K <= 131071).
    """
    a = np.asarray(a, dtype=np.int8)
    b = np.asarray(b, dtype=np.int8)
    assert a.ndim == 2 and b.ndim == 2 and a.shape[1] == b.shape[0]
    # float64 matmul is exact here (|acc| < 2^53) and uses BLAS.
    c = a.astype(np.float64) @ b.astype(np.float64)
    return c.astype(np.int64).astype(np.int32)


def conv2d_int8(x, w, stride=1, padding=0):
    """Direct int8 convolution, int32 accumulate, NCHW / OIHW.

    cuda_kernels/conv2d_int8/conv2d_w8a8.cu:61-77 for the sum and :91-92 for the output
    size.  The reference kernel's inner loop ignores `stride`; like its own test
    (conv2d_int8/test.py:85,97) this restatement uses F.conv2d semantics, i.e. the input
    index is y*stride + i - padding.  Dilation 1, groups 1.
    """
    x = np.asarray(x, dtype=np.int8)
    w = np.asarray(w, dtype=np.int8)
    B, C, H, W = x.shape
    Co, Ci, KH, KW = w.shape
    assert C == Ci
    Ho = (H + 2 * padding - KH) // stride + 1
    Wo = (W + 2 * padding - KW) // stride + 1
    xp = np.zeros((B, C, H + 2 * padding, W + 2 * padding), dtype=np.float64)
    xp[:, :, padding:padding + H, padding:padding + W] = x
    wf = w.astype(np.float64)
    out = np.zeros((B, Co, Ho, Wo), dtype=np.float64)
    for i in range(KH):
        for j in range(KW):
            patch = xp[:, :, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride]
            out += np.einsum("bchw,oc->bohw", patch, wf[:, :, i, j], optimize=True)
    return out.astype(np.int64).astype(np.int32)


# --------------------------------------------------------------------------------------
# epilogues applied after the integer accumulation
# --------------------------------------------------------------------------------------
def epilogue(acc, bias, alpha, beta, channel_axis=1):
    """out = (float32(acc) + bias[n]) * alpha[n] + beta, every step rounded to fp32.

    Weight-only (quant/Weight_Only/Quantizer.py:79-86 + Qop.py:179): bias = layer bias
    (added BEFORE the scale), alpha = weight scale, beta = 0 (zero point is 0).
    Weight+activation (quant/Weight_Activation/Quantizer.py:58-60): bias = folded bias,
    alpha = scalers, beta = output_zero_point.
    """
    acc = np.asarray(acc)
    shape = [1] * acc.ndim
    shape[channel_axis] = acc.shape[channel_axis]
    y = acc.astype(np.float32)
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float32).reshape(shape)
    y = y * np.asarray(alpha, dtype=np.float32).reshape(shape)
    y = y + np.float32(beta)
    return y.astype(np.float32)


def wo_linear_forward(x_i8, wq, scale, bias=None):
    """Weight-only Quantizer.forward for a Linear layer on integer-valued activations.

    quant/Weight_Only/Quantizer.py:79-86: y = scale * ((x @ Wq^T + b) - 0).
    x_i8 int8 [M,K], wq int8 [N,K], scale fp32 [N] (or [N,1]), bias fp32 [N] or None.
    """
    acc = linear_w8a8(x_i8, np.ascontiguousarray(np.asarray(wq).T))
    return epilogue(acc, bias, np.asarray(scale).reshape(-1), 0.0, channel_axis=1)


def wo_linear_forward_float(x, wq, scale, bias=None):
    """Weight-only Quantizer.forward for a Linear layer on real-valued activations.

    quant/Weight_Only/Quantizer.py:79-91: F.linear(x, weight.to(x.dtype)) in the activation's precision, `+ bias`,
    then Qop.dequantize (Qop.py:164-181): scale * (y - 0).  The reference's fp32 product sums in an order its BLAS
    chooses; this restatement sums in float64 and rounds once, which every fp32 summation order agrees with to
    about 1e-6 relative to sum_k |x*w| -- far inside the 1e-3 the outputs are compared at.
    x float [M,K], wq int8 [N,K], scale fp32 [N] (or [N,1]), bias fp32 [N] or None.
    """
    acc = np.asarray(x, dtype=np.float64) @ np.asarray(wq, dtype=np.float64).T
    return epilogue(acc.astype(np.float32), bias, np.asarray(scale).reshape(-1), 0.0, channel_axis=1)


def wo_conv2d_forward(x_i8, wq, scale, bias=None, stride=1, padding=0):
    """Weight-only Quantizer.forward for a Conv2d layer (quant/Weight_Only/Quantizer.py:72-86)."""
    acc = conv2d_int8(x_i8, wq, stride, padding)
    return epilogue(acc, bias, np.asarray(scale).reshape(-1), 0.0, channel_axis=1)


def wo_conv2d_forward_float(x, wq, scale, bias=None, stride=1, padding=0):
    """Weight-only Quantizer.forward for a Conv2d layer on real-valued activations
    (quant/Weight_Only/Quantizer.py:72-91: F.conv2d(x, weight.to(x.dtype), stride, padding), `+ bias`, dequantize).
    Summed in float64 and rounded once, as wo_linear_forward_float."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(wq, dtype=np.float64)
    B, C, H, W = x.shape
    Co, _, KH, KW = w.shape
    xp = np.zeros((B, C, H + 2 * padding, W + 2 * padding), dtype=np.float64)
    xp[:, :, padding:padding + H, padding:padding + W] = x
    Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    acc = np.zeros((B, Co, Ho, Wo), dtype=np.float64)
    for i in range(KH):
        for j in range(KW):
            patch = xp[:, :, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride]
            acc += np.einsum("bchw,oc->bohw", patch, w[:, :, i, j])
    return epilogue(acc.astype(np.float32), bias, np.asarray(scale).reshape(-1), 0.0, channel_axis=1)


def dequantize(q, scales, zero_point):
    """Qop.dequantize (quant/Weight_Only/Qop.py:179): scales * (float(q) - zero_point)."""
    return (np.asarray(scales, dtype=np.float32) *
            (np.asarray(q).astype(np.float32) - np.asarray(zero_point).astype(np.float32))
            ).astype(np.float32)


# --------------------------------------------------------------------------------------
# multi-GPU partition (pure index arithmetic; no reference counterpart)
# --------------------------------------------------------------------------------------
def shard_rows(n, world, rank):
    """[lo, hi) of the output channels rank `rank` of `world` owns (contiguous, balanced)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def cast_f32_to_int8(v):
    """float32 -> int8 as torch's CPU `.to(torch.int8)` does (the C cast chain float -> int32 ->
    int8): truncate toward zero, keep the low byte (200.0 -> -56); NaN and |v| >= 2^31 -> 0."""
    v = np.asarray(v, dtype=np.float32)
    ok = np.abs(v) < np.float32(2147483648.0)          # False for NaN
    i32 = np.where(ok, v, np.float32(0)).astype(np.int32)
    return (i32 & 0xFF).astype(np.uint8).view(np.int8)


def pool2d_to_int8(x, kernel_size, stride=None, padding=0, mode="max"):
    """QuantizedMaxPool2d / QuantizedAvgPool2d.forward (quant/Weight_Activation/Quantizer.py:122-130):
    nn.MaxPool2d / nn.AvgPool2d on x.float() with the module defaults (dilation 1, ceil_mode False,
    count_include_pad True), then .to(torch.int8).  x: [B, C, H, W], any real dtype (values are
    taken as float32).  The average is a float32 running sum over the window in row-major order
    divided by kh*kw, as torch's CPU kernel computes it."""
    def pair(v):
        return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))
    kh, kw = pair(kernel_size)
    sh, sw = pair(kernel_size if stride is None else stride)
    ph, pw = pair(padding)
    xf = np.asarray(x).astype(np.float32)
    B, C, H, W = xf.shape
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    res = np.full((B, C, Ho, Wo), -np.inf if mode == "max" else 0.0, dtype=np.float32)
    for i in range(kh):
        for j in range(kw):
            # output (ho, wo) reads input (ho*sh - ph + i, wo*sw - pw + j) when that lies inside x
            ho_lo = max(0, -((i - ph) // sh)) if i < ph else 0
            wo_lo = max(0, -((j - pw) // sw)) if j < pw else 0
            ho_hi = min(Ho, (H - 1 + ph - i) // sh + 1)
            wo_hi = min(Wo, (W - 1 + pw - j) // sw + 1)
            if ho_hi <= ho_lo or wo_hi <= wo_lo:
                continue
            hs = slice(ho_lo * sh - ph + i, (ho_hi - 1) * sh - ph + i + 1, sh)
            ws = slice(wo_lo * sw - pw + j, (wo_hi - 1) * sw - pw + j + 1, sw)
            v = xf[:, :, hs, ws]
            dst = res[:, :, ho_lo:ho_hi, wo_lo:wo_hi]
            if mode == "max":
                np.copyto(dst, np.where((v > dst) | np.isnan(v), v, dst))
            else:
                np.copyto(dst, (dst + v).astype(np.float32))   # one fp32 rounding per term, row-major order
    if mode != "max":
        res = (res / np.float32(kh * kw)).astype(np.float32)
    return cast_f32_to_int8(res)


def quantize_tensor_affine_int8(x):
    """Per-tensor asymmetric int8 quantization with the reference's formulas: min / max over the whole
    tensor (quant/Weight_Only/Qop.py:97-99), calculate_params' per-tensor affine branch (Qop.py:62-65),
    q = round(x / scale + zero_point).clamp(-128, 127) (Qop.py:157), all in float32.
    Returns (q int8, scale [1] float32, zero_point [1] int32)."""
    x = np.asarray(x, dtype=np.float32)
    scale, zp = calculate_params_affine_tensor(x.min(), x.max())
    v = (x / scale[0]).astype(np.float32) + np.float32(zp[0])
    q = np.clip(np.rint(v.astype(np.float32)), INT8_MIN, INT8_MAX).astype(np.int8)
    return q, scale, zp
