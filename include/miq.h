/* miq - MI355X (gfx950) int8 quantization hot path, C ABI.
 *
 * Drop-in boundary for the quantized hot path of satabios/quantization: the per-channel
 * int8 weight quantizer, the int8 x int8 -> int32 Linear GEMM with a fused per-channel
 * dequant epilogue, and the int8 Conv2d as an implicit GEMM.  Each entry point names the
 * reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer on the current HIP device unless it says "host".
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Work is
 *    enqueued on it; no entry point synchronises or allocates device memory.
 *  - Return value: 0 on success, a negative MIQ_E* code on failure; miq_last_error() then
 *    returns a thread-local, human-readable message.  Arguments are validated before
 *    anything is launched, so a failed call has no effect on device memory.
 *  - Tensors are dense and contiguous in the stated layout.
 *  - Sizes may be any int64: an entry point bounds them before it does arithmetic on them (a dimension at or beyond
 *    2^31, a filter size / stride / dilation / padding at or beyond 2^15, or a byte count beyond what a kernel indexes
 *    gives MIQ_ERANGE or MIQ_EINVAL), and a *_workspace / *_bytes query answers 0 for what the call would refuse.
 */
#ifndef MIQ_H
#define MIQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIQ_VERSION 200 /* major*10000 + minor*100 + patch */

/* error codes */
#define MIQ_OK 0
#define MIQ_EINVAL (-1)    /* bad argument: null pointer, negative size, unknown enum     */
#define MIQ_EALIGN (-2)    /* pointer / leading dimension not aligned as the path needs  */
#define MIQ_EWORKSPACE (-3) /* workspace missing or too small                             */
#define MIQ_ELAUNCH (-4)   /* hipGetLastError() != hipSuccess after a launch             */
#define MIQ_ERANGE (-5)    /* size beyond what the kernels index (2^31 elements / rows)  */

/* element types of floating inputs / outputs */
#define MIQ_F32 0
#define MIQ_F16 1
#define MIQ_BF16 2
#define MIQ_I32 3 /* outputs only: raw accumulators, no epilogue */
#define MIQ_I8 4  /* int8 activations in; as an output only of miq_conv2d_i8_nhwc and miq_gemm_i8_ex */

/* layout of the weight operand of miq_gemm_i8 */
#define MIQ_W_NK 0 /* [N,K], K contiguous: what Quantizer.weight holds (Quantizer.py:33) */
#define MIQ_W_KN 1 /* [K,N], N contiguous: what linear_w8a8 takes (linear_w8a8_kernel.cu:18) */

int miq_version(void);
const char* miq_last_error(void);

/* Per-output-channel symmetric int8 weight quantizer.
 *
 * Replaces Qop.quantize() with affine='channel', affine_dim=0, symmetric=True, dtype=int8
 * (quant/Weight_Only/Qop.py:139-162, :74-102, :37-54; identical code in
 * quant/Weight_Activation/Qop.py:137-160), i.e. for each row n of w[rows, cols]:
 *     m        = max_k w[n,k]                      (signed maximum, not abs-max)
 *     scale[n] = max(max(m, 0) / 127.5, 2^-23)     (IEEE division; for f16/bf16 inputs the
 *                                                   quotient and the floor are rounded to
 *                                                   that type, returned widened to fp32)
 *     zp[n]    = 0                                 (int64, as the reference stores it)
 *     q[n,k]   = int8(clamp(rint(w[n,k] / scale[n]), -128, 127))   (IEEE division,
 *                                                   round-half-to-even; 16-bit inputs round
 *                                                   the quotient to their type first)
 *     row_max[n] = m                               (optional, may be NULL: Qop.max_val)
 * A 4-D conv weight [Cout,Cin,KH,KW] is passed as rows=Cout, cols=Cin*KH*KW.
 * w_dtype: MIQ_F32 | MIQ_F16 | MIQ_BF16.  One pass over HBM: 4 B read + 1 B written per
 * element for fp32.  NaN inputs are outside the contract (the reference's result for
 * them is a C undefined conversion).
 */
int miq_quantize_rows_sym_i8(const void* w, int w_dtype, int64_t rows, int64_t cols, int8_t* q,
                             float* scale, int64_t* zp, float* row_max, void* stream);

/* The pieces of the quantizer on their own, for callers that use Qop's methods one at a time.
 *
 * miq_row_max: row_max[n] = max_k w[n,k], the reduction of Qop.compute_scales_zero_point's
 * channel branch (quant/Weight_Only/Qop.py:78-92).
 * miq_quantize_rows_given_i8: Qop.quantize when scales / zero points were computed earlier
 * (Qop.py:144,154-157): q = int8(clamp(rint(w / scale + zp), -128, 127)); scale (and zp, which
 * may be NULL for the symmetric form) hold `rows` entries when per_row != 0, else one. */
int miq_row_max(const void* w, int w_dtype, int64_t rows, int64_t cols, float* row_max, void* stream);
int miq_quantize_rows_given_i8(const void* w, int w_dtype, int64_t rows, int64_t cols, const float* scale,
                               const float* zp, int per_row, int8_t* q, void* stream);

/* Weight dequantisation, Qop.dequantize(q) (quant/Weight_Only/Qop.py:179):
 *     out[n,k] = scale[n] * (float(q[n,k]) - float(zp[n]))        fp32 out
 * zp may be NULL (treated as zeros). */
int miq_dequantize_rows_i8(const int8_t* q, const float* scale, const int64_t* zp, int64_t rows,
                           int64_t cols, float* out, void* stream);

/* Activation-side dequantisation, Qop.dequantize(y, activation=True) (Qop.py:175-179), for a
 * float tensor y[outer, channels, inner] (inner = 1 for Linear outputs, Ho*Wo for Conv2d):
 *     out = scale[c] * (y - zp[c])          zp fp32 [channels] or NULL. */
int miq_dequantize_channels_f32(const float* y, const float* scale, const float* zp, int64_t outer,
                                int64_t channels, int64_t inner, float* out, void* stream);

/* The layer epilogue applied to a product that was computed in floating point (the
 * reference's Quantizer.forward on float activations, quant/Weight_Only/Quantizer.py:73-86,
 * quant/Weight_Activation/Quantizer.py:58-60):
 *     out[o,c,i] = (y[o,c,i] + bias[c]) * alpha[c] + beta      bias may be NULL. */
int miq_epilogue_channels_f32(const float* y, const float* bias, const float* alpha, float beta,
                              int64_t outer, int64_t channels, int64_t inner, float* out, void* stream);

/* int8 Linear: out[m,n] = epilogue( sum_k int32(A[m,k]) * int32(W[n,k]) ).
 *
 * Replaces linear_w8a8(input, weights) (cuda_kernels/linear_int8/linear_w8a8.cpp:11-20,
 * linear_w8a8_kernel.cu:5-39) when out_dtype = MIQ_I32 and w_layout = MIQ_W_KN, and the
 * composition of that product with the dequantisation the Python layers apply after it:
 *   weight-only   (quant/Weight_Only/Quantizer.py:79-86, Qop.py:179):
 *       out = scale[n] * ((acc + b[n]) - 0)      -> bias = b, alpha = scale, beta = 0
 *   weight+activation (quant/Weight_Activation/Quantizer.py:58-60):
 *       out = (acc + bias'[n]) * scalers[n] + output_zero_point
 * Epilogue, evaluated in fp32 with one rounding per step (no fused multiply-add):
 *       out[m,n] = (float(acc[m,n]) + bias[n]) * alpha[n] + beta
 * A: int8 [M,K] row-major.  W: int8, [N,K] (MIQ_W_NK) or [K,N] (MIQ_W_KN).
 * bias: fp32 [N] or NULL (no add).  alpha: fp32 [N] (required unless out_dtype = MIQ_I32).
 * out: [M,N] row-major of out_dtype MIQ_F32 | MIQ_BF16 | MIQ_I32.
 * workspace: device scratch of at least miq_gemm_i8_workspace(...) bytes, 256-byte
 * aligned; may be NULL when that size is 0.  It holds the weight re-laid [N,K] (MIQ_W_KN
 * only; the caller may instead cache the re-laid weight, see miq_transpose_i8) and, for
 * thin problems (few output tiles, deep K), the int32 slabs of a split-K run.
 */
size_t miq_gemm_i8_workspace(int w_layout, int64_t M, int64_t N, int64_t K);
int miq_gemm_i8(const int8_t* A, const int8_t* W, int w_layout, int64_t M, int64_t N, int64_t K,
                const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                void* workspace, size_t workspace_bytes, void* stream);

/* miq_gemm_i8 with the output placed and typed as a column-sharded layer needs it (one rank of a layer split by output
 * channel over the GPUs of a node computes out[:, lo:hi] and all-gathers the blocks; the reference has no multi-GPU
 * code, and its single-GPU contract returns one row-major matrix, cuda_kernels/linear_int8/linear_w8a8.cpp:15-20):
 *   out_layout MIQ_OUT_MN: out[m * ldo + n], ldo >= N: the rank's columns written straight into a wider row-major matrix.
 *   out_layout MIQ_OUT_NM: out[n * ldo + m], ldo >= M: the TRANSPOSED result.  A rank's block is then rows lo..hi of one
 *     [N_total, M] matrix, so the ranks' blocks are contiguous in rank order and a plain all-gather leaves the whole
 *     (transposed) output dense, with no re-arranging pass.  bias / alpha are per output channel n as always.
 *   out_dtype MIQ_I8 (besides those of miq_gemm_i8): out = int8(clamp(rint(y / requant_scale), -128, 127)) with
 *     y = (acc + bias[n]) * alpha[n] + beta in fp32, IEEE division, round-half-even: the per-tensor form of Qop.quantize
 *     (quant/Weight_Only/Qop.py:154-157) with a static activation scale, which is what a following int8 layer consumes;
 *     a quarter of the fp32 bytes on the links.  requant_scale must be positive; other out_dtypes ignore it.
 * Same arithmetic, kernels and workspace rules as miq_gemm_i8 (miq_gemm_i8_ex_workspace bytes). */
#define MIQ_OUT_MN 0
#define MIQ_OUT_NM 1
size_t miq_gemm_i8_ex_workspace(int w_layout, int out_layout, int64_t M, int64_t N, int64_t K);
int miq_gemm_i8_ex(const int8_t* A, const int8_t* W, int w_layout, int64_t M, int64_t N, int64_t K,
                   const float* bias, const float* alpha, float beta, void* out, int out_dtype, int out_layout,
                   int64_t ldo, float requant_scale, void* workspace, size_t workspace_bytes, void* stream);

/* miq_gemm_i8 for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments and
 * results as miq_gemm_i8, without the workspace: the scratch a call may need (the re-laid [N,K] copy of a
 * MIQ_W_KN weight, the int32 slabs of a split-K run) lives in a buffer of
 * miq_gemm_i8_noalloc_max_workspace() bytes (32 MiB) that is part of the library's code object on every GPU
 * it runs on.  A call whose miq_gemm_i8_workspace() exceeds that returns MIQ_EWORKSPACE and launches
 * nothing.  The call allocates and frees nothing and does not synchronise.  Every call on a GPU shares that
 * GPU's buffer: calls that overlap in time on one GPU (different streams, different host threads) race;
 * calls on one stream are ordered and safe. */
size_t miq_gemm_i8_noalloc_max_workspace(void);
int miq_gemm_i8_noalloc(const int8_t* A, const int8_t* W, int w_layout, int64_t M, int64_t N, int64_t K,
                        const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                        void* stream);

/* Weight-only Linear on floating-point activations:
 *       out[m,n] = (sum_k x[m,k] * float(W[n,k]) + bias[n]) * alpha[n] + beta
 *
 * Replaces the weight-only Quantizer.forward for nn.Linear (quant/Weight_Only/Quantizer.py:79-91):
 * `weight.to(x.dtype)` on every call, F.linear, `+ bias`, and Qop.dequantize (Qop.py:164-181) with
 * alpha = scale, beta = 0 (zero_point is 0 for the symmetric int8 weights Qop.quantize produces).
 * The int8 weight is read as it is stored (one byte per element) and widened in registers; the
 * product runs on the bf16 (fp16 activations: the fp16) MFMA with fp32 accumulation.  fp32 activations
 * are carried as two bf16 planes (hi + lo), which keeps |x - hi - lo| below 2^-17 |x|, so the result
 * agrees with the reference's fp32 product far inside 1e-3 relative; bf16 and fp16 activations are
 * used as they are (every product exact in the accumulator).  The epilogue is evaluated in fp32 with
 * one rounding per step.
 * x: [M,K] row-major of x_dtype MIQ_F32 | MIQ_F16 | MIQ_BF16.  W: int8 [N,K].  bias: fp32 [N] or NULL.
 * alpha: fp32 [N].  out: [M,N] of out_dtype MIQ_F32 | MIQ_BF16.
 * workspace: at least miq_gemm_w8_workspace(...) bytes, 256-byte aligned (the bf16 planes of x and,
 * for thin problems, the fp32 slabs of a split-K run); may be NULL when that size is 0. */
size_t miq_gemm_w8_workspace(int x_dtype, int64_t M, int64_t N, int64_t K);
int miq_gemm_w8(const void* x, int x_dtype, const int8_t* W, int64_t M, int64_t N, int64_t K,
                const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                void* workspace, size_t workspace_bytes, void* stream);

/* Weight-only Conv2d on floating-point activations, NCHW in, OIHW int8 weights, NCHW out:
 *       out[b,o,y,x] = (sum_{c,i,j} x[b,c,y*s+i-p,x*s+j-p] * float(w[o,c,i,j]) + bias[o]) * alpha[o] + beta
 *
 * Replaces the weight-only Quantizer.forward for nn.Conv2d (quant/Weight_Only/Quantizer.py:72-91):
 * `weight.to(x.dtype)`, F.conv2d(x, ., stride, padding), `+ bias`, Qop.dequantize.  An implicit GEMM on the bf16
 * MFMA with the activation carried as bf16 planes exactly as in miq_gemm_w8 (fp32-accurate); dilation 1,
 * groups 1, one stride and one padding for both axes, zero padding.  Output size as F.conv2d:
 * Ho = (H + 2*pad - KH) / stride + 1.
 * workspace: at least miq_conv2d_w8_workspace(...) bytes, 256-byte aligned (padded channels-last planes of x
 * and the re-laid weight; for a 1 x 1 filter with a stride and no padding only the pixels the filter visits). */
size_t miq_conv2d_w8_workspace(int x_dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                               int64_t KW, int64_t stride, int64_t pad);
int miq_conv2d_w8(const void* x, int x_dtype, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                  int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias,
                  const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                  size_t workspace_bytes, void* stream);

/* miq_conv2d_w8 with the filter re-laid ONCE, when the layer is built, instead of by every call (the reference's layers
 * fix their int8 weight in Quantizer.from_float, quant/Weight_Only/Quantizer.py:93-112, the one place it changes; its
 * forward re-casts the whole weight on every call, Quantizer.py:73).
 *   miq_conv2d_w8_pack_weight: OIHW int8 w -> wp (miq_conv2d_w8_packed_weight_bytes bytes, 256-byte aligned), the forms
 *     the kernels stage from for activations of x_dtype (im2col-order rows; for 3 x 3 filters over whole 32-channel groups
 *     and for stems the filters widened to bf16 / fp16 in the layout of the LDS stage).  wp depends on the filter's shape
 *     and on x_dtype only, not on the images the layer is later given.
 *   miq_conv2d_w8_prepacked: miq_conv2d_w8 with `wp` in place of `w`; same results bit for bit.  workspace: at least
 *     miq_conv2d_w8_prepacked_workspace(...) bytes (the activation planes only). */
size_t miq_conv2d_w8_packed_weight_bytes(int x_dtype, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW);
int miq_conv2d_w8_pack_weight(const int8_t* w, int x_dtype, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, void* wp,
                              void* stream);
size_t miq_conv2d_w8_prepacked_workspace(int x_dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                         int64_t KH, int64_t KW, int64_t stride, int64_t pad);
int miq_conv2d_w8_prepacked(const void* x, int x_dtype, const void* wp, int64_t B, int64_t Cin, int64_t H, int64_t W,
                            int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias,
                            const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                            size_t workspace_bytes, void* stream);

/* dst[c,r] = src[r,c] for an int8 matrix src[rows, cols] (both dense).  Used to turn the
 * [K,N] weight of linear_w8a8 into the [N,K] layout once, ahead of repeated GEMMs. */
int miq_transpose_i8(const int8_t* src, int64_t rows, int64_t cols, int8_t* dst, void* stream);

/* int8 Conv2d as an implicit GEMM, NCHW activations, OIHW weights, NCHW output.
 *
 * Replaces conv2d_int8(input, kernel, output, stride, padding)
 * (cuda_kernels/conv2d_int8/conv2d_w8a8.cu:81-114; kernel :6-78) when out_dtype = MIQ_I32,
 * and the weight-only / weight+activation Conv2d forward (quant/Weight_Only/Quantizer.py:72-86,
 * quant/Weight_Activation/Quantizer.py:53-60) with the epilogue of miq_gemm_i8 applied per
 * output channel.  Output size Ho = (H + 2*pad - KH)/stride + 1 (conv2d_w8a8.cu:91-92); the
 * input index is y*stride + i - pad as in F.conv2d (the reference kernel's inner loop drops
 * the stride; its own test compares with F.conv2d).  Dilation 1, groups 1, square stride
 * and padding, as the reference kernel.
 * workspace: at least miq_conv2d_i8_workspace(...) bytes, 256-byte aligned: holds the
 * zero-padded channels-last copy of x and the re-laid weight (for a 1 x 1 filter with a stride
 * and no padding only the pixels the filter visits).
 */
size_t miq_conv2d_i8_workspace(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                               int64_t KW, int64_t stride, int64_t pad);
int miq_conv2d_i8(const int8_t* x, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                  int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad,
                  const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                  void* workspace, size_t workspace_bytes, void* stream);

/* miq_conv2d_i8 with the filter re-laid once, when the layer is built (Quantizer.from_float,
 * quant/Weight_Only/Quantizer.py:93-112, is the one place the reference's weight changes): `wp` is the packed weight of
 * miq_conv2d_i8_pack_weight below (it depends on the filter's shape only), the workspace holds the padded copy of x only
 * (miq_conv2d_i8_prepacked_workspace bytes).  Same results as miq_conv2d_i8 bit for bit. */
size_t miq_conv2d_i8_prepacked_workspace(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                                         int64_t KW, int64_t stride, int64_t pad);
int miq_conv2d_i8_prepacked(const int8_t* x, const int8_t* wp, int64_t B, int64_t Cin, int64_t H, int64_t W,
                            int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad,
                            const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                            void* workspace, size_t workspace_bytes, void* stream);

/* int8 Conv2d on channels-last activations: the form in which a stack of int8 layers hands its activations from
 * one layer to the next without any re-layout pass (miq_conv2d_i8 above spends about a quarter of a ResNet-18 stack
 * turning NCHW into this form, layer by layer).  Same arithmetic and epilogue as miq_conv2d_i8, which replaces
 * conv2d_int8 (cuda_kernels/conv2d_int8/conv2d_w8a8.cu:81-114) and the Conv2d forward of the Python layers; the
 * channels-last layout and the int8 output have no counterpart in the reference, whose kernel is NCHW only.
 *   x:   int8 [B][H + 2*pad][W + 2*pad][Cin], the border already present and zero, Cin % 4 == 0, 16-byte aligned.
 *   wp:  the OIHW weight re-laid once by miq_conv2d_i8_pack_weight (miq_conv2d_i8_packed_weight_bytes bytes, 16-byte
 *        aligned; the same packed weight serves miq_conv2d_i8_prepacked).
 *   out: [B][Ho + 2*out_pad][Wo + 2*out_pad][Cout], Cout % 4 == 0; only the Ho x Wo interior is written, so a buffer
 *        whose border was zeroed once serves as the next layer's x as it is.  out_dtype MIQ_I32 (raw sums), MIQ_F32 /
 *        MIQ_BF16 ((acc + bias[c]) * alpha[c] + beta), or MIQ_I8: that value requantised, clamp(rint(.), -128, 127),
 *        round-half-even: the per-tensor form of Qop.quantize (quant/Weight_Only/Qop.py:154-157) with the next
 *        layer's activation scale folded into alpha.
 * No workspace, no allocation; dilation 1, groups 1, one stride and one padding for both axes. */
size_t miq_conv2d_i8_packed_weight_bytes(int64_t Cin, int64_t Cout, int64_t KH, int64_t KW);
int miq_conv2d_i8_pack_weight(const int8_t* w, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, int8_t* wp, void* stream);
int miq_conv2d_i8_nhwc(const int8_t* x, const int8_t* wp, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                       int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias, const float* alpha,
                       float beta, void* out, int out_dtype, int64_t out_pad, void* stream);

/* Conv2d with every argument nn.Conv2d has: a stride, a dilation and a padding per axis, groups, and a padding that may
 * differ before and after (padding='same' on an even filter extent).  NCHW activations of x_dtype MIQ_I8 | MIQ_F32 |
 * MIQ_F16 | MIQ_BF16, int8 weights [Cout][Cin / groups][KH][KW], NCHW output:
 *     out[b,o,y,x] = (sum_{c,i,j} x[b, g(o)*Cin/groups + c, y*stride_h + i*dil_h - pad_top, x*stride_w + j*dil_w - pad_left]
 *                     * w[o,c,i,j] + bias[o]) * alpha[o] + beta          (positions outside the image count as 0)
 *     Ho = (H + pad_top + pad_bottom - dil_h*(KH-1) - 1) / stride_h + 1, Wo likewise.
 * Replaces F.conv2d(x, weight.to(x.dtype), stride, padding, dilation, groups) + bias + Qop.dequantize in the layers'
 * forward (quant/Weight_Only/Quantizer.py:73-86, quant/Weight_Activation/Quantizer.py:53-60) for the layers that
 * miq_conv2d_i8 / miq_conv2d_w8 (dilation 1, groups 1, one stride and padding: all that conv2d_int8 of
 * cuda_kernels/conv2d_int8/conv2d_w8a8.cu:81-114 knows) do not take.  Dense layers (groups = 1) run on the same
 * implicit-GEMM MFMA kernels as those two, with the arithmetic of miq_conv2d_i8 (exact int32) / miq_conv2d_w8 (bf16
 * planes, fp32 accumulation); grouped and depthwise layers on a direct convolution on the vector ALUs (int32, or fp32
 * fused multiply-adds).
 * out_dtype MIQ_F32 | MIQ_BF16, or MIQ_I32 (raw sums, int8 activations only; bias / alpha unused).
 * workspace: at least miq_conv2d_general_workspace(...) bytes, 256-byte aligned (the padded channels-last copy of x and
 * the re-laid weight of a dense layer); 0 bytes, and then NULL will do, for grouped layers. */
size_t miq_conv2d_general_workspace(int x_dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                                    int64_t KW, int64_t stride_h, int64_t stride_w, int64_t pad_top, int64_t pad_bottom,
                                    int64_t pad_left, int64_t pad_right, int64_t dil_h, int64_t dil_w, int64_t groups);
int miq_conv2d_general(const void* x, int x_dtype, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                       int64_t Cout, int64_t KH, int64_t KW, int64_t stride_h, int64_t stride_w, int64_t pad_top,
                       int64_t pad_bottom, int64_t pad_left, int64_t pad_right, int64_t dil_h, int64_t dil_w,
                       int64_t groups, const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                       void* workspace, size_t workspace_bytes, void* stream);

/* 2-D pooling with an int8 result, NCHW: what QuantizedMaxPool2d / QuantizedAvgPool2d compute
 * (quant/Weight_Activation/Quantizer.py:122-130: nn.MaxPool2d / nn.AvgPool2d on x.float(), then
 * .to(torch.int8)), in one pass over x.
 *
 * x_dtype: MIQ_F32, MIQ_F16, MIQ_BF16 or MIQ_I8.  mode MIQ_POOL_MAX: maximum over the window,
 * padding never wins.  mode MIQ_POOL_AVG: fp32 sum over the window in row-major order divided by
 * kh*kw (padding counts as zeros: count_include_pad=True, the module's default).  Output size
 * (H + 2*ph - kh)/sh + 1 (ceil_mode=False), dilation 1: the arguments the reference's Chunker
 * forwards (quant/Weight_Activation/Chunker.py:75-77).  The cast to int8 truncates toward zero
 * and wraps modulo 256 as the host's cast does (200.0 -> -56); NaN and |v| >= 2^31 give 0.
 */
#define MIQ_POOL_MAX 0
#define MIQ_POOL_AVG 1
int miq_pool2d_to_i8(const void* x, int x_dtype, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw,
                     int sh, int sw, int ph, int pw, int mode, int8_t* out, void* stream);

/* min_max[0] = min(x), min_max[1] = max(x) over all n elements of x (fp32 results, on the device).
 *
 * The reduction of the per-tensor branch of Qop.compute_scales_zero_point
 * (quant/Weight_Only/Qop.py:97-99: tensor.max(), tensor.min()), the form the reference configures
 * its activation quantizer with (quant/Weight_Only/Quantizer.py:38-43).  One read of x.  The
 * scale / zero point follow on the host from Qop.calculate_params (Qop.py:62-65) and the int8
 * tensor from miq_quantize_rows_given_i8 with per_row = 0.
 * workspace: at least miq_tensor_minmax_workspace(n) bytes (a few KiB of partial results).
 */
size_t miq_tensor_minmax_workspace(int64_t n);
int miq_tensor_minmax(const void* x, int x_dtype, int64_t n, float* min_max, void* workspace,
                      size_t workspace_bytes, void* stream);

/* From min_max (as miq_tensor_minmax leaves it), on the device and in one launch:
 *   scale_zp[0] = max((max(max,0) - min(min,0)) / 255, 2^-23)            Qop.calculate_params, per-tensor
 *   scale_zp[1] = clamp(trunc(-128 - rint(min(min,0) / scale)), -128, 127)  affine branch (Qop.py:40-41, 62-65)
 * and, for a weight-only layer (quant/Weight_Only/Quantizer.py:79-86) evaluated on the activation
 * x_hat = scale * (q - zero_point) that miq_quantize_rows_given_i8(x, scale_zp, scale_zp + 1, per_row = 0)
 * produces, the per-channel vectors its int8 product needs as miq_gemm_i8's `alpha` and `bias`:
 *   alpha[n] = w_scale[n] * scale
 *   corr[n]  = bias[n] / scale - zero_point * w_rowsum[n]        (bias may be NULL)
 * w_rowsum[n] = sum_k Wq[n,k] as fp32.  No counterpart runs in the reference: its activation quantizer is
 * configured (Quantizer.py:38-44) but its call sites are commented out (Quantizer.py:69-70). */
int miq_dynamic_epilogue_i8(const float* min_max, const float* w_scale, const float* bias, const float* w_rowsum,
                            int64_t n, float* scale_zp, float* alpha, float* corr, void* stream);

/* The two calls above and the statistics before them as ONE launch, and the quantization as a second: what a layer with a
 * dynamic per-tensor activation quantizer runs per forward (the step the reference configures, Quantizer.py:38-44, and
 * leaves commented out, Quantizer.py:69-70).
 *   miq_dynamic_quant_params_i8: min / max over the n elements of x; scale_zp[0] = scale, scale_zp[1] = zero point as a
 *     float, scale_zp[2] = the zero point's int32 bits (formulas of miq_dynamic_epilogue_i8); and, when channels > 0,
 *     alpha[c] / corr[c] of miq_dynamic_epilogue_i8.  workspace: miq_dynamic_quant_workspace(n) bytes.  counter: one
 *     4-byte device word that is ZERO on entry and is left zero (the workgroups count themselves in; the last one
 *     finishes the reduction): allocate and clear it once; calls that share a counter must not overlap in time.
 *   miq_quantize_tensor_i8: q = int8(clamp(rint(x / scale + zero_point), -128, 127)) (Qop.py:157; IEEE division, fp16 /
 *     bf16 inputs round to their type after the division and after the addition, as tensor arithmetic does) for x
 *     [planes, H, W] with scale = scale_zp[0], zero_point = scale_zp[1] read on the device; q is
 *     [planes, H + pad_top + pad_bottom, W + pad_left + pad_right] whose frame is filled with the zero point: a padded
 *     position of a convolution holds the real value 0, whose quantized form is the zero point. */
size_t miq_dynamic_quant_workspace(int64_t n);
int miq_dynamic_quant_params_i8(const void* x, int x_dtype, int64_t n, const float* w_scale, const float* bias,
                                const float* w_rowsum, int64_t channels, float* scale_zp, float* alpha, float* corr,
                                void* workspace, size_t workspace_bytes, unsigned* counter, void* stream);
int miq_quantize_tensor_i8(const void* x, int x_dtype, int64_t planes, int64_t H, int64_t W, int64_t pad_top,
                           int64_t pad_bottom, int64_t pad_left, int64_t pad_right, const float* scale_zp, int8_t* q,
                           void* stream);

/* miq_dynamic_quant_params_i8 followed by miq_quantize_tensor_i8 on the same tensor, as one call (two launches): the
 * arguments of both (x is [planes, H, W], its statistics are taken over all planes * H * W elements). */
int miq_dynamic_quantize_i8(const void* x, int x_dtype, int64_t planes, int64_t H, int64_t W, int64_t pad_top,
                            int64_t pad_bottom, int64_t pad_left, int64_t pad_right, const float* w_scale, const float* bias,
                            const float* w_rowsum, int64_t channels, float* scale_zp, float* alpha, float* corr, int8_t* q,
                            void* workspace, size_t workspace_bytes, unsigned* counter, void* stream);

/* A weight-only Linear on a floating-point activation through the int8 product, as ONE call (three launches:
 * miq_dynamic_quant_params_i8, miq_quantize_tensor_i8, miq_gemm_i8 with alpha / corr as its epilogue vectors): what
 * Quantizer.forward computes when the activation quantizer the reference configures (quant/Weight_Only/Quantizer.py:38-44,
 * call sites commented out at :69-70) is switched on:
 *     out[m,n] = w_scale[n] * (sum_k x_hat[m,k] * W[n,k] + bias[n]),  x_hat = scale * (q - zero_point) the activation as
 *     its per-tensor affine int8 form represents it.
 * x: [M,K] of x_dtype MIQ_F32 | MIQ_F16 | MIQ_BF16; W int8 [N,K]; w_rowsum[n] = sum_k W[n,k] as fp32 (once per layer);
 * out: [M,N] of out_dtype MIQ_F32 | MIQ_BF16.  workspace: miq_linear_dynamic_w8a8_workspace(M, N, K) bytes, 256-byte
 * aligned; counter: as for miq_dynamic_quant_params_i8. */
size_t miq_linear_dynamic_w8a8_workspace(int64_t M, int64_t N, int64_t K);
int miq_linear_dynamic_w8a8(const void* x, int x_dtype, const int8_t* W, int64_t M, int64_t N, int64_t K,
                            const float* w_scale, const float* bias, const float* w_rowsum, void* out, int out_dtype,
                            void* workspace, size_t workspace_bytes, unsigned* counter, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIQ_H */
