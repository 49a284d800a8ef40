/* CPU restatement, in plain C, of the reference's int8 quantization hot path.
 *
 * TEST INFRASTRUCTURE ONLY: the checker the HIP kernels are compared with (tests/,
 * __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing in quantization_amd/
 * links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py (test_c_oracle_*) checks these functions against
 * fixtures produced by the reference's own Python (oracle/gen_golden.py), and against the
 * numpy restatement in oracle/oracle_np.py on random inputs.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no fast-math: the quantizer needs
 * IEEE division and round-half-to-even).  Paths cited are relative to the reference checkout.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MIQ_EPS_F32 1.1920928955078125e-07f /* torch.finfo(float32).eps, Qop.py:22 */

/* quant/Weight_Only/Qop.py:74-102 (signed row max, min_val = 0), :37-54 (scale =
 * max(max(amax,0)/127.5, eps), zero_point = 0 as int64), :154-162 (q = clamp(round(w/s))).
 * w: [rows, cols] fp32 contiguous. */
void miq_oracle_quantize_rows_sym_i8(const float* w, int64_t rows, int64_t cols, int8_t* q,
                                     float* scale, int64_t* zp) {
    for (int64_t r = 0; r < rows; ++r) {
        const float* row = w + r * cols;
        float mx = row[0];
        for (int64_t c = 1; c < cols; ++c) mx = row[c] > mx ? row[c] : mx; /* torch.amax */
        float pos = mx > 0.0f ? mx : 0.0f;         /* Qop.py:41,49 with min_val = 0 */
        float s = pos / 127.5f;                    /* Qop.py:50, true division */
        s = s > MIQ_EPS_F32 ? s : MIQ_EPS_F32;     /* Qop.py:51 */
        scale[r] = s;
        zp[r] = 0;                                 /* Qop.py:45 */
        for (int64_t c = 0; c < cols; ++c) {
            float v = nearbyintf(row[c] / s);      /* torch.round = half-to-even, Qop.py:155 */
            v = v < -128.0f ? -128.0f : v;
            v = v > 127.0f ? 127.0f : v;
            q[r * cols + c] = (int8_t)v;
        }
    }
}

/* cuda_kernels/linear_int8/linear_w8a8_kernel.cu:15-21.  A [M,K], B [K,N], C [M,N]. */
void miq_oracle_linear_w8a8(const int8_t* A, const int8_t* B, int32_t* C, int64_t m, int64_t n,
                            int64_t k) {
    for (int64_t row = 0; row < m; ++row) {
        int32_t* c = C + row * n;
        memset(c, 0, (size_t)n * sizeof(int32_t));
        for (int64_t i = 0; i < k; ++i) {          /* k outer so the inner loop vectorises */
            const int32_t a = (int32_t)A[row * k + i];
            const int8_t* b = B + i * n;
            for (int64_t col = 0; col < n; ++col) c[col] += a * (int32_t)b[col];
        }
    }
}

/* Same product with the weight stored [N,K] as Quantizer.weight holds it
 * (quant/Weight_Only/Quantizer.py:33,79: F.linear(x, weight)). */
void miq_oracle_linear_i8_nt(const int8_t* A, const int8_t* W, int32_t* C, int64_t m, int64_t n,
                             int64_t k) {
    for (int64_t row = 0; row < m; ++row)
        for (int64_t col = 0; col < n; ++col) {
            int32_t sum = 0;
            const int8_t* a = A + row * k;
            const int8_t* w = W + col * k;
            for (int64_t i = 0; i < k; ++i) sum += (int32_t)a[i] * (int32_t)w[i];
            C[row * n + col] = sum;
        }
}

/* cuda_kernels/conv2d_int8/conv2d_w8a8.cu:61-77 (sum) and :91-92 (output size), with the
 * stride applied to the input index as F.conv2d does (the reference's own test compares
 * with F.conv2d, conv2d_int8/test.py:85,97).  NCHW int8, OIHW int8, NCHW int32. */
void miq_oracle_conv2d_i8(const int8_t* in, const int8_t* ker, int32_t* out, int64_t batch,
                          int64_t cin, int64_t H, int64_t W, int64_t cout, int64_t kh, int64_t kw,
                          int64_t stride, int64_t padding) {
    const int64_t Ho = (H + 2 * padding - kh) / stride + 1;
    const int64_t Wo = (W + 2 * padding - kw) / stride + 1;
    for (int64_t b = 0; b < batch; ++b)
        for (int64_t oc = 0; oc < cout; ++oc)
            for (int64_t y = 0; y < Ho; ++y)
                for (int64_t x = 0; x < Wo; ++x) {
                    int32_t sum = 0;
                    for (int64_t c = 0; c < cin; ++c)
                        for (int64_t i = 0; i < kh; ++i) {
                            const int64_t iy = y * stride + i - padding;
                            if (iy < 0 || iy >= H) continue;
                            for (int64_t j = 0; j < kw; ++j) {
                                const int64_t ix = x * stride + j - padding;
                                if (ix < 0 || ix >= W) continue;
                                sum += (int32_t)in[((b * cin + c) * H + iy) * W + ix] *
                                       (int32_t)ker[((oc * cin + c) * kh + i) * kw + j];
                            }
                        }
                    out[((b * cout + oc) * Ho + y) * Wo + x] = sum;
                }
}

/* out = (float(acc) + bias[n]) * alpha[n] + beta, each step rounded to fp32.
 * Weight-only: quant/Weight_Only/Quantizer.py:79-86 + Qop.py:179 (bias before scale, zp 0).
 * Weight+activation: quant/Weight_Activation/Quantizer.py:58-60.
 * acc is [outer, channels, inner] (inner = 1 for Linear, Ho*Wo for Conv2d). */
void miq_oracle_epilogue_f32(const int32_t* acc, const float* bias, const float* alpha, float beta,
                             float* out, int64_t outer, int64_t channels, int64_t inner) {
    for (int64_t o = 0; o < outer; ++o)
        for (int64_t c = 0; c < channels; ++c) {
            const float b = bias ? bias[c] : 0.0f;
            const float a = alpha[c];
            for (int64_t i = 0; i < inner; ++i) {
                const int64_t idx = (o * channels + c) * inner + i;
                float y = (float)acc[idx];
                y = y + b;
                y = y * a;
                y = y + beta;
                out[idx] = y;
            }
        }
}
