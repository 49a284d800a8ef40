// Dense (groups = 1) convolutions with a stride / padding per axis or a dilation, on the implicit-GEMM MFMA kernels:
// defined beside those kernels (miq_conv2d_i8.hip, miq_gemm_w8f.hip), called by miq_conv2d_general.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace miq {

size_t conv2d_i8_dense_workspace(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW, int64_t sh,
                                 int64_t sw, int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh, int64_t dw);
int conv2d_i8_dense(const int8_t* x, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                    int64_t KW, int64_t sh, int64_t sw, int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh, int64_t dw,
                    const float* bias, const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                    size_t workspace_bytes, hipStream_t st);

size_t conv2d_w8_dense_workspace(int x_dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW,
                                 int64_t sh, int64_t sw, int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh, int64_t dw);
int conv2d_w8_dense(const void* x, int x_dtype, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                    int64_t KH, int64_t KW, int64_t sh, int64_t sw, int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh,
                    int64_t dw, const float* bias, const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                    size_t workspace_bytes, hipStream_t st);

}  // namespace miq
