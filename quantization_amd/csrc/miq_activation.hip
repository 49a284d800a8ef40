// Whole-tensor minimum and maximum: the reduction behind the per-tensor branch of
// Qop.compute_scales_zero_point (quant/Weight_Only/Qop.py:97-99: tensor.max(), tensor.min()), which
// the reference's activation quantizer is configured with (quant/Weight_Only/Quantizer.py:38-43).
// HBM-bandwidth bound: one read of the tensor with 16-byte loads; every workgroup leaves one
// (min, max) pair in the workspace and a second, single-workgroup launch folds the pairs.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "miq_common.h"
#include "miq_gemm_core.h"

namespace {

using namespace miq;

constexpr int kBlock = 256;
constexpr int kMaxParts = 2048;

template <int DT>
struct Elems;
template <>
struct Elems<MIQ_F32> {
    static constexpr int kPerVec = 4;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ float load1(const void* p, int64_t i) { return static_cast<const float*>(p)[i]; }
};
template <>
struct Elems<MIQ_F16> {
    static constexpr int kPerVec = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __half2float(__ushort_as_half(static_cast<unsigned short>(w[i] & 0xffffu)));
            f[2 * i + 1] = __half2float(__ushort_as_half(static_cast<unsigned short>(w[i] >> 16)));
        }
    }
    static __device__ __forceinline__ float load1(const void* p, int64_t i) {
        return __half2float(static_cast<const __half*>(p)[i]);
    }
};
template <>
struct Elems<MIQ_BF16> {
    static constexpr int kPerVec = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ float load1(const void* p, int64_t i) {
        return __uint_as_float(static_cast<uint32_t>(static_cast<const unsigned short*>(p)[i]) << 16);
    }
};

__device__ __forceinline__ void block_minmax(float& lo, float& hi) {
    __shared__ float slo[kBlock / 64], shi[kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off, 64));
        hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6] = lo;
        shi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    lo = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
    hi = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
}

// parts[2 * b] / parts[2 * b + 1] = min / max of the elements workgroup b visited.
template <int DT, bool VECTOR>
__global__ __launch_bounds__(kBlock) void minmax_partial_kernel(const void* __restrict__ x, int64_t n,
                                                                float* __restrict__ parts) {
    using E = Elems<DT>;
    float lo = INFINITY, hi = -INFINITY;
    const int64_t tid = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
    const int64_t nthreads = static_cast<int64_t>(gridDim.x) * kBlock;
    if constexpr (VECTOR) {
        const int64_t nvec = n / E::kPerVec;
        const uint4* src = static_cast<const uint4*>(x);
        for (int64_t i = tid; i < nvec; i += nthreads) {
            float f[E::kPerVec];
            E::unpack(src[i], f);
#pragma unroll
            for (int e = 0; e < E::kPerVec; ++e) {
                lo = fminf(lo, f[e]);
                hi = fmaxf(hi, f[e]);
            }
        }
        for (int64_t i = nvec * E::kPerVec + tid; i < n; i += nthreads) {   // tail
            const float v = E::load1(x, i);
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    } else {
        for (int64_t i = tid; i < n; i += nthreads) {
            const float v = E::load1(x, i);
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
    block_minmax(lo, hi);
    if (threadIdx.x == 0) {
        parts[2 * blockIdx.x] = lo;
        parts[2 * blockIdx.x + 1] = hi;
    }
}

__global__ __launch_bounds__(kBlock) void minmax_final_kernel(const float* __restrict__ parts, int nparts,
                                                              float* __restrict__ out) {
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += kBlock) {
        lo = fminf(lo, parts[2 * i]);
        hi = fmaxf(hi, parts[2 * i + 1]);
    }
    block_minmax(lo, hi);
    if (threadIdx.x == 0) {
        out[0] = lo;
        out[1] = hi;
    }
}

// scale_zp[0] = scale, scale_zp[1] = zero point (as a float) of the per-tensor affine int8 form
// (Qop.calculate_params, quant/Weight_Only/Qop.py:40-41, 62-65), and the epilogue vectors of a
// weight-only layer evaluated on the dequantized activation:
//   alpha[n] = w_scale[n] * scale,   corr[n] = bias[n] / scale - zero_point * w_rowsum[n].
__global__ __launch_bounds__(kBlock) void dynamic_epilogue_kernel(const float* __restrict__ min_max,
                                                                  const float* __restrict__ w_scale,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ w_rowsum, int64_t n,
                                                                  float* __restrict__ scale_zp,
                                                                  float* __restrict__ alpha, float* __restrict__ corr) {
    const float lo = fminf(min_max[0], 0.0f), hi = fmaxf(min_max[1], 0.0f);
    const float scale = fmaxf(__fsub_rn(hi, lo) / 255.0f, 1.1920928955078125e-07f);
    float zp = __fsub_rn(-128.0f, rintf(lo / scale));
    zp = fminf(fmaxf(truncf(zp), -128.0f), 127.0f);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scale_zp[0] = scale;
        scale_zp[1] = zp;
    }
    for (int64_t i = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
        alpha[i] = __fmul_rn(w_scale[i], scale);
        const float c = __fmul_rn(-zp, w_rowsum[i]);
        corr[i] = bias ? __fadd_rn(c, bias[i] / scale) : c;
    }
}


// ---------------------------------------------------------------------------------------
// The dynamic activation quantizer in two launches.
//
// dynamic_params_kernel: every workgroup leaves the (min, max) of its share of x in `parts` (device-scope stores) and
// takes a ticket from `counter` (a device-scope atomic, after its stores have completed: see store_shared in
// miq_gemm_core.h).  The workgroup that draws the last ticket folds the pairs, derives the per-tensor affine
// parameters (Qop.calculate_params, quant/Weight_Only/Qop.py:40-41, 62-65) and, for a weight-only layer, the epilogue
// vectors of its int8 product, and sets the counter back to zero: one launch where the statistics, the parameters and
// the vectors took three.  quantize_tensor_kernel then applies Qop.py:157 with those parameters, optionally into the
// interior of a frame filled with the zero point (a convolution's padding made explicit).
// ---------------------------------------------------------------------------------------
template <int DT, bool VECTOR>
__global__ __launch_bounds__(kBlock) void dynamic_params_kernel(const void* __restrict__ x, int64_t n, float* __restrict__ parts,
                                                                unsigned* __restrict__ counter, const float* __restrict__ w_scale,
                                                                const float* __restrict__ bias, const float* __restrict__ w_rowsum,
                                                                int64_t nch, float* __restrict__ scale_zp,
                                                                float* __restrict__ alpha, float* __restrict__ corr) {
    using E = Elems<DT>;
    float lo = INFINITY, hi = -INFINITY;
    const int64_t tid = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
    const int64_t nthreads = static_cast<int64_t>(gridDim.x) * kBlock;
    if constexpr (VECTOR) {
        const int64_t nvec = n / E::kPerVec;
        const uint4* src = static_cast<const uint4*>(x);
        for (int64_t i = tid; i < nvec; i += nthreads) {
            float f[E::kPerVec];
            E::unpack(src[i], f);
#pragma unroll
            for (int e = 0; e < E::kPerVec; ++e) {
                lo = fminf(lo, f[e]);
                hi = fmaxf(hi, f[e]);
            }
        }
        for (int64_t i = nvec * E::kPerVec + tid; i < n; i += nthreads) {   // tail
            const float v = E::load1(x, i);
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    } else {
        for (int64_t i = tid; i < n; i += nthreads) {
            const float v = E::load1(x, i);
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
    block_minmax(lo, hi);
    if (threadIdx.x == 0) {                     // device-scope stores: visible to whoever draws the last ticket
        store_shared(parts + 2 * blockIdx.x, lo);
        store_shared(parts + 2 * blockIdx.x + 1, hi);
    }
    if (take_ticket(counter) != gridDim.x - 1) return;   // (uniform over the workgroup)
    lo = INFINITY;
    hi = -INFINITY;
    for (int i = threadIdx.x; i < static_cast<int>(gridDim.x); i += kBlock) {
        lo = fminf(lo, load_shared(parts + 2 * i));
        hi = fmaxf(hi, load_shared(parts + 2 * i + 1));
    }
    __syncthreads();                            // block_minmax's LDS words are read by every thread above
    block_minmax(lo, hi);
    lo = fminf(lo, 0.0f);
    hi = fmaxf(hi, 0.0f);
    const float scale = fmaxf(__fsub_rn(hi, lo) / 255.0f, 1.1920928955078125e-07f);
    float zp = __fsub_rn(-128.0f, rintf(lo / scale));
    zp = fminf(fmaxf(truncf(zp), -128.0f), 127.0f);
    if (threadIdx.x == 0) {
        scale_zp[0] = scale;
        scale_zp[1] = zp;
        scale_zp[2] = __int_as_float(static_cast<int>(zp));   // the zero point as int32 bits (Qop keeps an integer tensor)
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    }
    // the epilogue vectors: four channels a thread and step where the vectors are whole 16-byte groups, several steps'
    // loads in flight (one workgroup does this for up to tens of thousands of channels: a dependent load per step
    // would cost a memory round trip each)
    const bool vec4 = (nch & 3) == 0 && ((reinterpret_cast<uintptr_t>(w_scale) | reinterpret_cast<uintptr_t>(w_rowsum) |
                                          reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(alpha) |
                                          reinterpret_cast<uintptr_t>(corr)) & 15) == 0;
    if (vec4) {
        const int64_t n4 = nch >> 2;
        const float4* ws4 = reinterpret_cast<const float4*>(w_scale);
        const float4* rs4 = reinterpret_cast<const float4*>(w_rowsum);
        const float4* b4 = reinterpret_cast<const float4*>(bias);
#pragma unroll 4
        for (int64_t i = threadIdx.x; i < n4; i += kBlock) {
            const float4 w = ws4[i], r = rs4[i];
            const float4 b = bias ? b4[i] : float4{0.f, 0.f, 0.f, 0.f};
            reinterpret_cast<float4*>(alpha)[i] = float4{__fmul_rn(w.x, scale), __fmul_rn(w.y, scale), __fmul_rn(w.z, scale), __fmul_rn(w.w, scale)};
            float4 c{__fmul_rn(-zp, r.x), __fmul_rn(-zp, r.y), __fmul_rn(-zp, r.z), __fmul_rn(-zp, r.w)};
            if (bias) c = float4{__fadd_rn(c.x, b.x / scale), __fadd_rn(c.y, b.y / scale), __fadd_rn(c.z, b.z / scale), __fadd_rn(c.w, b.w / scale)};
            reinterpret_cast<float4*>(corr)[i] = c;
        }
        return;
    }
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < nch; i += kBlock) {
        alpha[i] = __fmul_rn(w_scale[i], scale);
        const float c = __fmul_rn(-zp, w_rowsum[i]);
        corr[i] = bias ? __fadd_rn(c, bias[i] / scale) : c;
    }
}

// The storage-type rounding the reference's tensor arithmetic applies between operations (a bf16 / fp16 tensor divided
// by a scalar tensor is rounded to bf16 / fp16 before the next operation).
template <int DT>
__device__ __forceinline__ float to_storage(float v) {
    if constexpr (DT == MIQ_F16) return __half2float(__float2half_rn(v));
    if constexpr (DT == MIQ_BF16) return __bfloat162float(__float2bfloat16(v));
    return v;
}

template <int DT>
__device__ __forceinline__ float quant1(float v, float s, float z) {
    v = to_storage<DT>(v / s);
    v = to_storage<DT>(__fadd_rn(v, z));
    return fminf(fmaxf(rintf(v), -128.0f), 127.0f);
}

__device__ __forceinline__ uint32_t pack4_i8(float a, float b, float c, float d) {
    return (static_cast<uint32_t>(static_cast<int>(a)) & 0xffu) | ((static_cast<uint32_t>(static_cast<int>(b)) & 0xffu) << 8) |
           ((static_cast<uint32_t>(static_cast<int>(c)) & 0xffu) << 16) | (static_cast<uint32_t>(static_cast<int>(d)) << 24);
}

// q = int8(clamp(rint(x / scale + zp))), parameters read from the device (scale_zp[0], scale_zp[1]): one 16-byte load
// and one 4- or 8-byte store per thread and step.
template <int DT>
__global__ __launch_bounds__(kBlock) void quantize_tensor_kernel(const void* __restrict__ x, int64_t n,
                                                                 const float* __restrict__ scale_zp, int8_t* __restrict__ q,
                                                                 int vector) {
    using E = Elems<DT>;
    const float s = scale_zp[0], z = scale_zp[1];
    const int64_t tid = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
    const int64_t nthreads = static_cast<int64_t>(gridDim.x) * kBlock;
    int64_t done = 0;
    if (vector) {
        const int64_t nvec = n / E::kPerVec;
        const uint4* src = static_cast<const uint4*>(x);
        for (int64_t i = tid; i < nvec; i += nthreads) {
            float f[E::kPerVec];
            E::unpack(src[i], f);
#pragma unroll
            for (int e = 0; e < E::kPerVec; ++e) f[e] = quant1<DT>(f[e], s, z);
            if constexpr (E::kPerVec == 4) {
                reinterpret_cast<uint32_t*>(q)[i] = pack4_i8(f[0], f[1], f[2], f[3]);
            } else {
                reinterpret_cast<uint2*>(q)[i] = uint2{pack4_i8(f[0], f[1], f[2], f[3]), pack4_i8(f[4], f[5], f[6], f[7])};
            }
        }
        done = nvec * E::kPerVec;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) q[i] = static_cast<int8_t>(static_cast<int>(quant1<DT>(E::load1(x, i), s, z)));
}

// The same into a frame: x is [planes, H, W], q is [planes, H + pt + pb, W + pl + pr]; the interior holds the quantized
// image, the frame the zero point (the quantized form of the real value 0 a convolution pads with).  One thread per
// four consecutive bytes of an output row.
template <int DT>
__global__ __launch_bounds__(kBlock) void quantize_framed_kernel(const void* __restrict__ x, uint32_t rows_out, int H, int W,
                                                                 int pt, int pl, int Hf, int Wf, const float* __restrict__ scale_zp,
                                                                 int8_t* __restrict__ q) {
    using E = Elems<DT>;
    const float s = scale_zp[0], z = scale_zp[1];
    const uint32_t segs = (static_cast<uint32_t>(Wf) + 3u) >> 2;
    const uint64_t total = static_cast<uint64_t>(rows_out) * segs;
    for (uint64_t idx = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; idx < total;
         idx += static_cast<uint64_t>(gridDim.x) * kBlock) {
        const uint32_t row = static_cast<uint32_t>(idx / segs), seg = static_cast<uint32_t>(idx - static_cast<uint64_t>(row) * segs);
        const uint32_t plane = row / static_cast<uint32_t>(Hf), hf = row - plane * static_cast<uint32_t>(Hf);
        const int h = static_cast<int>(hf) - pt;
        const bool row_in = h >= 0 && h < H;
        const int64_t src_row = (static_cast<int64_t>(plane) * H + h) * W;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int w = static_cast<int>(seg * 4 + k) - pl;
            v[k] = (row_in && w >= 0 && w < W) ? quant1<DT>(E::load1(x, src_row + w), s, z) : z;
        }
        const uint64_t o = static_cast<uint64_t>(row) * Wf + seg * 4;
        if (seg * 4 + 3 < static_cast<uint32_t>(Wf) && (o & 3) == 0) {
            *reinterpret_cast<uint32_t*>(q + o) = pack4_i8(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (seg * 4 + k < static_cast<uint32_t>(Wf)) q[o + k] = static_cast<int8_t>(static_cast<int>(v[k]));
        }
    }
}

int parts_for(int64_t n) {
    // >= 16 elements per thread.  (More, smaller workgroups read a small tensor faster but then all finish together and
    // queue on the arrival counter: 512 workgroups on a [256,4096] tensor took 14.6 us where 128 take 11.)
    constexpr int64_t per = static_cast<int64_t>(kBlock) * 16;
    const int64_t want = n / per + (n % per != 0);   // (ceil_div would add to an n that may be any int64)
    return static_cast<int>(want < 1 ? 1 : want > kMaxParts ? kMaxParts : want);
}

template <int DT>
void launch_partial(const void* x, int64_t n, float* parts, int nparts, hipStream_t st) {
    if (aligned(x, 16)) {
        minmax_partial_kernel<DT, true><<<nparts, kBlock, 0, st>>>(x, n, parts);
    } else {
        minmax_partial_kernel<DT, false><<<nparts, kBlock, 0, st>>>(x, n, parts);
    }
}

}  // namespace

extern "C" size_t miq_tensor_minmax_workspace(int64_t n) {
    return n > 0 ? static_cast<size_t>(parts_for(n)) * 2 * sizeof(float) : 0;
}

extern "C" int miq_tensor_minmax(const void* x, int x_dtype, int64_t n, float* min_max, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    MIQ_REQUIRE(n > 0, MIQ_EINVAL, "miq_tensor_minmax: the tensor is empty (n=%lld)", (long long)n);
    MIQ_REQUIRE(x && min_max, MIQ_EINVAL, "miq_tensor_minmax: null pointer");
    MIQ_REQUIRE(x_dtype == MIQ_F32 || x_dtype == MIQ_F16 || x_dtype == MIQ_BF16, MIQ_EINVAL,
                "miq_tensor_minmax: x_dtype must be MIQ_F32, MIQ_F16 or MIQ_BF16 (got %d)", x_dtype);
    MIQ_REQUIRE(aligned(x, x_dtype == MIQ_F32 ? 4 : 2) && aligned(min_max, 4), MIQ_EALIGN,
                "miq_tensor_minmax: x / min_max not aligned to their element size");
    const size_t need = miq_tensor_minmax_workspace(n);
    MIQ_REQUIRE(workspace && workspace_bytes >= need && aligned(workspace, 4), MIQ_EWORKSPACE,
                "miq_tensor_minmax: this call needs %zu bytes of 4-byte aligned workspace (got %zu)", need, workspace_bytes);
    hipStream_t st = as_stream(stream);
    float* parts = static_cast<float*>(workspace);
    const int nparts = parts_for(n);
    switch (x_dtype) {
        case MIQ_F32: launch_partial<MIQ_F32>(x, n, parts, nparts, st); break;
        case MIQ_F16: launch_partial<MIQ_F16>(x, n, parts, nparts, st); break;
        default: launch_partial<MIQ_BF16>(x, n, parts, nparts, st); break;
    }
    if (int rc = check_launch("minmax_partial_kernel")) return rc;
    minmax_final_kernel<<<1, kBlock, 0, st>>>(parts, nparts, min_max);
    return check_launch("minmax_final_kernel");
}

extern "C" int miq_dynamic_epilogue_i8(const float* min_max, const float* w_scale, const float* bias,
                                       const float* w_rowsum, int64_t n, float* scale_zp, float* alpha, float* corr,
                                       void* stream) {
    MIQ_REQUIRE(n >= 0, MIQ_EINVAL, "miq_dynamic_epilogue_i8: negative size");
    MIQ_REQUIRE(min_max && scale_zp && (n == 0 || (w_scale && w_rowsum && alpha && corr)), MIQ_EINVAL,
                "miq_dynamic_epilogue_i8: null pointer");
    MIQ_REQUIRE(aligned(min_max, 4) && aligned(w_scale, 4) && aligned(bias, 4) && aligned(w_rowsum, 4) &&
                    aligned(scale_zp, 4) && aligned(alpha, 4) && aligned(corr, 4),
                MIQ_EALIGN, "miq_dynamic_epilogue_i8: pointers must be 4-byte aligned");
    const int64_t blocks = n / kBlock + 1;
    dynamic_epilogue_kernel<<<static_cast<int>(blocks < 1024 ? blocks : 1024), kBlock, 0, as_stream(stream)>>>(
        min_max, w_scale, bias, w_rowsum, n, scale_zp, alpha, corr);
    return check_launch("dynamic_epilogue_kernel");
}

extern "C" size_t miq_dynamic_quant_workspace(int64_t n) {
    return n > 0 ? static_cast<size_t>(parts_for(n)) * 2 * sizeof(float) : 0;
}

extern "C" int miq_dynamic_quant_params_i8(const void* x, int x_dtype, int64_t n, const float* w_scale, const float* bias,
                                           const float* w_rowsum, int64_t channels, float* scale_zp, float* alpha, float* corr,
                                           void* workspace, size_t workspace_bytes, unsigned* counter, void* stream) {
    MIQ_REQUIRE(n > 0 && channels >= 0, MIQ_EINVAL, "miq_dynamic_quant_params_i8: the tensor is empty (n=%lld)", (long long)n);
    MIQ_REQUIRE(x && scale_zp && counter && (channels == 0 || (w_scale && w_rowsum && alpha && corr)), MIQ_EINVAL,
                "miq_dynamic_quant_params_i8: null pointer");
    MIQ_REQUIRE(x_dtype == MIQ_F32 || x_dtype == MIQ_F16 || x_dtype == MIQ_BF16, MIQ_EINVAL,
                "miq_dynamic_quant_params_i8: x_dtype must be MIQ_F32, MIQ_F16 or MIQ_BF16 (got %d)", x_dtype);
    MIQ_REQUIRE(aligned(x, x_dtype == MIQ_F32 ? 4 : 2) && aligned(scale_zp, 4) && aligned(counter, 4) && aligned(w_scale, 4) &&
                    aligned(bias, 4) && aligned(w_rowsum, 4) && aligned(alpha, 4) && aligned(corr, 4),
                MIQ_EALIGN, "miq_dynamic_quant_params_i8: pointers must be aligned to their element size");
    const size_t need = miq_dynamic_quant_workspace(n);
    MIQ_REQUIRE(workspace && workspace_bytes >= need && aligned(workspace, 4), MIQ_EWORKSPACE,
                "miq_dynamic_quant_params_i8: this call needs %zu bytes of 4-byte aligned workspace (got %zu)", need,
                workspace_bytes);
    hipStream_t st = as_stream(stream);
    float* parts = static_cast<float*>(workspace);
    const int nparts = parts_for(n);
    const bool vec = aligned(x, 16);
#define MIQ_PARAMS(DT)                                                                                                       \
    do {                                                                                                                     \
        if (vec)                                                                                                             \
            dynamic_params_kernel<DT, true><<<nparts, kBlock, 0, st>>>(x, n, parts, counter, w_scale, bias, w_rowsum, channels, \
                                                                       scale_zp, alpha, corr);                              \
        else                                                                                                                 \
            dynamic_params_kernel<DT, false><<<nparts, kBlock, 0, st>>>(x, n, parts, counter, w_scale, bias, w_rowsum, channels, \
                                                                        scale_zp, alpha, corr);                             \
    } while (0)
    switch (x_dtype) {
        case MIQ_F32: MIQ_PARAMS(MIQ_F32); break;
        case MIQ_F16: MIQ_PARAMS(MIQ_F16); break;
        default: MIQ_PARAMS(MIQ_BF16); break;
    }
#undef MIQ_PARAMS
    return check_launch("dynamic_params_kernel");
}

extern "C" int miq_quantize_tensor_i8(const void* x, int x_dtype, int64_t planes, int64_t H, int64_t W, int64_t pad_top,
                                      int64_t pad_bottom, int64_t pad_left, int64_t pad_right, const float* scale_zp, int8_t* q,
                                      void* stream) {
    MIQ_REQUIRE(planes >= 0 && H >= 0 && W >= 0 && pad_top >= 0 && pad_bottom >= 0 && pad_left >= 0 && pad_right >= 0, MIQ_EINVAL,
                "miq_quantize_tensor_i8: negative size");
    MIQ_REQUIRE(x_dtype == MIQ_F32 || x_dtype == MIQ_F16 || x_dtype == MIQ_BF16, MIQ_EINVAL,
                "miq_quantize_tensor_i8: x_dtype must be MIQ_F32, MIQ_F16 or MIQ_BF16 (got %d)", x_dtype);
    // (an unframed tensor may be one row of more than 2^31 elements: the sums below only need room for the paddings)
    MIQ_REQUIRE(planes < (1ll << 62) && H < (1ll << 62) && W < (1ll << 62) && all_below_2_31({pad_top, pad_bottom, pad_left, pad_right}),
                MIQ_ERANGE, "miq_quantize_tensor_i8: sizes must stay below 2^62 and the paddings below 2^31");
    const int64_t Hf = H + pad_top + pad_bottom, Wf = W + pad_left + pad_right;
    const int64_t n_out = size_product({planes, Hf, Wf}), n_in = size_product({planes, H, W});
    MIQ_REQUIRE(n_out >= 0, MIQ_ERANGE, "miq_quantize_tensor_i8: the output's element count does not fit 62 bits");
    if (n_out == 0) return MIQ_OK;
    MIQ_REQUIRE(scale_zp && q && (x || n_in == 0), MIQ_EINVAL, "miq_quantize_tensor_i8: null pointer");
    MIQ_REQUIRE(aligned(x, x_dtype == MIQ_F32 ? 4 : 2) && aligned(scale_zp, 4), MIQ_EALIGN,
                "miq_quantize_tensor_i8: x / scale_zp not aligned to their element size");
    hipStream_t st = as_stream(stream);
    const bool framed = pad_top || pad_bottom || pad_left || pad_right;
    if (!framed) {
        const int64_t n = n_in;
        const int per = x_dtype == MIQ_F32 ? 4 : 8;
        const int vector = aligned(x, 16) && aligned(q, per);
        const int64_t blocks = ceil_div(ceil_div(n, per), kBlock);
        const int grid = static_cast<int>(blocks < 1 ? 1 : blocks > 8192 ? 8192 : blocks);
        switch (x_dtype) {
            case MIQ_F32: quantize_tensor_kernel<MIQ_F32><<<grid, kBlock, 0, st>>>(x, n, scale_zp, q, vector); break;
            case MIQ_F16: quantize_tensor_kernel<MIQ_F16><<<grid, kBlock, 0, st>>>(x, n, scale_zp, q, vector); break;
            default: quantize_tensor_kernel<MIQ_BF16><<<grid, kBlock, 0, st>>>(x, n, scale_zp, q, vector); break;
        }
        return check_launch("quantize_tensor_kernel");
    }
    const int64_t rows_out = size_product({planes, Hf});
    MIQ_REQUIRE(rows_out >= 0 && rows_out < (1ll << 32) && Hf < (1ll << 31) && Wf < (1ll << 31), MIQ_ERANGE,
                "miq_quantize_tensor_i8: at most 2^32 output rows");
    const int64_t items = planes * Hf * ceil_div(Wf, 4);
    const int64_t blocks = ceil_div(items, kBlock);
    const int grid = static_cast<int>(blocks > 16384 ? 16384 : blocks);
#define MIQ_FRAMED(DT)                                                                                                      \
    quantize_framed_kernel<DT><<<grid, kBlock, 0, st>>>(x, static_cast<uint32_t>(planes * Hf), (int)H, (int)W, (int)pad_top, \
                                                        (int)pad_left, (int)Hf, (int)Wf, scale_zp, q)
    switch (x_dtype) {
        case MIQ_F32: MIQ_FRAMED(MIQ_F32); break;
        case MIQ_F16: MIQ_FRAMED(MIQ_F16); break;
        default: MIQ_FRAMED(MIQ_BF16); break;
    }
#undef MIQ_FRAMED
    return check_launch("quantize_framed_kernel");
}

// Statistics + parameters (+ epilogue vectors) and the quantization (optionally into a zero-point frame): the two launches
// of the dynamic activation quantizer behind one call.
extern "C" int miq_dynamic_quantize_i8(const void* x, int x_dtype, int64_t planes, int64_t H, int64_t W, int64_t pad_top,
                                       int64_t pad_bottom, int64_t pad_left, int64_t pad_right, const float* w_scale,
                                       const float* bias, const float* w_rowsum, int64_t channels, float* scale_zp, float* alpha,
                                       float* corr, int8_t* q, void* workspace, size_t workspace_bytes, unsigned* counter,
                                       void* stream) {
    MIQ_REQUIRE(planes >= 0 && H >= 0 && W >= 0, MIQ_EINVAL, "miq_dynamic_quantize_i8: negative size");
    const int64_t n = size_product({planes, H, W});
    MIQ_REQUIRE(n >= 0, MIQ_ERANGE, "miq_dynamic_quantize_i8: the tensor's element count does not fit 62 bits");
    if (int rc = miq_dynamic_quant_params_i8(x, x_dtype, n, w_scale, bias, w_rowsum, channels, scale_zp, alpha, corr,
                                             workspace, workspace_bytes, counter, stream))
        return rc;
    return miq_quantize_tensor_i8(x, x_dtype, planes, H, W, pad_top, pad_bottom, pad_left, pad_right, scale_zp, q, stream);
}

// ---- a weight-only Linear on a dynamically quantized activation, as one call ---------------------------------------
namespace {
struct DynLayout {
    size_t vectors, parts, xq, gemm, total;   // byte offsets of the pieces in the workspace; total size
};
DynLayout dyn_layout(int64_t M, int64_t N, int64_t K) {
    DynLayout L{};
    size_t at = 256;                                                                 // [scale, zero point, its int32 bits]
    L.vectors = at; at += static_cast<size_t>(round_up(2 * N * 4, 256));             // alpha[N], corr[N]
    L.parts = at;   at += static_cast<size_t>(round_up(miq_dynamic_quant_workspace(M * K), 256));
    L.xq = at;      at += static_cast<size_t>(round_up(M * K, 256));                 // the quantized activation
    L.gemm = at;    at += miq_gemm_i8_workspace(MIQ_W_NK, M, N, K);
    L.total = at;
    return L;
}
}  // namespace

extern "C" size_t miq_linear_dynamic_w8a8_workspace(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!all_below_2_31({M, N, K}) || M * K >= (1ll << 31) || N * K >= (1ll << 31)) return 0;   // (the call answers MIQ_ERANGE)
    return dyn_layout(M, N, K).total;
}

extern "C" int miq_linear_dynamic_w8a8(const void* x, int x_dtype, const int8_t* W, int64_t M, int64_t N, int64_t K,
                                       const float* w_scale, const float* bias, const float* w_rowsum, void* out, int out_dtype,
                                       void* workspace, size_t workspace_bytes, unsigned* counter, void* stream) {
    MIQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, MIQ_EINVAL, "miq_linear_dynamic_w8a8: negative size");
    if (M == 0 || N == 0) return MIQ_OK;
    MIQ_REQUIRE(K > 0, MIQ_EINVAL, "miq_linear_dynamic_w8a8: K must be positive (the activation's statistics need elements)");
    MIQ_REQUIRE(all_below_2_31({M, N, K}) && M * K < (1ll << 31) && N * K < (1ll << 31), MIQ_ERANGE,
                "miq_linear_dynamic_w8a8: M * K and N * K must stay below 2^31");
    const DynLayout L = dyn_layout(M, N, K);
    MIQ_REQUIRE(workspace && workspace_bytes >= L.total, MIQ_EWORKSPACE,
                "miq_linear_dynamic_w8a8: needs %zu bytes of workspace (got %zu)", L.total, workspace_bytes);
    MIQ_REQUIRE(aligned(workspace, 256), MIQ_EALIGN, "miq_linear_dynamic_w8a8: workspace must be 256-byte aligned");
    char* ws = static_cast<char*>(workspace);
    float* scale_zp = reinterpret_cast<float*>(ws);
    float* alpha = reinterpret_cast<float*>(ws + L.vectors);
    float* corr = alpha + N;
    int8_t* xq = reinterpret_cast<int8_t*>(ws + L.xq);
    if (int rc = miq_dynamic_quant_params_i8(x, x_dtype, M * K, w_scale, bias, w_rowsum, N, scale_zp, alpha, corr, ws + L.parts,
                                             L.xq - L.parts, counter, stream))
        return rc;
    if (int rc = miq_quantize_tensor_i8(x, x_dtype, 1, 1, M * K, 0, 0, 0, 0, scale_zp, xq, stream)) return rc;
    return miq_gemm_i8(xq, W, MIQ_W_NK, M, N, K, corr, alpha, 0.0f, out, out_dtype, ws + L.gemm, L.total - L.gemm, stream);
}
