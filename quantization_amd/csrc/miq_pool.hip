// 2-D pooling with an int8 result (NCHW), HBM-bandwidth bound: one pass over the input.
//
// What the reference's QuantizedMaxPool2d / QuantizedAvgPool2d compute
// (quant/Weight_Activation/Quantizer.py:122-130): the torch pooling module on x.float(), then
// .to(torch.int8).  A thread produces four horizontally adjacent outputs (one packed 4-byte store
// when the output rows allow it); the window rows it reads are contiguous in x and overlap its
// neighbours', which the caches absorb.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "miq_common.h"

namespace {

using namespace miq;

struct PoolArgs {
    const void* x;
    int8_t* out;
    int64_t planes;  // B * C
    int H, W, Ho, Wo;
    int kh, kw, sh, sw, ph, pw;
    float divisor;   // kh * kw
};

template <int DT>
__device__ __forceinline__ float load_f32(const void* p, int64_t i);
template <>
__device__ __forceinline__ float load_f32<MIQ_F32>(const void* p, int64_t i) {
    return static_cast<const float*>(p)[i];
}
template <>
__device__ __forceinline__ float load_f32<MIQ_F16>(const void* p, int64_t i) {
    return __half2float(static_cast<const __half*>(p)[i]);
}
template <>
__device__ __forceinline__ float load_f32<MIQ_BF16>(const void* p, int64_t i) {
    return __uint_as_float(static_cast<uint32_t>(static_cast<const unsigned short*>(p)[i]) << 16);
}
template <>
__device__ __forceinline__ float load_f32<MIQ_I8>(const void* p, int64_t i) {
    return static_cast<float>(static_cast<const int8_t*>(p)[i]);
}

// float -> int8 as the host's static_cast chain does: truncate to int32, keep the low byte;
// NaN and values that do not fit int32 give 0 (the x86 "integer indefinite" 0x80000000).
__device__ __forceinline__ uint32_t to_i8_bits(float v) {
    return fabsf(v) < 2147483648.0f ? (static_cast<uint32_t>(static_cast<int>(v)) & 0xffu) : 0u;
}

// One 32-bit word of storage elements -> floats (1, 2 or 4 of them).
template <int DT>
struct Word;
template <>
struct Word<MIQ_F32> {
    static constexpr int kElems = 1;
    static __device__ __forceinline__ void unpack(uint32_t w, float* f) { f[0] = __uint_as_float(w); }
};
template <>
struct Word<MIQ_F16> {
    static constexpr int kElems = 2;
    static __device__ __forceinline__ void unpack(uint32_t w, float* f) {
        f[0] = __half2float(__ushort_as_half(static_cast<unsigned short>(w & 0xffffu)));
        f[1] = __half2float(__ushort_as_half(static_cast<unsigned short>(w >> 16)));
    }
};
template <>
struct Word<MIQ_BF16> {
    static constexpr int kElems = 2;
    static __device__ __forceinline__ void unpack(uint32_t w, float* f) {
        f[0] = __uint_as_float(w << 16);
        f[1] = __uint_as_float(w & 0xffff0000u);
    }
};
template <>
struct Word<MIQ_I8> {
    static constexpr int kElems = 4;
    static __device__ __forceinline__ void unpack(uint32_t w, float* f) {
        const int v = static_cast<int>(w);
        f[0] = static_cast<float>((v << 24) >> 24);
        f[1] = static_cast<float>((v << 16) >> 24);
        f[2] = static_cast<float>((v << 8) >> 24);
        f[3] = static_cast<float>(v >> 24);
    }
};

// A chunk of E elements (4, 8 or 16 bytes... 16 bytes at most) starting at element index i, as floats.
template <int DT, int E>
__device__ __forceinline__ void load_chunk(const void* p, int64_t i, float (&f)[E]) {
    constexpr int kWords = E / Word<DT>::kElems;
    static_assert(kWords == 1 || kWords == 2 || kWords == 4, "a chunk is one 4-, 8- or 16-byte load");
    const char* src = static_cast<const char*>(p) + i * (4 / Word<DT>::kElems);
    uint32_t w[kWords];
    if constexpr (kWords == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(src);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else if constexpr (kWords == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(src);
        w[0] = v.x; w[1] = v.y;
    } else {
        w[0] = *reinterpret_cast<const uint32_t*>(src);
    }
#pragma unroll
    for (int k = 0; k < kWords; ++k) Word<DT>::unpack(w[k], &f[k * Word<DT>::kElems]);
}

// Square window K, stride S, padding P at compile time and rows that are a whole number of aligned
// E-element chunks (W % E == 0, x aligned to a chunk): a thread produces E adjacent outputs; it loads
// the NCH chunks that cover the span of their windows with one vector load each and picks the taps
// out of registers, and stores its E int8 results with one store when Wo % E == 0.
template <int DT, bool MAX, int K, int S, int P, int E>
__global__ __launch_bounds__(256) void pool2d_to_i8_chunk_kernel(PoolArgs a) {
    constexpr int PC = (P + E - 1) / E;        // chunks to the left of the first output's column
    constexpr int OFF = E * PC - P;            // first tap's position inside the first chunk
    constexpr int NCH = (OFF + (E - 1) * S + K + E - 1) / E;
    const int wq = (a.Wo + E - 1) / E;
    const int wc = a.W / E;
    const int64_t total = a.planes * a.Ho * wq;
    const float neutral = MAX ? -INFINITY : 0.0f;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += 256ll * gridDim.x) {
        const int g = static_cast<int>(idx % wq);
        const int64_t row = idx / wq;
        const int ho = static_cast<int>(row % a.Ho);
        const int64_t base = (row / a.Ho) * a.H * a.W;
        float r[E];
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = neutral;
        const int h0 = ho * S - P;
        const int c0 = S * g - PC;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int ih = h0 + i;
            const bool hok = static_cast<unsigned>(ih) < static_cast<unsigned>(a.H);
            const int64_t rbase = base + static_cast<int64_t>(hok ? ih : 0) * a.W;
            float v[NCH * E];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cc = c0 + c;
                const bool ok = hok && static_cast<unsigned>(cc) < static_cast<unsigned>(wc);
                float f[E];
                load_chunk<DT, E>(a.x, rbase + static_cast<int64_t>(E) * (ok ? cc : 0), f);
#pragma unroll
                for (int t = 0; t < E; ++t) v[E * c + t] = ok ? f[t] : neutral;
            }
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float t = v[OFF + e * S + j];
                    if constexpr (MAX) {
                        r[e] = (t > r[e] || t != t) ? t : r[e];
                    } else {
                        r[e] = __fadd_rn(r[e], t);
                    }
                }
        }
        uint32_t pk[E / 4];
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            uint32_t b[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = to_i8_bits(MAX ? r[4 * q + e] : r[4 * q + e] / a.divisor);
            pk[q] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        }
        int8_t* o = a.out + row * a.Wo + g * E;
        if (a.Wo % E == 0) {
            if constexpr (E == 16) {
                *reinterpret_cast<uint4*>(o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            } else if constexpr (E == 8) {
                *reinterpret_cast<uint2*>(o) = make_uint2(pk[0], pk[1]);
            } else {
                *reinterpret_cast<uint32_t*>(o) = pk[0];
            }
        } else if ((a.Wo & 3) == 0) {   // rows of whole dwords (out is 4-byte aligned: checked by the caller)
#pragma unroll
            for (int q = 0; q < E / 4; ++q)
                if (g * E + 4 * q < a.Wo) reinterpret_cast<uint32_t*>(o)[q] = pk[q];
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (g * E + e < a.Wo) o[e] = static_cast<int8_t>((pk[e / 4] >> (8 * (e % 4))) & 0xffu);
        }
    }
}

// K, S > 0: square window K x K and stride S known at compile time (the loops unroll, the loads of a
// thread's four overlapping windows are issued together and shared); K = 0: any window, from the args.
// Loads are unconditional (out-of-range taps read element 0 of the row and are replaced by the
// neutral value), so no tap waits for a branch.
template <int DT, bool MAX, int K, int S>
__global__ __launch_bounds__(256) void pool2d_to_i8_kernel(PoolArgs a) {
    const int kh = K ? K : a.kh, kw = K ? K : a.kw, sh = K ? S : a.sh, sw = K ? S : a.sw;
    const int wq = (a.Wo + 3) / 4;   // groups of four outputs per output row
    const int64_t total = a.planes * a.Ho * wq;
    const float neutral = MAX ? -INFINITY : 0.0f;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += 256ll * gridDim.x) {
        const int g = static_cast<int>(idx % wq);
        const int64_t row = idx / wq;   // (plane, ho)
        const int ho = static_cast<int>(row % a.Ho);
        const int64_t plane = row / a.Ho;
        const int64_t base = plane * a.H * a.W;
        const int wo0 = g * 4;
        float r[4] = {neutral, neutral, neutral, neutral};
        const int h0 = ho * sh - a.ph;
        const int w0 = wo0 * sw - a.pw;
#pragma unroll
        for (int i = 0; i < kh; ++i) {
            const int ih = h0 + i;
            const bool hok = static_cast<unsigned>(ih) < static_cast<unsigned>(a.H);
            const int64_t rbase = base + static_cast<int64_t>(hok ? ih : 0) * a.W;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < kw; ++j) {
                    const int iw = w0 + e * sw + j;
                    const bool ok = hok && static_cast<unsigned>(iw) < static_cast<unsigned>(a.W);
                    float v = load_f32<DT>(a.x, rbase + (ok ? iw : 0));
                    v = ok ? v : neutral;
                    if constexpr (MAX) {
                        r[e] = (v > r[e] || v != v) ? v : r[e];   // NaN wins, as in torch's max_pool2d
                    } else {
                        r[e] = __fadd_rn(r[e], v);
                    }
                }
            }
        }
        uint32_t b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = to_i8_bits(MAX ? r[e] : r[e] / a.divisor);
        int8_t* o = a.out + row * a.Wo + wo0;
        if ((a.Wo & 3) == 0) {
            *reinterpret_cast<uint32_t*>(o) = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (wo0 + e < a.Wo) o[e] = static_cast<int8_t>(b[e]);
        }
    }
}

template <int DT, bool MAX>
int launch_mode(const PoolArgs& a, hipStream_t st) {
    const int64_t total = a.planes * a.Ho * ((a.Wo + 3) / 4);
    const int grid = static_cast<int>(ceil_div(total, 256) < (1 << 20) ? ceil_div(total, 256) : (1 << 20));
    const bool square = a.kh == a.kw && a.sh == a.sw && a.ph == a.pw;
    constexpr int kElemBytes = DT == MIQ_F32 ? 4 : DT == MIQ_I8 ? 1 : 2;
    constexpr int kE = 16 / kElemBytes;   // elements in a 16-byte chunk
    // 16-byte chunks need rows of whole chunks and outputs the vector store can address
    if (square && a.W % kE == 0 && aligned(a.x, 16) && (a.Wo % kE != 0 || aligned(a.out, kE))) {
        const int64_t work = a.planes * a.Ho * ((a.Wo + kE - 1) / kE);
        const int vgrid = static_cast<int>(ceil_div(work, 256) < (1 << 20) ? ceil_div(work, 256) : (1 << 20));
        if (a.kh == 3 && a.sh == 2 && a.ph == 1) {
            pool2d_to_i8_chunk_kernel<DT, MAX, 3, 2, 1, kE><<<vgrid, 256, 0, st>>>(a);
            return check_launch("pool2d_to_i8_chunk_kernel");
        }
        if (a.kh == 2 && a.sh == 2 && a.ph == 0) {
            pool2d_to_i8_chunk_kernel<DT, MAX, 2, 2, 0, kE><<<vgrid, 256, 0, st>>>(a);
            return check_launch("pool2d_to_i8_chunk_kernel");
        }
        if (a.kh == 3 && a.sh == 1 && a.ph == 1) {
            pool2d_to_i8_chunk_kernel<DT, MAX, 3, 1, 1, kE><<<vgrid, 256, 0, st>>>(a);
            return check_launch("pool2d_to_i8_chunk_kernel");
        }
    }
    if (square && a.kh == 2 && a.sh == 2) {
        pool2d_to_i8_kernel<DT, MAX, 2, 2><<<grid, 256, 0, st>>>(a);
    } else if (square && a.kh == 3 && a.sh == 2) {
        pool2d_to_i8_kernel<DT, MAX, 3, 2><<<grid, 256, 0, st>>>(a);
    } else if (square && a.kh == 3 && a.sh == 1) {
        pool2d_to_i8_kernel<DT, MAX, 3, 1><<<grid, 256, 0, st>>>(a);
    } else {
        pool2d_to_i8_kernel<DT, MAX, 0, 0><<<grid, 256, 0, st>>>(a);
    }
    return check_launch("pool2d_to_i8_kernel");
}

template <int DT>
int launch(const PoolArgs& a, int mode, hipStream_t st) {
    return mode == MIQ_POOL_MAX ? launch_mode<DT, true>(a, st) : launch_mode<DT, false>(a, st);
}

}  // namespace

extern "C" int miq_pool2d_to_i8(const void* x, int x_dtype, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw,
                                int sh, int sw, int ph, int pw, int mode, int8_t* out, void* stream) {
    MIQ_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0, MIQ_EINVAL, "miq_pool2d_to_i8: bad shape (B=%lld C=%lld H=%lld W=%lld)",
                (long long)B, (long long)C, (long long)H, (long long)W);
    MIQ_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0, MIQ_EINVAL,
                "miq_pool2d_to_i8: bad window (kernel %dx%d stride %dx%d pad %dx%d)", kh, kw, sh, sw, ph, pw);
    MIQ_REQUIRE(kh < (1 << 15) && kw < (1 << 15) && sh < (1 << 15) && sw < (1 << 15) && ph < (1 << 15) && pw < (1 << 15), MIQ_ERANGE,
                "miq_pool2d_to_i8: window, stride and padding must stay below 32768");
    // torch's rule (pad should be at most half of the kernel): no window lies wholly in the padding
    MIQ_REQUIRE(2 * ph <= kh && 2 * pw <= kw, MIQ_EINVAL, "miq_pool2d_to_i8: pad must be at most half the kernel size");
    MIQ_REQUIRE(mode == MIQ_POOL_MAX || mode == MIQ_POOL_AVG, MIQ_EINVAL, "miq_pool2d_to_i8: unknown mode %d", mode);
    MIQ_REQUIRE(x_dtype == MIQ_F32 || x_dtype == MIQ_F16 || x_dtype == MIQ_BF16 || x_dtype == MIQ_I8, MIQ_EINVAL,
                "miq_pool2d_to_i8: x_dtype must be MIQ_F32, MIQ_F16, MIQ_BF16 or MIQ_I8 (got %d)", x_dtype);
    const int64_t planes = miq::size_product({B, C});
    MIQ_REQUIRE(H < (1ll << 31) && W < (1ll << 31) && planes >= 0 && planes < (1ll << 40), MIQ_ERANGE,
                "miq_pool2d_to_i8: shape too large");
    MIQ_REQUIRE(H + 2 * ph >= kh && W + 2 * pw >= kw, MIQ_EINVAL, "miq_pool2d_to_i8: window larger than the padded input");
    if (B == 0 || C == 0) return MIQ_OK;
    MIQ_REQUIRE(x && out, MIQ_EINVAL, "miq_pool2d_to_i8: null pointer");
    const size_t esz = x_dtype == MIQ_F32 ? 4 : x_dtype == MIQ_I8 ? 1 : 2;
    MIQ_REQUIRE(miq::aligned(x, esz), MIQ_EALIGN, "miq_pool2d_to_i8: x not aligned to its element size");
    PoolArgs a{};
    a.x = x;
    a.out = out;
    a.planes = planes;
    a.H = static_cast<int>(H);
    a.W = static_cast<int>(W);
    a.Ho = static_cast<int>((H + 2 * ph - kh) / sh + 1);
    a.Wo = static_cast<int>((W + 2 * pw - kw) / sw + 1);
    a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw;
    a.divisor = static_cast<float>(kh * kw);
    // packed 4-byte stores need 4-byte aligned output rows
    if ((a.Wo & 3) == 0) MIQ_REQUIRE(miq::aligned(out, 4), MIQ_EALIGN, "miq_pool2d_to_i8: out must be 4-byte aligned");
    hipStream_t st = miq::as_stream(stream);
    switch (x_dtype) {
        case MIQ_F32: return launch<MIQ_F32>(a, mode, st);
        case MIQ_F16: return launch<MIQ_F16>(a, mode, st);
        case MIQ_BF16: return launch<MIQ_BF16>(a, mode, st);
        default: return launch<MIQ_I8>(a, mode, st);
    }
}
