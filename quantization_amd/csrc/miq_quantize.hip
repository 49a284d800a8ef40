// Per-output-channel symmetric int8 weight quantizer for gfx950 (HBM-bandwidth bound).
//
// One pass over HBM: a row (one output channel) is loaded once with 16-byte coalesced loads
// and kept in registers while its signed maximum is reduced (wave64 shuffles, then LDS
// across the 4 waves of a workgroup); the scale is derived in-kernel with IEEE division and
// the same registers are divided, rounded half-to-even, clamped and stored as packed int8.
// Algorithmic traffic: sizeof(T) bytes read + 1 byte written per element, 12 bytes per row.
//
// Semantics follow the reference's Qop (quant/Weight_Only/Qop.py:37-54, 74-102, 139-162):
// signed row maximum (not abs-max), divisor 127.5, floor 2^-23, zero point 0 as int64,
// q = clamp(rint(w / scale)) with a true division.  (The kernel multiplies by 1 / scale where it
// can prove the int8 equal to the true division's and divides otherwise: quantize_vec.)
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>

#include "miq_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr float kEps = 1.1920928955078125e-07f;  // 2^-23 = torch.finfo(float32).eps
constexpr int kBlock = 256;

// ---- storage-precision rounding of an fp32 value (result widened back to fp32) ----------
template <int DT>
__device__ __forceinline__ float round_to_storage(float x);
template <>
__device__ __forceinline__ float round_to_storage<MIQ_F32>(float x) {
    return x;
}
template <>
__device__ __forceinline__ float round_to_storage<MIQ_F16>(float x) {
    return __half2float(__float2half_rn(x));
}
template <>
__device__ __forceinline__ float round_to_storage<MIQ_BF16>(float x) {
    return __bfloat162float(__float2bfloat16(x));
}

// ---- 16-byte vector of storage elements -> floats --------------------------------------
template <int DT>
struct Vec;
template <>
struct Vec<MIQ_F32> {
    static constexpr int kElems = 4;
    using Store = uint32_t;  // 4 packed int8
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x);
        f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z);
        f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ float load1(const void* p, int64_t i) {
        return static_cast<const float*>(p)[i];
    }
};
template <>
struct Vec<MIQ_F16> {
    static constexpr int kElems = 8;
    using Store = uint2;  // 8 packed int8
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
struct Vec<MIQ_BF16> {
    static constexpr int kElems = 8;
    using Store = uint2;
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

// scale = max(max(m, 0) / 127.5, eps), each step in the weight's storage precision
// (Qop.py:41,49-51; eps is a 0-dim tensor and does not promote a 16-bit scale).
template <int DT>
__device__ __forceinline__ float scale_from_max(float m) {
    const float pos = fmaxf(m, 0.0f);
    const float s = round_to_storage<DT>(pos / 127.5f);
    return fmaxf(s, kEps);  // 2^-23 is exactly representable in f16 (subnormal) and bf16
}

// q = clamp(rint(w / s), -128, 127) as an int in [-128,127] (Qop.py:155,162).
template <int DT>
__device__ __forceinline__ int quantize_one(float w, float s) {
    float v = rintf(round_to_storage<DT>(w / s));
    v = fminf(fmaxf(v, -128.0f), 127.0f);
    return static_cast<int>(v);
}

__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    return (static_cast<uint32_t>(a) & 0xffu) | ((static_cast<uint32_t>(b) & 0xffu) << 8) |
           ((static_cast<uint32_t>(c) & 0xffu) << 16) | (static_cast<uint32_t>(d) << 24);
}

// clamp(rint(x), -128, 127) of four floats as four packed int8.  The clamp goes first (its bounds
// are integers, so it commutes with the rounding); adding 1.5 * 2^23 then rounds half-to-even and
// leaves the two's-complement byte in the low 8 bits of the float's bit pattern.
__device__ __forceinline__ uint32_t round_clamp_pack4(float a, float b, float c, float d) {
    constexpr float kMagic = 12582912.0f;
    const uint32_t ua = __float_as_uint(__fadd_rn(__builtin_amdgcn_fmed3f(a, -128.0f, 127.0f), kMagic));
    const uint32_t ub = __float_as_uint(__fadd_rn(__builtin_amdgcn_fmed3f(b, -128.0f, 127.0f), kMagic));
    const uint32_t uc = __float_as_uint(__fadd_rn(__builtin_amdgcn_fmed3f(c, -128.0f, 127.0f), kMagic));
    const uint32_t ud = __float_as_uint(__fadd_rn(__builtin_amdgcn_fmed3f(d, -128.0f, 127.0f), kMagic));
    const uint32_t lo = __builtin_amdgcn_perm(ub, ua, 0x0c0c0400u);   // a.b0, b.b0, 0, 0
    const uint32_t hi = __builtin_amdgcn_perm(ud, uc, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// One 16-byte vector of weights -> packed int8, bit-identical to quantize_one on every element.
//
// The true division (about a dozen instructions) is replaced by a multiplication with
// r = fl(1 / s) wherever that provably gives the same int8.  q' = fl(w * r) lies within three units
// in the last place of the correctly rounded quotient q = fl(w / s) (one rounding each in r, in the
// product and in q: relative error <= 3 * 2^-24).  The int8 can differ only if a rounding boundary lies between
// q' and q or on q: a half-integer for fp32 weights, a midpoint of the 16-bit storage format
// for f16 / bf16 weights (their quotient is rounded to storage before rint).  Lanes whose q' sits
// within a margin of such a boundary are "unsure"; if any lane of the wave is, the whole vector is
// redone with the true division (a few per cent of the vectors), otherwise q' is used.
template <int DT>
__device__ __forceinline__ typename Vec<DT>::Store quantize_vec(const float (&f)[Vec<DT>::kElems], float s, float r) {
    using V = Vec<DT>;
    float qf[V::kElems];
    bool unsure = false;
    if constexpr (DT == MIQ_F32) {
        float margin = 1.0f;   // min over the elements of the distance from q' to the nearest half-integer
#pragma unroll
        for (int e = 0; e < V::kElems; ++e) {
            qf[e] = __fmul_rn(f[e], r);
            margin = fminf(margin, 0.5f - fabsf(__fsub_rn(qf[e], rintf(qf[e]))));
        }
        // |q' - q| < 129 * 2^-22 wherever the clamp does not decide the result anyway
        unsure = margin < 0x1p-13f;
    } else {
        // |q' - q| <= 3 units in the last place of fp32: unsure when the bits the storage format drops
        // lie within 4 of their midpoint, i.e. (bits - (mid - 4)) mod 2^dropped <= 8.
        constexpr uint32_t kLowMask = DT == MIQ_BF16 ? 0xffffu : 0x1fffu;
        constexpr uint32_t kFrom = (DT == MIQ_BF16 ? 0x8000u : 0x1000u) - 4u;
        uint32_t margin = 0xffffffffu;
#pragma unroll
        for (int e = 0; e < V::kElems; e += 2) {
            const float q0 = __fmul_rn(f[e], r), q1 = __fmul_rn(f[e + 1], r);
            margin = min(margin, min((__float_as_uint(q0) - kFrom) & kLowMask, (__float_as_uint(q1) - kFrom) & kLowMask));
            if constexpr (DT == MIQ_BF16) {   // one v_cvt_pk_bf16_f32 for the pair
                const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{q0, q1}, bf16x2));
                qf[e] = __uint_as_float(u << 16);
                qf[e + 1] = __uint_as_float(u & 0xffff0000u);
            } else {
                qf[e] = round_to_storage<DT>(q0);
                qf[e + 1] = round_to_storage<DT>(q1);
            }
        }
        unsure = margin <= 8u;
    }
    if (__any(unsure)) {
#pragma unroll
        for (int e = 0; e < V::kElems; ++e) qf[e] = round_to_storage<DT>(f[e] / s);
    }
    if constexpr (V::kElems == 4) {
        return round_clamp_pack4(qf[0], qf[1], qf[2], qf[3]);
    } else {
        return make_uint2(round_clamp_pack4(qf[0], qf[1], qf[2], qf[3]), round_clamp_pack4(qf[4], qf[5], qf[6], qf[7]));
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Row held in registers.  TPR threads cooperate on one row (64 = one wave, 256 or 1024 = the whole
// workgroup); each keeps NV 16-byte vectors.  The launcher gives every row group its own workgroup
// (the loop below runs once); the stride form only keeps the kernel correct for any grid.
template <int DT, int NV, int TPR, bool NT = false, int BLOCK = (TPR > kBlock ? TPR : kBlock)>
__global__ __launch_bounds__(BLOCK) void quantize_rows_reg_kernel(const uint4* __restrict__ w,
                                                                   int64_t rows, int nvec, int8_t* __restrict__ q,
                                                                   float* __restrict__ scale,
                                                                   int64_t* __restrict__ zp,
                                                                   float* __restrict__ row_max) {
    using V = Vec<DT>;
    constexpr int kRowsPerBlock = BLOCK / TPR;
    constexpr int kWavesPerRow = TPR / 64;
    __shared__ float red[2][BLOCK / 64];
    const int t = threadIdx.x % TPR;
    const int sub = threadIdx.x / TPR;
    int phase = 0;
    for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * kRowsPerBlock; row0 < rows;
         row0 += static_cast<int64_t>(gridDim.x) * kRowsPerBlock) {
        const int64_t row = row0 + sub;
        const bool live = row < rows;
        const uint4* src = w + row * nvec;
        uint4 v[NV];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = t + j * TPR;
            if (live && idx < nvec) {
                if constexpr (NT) {
                    const u32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&src[idx]));
                    v[j] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                } else {
                    v[j] = src[idx];
                }
                float f[V::kElems];
                V::unpack(v[j], f);
#pragma unroll
                for (int e = 0; e < V::kElems; ++e) m = fmaxf(m, f[e]);
            }
        }
        m = wave_max(m);
        if constexpr (TPR > 64) {
            if ((threadIdx.x & 63) == 0) red[phase][threadIdx.x >> 6] = m;
            __syncthreads();
            m = red[phase][sub * kWavesPerRow];
#pragma unroll
            for (int k = 1; k < kWavesPerRow; ++k) m = fmaxf(m, red[phase][sub * kWavesPerRow + k]);
            phase ^= 1;  // the other slot is free again after the next row's barrier
        }
        const float s = scale_from_max<DT>(m);
        if (live && t == 0) {
            scale[row] = s;
            zp[row] = 0;
            if (row_max) row_max[row] = m;
        }
        typename V::Store* dst = reinterpret_cast<typename V::Store*>(q) + row * nvec;
        const float rcp_s = 1.0f / s;   // correctly rounded
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = t + j * TPR;
            if (live && idx < nvec) {
                float f[V::kElems];
                V::unpack(v[j], f);
                const typename V::Store pk = quantize_vec<DT>(f, s, rcp_s);
                if constexpr (NT) {
                    if constexpr (V::kElems == 4) {
                        __builtin_nontemporal_store(pk, &dst[idx]);
                    } else {
                        __builtin_nontemporal_store(u32x2{pk.x, pk.y}, reinterpret_cast<u32x2*>(&dst[idx]));
                    }
                } else {
                    dst[idx] = pk;
                }
            }
        }
    }
}

// Any row length and alignment: one workgroup per row, the row is read twice (the second
// read is served from L2: a row is at most a few hundred KB).  VECTOR selects 16-byte
// accesses (needs 16-byte aligned rows); otherwise element-wise accesses.
template <int DT, bool VECTOR>
__global__ __launch_bounds__(kBlock) void quantize_rows_stream_kernel(const void* __restrict__ w, int64_t rows,
                                                                      int64_t cols, int8_t* __restrict__ q,
                                                                      float* __restrict__ scale,
                                                                      int64_t* __restrict__ zp,
                                                                      float* __restrict__ row_max) {
    using V = Vec<DT>;
    __shared__ float red[2][kBlock / 64];
    int phase = 0;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        float m = -INFINITY;
        if constexpr (VECTOR) {
            const int64_t nvec = cols / V::kElems;
            const uint4* src = static_cast<const uint4*>(w) + row * nvec;
            for (int64_t i = threadIdx.x; i < nvec; i += kBlock) {
                float f[V::kElems];
                V::unpack(src[i], f);
#pragma unroll
                for (int e = 0; e < V::kElems; ++e) m = fmaxf(m, f[e]);
            }
        } else {
            for (int64_t i = threadIdx.x; i < cols; i += kBlock) m = fmaxf(m, V::load1(w, row * cols + i));
        }
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) red[phase][threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[phase][0], red[phase][1]), fmaxf(red[phase][2], red[phase][3]));
        phase ^= 1;
        const float s = scale_from_max<DT>(m);
        if (threadIdx.x == 0) {
            scale[row] = s;
            zp[row] = 0;
            if (row_max) row_max[row] = m;
        }
        if constexpr (VECTOR) {
            const int64_t nvec = cols / V::kElems;
            const uint4* src = static_cast<const uint4*>(w) + row * nvec;
            typename V::Store* dst = reinterpret_cast<typename V::Store*>(q) + row * nvec;
            const float rcp_s = 1.0f / s;   // correctly rounded
            // Uniform trip count: quantize_vec votes across the wave, so every lane must reach it.
            const int64_t nvec_up = (nvec + kBlock - 1) / kBlock * kBlock;
            for (int64_t i = threadIdx.x; i < nvec_up; i += kBlock) {
                float f[V::kElems] = {};
                if (i < nvec) V::unpack(src[i], f);
                const typename V::Store pk = quantize_vec<DT>(f, s, rcp_s);
                if (i < nvec) dst[i] = pk;
            }
        } else {
            for (int64_t i = threadIdx.x; i < cols; i += kBlock)
                q[row * cols + i] = static_cast<int8_t>(quantize_one<DT>(V::load1(w, row * cols + i), s));
        }
    }
}

// out[n,k] = scale[n] * (float(q[n,k]) - float(zp[n]))  (Qop.py:179)
__global__ __launch_bounds__(kBlock) void dequantize_rows_kernel(const int8_t* __restrict__ q,
                                                                 const float* __restrict__ scale,
                                                                 const int64_t* __restrict__ zp, int64_t rows,
                                                                 int64_t cols, float* __restrict__ out) {
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float s = scale[row];
        const float z = zp ? static_cast<float>(zp[row]) : 0.0f;
        for (int64_t i = threadIdx.x; i < cols; i += kBlock)
            out[row * cols + i] = __fmul_rn(s, __fsub_rn(static_cast<float>(q[row * cols + i]), z));
    }
}

// Signed row maximum only (Qop.compute_scales_zero_point, channel branch: Qop.py:78-92).
template <int DT>
__global__ __launch_bounds__(kBlock) void row_max_kernel(const void* __restrict__ w, int64_t rows, int64_t cols,
                                                         float* __restrict__ row_max) {
    using V = Vec<DT>;
    __shared__ float red[2][kBlock / 64];
    int phase = 0;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        float m = -INFINITY;
        for (int64_t i = threadIdx.x; i < cols; i += kBlock) m = fmaxf(m, V::load1(w, row * cols + i));
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) red[phase][threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[phase][0], red[phase][1]), fmaxf(red[phase][2], red[phase][3]));
        phase ^= 1;
        if (threadIdx.x == 0) row_max[row] = m;
    }
}

// q = clamp(rint(w / scale[r] + zp[r]), -128, 127) with caller-supplied parameters
// (Qop.quantize with pre-computed scales: Qop.py:144,154-157).  nscale is `rows` or 1.
template <int DT>
__global__ __launch_bounds__(kBlock) void quantize_given_kernel(const void* __restrict__ w, int64_t rows, int64_t cols,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ zp, int per_row,
                                                                int8_t* __restrict__ q) {
    using V = Vec<DT>;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float s = scale[per_row ? row : 0];
        const bool has_zp = zp != nullptr;
        const float z = has_zp ? zp[per_row ? row : 0] : 0.0f;
        for (int64_t i = threadIdx.x; i < cols; i += kBlock) {
            float v = round_to_storage<DT>(V::load1(w, row * cols + i) / s);
            if (has_zp) v = round_to_storage<DT>(__fadd_rn(v, z));
            v = fminf(fmaxf(rintf(v), -128.0f), 127.0f);
            q[row * cols + i] = static_cast<int8_t>(static_cast<int>(v));
        }
    }
}

// out[o,c,i] = scale[c] * (y[o,c,i] - zp[c]): Qop.dequantize(y, activation=True), Qop.py:175-179.
__global__ __launch_bounds__(kBlock) void dequantize_channels_kernel(const float* __restrict__ y,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ zp, int64_t outer,
                                                                     int64_t channels, int64_t inner,
                                                                     float* __restrict__ out) {
    const int64_t total = outer * channels * inner;
    for (int64_t idx = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(kBlock) * gridDim.x) {
        const int64_t c = (idx / inner) % channels;
        out[idx] = __fmul_rn(scale[c], __fsub_rn(y[idx], zp ? zp[c] : 0.0f));
    }
}

int grid_for(int64_t blocks_needed) {
    // 256 CUs x 8 resident workgroups; the kernels stride over what is left.
    const int64_t cap = 256 * 8;
    return static_cast<int>(blocks_needed < cap ? (blocks_needed < 1 ? 1 : blocks_needed) : cap);
}

template <int DT, int NV, int TPR>
void launch_reg(const void* w, int64_t rows, int nvec, int8_t* q, float* scale, int64_t* zp, float* row_max,
                hipStream_t st) {
    constexpr int kThreads = TPR > kBlock ? TPR : kBlock;
    // One row group per workgroup, no grid stride: with every load of a row issued up front, the
    // hardware dispatcher refilling CUs keeps HBM busier than a resident-grid loop does.
    const int64_t blocks = miq::ceil_div(rows, kThreads / TPR);
    quantize_rows_reg_kernel<DT, NV, TPR, true>
        <<<static_cast<unsigned>(blocks), kThreads, 0, st>>>(static_cast<const uint4*>(w), rows, nvec, q, scale, zp,
                                                              row_max);
}

template <int DT>
int dispatch(const void* w, int64_t rows, int64_t cols, int8_t* q, float* scale, int64_t* zp, float* row_max,
             hipStream_t st) {
    constexpr int kElems = Vec<DT>::kElems;
    constexpr int kStoreBytes = kElems;  // int8 bytes produced per 16-byte input vector
    const bool vec_ok = cols % kElems == 0 && miq::aligned(w, 16) && miq::aligned(q, kStoreBytes);
    if (!vec_ok) {
        quantize_rows_stream_kernel<DT, false><<<grid_for(rows), kBlock, 0, st>>>(w, rows, cols, q, scale, zp, row_max);
        return miq::check_launch("quantize_rows_stream_kernel");
    }
    // Threads per row grow with the row so that a lane keeps 4-8 vectors (64-128 B) in flight.
    const int nvec = static_cast<int>(cols / kElems);
    if (cols / kElems > 8192) {
        quantize_rows_stream_kernel<DT, true><<<grid_for(rows), kBlock, 0, st>>>(w, rows, cols, q, scale, zp, row_max);
        return miq::check_launch("quantize_rows_stream_kernel");
    } else if (nvec <= 64) {
        launch_reg<DT, 1, 64>(w, rows, nvec, q, scale, zp, row_max, st);
    } else if (nvec <= 128) {
        launch_reg<DT, 2, 64>(w, rows, nvec, q, scale, zp, row_max, st);
    } else if (nvec <= 256) {
        launch_reg<DT, 4, 64>(w, rows, nvec, q, scale, zp, row_max, st);
    } else if (nvec <= 512) {
        // 16-bit weights (twice the arithmetic per byte): two rows per workgroup, four vectors a lane: 37.1 -> 35.1 us on
        // bf16 [14336,4096] (profiles/r3_quantizer.txt; fp32 rows of this length measured equal either way)
        if (DT != MIQ_F32) launch_reg<DT, 4, 128>(w, rows, nvec, q, scale, zp, row_max, st);
        else launch_reg<DT, 2, 256>(w, rows, nvec, q, scale, zp, row_max, st);
    } else if (nvec <= 1024) {
        launch_reg<DT, 4, 256>(w, rows, nvec, q, scale, zp, row_max, st);
    } else if (nvec <= 2048) {
        launch_reg<DT, 8, 256>(w, rows, nvec, q, scale, zp, row_max, st);
    } else if (nvec <= 4096) {
        launch_reg<DT, 4, 1024>(w, rows, nvec, q, scale, zp, row_max, st);
    } else {
        launch_reg<DT, 8, 1024>(w, rows, nvec, q, scale, zp, row_max, st);
    }
    return miq::check_launch("quantize_rows_reg_kernel");
}

}  // namespace

extern "C" int miq_quantize_rows_sym_i8(const void* w, int w_dtype, int64_t rows, int64_t cols, int8_t* q,
                                        float* scale, int64_t* zp, float* row_max, void* stream) {
    MIQ_REQUIRE(rows >= 0 && cols >= 0, MIQ_EINVAL, "miq_quantize_rows_sym_i8: negative size (%lld x %lld)",
                (long long)rows, (long long)cols);
    if (rows == 0) return MIQ_OK;
    MIQ_REQUIRE(cols > 0, MIQ_EINVAL, "miq_quantize_rows_sym_i8: rows of zero length have no maximum");
    MIQ_REQUIRE(w && q && scale && zp, MIQ_EINVAL, "miq_quantize_rows_sym_i8: null pointer");
    MIQ_REQUIRE(miq::aligned(scale, 4) && miq::aligned(zp, 8) && miq::aligned(row_max, 4), MIQ_EALIGN,
                "miq_quantize_rows_sym_i8: scale/row_max must be 4-byte and zp 8-byte aligned");
    MIQ_REQUIRE(rows < (1ll << 31) && cols < (1ll << 31), MIQ_ERANGE,
                "miq_quantize_rows_sym_i8: at most 2^31-1 rows and columns");
    hipStream_t st = miq::as_stream(stream);
    switch (w_dtype) {
        case MIQ_F32:
            return dispatch<MIQ_F32>(w, rows, cols, q, scale, zp, row_max, st);
        case MIQ_F16:
            MIQ_REQUIRE(miq::aligned(w, 2), MIQ_EALIGN, "miq_quantize_rows_sym_i8: f16 input not 2-byte aligned");
            return dispatch<MIQ_F16>(w, rows, cols, q, scale, zp, row_max, st);
        case MIQ_BF16:
            MIQ_REQUIRE(miq::aligned(w, 2), MIQ_EALIGN, "miq_quantize_rows_sym_i8: bf16 input not 2-byte aligned");
            return dispatch<MIQ_BF16>(w, rows, cols, q, scale, zp, row_max, st);
        default:
            return miq::fail(MIQ_EINVAL, "miq_quantize_rows_sym_i8: unknown w_dtype %d", w_dtype);
    }
}

extern "C" int miq_dequantize_rows_i8(const int8_t* q, const float* scale, const int64_t* zp, int64_t rows,
                                      int64_t cols, float* out, void* stream) {
    MIQ_REQUIRE(rows >= 0 && cols >= 0, MIQ_EINVAL, "miq_dequantize_rows_i8: negative size");
    if (rows == 0 || cols == 0) return MIQ_OK;
    MIQ_REQUIRE(q && scale && out, MIQ_EINVAL, "miq_dequantize_rows_i8: null pointer");
    dequantize_rows_kernel<<<grid_for(rows), kBlock, 0, miq::as_stream(stream)>>>(q, scale, zp, rows, cols, out);
    return miq::check_launch("dequantize_rows_kernel");
}

extern "C" int miq_row_max(const void* w, int w_dtype, int64_t rows, int64_t cols, float* row_max, void* stream) {
    MIQ_REQUIRE(rows >= 0 && cols >= 0, MIQ_EINVAL, "miq_row_max: negative size");
    if (rows == 0) return MIQ_OK;
    MIQ_REQUIRE(cols > 0, MIQ_EINVAL, "miq_row_max: rows of zero length have no maximum");
    MIQ_REQUIRE(w && row_max, MIQ_EINVAL, "miq_row_max: null pointer");
    hipStream_t st = miq::as_stream(stream);
    switch (w_dtype) {
        case MIQ_F32:
            row_max_kernel<MIQ_F32><<<grid_for(rows), kBlock, 0, st>>>(w, rows, cols, row_max);
            break;
        case MIQ_F16:
            row_max_kernel<MIQ_F16><<<grid_for(rows), kBlock, 0, st>>>(w, rows, cols, row_max);
            break;
        case MIQ_BF16:
            row_max_kernel<MIQ_BF16><<<grid_for(rows), kBlock, 0, st>>>(w, rows, cols, row_max);
            break;
        default:
            return miq::fail(MIQ_EINVAL, "miq_row_max: unknown w_dtype %d", w_dtype);
    }
    return miq::check_launch("row_max_kernel");
}

extern "C" int miq_quantize_rows_given_i8(const void* w, int w_dtype, int64_t rows, int64_t cols, const float* scale,
                                          const float* zp, int per_row, int8_t* q, void* stream) {
    MIQ_REQUIRE(rows >= 0 && cols >= 0, MIQ_EINVAL, "miq_quantize_rows_given_i8: negative size");
    if (rows == 0 || cols == 0) return MIQ_OK;
    MIQ_REQUIRE(w && scale && q, MIQ_EINVAL, "miq_quantize_rows_given_i8: null pointer");
    hipStream_t st = miq::as_stream(stream);
    switch (w_dtype) {
        case MIQ_F32:
            quantize_given_kernel<MIQ_F32><<<grid_for(rows), kBlock, 0, st>>>(w, rows, cols, scale, zp, per_row, q);
            break;
        case MIQ_F16:
            quantize_given_kernel<MIQ_F16><<<grid_for(rows), kBlock, 0, st>>>(w, rows, cols, scale, zp, per_row, q);
            break;
        case MIQ_BF16:
            quantize_given_kernel<MIQ_BF16><<<grid_for(rows), kBlock, 0, st>>>(w, rows, cols, scale, zp, per_row, q);
            break;
        default:
            return miq::fail(MIQ_EINVAL, "miq_quantize_rows_given_i8: unknown w_dtype %d", w_dtype);
    }
    return miq::check_launch("quantize_given_kernel");
}

extern "C" int miq_dequantize_channels_f32(const float* y, const float* scale, const float* zp, int64_t outer,
                                           int64_t channels, int64_t inner, float* out, void* stream) {
    MIQ_REQUIRE(outer >= 0 && channels >= 0 && inner >= 0, MIQ_EINVAL, "miq_dequantize_channels_f32: negative size");
    const int64_t total = miq::size_product({outer, channels, inner});
    MIQ_REQUIRE(total >= 0, MIQ_ERANGE, "miq_dequantize_channels_f32: outer * channels * inner does not fit 62 bits");
    if (total == 0) return MIQ_OK;
    MIQ_REQUIRE(y && scale && out, MIQ_EINVAL, "miq_dequantize_channels_f32: null pointer");
    dequantize_channels_kernel<<<grid_for(miq::ceil_div(total, kBlock)), kBlock, 0, miq::as_stream(stream)>>>(
        y, scale, zp, outer, channels, inner, out);
    return miq::check_launch("dequantize_channels_kernel");
}

namespace {
// out[o,c,i] = (y[o,c,i] + bias[c]) * alpha[c] + beta: the layer epilogue on a float product.
__global__ __launch_bounds__(kBlock) void epilogue_channels_kernel(const float* __restrict__ y,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ alpha, float beta,
                                                                   int64_t total, int64_t channels, int64_t inner,
                                                                   float* __restrict__ out) {
    for (int64_t idx = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(kBlock) * gridDim.x) {
        const int64_t c = (idx / inner) % channels;
        out[idx] = __fadd_rn(__fmul_rn(__fadd_rn(y[idx], bias ? bias[c] : 0.0f), alpha[c]), beta);
    }
}
}  // namespace

extern "C" int miq_epilogue_channels_f32(const float* y, const float* bias, const float* alpha, float beta,
                                         int64_t outer, int64_t channels, int64_t inner, float* out, void* stream) {
    MIQ_REQUIRE(outer >= 0 && channels >= 0 && inner >= 0, MIQ_EINVAL, "miq_epilogue_channels_f32: negative size");
    const int64_t total = miq::size_product({outer, channels, inner});
    MIQ_REQUIRE(total >= 0, MIQ_ERANGE, "miq_epilogue_channels_f32: outer * channels * inner does not fit 62 bits");
    if (total == 0) return MIQ_OK;
    MIQ_REQUIRE(y && alpha && out, MIQ_EINVAL, "miq_epilogue_channels_f32: null pointer");
    epilogue_channels_kernel<<<grid_for(miq::ceil_div(total, kBlock)), kBlock, 0, miq::as_stream(stream)>>>(
        y, bias, alpha, beta, total, channels, inner, out);
    return miq::check_launch("epilogue_channels_kernel");
}
