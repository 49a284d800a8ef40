// Weight-only Linear forward: floating-point activations x int8 weights on the CDNA4 bf16 MFMA
// (v_mfma_f32_16x16x32_bf16) with the per-channel dequant epilogue fused:
//     out[m,n] = (sum_k x[m,k] * float(W[n,k]) + bias[n]) * alpha[n] + beta.
//
// Replaces what quant/Weight_Only/Quantizer.py:79-91 does per forward call: `weight.to(x.dtype)` (N*K*4
// bytes written and read again), a float GEMM, the bias add and Qop.dequantize (Qop.py:164-181).  Here
// the weight stays int8 all the way into LDS (one byte per element from HBM) and becomes bf16, exactly,
// in registers; the activation is carried as bf16 PLANES so that nothing is lost against the
// reference's fp32 product:
//   * bf16 activations: one plane, the tensor itself;
//   * fp32 activations: hi = bf16(x), lo = bf16(x - hi): x - hi - lo is below 2^-17 |x|, every
//     bf16 x int8 product is exact in fp32 and the MFMA accumulates in fp32;
//   * fp16 activations: the same two planes hold an fp16 value exactly (11 significant bits).
// A small pre-pass (split_planes_kernel) writes the planes into the caller's workspace.
//
// Tiling follows the int8 GEMM (miq_gemm_i8.hip): both operands K-contiguous, 128-byte LDS rows with
// the same chunk swizzle, LDS-DMA staging with out-of-range zero fill for every ragged edge, the
// weight as the MFMA's first operand.  A K-tile is 128 ELEMENTS: one 128-byte row piece of the weight
// and, per plane, two 128-byte sub-tiles (k 0-63 and 64-127) of the activation.  A lane reads 16
// weight bytes (k = 16c .. 16c+15, c = (lane>>4) + 4h) per 16-channel fragment and sub-tile h: two
// MFMA operands of eight k each, whose activation counterparts are the 16-byte chunks 2*(lane>>4) and
// 2*(lane>>4)+1 of sub-tile h.  The k order inside an MFMA never has to be known: both operands use
// the same k -> (lane, slot) map.
#include <algorithm>
#include <type_traits>

#include "miq_common.h"
#include "miq_conv_dense.h"
#include "miq_gemm_core.h"
#include "miq_patch.h"

namespace {

using namespace miq;

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

enum : int { kOutRaw = 7 };   // split-K slices: raw fp32 partial sums into a slab

struct W8fArgs {
    const void* X;      // NP bf16 planes [M,K], plane_stride bytes apart
    const int8_t* W;    // [N,K]
    const float* bias;
    const float* alpha;
    void* out;
    float beta;
    int M, N, K;
    uint32_t plane_stride;
    int tiles_m, tiles_n;
    int kt_per_split;   // K-tiles per blockIdx.y slice; 0 = all of K
    // Implicit-GEMM convolution (CONV kernels only): row m of X is output pixel m, whose K axis is KH runs of
    // Lp elements (L = KW * Cp of them real) starting row_bytes apart in the padded channels-last planes.
    int HoWo, Wo, cstride, Lp, L;
    uint32_t row_bytes, pix_bytes, img_bytes, lp_magic;
    int ldc = 0;        // elements between two output rows (GEMM kernels; = N)
    // Convolutions with a stride per axis, a dilation or unequal paddings (miq::conv2d_w8_dense); zeros = the plain form.
    int cstride_x = 0;          // stride along x (0: cstride serves both axes)
    uint32_t krow_bytes = 0;    // bytes between the runs of two consecutive filter rows (0: row_bytes; dilation along y)
    uint32_t origin = 0;        // byte offset of the first window in the padded copy (its border may be wider than the padding)
    int kwt = 0;                // > 0 (dilation along x): every tap is a run of its own, kwt taps per filter row, ...
    uint32_t kwt_magic = 0, tap_bytes = 0;   // ... tap_bytes apart
};

// Byte offset of run t of a window's K axis: a filter row, or a single tap under a dilation along x.
__device__ __forceinline__ uint32_t w8f_run_offset(const W8fArgs& g, uint32_t t) {
    const uint32_t rb = g.krow_bytes ? g.krow_bytes : g.row_bytes;
    if (g.kwt == 0) return t * rb;
    const uint32_t kh = __umulhi(t, g.kwt_magic);
    return kh * rb + (t - kh * g.kwt) * g.tap_bytes;
}

__device__ __forceinline__ float bf16_round(float x) { return static_cast<float>(static_cast<__bf16>(x)); }

// Eight consecutive activations as fp32 (the rows are 16-byte aligned on this path).
__device__ __forceinline__ void load8(const float* __restrict__ x, float (&f)[8]) {
    const v4f a = reinterpret_cast<const v4f*>(x)[0], b = reinterpret_cast<const v4f*>(x)[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = a[i];
        f[4 + i] = b[i];
    }
}
__device__ __forceinline__ void load8(const _Float16* __restrict__ x, float (&f)[8]) {
    typedef _Float16 v8h __attribute__((ext_vector_type(8)));
    const v8h a = *reinterpret_cast<const v8h*>(x);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = static_cast<float>(a[i]);
}

// hi + lo planes of eight consecutive activations.
template <typename T>
__device__ __forceinline__ void split8(const T* __restrict__ x, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    float f[8];
    load8(x, f);
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float fh[2], fl[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = f[2 * i + e];
            fh[e] = bf16_round(v);
            if (__builtin_isinf(fh[e]) && !__builtin_isinf(v))   // rounded past the largest bf16: truncate instead
                fh[e] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & 0xffff0000u);
            // (an infinite or NaN value is carried by hi alone)
            fl[e] = (__builtin_isinf(fh[e]) || fh[e] != fh[e]) ? 0.0f : __fsub_rn(v, fh[e]);
        }
        h[i] = pack2_bf16(fh[0], fh[1]);
        l[i] = pack2_bf16(fl[0], fl[1]);
    }
    *reinterpret_cast<uint4*>(hi) = uint4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<uint4*>(lo) = uint4{l[0], l[1], l[2], l[3]};
}

template <typename T>
__global__ __launch_bounds__(256) void split_planes_kernel(const T* __restrict__ x, uint16_t* __restrict__ hi,
                                                           uint16_t* __restrict__ lo, int64_t groups) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < groups; i += 256ll * gridDim.x)
        split8(x + i * 8, hi + i * 8, lo + i * 8);
}

// Eight int8 (two dwords) -> eight bf16, exact.
__device__ __forceinline__ v8bf int8x8_to_bf16(int a, int b) {
    uint32_t p[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = i ? b : a;
        p[2 * i] = pack2_bf16(static_cast<float>(static_cast<int8_t>(v)), static_cast<float>(static_cast<int8_t>(v >> 8)));
        p[2 * i + 1] = pack2_bf16(static_cast<float>(static_cast<int8_t>(v >> 16)), static_cast<float>(v >> 24));
    }
    return __builtin_bit_cast(v8bf, uint4{p[0], p[1], p[2], p[3]});
}

// Eight int8 (two dwords) -> eight fp16, exact: the byte b ^ 0x80 under the exponent byte 0x64 is the fp16 number
// 1024 + (b + 128), from which 1152 is subtracted two at a time.
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v8h int8x8_to_f16(int a, int b) {
    uint32_t p[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t v = static_cast<uint32_t>(i ? b : a) ^ 0x80808080u;
        const uint32_t lo = __builtin_amdgcn_perm(0x64646464u, v, 0x05010400u);   // b0 64 b1 64
        const uint32_t hi = __builtin_amdgcn_perm(0x64646464u, v, 0x07030602u);   // b2 64 b3 64
        const v2h bias{static_cast<_Float16>(1152.0f), static_cast<_Float16>(1152.0f)};
        p[2 * i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2h, lo) - bias);
        p[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2h, hi) - bias);
    }
    return __builtin_bit_cast(v8h, uint4{p[0], p[1], p[2], p[3]});
}

__device__ __forceinline__ uint32_t pack2_f16(float a, float b) {
    return __builtin_bit_cast(uint32_t, v2h{static_cast<_Float16>(a), static_cast<_Float16>(b)});
}
// fp16 activations carried as ONE plane keep their own format (T = _Float16, NP = 1); every other plane is bf16.
template <typename T, int NP>
constexpr bool kPlaneF16 = std::is_same_v<T, _Float16> && NP == 1;

// Two activations -> their packed 16-bit forms: the plane's own (bf16, or fp16 as it is) and, for two planes, the bf16 rest.
template <typename T, int NP>
__device__ __forceinline__ void split2(float v0, float v1, uint32_t& hi, uint32_t& lo) {
    if constexpr (kPlaneF16<T, NP>) {
        hi = pack2_f16(v0, v1);
        lo = 0u;
    } else {
        float fh[2], fl[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = e ? v1 : v0;
            fh[e] = bf16_round(v);
            fl[e] = 0.0f;
            if constexpr (NP == 2) {
                if (__builtin_isinf(fh[e]) && !__builtin_isinf(v))   // rounded past the largest bf16: truncate instead
                    fh[e] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & 0xffff0000u);
                fl[e] = (__builtin_isinf(fh[e]) || fh[e] != fh[e]) ? 0.0f : __fsub_rn(v, fh[e]);
            }
        }
        hi = pack2_bf16(fh[0], fh[1]);
        lo = pack2_bf16(fl[0], fl[1]);
    }
}

// The MFMA operand type of the weight-only GEMM kernels: bf16 (bf16 activations, or the hi / lo planes of fp32 ones), or fp16
// for fp16 activations, which then need no planes: fp16 x int8 products are exact in the fp32 accumulator either way.
template <bool F16>
struct W8Frag {
    using type = v8bf;
};
template <>
struct W8Frag<true> {
    using type = v8h;
};
template <bool F16>
__device__ __forceinline__ typename W8Frag<F16>::type int8x8_to_frag(int a, int b) {
    if constexpr (F16)
        return int8x8_to_f16(a, b);
    else
        return int8x8_to_bf16(a, b);
}
template <bool F16, typename F>
__device__ __forceinline__ v4f mfma_w8(F w, F x, v4f c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
}

__device__ __forceinline__ float dequant1f(float acc, float bias, float alpha, float beta) {
    return __fadd_rn(__fmul_rn(__fadd_rn(acc, bias), alpha), beta);
}

// STAGES = 2: double buffer, one __syncthreads per K-tile.  STAGES > 2: ring of K-tile buffers with
// STAGES-1 tiles in flight behind a counted vmcnt and one raw barrier per K-tile (the thin problems,
// whose K-tile of MFMAs is far shorter than a fetch).
template <int BM, int BN, int WAVES_M, int WAVES_N, int NP, int OUT, int STAGES, bool CONV = false, bool F16 = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_w8f_kernel(W8fArgs g) {
    using frag_t = typename W8Frag<F16>::type;
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int XSUB = BM * kTileK;          // one sub-tile of one plane
    constexpr int XBYTES = NP * 2 * XSUB;
    constexpr int STAGE = XBYTES + BN * kTileK;
    constexpr int XPIECES = NP * 2 * BM / 8, WPIECES = BN / 8;   // 8-row x 128-byte DMA pieces
    static_assert(XPIECES % NWAVES == 0 && WPIECES % NWAVES == 0, "pieces must divide among the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, 8, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const uint32_t kb = 2u * static_cast<uint32_t>(g.K);   // bytes per activation row
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.X, CONV ? NP * g.plane_stride : (NP - 1) * g.plane_stride + static_cast<uint32_t>(g.M) * kb);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.W, static_cast<uint32_t>(g.N) * g.K);

    // Convolution: byte offset of the window's first element for each tile row this lane stages.
    constexpr int kRowsPerLane = CONV ? (BM / 8) / NWAVES : 1;
    static_assert(!CONV || (BM / 8) % NWAVES == 0, "a lane must stage a fixed set of tile rows");
    uint32_t pixbase[kRowsPerLane];
    if constexpr (CONV) {
#pragma unroll
        for (int j = 0; j < kRowsPerLane; ++j) {
            const int m = m0 + (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + j * NWAVES) * 8 + ((threadIdx.x & 63) >> 3);
            if (m < g.M) {
                const int b = m / g.HoWo, pix = m - b * g.HoWo, ho = pix / g.Wo, wo = pix - ho * g.Wo;
                pixbase[j] = g.origin + static_cast<uint32_t>(b) * g.img_bytes + static_cast<uint32_t>(ho * g.cstride) * g.row_bytes +
                             static_cast<uint32_t>(wo * (g.cstride_x ? g.cstride_x : g.cstride)) * g.pix_bytes;
            } else {
                pixbase[j] = kOob;
            }
        }
    }

    v4f acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const int nt_all = (g.K + kTileK - 1) / kTileK;
    const int kt0 = g.kt_per_split ? static_cast<int>(blockIdx.y) * g.kt_per_split : 0;
    const int nt = g.kt_per_split ? (nt_all - kt0 < g.kt_per_split ? nt_all - kt0 : g.kt_per_split) : nt_all;
    if (g.kt_per_split) g.out = static_cast<float*>(g.out) + static_cast<size_t>(blockIdx.y) * g.M * g.N;

    const int r_in = lane >> 3, slot = lane & 7;
    auto stage = [&](char* buf, int kt) {
#pragma unroll
        for (int i = 0; i < XPIECES / NWAVES; ++i) {
            const int q = wave + i * NWAVES;              // wave-uniform
            const int sub = q / (BM / 8), pc = q % (BM / 8);
            const int p = sub >> 1, h = sub & 1;
            const int r = pc * 8 + r_in;
            const int c = slot ^ ((r >> 1) & 7);
            uint32_t off;
            if constexpr (CONV) {
                const uint32_t e0 = static_cast<uint32_t>(kt) * kTileK + 64 * h + 8 * c;   // first of this chunk's 8 elements
                const uint32_t kh = __umulhi(e0, g.lp_magic), inrow = e0 - kh * g.Lp;
                const uint32_t base = pixbase[i % kRowsPerLane];
                off = (base != kOob && e0 < static_cast<uint32_t>(g.K) && inrow < static_cast<uint32_t>(g.L))
                          ? p * g.plane_stride + base + w8f_run_offset(g, kh) + inrow * 2u : kOob;
            } else {
                const uint32_t kbyte = static_cast<uint32_t>(2 * kt + h) * kTileK + c * kChunk;
                const int row = m0 + r;
                off = (row < g.M && kbyte < kb) ? p * g.plane_stride + static_cast<uint32_t>(row) * kb + kbyte : kOob;
            }
            dma16(rx, buf + q * (8 * kTileK), off);
        }
#pragma unroll
        for (int i = 0; i < WPIECES / NWAVES; ++i) {
            const int piece = wave + i * NWAVES;
            const int r = piece * 8 + r_in;
            const int c = slot ^ ((r >> 1) & 7);
            const int k = kt * kTileK + c * kChunk;
            const int row = n0 + r;
            const uint32_t off = (row < g.N && k < g.K) ? static_cast<uint32_t>(row) * g.K + k : kOob;
            dma16(rw, buf + XBYTES + piece * (8 * kTileK), off);
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int wlo[2] = {frag_lane_offset(lane, 0), frag_lane_offset(lane, 1)};
    const int xlo[2] = {fr * kTileK + (((2 * fq) ^ (fr >> 1)) * kChunk), fr * kTileK + (((2 * fq + 1) ^ (fr >> 1)) * kChunk)};

    auto mma_tile = [&](const char* cur) {
        const char* wt = cur + XBYTES + wn * WN * kTileK;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            frag_t wf[2][FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const v4i raw = lds_read16(wt + j * 16 * kTileK + wlo[h]);
                wf[0][j] = int8x8_to_frag<F16>(raw[0], raw[1]);
                wf[1][j] = int8x8_to_frag<F16>(raw[2], raw[3]);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const char* xt = cur + (p * 2 + h) * XSUB + wm * WM * kTileK;
                    frag_t xf[FM];
#pragma unroll
                    for (int i = 0; i < FM; ++i) xf[i] = __builtin_bit_cast(frag_t, lds_read16(xt + i * 16 * kTileK + xlo[e]));
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = mfma_w8<F16>(wf[e][j], xf[i], acc[i][j]);
                }
        }
    };

    if constexpr (STAGES == 2) {
        stage(smem, kt0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt) stage(smem + ((t + 1) & 1) * STAGE, kt0 + t + 1);
            mma_tile(smem + (t & 1) * STAGE);
            __syncthreads();
        }
    } else {
        constexpr int kLoads = (XPIECES + WPIECES) / NWAVES;
        constexpr int kKeep = kLoads * (STAGES - 2);
        static_assert(kKeep <= 63, "vmcnt is a 6-bit counter");
        const int kt_end = kt0 + nt;   // tiles past the slice are requested past K: zero-filled, no traffic
        auto stage_or_skip = [&](char* buf, int kt) { stage(buf, kt < kt_end ? kt : nt_all); };
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s) stage_or_skip(smem + s * STAGE, kt0 + s);
        int cur = 0, fill = STAGES - 1;
        for (int t = 0; t < nt; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kKeep) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage_or_skip(smem + fill * STAGE, kt0 + t + STAGES - 1);
            mma_tile(smem + cur * STAGE);
            cur = cur + 1 == STAGES ? 0 : cur + 1;
            fill = fill + 1 == STAGES ? 0 : fill + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // D of a fragment: 4*(lane>>4) + reg -> output channel, lane&15 -> activation row.
    const int mrow = m0 + wm * WM + (lane & 15);
    const int ncol = n0 + wn * WN + (lane >> 4) * 4;
    const bool vec_ok = (g.N & 3) == 0;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = ncol + j * 16;
        float al[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (OUT != kOutRaw) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < g.N) {
                    al[e] = g.alpha[n + e];
                    if (g.bias) bi[e] = g.bias[n + e];
                }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mrow + i * 16;
            if (m >= g.M || n >= g.N) continue;
            const v4f a = acc[i][j];
            const size_t idx = static_cast<size_t>(m) * g.N + n;
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = OUT == kOutRaw ? a[e] : dequant1f(a[e], bi[e], al[e], g.beta);
            if constexpr (CONV) {   // NCHW: the 16 lanes of a group hold 16 consecutive pixels of a channel
                const int b = m / g.HoWo, pix = m - b * g.HoWo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= g.N) continue;
                    const size_t o = (static_cast<size_t>(b) * g.N + n + e) * g.HoWo + pix;
                    if constexpr (OUT == kOutBF16) {
                        static_cast<uint16_t*>(g.out)[o] = f32_to_bf16_bits(y[e]);
                    } else {
                        static_cast<float*>(g.out)[o] = y[e];
                    }
                }
            } else if constexpr (OUT == kOutBF16) {
                uint16_t* o = static_cast<uint16_t*>(g.out) + idx;
                if (vec_ok && n + 3 < g.N) {
                    *reinterpret_cast<uint2*>(o) = pack4_bf16(y[0], y[1], y[2], y[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < g.N) o[e] = f32_to_bf16_bits(y[e]);
                }
            } else {
                float* o = static_cast<float*>(g.out) + idx;
                if (vec_ok && n + 3 < g.N) {
                    *reinterpret_cast<v4f*>(o) = v4f{y[0], y[1], y[2], y[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < g.N) o[e] = y[e];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// 256 x 256 tile, software-pipelined, one activation plane (bf16 activations): large M.
//
// The schedule of the int8 GEMM's 256 x 256 kernel (miq_gemm_i8.hip) carried over: 8 waves as 2 (m) x 4 (n),
// a wave owns four 64 x 32 quadrants, a K-tile (here 64 elements) lives in LDS as four half-tiles X0 / X1
// (activation rows 0-127 / 128-255, 128 bytes each = 16 KiB) and W0 / W1 (weight rows likewise, 64 bytes each
// = 8 KiB), two K-tile buffers (96 KiB).  Per K-tile four phases of 16 MFMAs (one quadrant each); a phase is
// LDS fragment reads (+ widening the weight bytes) and one half-tile of LDS-DMA for a later K-tile, a counted
// vmcnt, a raw barrier, the MFMAs, a raw barrier:
//   P1: read W0 of tile t    | DMA W1(t+1) | vmcnt(6) | barrier | MFMA quadrant (0,0) | barrier
//   P2: read W1              | DMA X1(t+1) | vmcnt(6) | barrier | MFMA (0,1)          | barrier
//   P3: read X1              | DMA X0(t+2) | vmcnt(6) | barrier | MFMA (1,1)          | barrier
//   P4: read X0 of tile t+1  | DMA W0(t+2) | vmcnt(6) | barrier | MFMA (1,0)          | barrier
// Waves 4-7 run one barrier behind waves 0-3.  Every wave issues two DMA instructions per X half-tile and one
// per W half-tile, six per K-tile, so vmcnt(6) after a phase's issue means "the half-tile issued four phases
// ago has landed"; the wait that retires a half-tile sits one barrier before its first read, and a half-tile
// is re-staged at least two phases after its last read.
//
// That is SCHED 0, kept in the developer library for A/B runs.  SCHED 1, what the product runs, is the int8 kernel's
// later form: TWO segments of 32 MFMAs per K-tile (half the barriers) and the two k-halves of an accumulator
// issued back to back (less power per MFMA; the order of the sums inside an accumulator is unchanged, so the
// results are bit-equal):
//   S1: read W0(t) W1(t)    | DMA W0(t+1) W1(t+1) | vmcnt(2): X1(t) X0(t+1) landed   | barrier | 32 MFMAs | barrier
//   S2: read X1(t) X0(t+1)  | DMA X1(t+1) X0(t+2) | vmcnt(4): W0(t+1) W1(t+1) landed | barrier | 32 MFMAs | barrier
//
// A weight half-tile keeps TWO 64-byte weight rows per 128-byte LDS row (row pair j = n >> 1; the 16-byte
// chunk c of row n is logical slot q = (n & 1) * 4 + c, stored at slot q ^ ((j >> 1) & 3)): a 16-lane group
// of a fragment read then touches 16 different 16-byte slots of the 256-byte bank row, as the activation
// reads do under the 128-byte-row swizzle.
// ---------------------------------------------------------------------------------------
#define MIQ_W8F_BARRIER()                   \
    do {                                    \
        asm volatile("" ::: "memory");      \
        __builtin_amdgcn_s_barrier();       \
        asm volatile("" ::: "memory");      \
    } while (0)
#define MIQ_W8F_WAIT_DMA() asm volatile("s_waitcnt vmcnt(6)" ::: "memory")

template <int OUT, int NP, int SCHED = 1, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_w8f_256_kernel(W8fArgs g) {
    using frag_t = typename W8Frag<F16>::type;
    constexpr int kXHalf = 128 * kTileK;            // 128 rows x 64 bf16
    constexpr int kWHalf = 64 * kTileK;             // 128 rows x 64 int8 = 64 row pairs
    constexpr int kBuf = 2 * kXHalf + 2 * kWHalf;   // X0 X1 W0 W1
    constexpr int kStepK = 64;                      // elements per K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, 8, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;

    const uint32_t kb = 2u * static_cast<uint32_t>(g.K);   // bytes per activation row
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.X, (NP - 1) * g.plane_stride + static_cast<uint32_t>(g.M) * kb);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.W, static_cast<uint32_t>(g.N) * g.K);

    // Two activation planes (fp32 / fp16 activations): the loop walks 2 * ceil(K / 64) K-tiles, tile u pairing
    // plane u & 1 with the weight's K-tile u >> 1 (the weight half-tiles are staged once per plane).
    // Staging.  X: a wave-instruction moves 8 rows x 128 B, two per half-tile and wave (row groups `wave` and
    // `wave + 8`).  W: 8 row pairs x 128 B, one per half-tile and wave.
    const int srow = wave * 8 + (lane >> 3);
    const int xcbyte = ((lane & 7) ^ ((srow >> 1) & 7)) * kChunk;
    const int wq = (lane & 7) ^ ((srow >> 1) & 3);           // logical slot this lane's LDS slot holds
    const int wrow = 2 * srow + (wq >> 2), wcbyte = (wq & 3) * kChunk;
    uint32_t xbase[2][2], wbase[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = m0 + h * 128 + srow + i * 64;
            xbase[h][i] = r < g.M ? static_cast<uint32_t>(r) * kb + xcbyte : kOob;
        }
        const int n = n0 + h * 128 + wrow;
        wbase[h] = n < g.N ? static_cast<uint32_t>(n) * g.K + wcbyte : kOob;
    }
    auto stage_x = [&](char* dst, const uint32_t (&base)[2], int kt) {
        const uint32_t k = static_cast<uint32_t>(NP == 2 ? kt >> 1 : kt) * kTileK;
        const bool k_ok = k + xcbyte < kb;
        const uint32_t pk = k + (NP == 2 ? (kt & 1) * g.plane_stride : 0u);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rx, dst + (wave + 8 * i) * (8 * kTileK), (k_ok && base[i] != kOob) ? base[i] + pk : kOob);
    };
    auto stage_w = [&](char* dst, uint32_t base, int kt) {
        const int k = (NP == 2 ? kt >> 1 : kt) * kStepK;
        dma16(rw, dst + wave * (8 * kTileK), (k + wcbyte < g.K && base != kOob) ? base + k : kOob);
    };

    v4f acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const int nt = NP * ((g.K + kStepK - 1) / kStepK);
    const int fr = lane & 15, fq = lane >> 4;
    const int xlo[2] = {fr * kTileK + (((2 * fq) ^ (fr >> 1)) * kChunk), fr * kTileK + (((2 * fq + 1) ^ (fr >> 1)) * kChunk)};
    const int wlo = (fr >> 1) * kTileK + ((((fr & 1) * 4 + fq) ^ ((fr >> 2) & 3)) * kChunk);
    const int x_frag = wm * 64 * kTileK;   // this wave's rows inside an X half-tile
    const int w_frag = wn * 16 * kTileK;   // its 32 weight rows = 16 row pairs inside a W half-tile

    char* const b0 = smem;
    char* const b1 = smem + kBuf;
    // Prologue: all of tile 0 plus X0 / W0 of tile 1; wait for X0(0), W0(0) only.
    stage_x(b0, xbase[0], 0);
    stage_w(b0 + 2 * kXHalf, wbase[0], 0);
    stage_w(b0 + 2 * kXHalf + kWHalf, wbase[1], 0);
    stage_x(b0 + kXHalf, xbase[1], 0);
    stage_x(b1, xbase[0], 1);
    if constexpr (SCHED == 0) {
        stage_w(b1 + 2 * kXHalf, wbase[0], 1);
        MIQ_W8F_WAIT_DMA();
    } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // X0(0), W0(0), W1(0) landed; X1(0), X0(1) in flight
    }
    MIQ_W8F_BARRIER();
    if (wm == 1) MIQ_W8F_BARRIER();

    frag_t xf[2][4], xf2[2][4], wf0[2][2], wf1[2][2];   // [k half][fragment]
    auto read_w = [&](frag_t (&wf)[2][2], const char* half) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const v4i raw = lds_read16(half + w_frag + j * 8 * kTileK + wlo);
            wf[0][j] = int8x8_to_frag<F16>(raw[0], raw[1]);
            wf[1][j] = int8x8_to_frag<F16>(raw[2], raw[3]);
        }
    };
    auto read_x = [&](frag_t (&f)[2][4], const char* half) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) f[e][i] = __builtin_bit_cast(frag_t, lds_read16(half + x_frag + i * 16 * kTileK + xlo[e]));
    };
    auto mma = [&](v4f (&c)[4][2], const frag_t (&f)[2][4], const frag_t (&wf)[2][2]) {
        if constexpr (SCHED == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        c[i][j] = mfma_w8<F16>(wf[e][j], f[e][i], c[i][j]);
        } else {
            // the two k-halves of an accumulator back to back, the weight fragment alternating 0 1 1 0 (pinned with
            // scheduling barriers: left alone the compiler restores the k-major order)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = (i & 1) ? 1 - jj : jj;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        c[i][j] = mfma_w8<F16>(wf[e][j], f[e][i], c[i][j]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
    };

    read_x(xf, b0);   // X0(0)
    for (int t = 0; t < nt; ++t) {
        char* cur = smem + (t & 1) * kBuf;
        char* nxt = smem + ((t + 1) & 1) * kBuf;

        if constexpr (SCHED == 1) {
            read_w(wf0, cur + 2 * kXHalf);
            read_w(wf1, cur + 2 * kXHalf + kWHalf);
            stage_w(nxt + 2 * kXHalf, wbase[0], t + 1);
            stage_w(nxt + 2 * kXHalf + kWHalf, wbase[1], t + 1);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // X1(t), X0(t+1)
            MIQ_W8F_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            mma(acc[0][0], xf, wf0);
            mma(acc[0][1], xf, wf1);
            __builtin_amdgcn_s_setprio(0);
            MIQ_W8F_BARRIER();

            read_x(xf2, cur + kXHalf);
            read_x(xf, nxt);
            stage_x(nxt + kXHalf, xbase[1], t + 1);
            stage_x(cur, xbase[0], t + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // W0(t+1), W1(t+1)
            MIQ_W8F_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            mma(acc[1][1], xf2, wf1);
            mma(acc[1][0], xf2, wf0);
            __builtin_amdgcn_s_setprio(0);
            MIQ_W8F_BARRIER();
            continue;
        }

        read_w(wf0, cur + 2 * kXHalf);
        stage_w(nxt + 2 * kXHalf + kWHalf, wbase[1], t + 1);
        MIQ_W8F_WAIT_DMA();   // W1(t) has landed
        MIQ_W8F_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        mma(acc[0][0], xf, wf0);
        __builtin_amdgcn_s_setprio(0);
        MIQ_W8F_BARRIER();

        read_w(wf1, cur + 2 * kXHalf + kWHalf);
        stage_x(nxt + kXHalf, xbase[1], t + 1);
        MIQ_W8F_WAIT_DMA();   // X1(t)
        MIQ_W8F_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        mma(acc[0][1], xf, wf1);
        __builtin_amdgcn_s_setprio(0);
        MIQ_W8F_BARRIER();

        read_x(xf2, cur + kXHalf);
        stage_x(cur, xbase[0], t + 2);
        MIQ_W8F_WAIT_DMA();   // X0(t+1)
        MIQ_W8F_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        mma(acc[1][1], xf2, wf1);
        __builtin_amdgcn_s_setprio(0);
        MIQ_W8F_BARRIER();

        read_x(xf, nxt);      // X0(t+1): retired by the wait one phase ago
        stage_w(cur + 2 * kXHalf, wbase[0], t + 2);
        MIQ_W8F_WAIT_DMA();   // W0(t+1)
        MIQ_W8F_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        mma(acc[1][0], xf2, wf0);
        __builtin_amdgcn_s_setprio(0);
        MIQ_W8F_BARRIER();
    }
    if (wm == 0) MIQ_W8F_BARRIER();   // every wave executes the same number of barriers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs issued for tiles past K

    store_tile_256<OUT>(g, acc, m0, n0, wm, wn, lane);
}

// Sums the split-K slabs and applies the epilogue.
template <int OUT>
__global__ __launch_bounds__(256) void w8f_splitk_reduce_kernel(const float* __restrict__ slabs, int splits, W8fArgs g) {
    const int64_t total = static_cast<int64_t>(g.M) * g.N;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += 256ll * gridDim.x) {
        const int n = static_cast<int>(idx % g.N);
        float sum = 0.f;
        for (int s = 0; s < splits; ++s) sum = __fadd_rn(sum, slabs[s * total + idx]);
        const float y = dequant1f(sum, g.bias ? g.bias[n] : 0.0f, g.alpha[n], g.beta);
        if constexpr (OUT == kOutF32) {
            static_cast<float*>(g.out)[idx] = y;
        } else {
            static_cast<uint16_t*>(g.out)[idx] = f32_to_bf16_bits(y);
        }
    }
}

// Any K, any alignment: one thread per output, fp32 multiply-adds in k order.
template <typename T, int OUT>
__global__ __launch_bounds__(256) void gemm_w8f_any_kernel(const T* __restrict__ x, W8fArgs g) {
    const int64_t total = static_cast<int64_t>(g.M) * g.N;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += 256ll * gridDim.x) {
        const int64_t m = idx / g.N;
        const int n = static_cast<int>(idx % g.N);
        const T* xr = x + m * g.K;
        const int8_t* wr = g.W + static_cast<int64_t>(n) * g.K;
        float sum = 0.f;
        for (int k = 0; k < g.K; ++k) sum = __fadd_rn(sum, __fmul_rn(static_cast<float>(xr[k]), static_cast<float>(wr[k])));
        const float y = dequant1f(sum, g.bias ? g.bias[n] : 0.0f, g.alpha[n], g.beta);
        if constexpr (OUT == kOutF32) {
            static_cast<float*>(g.out)[idx] = y;
        } else {
            static_cast<uint16_t*>(g.out)[idx] = f32_to_bf16_bits(y);
        }
    }
}

// Deepest ring (at most four K-tiles) that fits the 160 KiB of LDS.  (Six- and eight-deep rings of the 16- and
// 32-row tiles were measured: equal at best, up to 1.7x slower where they cost a second workgroup per CU.)
constexpr int w8f_stages(int bm, int bn, int np) {
    const int stage = (np * 2 * bm + bn) * kTileK;
    const int fit = (160 * 1024) / stage;
    return fit > 4 ? 4 : fit;
}

// Tile shape by M, from a measured sweep on the Llama-3-8B MLP weights ([14336,4096] and [4096,14336];
// tools/w8f_sweep.py).  The bytes a launch moves from L2 into LDS are
//     N*K * ceil(M / bm)  (weight)  +  planes * 2*M*K * ceil(N / bn)  (activation planes),
// but up to M = 128 the kernels are bound by how many workgroups stream at once rather than by those
// bytes: 32-row tiles (two or more workgroups per CU) beat 64- and 128-row ones, and 256-channel tiles pay
// once the activation planes outweigh the weight tile.  Thin problems (few output tiles, deep K) are cut
// along K into fp32 slabs so that every CU streams.
struct W8fPlan {
    int bm, bn;
    int splits, kt_per_split;
};

W8fPlan plan_w8f(int64_t M, int64_t N, int64_t K, int planes) {
    W8fPlan p{128, 128, 1, 0};
    if (M <= 16) {
        p.bm = 16; p.bn = 64;
    } else if (M <= 32) {
        p.bm = 32; p.bn = planes == 2 ? 256 : 64;
    } else if (M <= 64) {
        p.bm = planes == 2 ? 64 : 32; p.bn = planes == 2 ? 256 : 64;
    } else if (M <= 128) {
        p.bm = 32; p.bn = 256;
    } else if (planes == 1 && N >= 512) {
        // One activation plane leaves LDS for wider tiles, which move fewer bytes per flop but come in fewer,
        // longer rounds over the 256 CUs (one workgroup per CU in every case).  Measured time of a round,
        // relative to a 128 x 128 round: 1.6 for 128 x 256 (twice the area) and 2.5 for the software-pipelined
        // 256 x 256 kernel (four times the area): the shape with the least rounds x time runs (8192^3: 820 us
        // on 256 x 256 against 1070 and 1290; M = 2048 on [4096,14336], 128 tiles of 256 x 256: 223 us on
        // 128 x 256 against 285; M = 256 on [14336,4096]: 43 us on 128 x 128 against 49).
        const double r128 = static_cast<double>(ceil_div(ceil_div(M, 128) * ceil_div(N, 128), 256));
        const double r128x256 = 1.6 * static_cast<double>(ceil_div(ceil_div(M, 128) * ceil_div(N, 256), 256));
        const double r256 = 2.5 * static_cast<double>(ceil_div(ceil_div(M, 256) * ceil_div(N, 256), 256));
        if (r256 < r128 && r256 < r128x256) {
            p.bm = 256; p.bn = 256;
        } else if (r128x256 < r128) {
            p.bn = 256;
        }
    }
    if (planes == 2 && M > 128 && N >= 512) {
        // Two planes: the pipelined 256 x 256 kernel walks them as 2 * K / 64 K-tiles; a round of it takes 2.9
        // rounds of 128 x 128 tiles (8192^3: 1705 against 2290 us; 128 tiles on [4096,14336]: 605 against 500).
        const double r128 = static_cast<double>(ceil_div(ceil_div(M, 128) * ceil_div(N, 128), 256));
        const double r256 = 2.9 * static_cast<double>(ceil_div(ceil_div(M, 256) * ceil_div(N, 256), 256));
        if (r256 < r128) { p.bm = 256; p.bn = 256; }
    }
    while (p.bm < 128 && p.bn > 64 && N < 2 * p.bn) p.bn >>= 1;   // narrow weights: no empty tile columns
    if (const char* t = dev_switch("MIQ_W8F_BM")) {   // developer A/B switches
        const int bm = atoi(t);
        if (bm == 16 || bm == 32 || bm == 64 || bm == 128 || bm == 256) { p.bm = bm; p.bn = bm >= 128 ? 128 : 64; }
    }
    if (const char* t = dev_switch("MIQ_W8F_BN")) {
        const int bn = atoi(t);
        if (p.bm < 128 && (bn == 64 || bn == 128 || bn == 256)) p.bn = bn;
        if (p.bm == 128 && planes == 1 && (bn == 128 || bn == 256)) p.bn = bn;
        if (p.bm == 256 && bn == 256) p.bn = bn;
    }
    if (p.bm == 256) p.bn = 256;
    if (p.bm == 256 && p.bn == 256) return p;   // the pipelined kernel sums all of K
    const int64_t tiles = ceil_div(M, p.bm) * ceil_div(N, p.bn);
    const int64_t nt = ceil_div(K, kTileK);
    // Split-K for thin problems: as many slices as keep every workgroup resident at once (a slice more would
    // start a second round on a few CUs: 288 workgroups took 97 us where 192 took 67).
    const int64_t lds = static_cast<int64_t>(w8f_stages(p.bm, p.bn, planes)) * (planes * 2 * p.bm + p.bn) * kTileK;
    const int64_t resident = 256 * std::min<int64_t>(2, (160 * 1024) / lds);
    int64_t splits = 1;
    if (tiles < 160 && nt >= 16) {   // (a grid that already covers most CUs once is left whole)
        splits = resident / tiles;
        if (splits > nt / 4) splits = nt / 4;                // at least 4 K-tiles (512 of K) per slice
        if (splits > K / (8 * M)) splits = K / (8 * M);      // slab traffic (2 * 4*splits*M*N) <= weight bytes N*K
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
    }
    // Two planes at M <= 16 with a grid that covers most CUs once: every workgroup is bound by what one workgroup can
    // pull through the LDS-DMA, and a second one per CU on half of K gains more than the slab reduction costs
    // ([14336,4096] fp32, M = 16: 22.5 -> 18.6 us; one plane loses: 13.2 -> 15.6 us; three or four slices gain less).
    if (planes == 2 && p.bm == 16 && tiles >= 160 && tiles <= 256 && nt >= 16 && resident >= 2 * tiles) splits = 2;
    if (const char* t = dev_switch("MIQ_W8F_SPLITS")) {   // developer A/B switch: force the number of K slices
        const int64_t f = atoi(t);
        if (f >= 1 && f <= 32 && f <= nt) splits = f;
    }
    if (splits > 1) {
        p.kt_per_split = static_cast<int>(ceil_div(nt, splits));
        p.splits = static_cast<int>(ceil_div(nt, p.kt_per_split));
    }
    return p;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NP, int OUT, bool F16 = false>
int launch_w8f_tile(W8fArgs g, const W8fPlan& plan, float* slabs, hipStream_t st) {
    constexpr int STAGES = w8f_stages(BM, BN, NP);
    static_assert(STAGES >= 2, "a tile needs two K-tile buffers");
    g.tiles_m = static_cast<int>(ceil_div(g.M, BM));
    g.tiles_n = static_cast<int>(ceil_div(g.N, BN));
    constexpr int kLds = STAGES * (NP * 2 * BM + BN) * kTileK;
    const int64_t nwg = static_cast<int64_t>(g.tiles_m) * g.tiles_n;
    if (nwg >= (1ll << 31)) return fail(MIQ_ERANGE, "miq_gemm_w8: too many tiles");
    if (plan.splits > 1) {
        W8fArgs part = g;
        part.out = slabs;
        part.kt_per_split = plan.kt_per_split;
        auto kern = gemm_w8f_kernel<BM, BN, WAVES_M, WAVES_N, NP, kOutRaw, STAGES, false, F16>;
        if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
        kern<<<dim3(static_cast<unsigned>(nwg), static_cast<unsigned>(plan.splits)), 64 * WAVES_M * WAVES_N, kLds, st>>>(part);
        if (int rc = check_launch("gemm_w8f_kernel (split-K)")) return rc;
        const int64_t total = static_cast<int64_t>(g.M) * g.N;
        const int rgrid = static_cast<int>(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
        w8f_splitk_reduce_kernel<OUT><<<rgrid, 256, 0, st>>>(slabs, plan.splits, g);
        return check_launch("w8f_splitk_reduce_kernel");
    }
    auto kern = gemm_w8f_kernel<BM, BN, WAVES_M, WAVES_N, NP, OUT, STAGES, false, F16>;
    if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
    kern<<<static_cast<int>(nwg), 64 * WAVES_M * WAVES_N, kLds, st>>>(g);
    return check_launch("gemm_w8f_kernel");
}

template <int NP, int OUT, bool F16 = false>
int launch_w8f(const W8fArgs& g, const W8fPlan& plan, float* slabs, hipStream_t st) {
    switch (plan.bm * 1000 + plan.bn) {
        case 16064: return launch_w8f_tile<16, 64, 1, 4, NP, OUT, F16>(g, plan, slabs, st);
        case 16128: return launch_w8f_tile<16, 128, 1, 4, NP, OUT, F16>(g, plan, slabs, st);
        case 16256: return launch_w8f_tile<16, 256, 1, 4, NP, OUT, F16>(g, plan, slabs, st);
        case 32064: return launch_w8f_tile<32, 64, 1, 4, NP, OUT, F16>(g, plan, slabs, st);
        case 32128: return launch_w8f_tile<32, 128, 1, 4, NP, OUT, F16>(g, plan, slabs, st);
        case 32256: return launch_w8f_tile<32, 256, 1, 8, NP, OUT, F16>(g, plan, slabs, st);
        case 64064: return launch_w8f_tile<64, 64, 2, 2, NP, OUT, F16>(g, plan, slabs, st);
        case 64128: return launch_w8f_tile<64, 128, 2, 2, NP, OUT, F16>(g, plan, slabs, st);
        case 64256: return launch_w8f_tile<64, 256, 2, 4, NP, OUT, F16>(g, plan, slabs, st);
        default: break;
    }
    if (plan.bm == 256 && plan.bn == 256) {
        W8fArgs a = g;
        a.tiles_m = static_cast<int>(ceil_div(g.M, 256));
        a.tiles_n = static_cast<int>(ceil_div(g.N, 256));
        constexpr int kLds = 2 * (2 * 128 + 2 * 64) * kTileK;
        auto kern = gemm_w8f_256_kernel<OUT, NP, 1, F16>;
#ifndef MIQ_NO_DEV_SWITCHES
        if (const char* sc = dev_switch("MIQ_W8F_SCHED"))   // developer A/B switch: 0 = the four-phase schedule
            if (sc[0] == '0') kern = gemm_w8f_256_kernel<OUT, NP, 0, F16>;
#endif
        if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
        kern<<<a.tiles_m * a.tiles_n, 512, kLds, st>>>(a);
        return check_launch("gemm_w8f_256_kernel");
    }
    if constexpr (NP == 1) {   // one activation plane leaves LDS for wider tiles
        if (plan.bm == 128 && plan.bn == 256) return launch_w8f_tile<128, 256, 2, 4, NP, OUT, F16>(g, plan, slabs, st);
    }
    return launch_w8f_tile<128, 128, 2, 4, NP, OUT, F16>(g, plan, slabs, st);
}

template <typename T, int OUT>
int launch_any(const void* x, const W8fArgs& g, hipStream_t st) {
    const int64_t total = static_cast<int64_t>(g.M) * g.N;
    const int grid = static_cast<int>(total / 256 + 1 < 65536 ? total / 256 + 1 : 65536);
    gemm_w8f_any_kernel<T, OUT><<<grid, 256, 0, st>>>(static_cast<const T*>(x), g);
    return check_launch("gemm_w8f_any_kernel");
}

// ---------------------------------------------------------------------------------------
// Weight-only Conv2d on floating-point activations: the same kernel as an implicit GEMM.  A pre-pass turns the
// NCHW activation into zero-padded channels-last bf16 planes xp[plane][b][hp][wp][Cp] (Cp = Cin rounded up to
// 8, so that a 16-byte chunk never leaves a pixel's channels) and the OIHW weight into wp[Cout][KH][Lp] with
// (kw, c) order inside a filter row and zero fill (Lp = KW * Cp rounded up to 16): the K axis of output pixel
// (b, ho, wo) is then KH contiguous runs of the padded copy, one padded row apart.  Nothing like an im2col
// buffer is materialised.
// ---------------------------------------------------------------------------------------
struct ConvW8Geom {
    int64_t B, Cin, H, W, Cout, KH, KW, stride, pad;
    int64_t Ho, Wo, Hp, Wp, Cp, L, Lp, K, M;
    int64_t plane_bytes, wp_bytes;
    // 1 x 1 filters with a stride and no padding visit only every stride-th pixel of every stride-th row: the
    // re-layout keeps just those (sub = the caller's stride, Hs x Ws = the caller's image) and the problem
    // continues as a stride-1 convolution over an Ho x Wo image (H, W, stride above are then the reduced ones).
    int64_t sub, Hs, Ws;
};

bool make_conv_w8_geom(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
                       int64_t pad, ConvW8Geom& g) {
    g = ConvW8Geom{B, Cin, H, W, Cout, KH, KW, stride, pad, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, H, W};
    if (B < 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout < 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return false;
    // (sizes first: below 2^31 each, filter / stride / padding below 2^15, so that the sums and two-factor products below
    // stay inside int64; longer products are held at 2^60 and refused by the range checks of whoever asked)
    if (!all_below_2_31({B, Cin, H, W, Cout}) || KH >= (1 << 15) || KW >= (1 << 15) || stride >= (1 << 15) || pad >= (1 << 15))
        return false;
    const char* ss = dev_switch("MIQ_CONV_SUBSAMPLE");   // developer A/B switch: 0 = re-lay the whole image
    if (KH == 1 && KW == 1 && pad == 0 && stride > 1 && !(ss && ss[0] == '0')) {
        g.sub = stride;
        g.H = H = (H - 1) / stride + 1;
        g.W = W = (W - 1) / stride + 1;
        g.stride = stride = 1;
    }
    g.Hp = H + 2 * pad;
    g.Wp = W + 2 * pad;
    if (g.Hp < KH || g.Wp < KW) return false;
    g.Ho = (g.Hp - KH) / stride + 1;
    g.Wo = (g.Wp - KW) / stride + 1;
    g.Cp = round_up(Cin, 8);
    g.L = KW * g.Cp;
    g.Lp = round_up(g.L, 16);
    g.K = KH * g.Lp;
    g.M = sat_product({B, g.Ho, g.Wo});
    g.plane_bytes = round_up(sat_product({B, g.Hp, g.Wp, g.Cp, 2}), 256);
    g.wp_bytes = round_up(sat_product({Cout, g.K}), 256);
    return true;
}

// The whole problem inside what the kernels index with 31 bits (the condition the entry points refuse on): the other
// geometries (patch, stem) are computed only for such problems, whose products then stay far inside int64.
inline bool conv_w8_in_range(const ConvW8Geom& g, int planes) {
    return sat_product({planes, g.plane_bytes}) < (1ll << 31) && g.wp_bytes < (1ll << 31) && g.M < (1ll << 31) &&
           sat_product({g.K, g.Lp}) < (1ll << 31);
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(_Float16 v) { return static_cast<float>(v); }
__device__ __forceinline__ float to_f32(__bf16 v) { return static_cast<float>(v); }

// OIHW int8 -> wp[Cout][KH][Lp], (kw, c) order inside a filter row, zero where padded.
__device__ __forceinline__ void conv_w8_pack_w_items(const int8_t* __restrict__ w, int8_t* __restrict__ wp, uint32_t total,
                                                     int Cin, int KH, int KW, int Cp, int L, int Lp, uint32_t first,
                                                     uint32_t step) {
    for (uint32_t idx = first; idx < total; idx += step) {
        const uint32_t row = idx / Lp, l = idx - row * Lp, n = row / KH, kh = row - n * KH;
        int8_t v = 0;
        if (l < static_cast<uint32_t>(L)) {
            const uint32_t kw = l / Cp, c = l - kw * Cp;
            if (c < static_cast<uint32_t>(Cin)) v = w[((static_cast<size_t>(n) * Cin + c) * KH + kh) * KW + kw];
        }
        wp[idx] = v;
    }
}

// What rides along with the activation re-layout: the weight re-layout, shared by the n_wpack workgroups that
// follow every image's interior and border workgroups.
struct ConvW8PackArgs {
    const int8_t* w;
    int8_t* wp;
    uint32_t total;
    int n_wpack, KH, KW, L, Lp;
};

// A workgroup moves 64 consecutive interior pixels (flat h * W + w of one image) x 64 channels through an LDS tile:
// reads run along the pixels of one channel plane (VB-byte vectors, 16 or 8, when VB != 0: H * W a multiple of the vector's
// pixels, an aligned base and no subsampling; single elements otherwise), writes along the channels of one pixel
// (eight lanes fill a pixel's 128 bytes per plane).  Channels past Cin are written as zeros.  The workgroups after
// an image's `n_interior` write the zero border of its padded copy, one 16-byte store per plane and item, and the
// `n_wpack` after those the packed weights.
template <typename T, int NP, int VB>
__global__ __launch_bounds__(256) void conv_w8_prep_x_kernel(const T* __restrict__ x, uint16_t* __restrict__ xp, uint32_t plane_elems,
                                                             int Cin, int H, int W, int Hp, int Wp, int Cp, int pad,
                                                             int cblocks, int n_interior, int n_border, int sub, int Hs, int Ws,
                                                             ConvW8PackArgs pk) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x;
    const uint32_t per_img = n_interior + n_border + pk.n_wpack;
    const uint32_t b = blockIdx.x / per_img;
    const int bx = blockIdx.x - b * per_img;
    if (bx >= n_interior + n_border) {
        const uint32_t nimg = gridDim.x / per_img;
        conv_w8_pack_w_items(pk.w, pk.wp, pk.total, Cin, pk.KH, pk.KW, Cp, pk.L, pk.Lp,
                             (b * pk.n_wpack + (bx - n_interior - n_border)) * 256u + t, nimg * pk.n_wpack * 256u);
        return;
    }
    if (bx >= n_interior) {
        const uint32_t cgs = Cp >> 3, border = 2 * pad * Wp + 2 * pad * H;   // border pixels of one image
        const uint32_t idx = (bx - n_interior) * 256u + t;
        if (idx >= border * cgs) return;
        const uint32_t e = idx / cgs, cg = idx - e * cgs;
        uint32_t hp, wp;
        if (e < static_cast<uint32_t>(2 * pad * Wp)) {   // full rows: the first `pad` and the last `pad`
            const uint32_t r = e / Wp;
            hp = r < static_cast<uint32_t>(pad) ? r : H + r;
            wp = e - r * Wp;
        } else {                                         // side columns of the H interior rows
            const uint32_t q = e - 2 * pad * Wp, r = q / (2 * pad), k = q - r * 2 * pad;
            hp = r + pad;
            wp = k < static_cast<uint32_t>(pad) ? k : W + k;
        }
        uint16_t* o = xp + ((static_cast<size_t>(b) * Hp + hp) * Wp + wp) * Cp + cg * 8;
        *reinterpret_cast<uint4*>(o) = uint4{0u, 0u, 0u, 0u};
        if constexpr (NP == 2) *reinterpret_cast<uint4*>(o + plane_elems) = uint4{0u, 0u, 0u, 0u};
        return;
    }
    const int HW = H * W;
    const int p0 = (bx / cblocks) * 64, c0 = (bx % cblocks) * 64;
    const size_t cstride = static_cast<size_t>(Hs) * Ws;   // elements between channel planes of the source
    const T* xb = x + static_cast<size_t>(b) * Cin * cstride;
    if constexpr (VB != 0) {
        typedef uint32_t vecw __attribute__((ext_vector_type(VB / 4)));
        constexpr int V = VB / sizeof(T);      // pixels per load
        constexpr int PV = 64 / V;             // loads per channel row of the tile
#pragma unroll
        for (int i = 0; i < (64 * PV) / 256; ++i) {
            const int item = t + 256 * i, pv = item % PV, cl = item / PV, c = c0 + cl;
            alignas(16) T v[V];
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = static_cast<T>(0.0f);
            if (c < Cin && p0 + pv * V < HW)
                *reinterpret_cast<vecw*>(v) = *reinterpret_cast<const vecw*>(xb + c * cstride + p0 + pv * V);
#pragma unroll
            for (int e = 0; e < V; ++e) tile[cl][pv * V + e] = to_f32(v[e]);
        }
    } else {
        const int p = p0 + (t & 63);
        const int h = p / W, w = p - h * W;
        const size_t src = static_cast<size_t>(h * sub) * Ws + w * sub;   // every sub-th pixel of every sub-th row is kept
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cl = (t >> 6) + 4 * i, c = c0 + cl;
            tile[cl][t & 63] = (p < HW && c < Cin) ? to_f32(xb[src + c * cstride]) : 0.0f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int it = t + 256 * k, pixl = it >> 3, cg = it & 7;
        const int p = p0 + pixl;
        if (p >= HW || c0 + cg * 8 >= Cp) continue;
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2<T, NP>(tile[cg * 8 + 2 * j][pixl], tile[cg * 8 + 2 * j + 1][pixl], hi[j], lo[j]);
        const int h = p / W, w = p - h * W;
        uint16_t* o = xp + ((static_cast<size_t>(b) * Hp + h + pad) * Wp + w + pad) * Cp + c0 + cg * 8;
        *reinterpret_cast<uint4*>(o) = uint4{hi[0], hi[1], hi[2], hi[3]};
        if constexpr (NP == 2) *reinterpret_cast<uint4*>(o + plane_elems) = uint4{lo[0], lo[1], lo[2], lo[3]};
    }
}

// Few channels (Cp <= 16: first layers): one item = eight channels of one padded pixel, pixels fastest, so that a
// wave reads runs of one channel plane; no LDS.
template <typename T, int NP>
__global__ __launch_bounds__(256) void conv_w8_prep_x_small_kernel(const T* __restrict__ x, uint16_t* __restrict__ xp, uint32_t plane_elems,
                                                             int Cin, int H, int W, int Hp, int Wp, int Cp, int pad,
                                                             uint32_t npix, uint32_t items, int sub, int Hs, int Ws) {
    for (uint32_t it = blockIdx.x * 256u + threadIdx.x; it < items; it += 256u * gridDim.x) {
        const uint32_t cg = it / npix, ppix = it - cg * npix;
        const uint32_t b = ppix / (Hp * Wp), rem = ppix - b * (Hp * Wp);
        const int hp = rem / Wp, wp = rem - hp * Wp, h = hp - pad, w = wp - pad;
        uint32_t hi[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
        if (h >= 0 && h < H && w >= 0 && w < W) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cg * 8 + j;
                f[j] = c < Cin ? to_f32(x[((static_cast<size_t>(b) * Cin + c) * Hs + h * sub) * Ws + w * sub]) : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) split2<T, NP>(f[2 * j], f[2 * j + 1], hi[j], lo[j]);
        }
        uint16_t* o = xp + static_cast<size_t>(ppix) * Cp + cg * 8;
        *reinterpret_cast<uint4*>(o) = uint4{hi[0], hi[1], hi[2], hi[3]};
        if constexpr (NP == 2) *reinterpret_cast<uint4*>(o + plane_elems) = uint4{lo[0], lo[1], lo[2], lo[3]};
    }
}

__global__ __launch_bounds__(256) void conv_w8_pack_w_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wp, uint32_t total,
                                                             int Cin, int KH, int KW, int Cp, int L, int Lp) {
    conv_w8_pack_w_items(w, wp, total, Cin, KH, KW, Cp, L, Lp, blockIdx.x * 256u + threadIdx.x, 256u * gridDim.x);
}

template <int BN, int WAVES_M, int WAVES_N, int NP, int OUT, int DEPTH = 0, int BM = 128, bool F16 = false>
int launch_conv_w8f(W8fArgs g, hipStream_t st) {
    constexpr int STAGES = DEPTH ? DEPTH : w8f_stages(BM, BN, NP);
    g.tiles_m = static_cast<int>(ceil_div(g.M, BM));
    g.tiles_n = static_cast<int>(ceil_div(g.N, BN));
    constexpr int kLds = STAGES * (NP * 2 * BM + BN) * kTileK;
    const int64_t nwg = static_cast<int64_t>(g.tiles_m) * g.tiles_n;
    if (nwg >= (1ll << 31)) return fail(MIQ_ERANGE, "miq_conv2d_w8: too many tiles");
    auto kern = gemm_w8f_kernel<BM, BN, WAVES_M, WAVES_N, NP, OUT, STAGES, true, F16>;
    if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
    kern<<<static_cast<int>(nwg), 64 * WAVES_M * WAVES_N, kLds, st>>>(g);
    return check_launch("gemm_w8f_kernel (convolution)");
}

template <int NP, int OUT, bool F16 = false>
int launch_conv_w8f_np(const W8fArgs& g, hipStream_t st) {
    // 64-channel tiles up to 64 output channels (no empty tile columns), 128-channel tiles beyond
    const char* d = dev_switch("MIQ_CONV_W8_TILE");   // developer A/B switch
    if (d && !F16) {
        switch (d[0]) {
            case 'a': return launch_conv_w8f<64, 2, 2, NP, OUT, 2>(g, st);            // 128 x 64, double-buffered
            case 'b': return launch_conv_w8f<128, 2, 4, NP, OUT, 2>(g, st);           // 128 x 128, double-buffered
            case 'c': return launch_conv_w8f<64, 2, 2, NP, OUT, 2, 64>(g, st);        // 64 x 64, double-buffered
            case 'd': return launch_conv_w8f<128, 2, 2, NP, OUT, 2, 64>(g, st);       // 64 x 128, double-buffered
            case 'e': return launch_conv_w8f<64, 2, 2, NP, OUT, 0, 64>(g, st);        // 64 x 64, deepest ring
            case 'f': return launch_conv_w8f<256, 2, 4, NP, OUT, 2, 64>(g, st);       // 64 x 256, double-buffered
            case 'g':                                                                 // 128 x 256, one plane only
                if constexpr (NP == 1) return launch_conv_w8f<256, 2, 4, NP, OUT, 2>(g, st);
                break;
            default: break;
        }
    }
    // Up to 64 output channels (the large-image layers, K of a few hundred): two double-buffered workgroups per CU
    // (80 KiB of LDS each) hide each other's fetches better than one with a deeper ring: 128 x 64 tiles for one plane
    // (239 against 354 us on 64->64 56x56 at batch 256), 64 x 64 for two (361 against 478 us).
    if (g.N <= 64) {
        if constexpr (NP == 1) return launch_conv_w8f<64, 2, 2, NP, OUT, 2, 128, F16>(g, st);
        return launch_conv_w8f<64, 2, 2, NP, OUT, 2, 64>(g, st);
    }
    // One plane and more than 128 output channels: 128 x 256 tiles halve the passes over the expanded activations,
    // which is what bounds these layers (ResNet-18 at batch 256, bf16: 256->256 14x14 128 -> 112 us, 512->512 7x7
    // 113 -> 99 us, 128->256 3x3/2 89 -> 77 us), once there are tiles for most CUs.  Two planes would need 192 KiB.
    if constexpr (NP == 1) {
        if (g.N > 128 && ceil_div(g.M, 128) * ceil_div(g.N, 256) >= 160) return launch_conv_w8f<256, 2, 4, NP, OUT, 2, 128, F16>(g, st);
    }
    return launch_conv_w8f<128, 2, 4, NP, OUT, 0, 128, F16>(g, st);
}

size_t plane_bytes(int x_dtype, int64_t M, int64_t K) {
    return x_dtype == MIQ_F32 ? static_cast<size_t>(round_up(M * K * 2, 256)) : 0;   // bf16 and fp16 activations are used as they are
}
// As for the int8 GEMM: a count of 256 x 256 tiles a little over whole rounds of the 256 CUs is cut along M into
// the whole rounds (pipelined kernel) and a small second problem.  Returns the rows of the first part, or M.
int64_t w8f_rows_in_whole_rounds(int64_t M, int64_t N, int64_t K, int planes) {
    const char* sw = dev_switch("MIQ_W8F_ROWSPLIT");   // developer A/B switch: 0 = never cut
    if ((sw && sw[0] == '0') || M <= 256 || N < 512) return M;
    const int64_t tn = ceil_div(N, 256), tm = ceil_div(M, 256), tiles = tm * tn;
    if (tiles <= 256 || tiles % 256 == 0) return M;
    const int64_t tm_main = (tiles / 256) * 256 / tn;
    if (tm_main < 1 || tm_main >= tm) return M;
    const int64_t main_tiles = tm_main * tn, tail_tiles = ceil_div(M - tm_main * 256, 256) * tn;
    if ((main_tiles % 256 != 0 && main_tiles % 256 < 200) || tail_tiles > 96) return M;
    const W8fPlan head = plan_w8f(tm_main * 256, N, K, planes);
    return head.bm == 256 ? tm_main * 256 : M;   // only where the whole rounds do run on the pipelined kernel
}
size_t w8f_slab_bytes(int64_t M, int64_t N, int64_t K, int planes) {
    if (K % 16 != 0) return 0;
    const int64_t m_main = w8f_rows_in_whole_rounds(M, N, K, planes);
    if (m_main < M) M -= m_main;   // only the second part can need slabs
    const W8fPlan p = plan_w8f(M, N, K, planes);
    return p.splits > 1 ? static_cast<size_t>(round_up(static_cast<int64_t>(p.splits) * M * N * 4, 256)) : 0;
}

// ---------------------------------------------------------------------------------------
// 3 x 3 / stride 1 layers: the pixel patch of a tile stays in LDS (the int8 convolution's conv3x3_patch_kernel,
// miq_conv2d_i8.hip, carried over to the bf16 MFMA).
//
// In im2col order such a layer stages every activation byte nine times (64 -> 64 56x56 at batch 256, fp32: 9 taps x
// 2 planes x 110 MB L2 -> LDS per call, which is what bounded it).  Here the padded copy is kept in CHUNK PLANES of eight
// bf16 channels, xp[plane][b][hp][pitch][8] (row pitch a multiple of 4 slots), so the patch of 448 consecutive output
// pixels of one image -- every padded row their windows touch -- is one contiguous run per plane, copied by LDS-DMA as it
// lies; the convolution is then nine shifted GEMMs over the patch (fragment address = per-row base + the tap's uniform
// shift), 32 "virtual" channels (four planes) per stage:
//   * bf16 activations: group g = channels 32g .. 32g+31, plane 4g + q = channels 32g + 8q .. +7;
//   * fp32 / fp16 activations: group g = channels 16g .. 16g+15 as hi and lo planes (hi = bf16(x), lo = bf16(x - hi)):
//     planes 4g + {0, 1} = hi of channels 16g .. +7 / +8 .. +15, planes 4g + {2, 3} = lo of the same, and the weight
//     fragment of k-group q is that of q & 1, so one MFMA sums (hi + lo) * w over sixteen channels.
// The filters of a group sit beside the patch as [tap][k-group][output channel][8 bf16], widened from int8 (exactly) by
// the filter re-layout, so a tap carries no conversion (widened in registers, as the GEMM kernels do, a tap carried 40
// vector instructions beside its 14 MFMAs).  Everything else is the int8 kernel's structure: one persistent workgroup per CU (eight waves, 256
// registers, two LDS stages), the next stage fetched piece by piece among the nine taps, tap t + 1's fragments read under
// tap t's MFMAs, a finished unit's accumulators handed to a second register set and stored one fragment row per tap of
// the next unit, one barrier per stage; small images take several whole images per tile.
// ---------------------------------------------------------------------------------------
constexpr int kPatchW8Bytes = 4 * kPlaneSlots * kChunk;   // 44 KiB: four planes of a group

struct PatchW8Args {
    const void* xp;       // chunk planes
    const void* wp;       // [Cout / 64][groups][tap][k-group][64][8 bf16]
    const float* bias;
    const float* alpha;
    void* out;
    float beta;
    int Cout, ncg, tiles_n, tiles_img, units;
    int pitch, Hp, Wo, HoWo;
    uint32_t ptot;        // slots of one chunk plane: B * Hp * pitch
    uint32_t wo_magic, hw_magic;
    uint32_t xp_bytes, wp_bytes;
    int group, B, img_slots, pix_end;
    int ragged;           // HoWo % 4 != 0 (7 x 7 images): a lane's four pixels may lie in two images, stored one by one
    int ablate;           // developer library, timing only (wrong results): 1 = every patch fetch out of range (no memory
                          // traffic), 2 = every output store out of range (dropped)
};

// NCHW activation -> chunk planes (zero border, zero slots behind a row's last pixel).  One thread per eight channels of
// one padded pixel: its eight loads run along pixels with its neighbours' (coalesced per channel plane), its one or two
// 16-byte stores are consecutive with theirs.
template <typename T, int NP>
__device__ __forceinline__ void patch_w8_prep_items(const T* __restrict__ x, uint16_t* __restrict__ xp, int Cin, int H, int W, int Hp,
                                                    int pitch, int pad, uint32_t ptot, uint32_t total, uint32_t first, uint32_t step) {
    for (uint32_t idx = first; idx < total; idx += step) {
        const uint32_t c8 = idx / ptot, slot = idx - c8 * ptot;          // channel octet, flat padded pixel
        const uint32_t rowi = slot / pitch, s = slot - rowi * pitch;    // b * Hp + hp, column
        const uint32_t b = rowi / Hp, hp = rowi - b * Hp;
        const int h = static_cast<int>(hp) - pad, w = static_cast<int>(s) - pad;
        uint32_t hi[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
        if (h >= 0 && h < H && w >= 0 && w < W) {
            const T* src = x + ((static_cast<size_t>(b) * Cin + c8 * 8) * H + h) * W + w;
            const size_t cs = static_cast<size_t>(H) * W;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = to_f32(src[j * cs]);
#pragma unroll
            for (int j = 0; j < 4; ++j) split2<T, NP>(f[2 * j], f[2 * j + 1], hi[j], lo[j]);
        }
        // plane of the hi chunk: bf16: the octet itself; two planes: group c8 >> 1, k-group c8 & 1 (lo: + 2)
        const uint32_t plane = NP == 1 ? c8 : (c8 >> 1) * 4 + (c8 & 1);
        uint4* o = reinterpret_cast<uint4*>(xp) + static_cast<size_t>(plane) * ptot + slot;
        *o = uint4{hi[0], hi[1], hi[2], hi[3]};
        if constexpr (NP == 2) o[2 * static_cast<size_t>(ptot)] = uint4{lo[0], lo[1], lo[2], lo[3]};
    }
}

// OIHW int8 (3 x 3) -> wp[Cout / 64][groups][tap][k-group][channel % 64][8 bf16]: per 64-output-channel tile and group
// one contiguous image of the LDS weight stage (zero past Cout), already widened to bf16 (exact) so that the kernel's
// taps carry no conversion.  One item per 16-byte chunk: eight consecutive input channels of one tap.
template <int NP, bool F16 = false>
__device__ __forceinline__ void patch_w8_pack_items(const int8_t* __restrict__ w, uint4* __restrict__ wp, uint32_t total_chunks,
                                                    int Cin, int Cout, uint32_t first, uint32_t step) {
    constexpr uint32_t NQ = NP == 1 ? 4 : 2, CPG = NQ * 8;   // k-groups and channels of a group
    const uint32_t ncg = static_cast<uint32_t>(Cin) / CPG;
    for (uint32_t idx = first; idx < total_chunks; idx += step) {
        const uint32_t nl = idx & 63;
        uint32_t rest = idx >> 6;
        const uint32_t q = rest % NQ;
        rest /= NQ;
        const uint32_t tap = rest % 9, grp = rest / 9, cg = grp % ncg, n = (grp / ncg) * 64 + nl;
        const uint32_t c = cg * CPG + q * 8;
        uint32_t v[4] = {0u, 0u, 0u, 0u};
        if (n < static_cast<uint32_t>(Cout)) {
            const int8_t* src = w + (static_cast<size_t>(n) * Cin + c) * 9 + tap;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w0 = static_cast<float>(src[(2 * j) * 9]), w1 = static_cast<float>(src[(2 * j + 1) * 9]);
                v[j] = F16 ? pack2_f16(w0, w1) : pack2_bf16(w0, w1);
            }
        }
        wp[idx] = uint4{v[0], v[1], v[2], v[3]};
    }
}

// Both re-layouts in one launch: workgroups [0, n_prep) re-lay the activation, the n_pack after them the filters.
template <typename T, int NP>
__global__ __launch_bounds__(256) void patch_w8_prep_kernel(const T* __restrict__ x, uint16_t* __restrict__ xp, int Cin, int H,
                                                            int W, int Hp, int pitch, int pad, uint32_t ptot, uint32_t total,
                                                            int n_prep, const int8_t* __restrict__ w, uint4* __restrict__ wp,
                                                            uint32_t total_chunks, int Cout) {
    const int n_pack = static_cast<int>(gridDim.x) - n_prep;
    if (static_cast<int>(blockIdx.x) < n_prep)
        patch_w8_prep_items<T, NP>(x, xp, Cin, H, W, Hp, pitch, pad, ptot, total, blockIdx.x * 256u + threadIdx.x, 256u * n_prep);
    else
        patch_w8_pack_items<NP, kPlaneF16<T, NP>>(w, wp, total_chunks, Cin, Cout, (blockIdx.x - n_prep) * 256u + threadIdx.x, 256u * n_pack);
}

template <int OUT, int NP, bool STREAM, bool F16 = false>
__global__ __launch_bounds__(512, 2) void conv3x3_patch_w8_kernel(PatchW8Args g) {
    using frag_t = typename W8Frag<F16>::type;   // fp16 activations as one plane: fp16 fragments (filters widened to fp16)
    constexpr int FM = 7, FN = 2;
    constexpr int NQ = NP == 1 ? 4 : 2;                  // k-groups of the weight stage
    constexpr int kWst = 9 * NQ * 64 * kChunk;           // bytes of one weight stage (bf16)
    constexpr int kWInstr = kWst / 1024;                 // wave-instructions that fetch it (36 or 18)
    constexpr int kPatch = kPatchW8Bytes;
    constexpr bool kPiecesFirst = STREAM && NP == 2;   // order of a stage's fetch pieces and output rows: see the taps
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2] patches, [2] weight stages
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.xp, (g.ablate & 1) ? 0u : g.xp_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.wp, g.wp_bytes);
    // Filters stay resident when those of all the layer's groups fit beside the two patches (up to 72 KiB: 64 input
    // channels): a workgroup keeps one channel tile, so they are fetched once; otherwise a group's filters travel with
    // its patch, stage by stage.
    const bool w_res = g.ncg * kWst <= 160 * 1024 - 2 * kPatch;

    struct Unit {
        uint32_t b, p0, y0, tn;
    };
    auto decode = [&](int unit) {
        Unit u;
        u.tn = static_cast<uint32_t>(unit % g.tiles_n);
        const uint32_t t = static_cast<uint32_t>(unit / g.tiles_n);
        const uint32_t bi = t / g.tiles_img;
        u.b = bi * g.group;
        u.p0 = (t - bi * g.tiles_img) * kTM;
        u.y0 = u.p0 / g.Wo;
        return u;
    };
    // One stage's fetch is 44 wave-instructions of patch and kWInstr of weights, dealt over the eight waves: piece k of a
    // wave is patch instruction wave + 8k (k < 6) and weight instruction wave + 8k.  Returns whether it issued any.
    auto issue_piece = [&](const Unit& u, int cg, int buf, bool weights, int k) -> bool {
        bool any = false;
        const int q = wave + 8 * k;
        if (q < 44) {
            const uint32_t first = (u.b * g.Hp + u.y0) * g.pitch;
            const int plane = q / 11, seg = q - plane * 11;
            const uint32_t off = ((static_cast<uint32_t>(cg) * 4 + plane) * g.ptot + first + seg * 64 + lane) * kChunk;
            dma16(rx, smem + buf * kPatch + plane * (kPlaneSlots * kChunk) + seg * 1024, off);
            any = true;
        }
        if (weights && q < kWInstr) {
            dma16(rw, smem + 2 * kPatch + (w_res ? cg : buf) * kWst + q * 1024,
                  ((u.tn * g.ncg + cg) * kWInstr + q) * 1024 + lane * kChunk);
            any = true;
        }
        return any;
    };

    // per-channel epilogue parameters of this workgroup's channel tile, once
    const int tn_mine = static_cast<int>(blockIdx.x) % g.tiles_n;
    float al[FN], bi[FN];
#pragma unroll
    for (int c = 0; c < FN; ++c) {
        const int n = tn_mine * 64 + wn * 32 + c * 16 + (lane & 15);
        al[c] = 1.f;
        bi[c] = 0.f;
        if (n < g.Cout) {
            al[c] = g.alpha[n];
            if (g.bias) bi[c] = g.bias[n];
        }
    }

    v4f acc[FM][FN], done[FM][FN];   // the unit being accumulated; the finished one on its way out
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = done[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    constexpr int kEsz = OUT == kOutBF16 ? 2 : 4;
    constexpr int kNt = STREAM ? 2 : 0;   // cache policy of the output stores: 2 = nontemporal
    // One fragment row of the finished unit, as buffer stores into the unit's image(s): a lane with nothing to store
    // gets an offset past the buffer, which the hardware drops (no branch: the count of store instructions is a constant).
    auto store_row = [&](__amdgpu_buffer_rsrc_t r, const Unit& u, auto I) -> int {
        constexpr int i = decltype(I)::value;
        const uint32_t p = u.p0 + wm * FM * 16 + i * 16 + (lane >> 4) * 4;   // HoWo % 4 == 0: four pixels in or out together
        const uint32_t im = __umulhi(p, g.hw_magic), pp = p - im * g.HoWo;
        const uint32_t at = im * g.Cout * g.HoWo + pp;
        if (g.ragged) {   // (wave-uniform) every pixel on its own: four times the store instructions, for small outputs only
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = u.tn * 64 + wn * 32 + j * 16 + (lane & 15);
                v4f a = done[i][j];
                asm volatile("" : "+v"(a));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t pe = p + e, ime = __umulhi(pe, g.hw_magic), ppe = pe - ime * g.HoWo;
                    const uint32_t off = pe < static_cast<uint32_t>(g.pix_end) && n < g.Cout ? ((ime * g.Cout + n) * g.HoWo + ppe) * kEsz : kOob;
                    const float y = dequant1f(a[e], bi[j], al[j], g.beta);
                    if constexpr (OUT == kOutF32)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(y), r, off, 0, kNt);
                    else
                        __builtin_amdgcn_raw_buffer_store_b16(static_cast<short>(f32_to_bf16_bits(y)), r, off, 0, kNt);
                }
            }
            return 4 * FN;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = u.tn * 64 + wn * 32 + j * 16 + (lane & 15);
            v4f a = done[i][j];
            asm volatile("" : "+v"(a));   // (opaque: the conversion stays beside the tap it was written for)
            const uint32_t off = p < static_cast<uint32_t>(g.pix_end) && n < g.Cout ? (n * g.HoWo + at) * kEsz : kOob;
            const float y0 = dequant1f(a[0], bi[j], al[j], g.beta), y1 = dequant1f(a[1], bi[j], al[j], g.beta);
            const float y2 = dequant1f(a[2], bi[j], al[j], g.beta), y3 = dequant1f(a[3], bi[j], al[j], g.beta);
            if constexpr (OUT == kOutF32) {
                __builtin_amdgcn_raw_buffer_store_b128(v4i{__float_as_int(y0), __float_as_int(y1), __float_as_int(y2), __float_as_int(y3)},
                                                       r, off, 0, kNt);
            } else {
                const uint2 h = pack4_bf16(y0, y1, y2, y3);
                __builtin_amdgcn_raw_buffer_store_b64(v2u{h.x, h.y}, r, off, 0, kNt);
            }
        }
        return FN;
    };
    auto store_step = [&](__amdgpu_buffer_rsrc_t r, const Unit& u, auto T) -> int {
        if constexpr (decltype(T)::value >= FM)
            return 0;
        else
            return store_row(r, u, T);
    };
    const uint32_t img_bytes = static_cast<uint32_t>(g.Cout) * g.HoWo * kEsz;
    auto image_of = [&](const Unit& u) {   // (the unit's images that exist: stores to one past the batch are dropped)
        const uint32_t n = static_cast<uint32_t>(g.B) - u.b < static_cast<uint32_t>(g.group) ? g.B - u.b : g.group;
        return make_rsrc(static_cast<char*>(g.out) + static_cast<size_t>(u.b) * img_bytes, (g.ablate & 2) ? 0u : n * img_bytes);
    };

    // weight fragment of k-group lane >> 4 (two planes: of (lane >> 4) & 1): eight bf16 per lane
    const int w_lane = ((((lane >> 4) & (NQ - 1)) * 64) + wn * 32 + (lane & 15)) * kChunk;

    if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // the second wave of a SIMD goes first whenever both want to issue
    int unit = blockIdx.x;
    Unit cur = decode(unit);
#pragma unroll 1
    for (int k = 0; k < 6; ++k) (void)issue_piece(cur, 0, 0, true, k);
    if (w_res) {   // the other groups' filters
#pragma unroll 1
        for (int cg = 1; cg < g.ncg; ++cg)
            for (int q = wave; q < kWInstr; q += 8)
                dma16(rw, smem + 2 * kPatch + cg * kWst + q * 1024, ((cur.tn * g.ncg + cg) * kWInstr + q) * 1024 + lane * kChunk);
    }
    int buf = 0;
    int stores = 0;         // store instructions this wave issued after its last piece of the next stage's fetch
    bool pending = false;   // `done` holds a unit that has not left yet
    Unit prev{};
    __amdgpu_buffer_rsrc_t rout = image_of(cur);
    for (; unit < g.units; unit += gridDim.x) {
        cur = decode(unit);
        int aoff[FM];   // patch address of each fragment row this lane supplies: plane lane>>4, pixel (y - y0) * pitch + x
        {
            uint32_t qfirst = cur.p0 + wm * FM * 16 + (lane & 15);
            asm volatile("" : "+v"(qfirst));
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                uint32_t p = qfirst + i * 16;
                if (p >= static_cast<uint32_t>(g.pix_end)) p = g.pix_end - 1;   // rows past the unit: computed, never stored
                const uint32_t im = __umulhi(p, g.hw_magic), pp = p - im * g.HoWo;
                const uint32_t y = __umulhi(pp, g.wo_magic), x = pp - y * g.Wo;
                aoff[i] = (lane >> 4) * (kPlaneSlots * kChunk) + (im * g.img_slots + (y - cur.y0) * g.pitch + x) * kChunk;
            }
        }
        for (int cg = 0; cg < g.ncg; ++cg) {
            // This wave's pieces of the stage have landed once all but the stores it issued after the last of them are
            // done; the barrier tells the same of the other waves' pieces, and that every wave has left the stage before.
            wait_vmcnt_upto16(stores);
            stores = 0;
            __syncthreads();
            int ncg = cg + 1, nunit = unit;
            if (ncg == g.ncg) {
                ncg = 0;
                nunit = unit + gridDim.x;
            }
            const bool fetch = nunit < g.units;
            const Unit nxt = decode(fetch ? nunit : unit);
            const bool flush = pending && cg == 0;
            const char* patch = smem + buf * kPatch;
            const char* wst = smem + 2 * kPatch + (w_res ? cg : buf) * kWst + w_lane;
            {
                frag_t af[2][FM], wf[2][FN];   // the fragments of tap t + 1 are read while the MFMAs of tap t run
                auto read_tap = [&](auto S, int kh, int kw) {
                    constexpr int s = decltype(S)::value;
                    const char* arow = patch + (kh * g.pitch + kw) * kChunk;
                    const char* wrow = wst + (kh * 3 + kw) * (NQ * 64 * kChunk);
#pragma unroll
                    for (int j = 0; j < FN; ++j) wf[s][j] = __builtin_bit_cast(frag_t, lds_read16(wrow + j * 16 * kChunk));
#pragma unroll
                    for (int i = 0; i < FM; ++i) af[s][i] = __builtin_bit_cast(frag_t, lds_read16(arow + aoff[i]));
                };
                auto tap = [&](auto T) {
                    constexpr int t = decltype(T)::value, s = t & 1;
                    // Large fp32 outputs of fp32 activations (STREAM, four stages a unit): the next stage's six pieces go out
                    // beside taps 0-2 and the finished unit's rows beside taps 2-8, every row behind the last piece.  Vector
                    // memory operations retire in order, so a row stored before a piece has to be written out to HBM before
                    // that piece counts as landed at the next stage's start (64->64 56x56 at batch 256: 240 -> 217 us whole
                    // call; with one plane, or outputs the caches absorb, the bunched pieces cost 1-5 % instead).
                    if constexpr (kPiecesFirst) {
                        if constexpr (t < 3) {
                            if (fetch) {
                                const bool a = issue_piece(nxt, ncg, buf ^ 1, !w_res, 2 * t), b = issue_piece(nxt, ncg, buf ^ 1, !w_res, 2 * t + 1);
                                if (a || b) stores = 0;
                            }
                        }
                    } else if constexpr (t < 6) {
                        if (fetch && issue_piece(nxt, ncg, buf ^ 1, !w_res, t)) stores = 0;
                    }
                    if constexpr (t + 1 < 9) read_tap(std::integral_constant<int, (t + 1) & 1>{}, (t + 1) / 3, (t + 1) % 3);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = mfma_w8<F16>(af[s][i], wf[s][j], acc[i][j]);
                    // the finished unit's row t leaves behind this tap's MFMAs, as a block of its own behind a branch
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (kPiecesFirst) {
                        // (rows spread over all the unit's stages instead of its first: 226 -> 308 us whole call on
                        // 64->64 56x56 at batch 256, every stage then waits for stores)
                        if constexpr (t >= 2) {
                            if (flush) stores += store_step(rout, prev, std::integral_constant<int, t - 2>{});
                        }
                    } else {
                        if (flush) stores += store_step(rout, prev, T);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                read_tap(std::integral_constant<int, 0>{}, 0, 0);
                tap(std::integral_constant<int, 0>{});
                tap(std::integral_constant<int, 1>{});
                tap(std::integral_constant<int, 2>{});
                tap(std::integral_constant<int, 3>{});
                tap(std::integral_constant<int, 4>{});
                tap(std::integral_constant<int, 5>{});
                tap(std::integral_constant<int, 6>{});
                tap(std::integral_constant<int, 7>{});
                tap(std::integral_constant<int, 8>{});
            }
            if (flush) pending = false;
            buf ^= 1;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                done[i][j] = acc[i][j];
                acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
            }
        prev = cur;
        rout = image_of(prev);
        pending = true;
    }
    if (pending) {
        (void)store_step(rout, prev, std::integral_constant<int, 0>{});
        (void)store_step(rout, prev, std::integral_constant<int, 1>{});
        (void)store_step(rout, prev, std::integral_constant<int, 2>{});
        (void)store_step(rout, prev, std::integral_constant<int, 3>{});
        (void)store_step(rout, prev, std::integral_constant<int, 4>{});
        (void)store_step(rout, prev, std::integral_constant<int, 5>{});
        (void)store_step(rout, prev, std::integral_constant<int, 6>{});
    }
}

// Geometry of the patch path, or ok = false when the layer does not take it: 3 x 3 / stride 1, whole 32-channel groups,
// enough pixels for every CU, and a patch that fits the LDS planes.
// Activation planes of the patch and im2col-order convolution kernels: fp32 as hi + lo bf16, bf16 and fp16 as they are.
inline int conv_w8_planes(int x_dtype) { return x_dtype == MIQ_F32 ? 2 : 1; }

struct PatchW8Geom {
    bool ok;
    int64_t Hp, Wp, pitch, Ho, Wo, ptot, group, ncg;
    int64_t xp_bytes, wp_bytes;
};

PatchW8Geom make_patch_w8_geom(int planes, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW,
                               int64_t stride, int64_t pad) {
    PatchW8Geom p{};
    const char* sw = dev_switch("MIQ_CONV_W8_PATCH");   // developer A/B switch: 0 = im2col-order kernel for every layer
    if (KH != 3 || KW != 3 || stride != 1 || Cin % 32 != 0 || Cout <= 0 || (sw && sw[0] == '0')) return p;
    p.Hp = H + 2 * pad;
    p.Wp = W + 2 * pad;
    if (p.Hp < 3 || p.Wp < 3) return p;
    p.Ho = p.Hp - 2;
    p.Wo = p.Wp - 2;
    const int64_t HoWo = p.Ho * p.Wo;
    if (B * HoWo < 4 * kTM || HoWo >= 65536 || B >= 65536) return p;
    p.pitch = p.Wp;   // (the re-layout writes pixel by pixel: no need for rows of whole dwords of pixels)
    p.ptot = B * p.Hp * p.pitch;
    if (!patch_tiling(HoWo, p.Wo, p.Hp, p.pitch, kPlaneSlots, p.group)) return p;
    const int64_t cpg = planes == 1 ? 32 : 16;
    p.ncg = Cin / cpg;
    p.xp_bytes = (Cin / 8) * planes * p.ptot * kChunk;
    p.wp_bytes = ceil_div(Cout, 64) * p.ncg * (9 * (planes == 1 ? 4 : 2) * 64 * kChunk);
    // 31-bit offsets: the planes, the packed filters, one image (or `group` images) of output, the tile count
    if (p.xp_bytes >= (1ll << 31) || p.wp_bytes >= (1ll << 31) || p.group * Cout * HoWo * 4 >= (1ll << 31) ||
        ceil_div(B, p.group) * ceil_div(HoWo, kTM) * ceil_div(Cout, 64) >= (1ll << 31))
        return p;
    p.ok = true;
    return p;
}

template <int OUT, int NP, bool F16 = false>
int launch_patch_w8(const PatchW8Args& a, bool stream_out, hipStream_t st) {
    void (*kern)(PatchW8Args) = conv3x3_patch_w8_kernel<OUT, NP, false, F16>;
    if (stream_out) kern = conv3x3_patch_w8_kernel<OUT, NP, true, F16>;
    constexpr int kWst = 9 * (NP == 1 ? 4 : 2) * 64 * kChunk;
    static_assert(2 * kPatchW8Bytes + 2 * kWst <= 160 * 1024, "LDS of one CU");
    // two filter stages, or every group's filters when they fit (the kernel makes the same choice)
    const int kLds = 2 * kPatchW8Bytes + (a.ncg * kWst <= 160 * 1024 - 2 * kPatchW8Bytes ? std::max(a.ncg, 2) : 2) * kWst;
    if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
    int grid = a.units < 256 ? a.units : 256 - 256 % a.tiles_n;   // one persistent workgroup per CU, each keeping one channel tile
    if (grid < a.tiles_n) grid = a.units < a.tiles_n ? a.units : a.tiles_n;
    kern<<<grid, 512, kLds, st>>>(a);
    return check_launch("conv3x3_patch_w8_kernel");
}

// The whole patch path: re-layout into chunk planes, filter re-layout, the kernel.  `workspace` holds xp_bytes (rounded
// up to 256) + wp_bytes.
int run_patch_w8(const PatchW8Geom& P, const void* x, int x_dtype, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                 int64_t Cout, int64_t pad, const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                 void* workspace, hipStream_t st, const void* wp_packed = nullptr) {
    const int planes = conv_w8_planes(x_dtype);
    uint16_t* xp = static_cast<uint16_t*>(workspace);
    // the filters in the kernel's form: re-laid by this call behind the planes, or handed over already packed
    uint4* wp = wp_packed ? static_cast<uint4*>(const_cast<void*>(wp_packed))
                          : reinterpret_cast<uint4*>(static_cast<char*>(workspace) + round_up(P.xp_bytes, 256));
    const int64_t items = (Cin / 8) * P.ptot;   // one thread per channel octet and padded pixel
    MIQ_REQUIRE(items < (1ll << 32), MIQ_ERANGE, "miq_conv2d_w8: too many padded pixels for the chunk-plane re-layout");
    const int pgrid = static_cast<int>(std::min<int64_t>(ceil_div(items, 256), 1 << 20));
    const uint32_t wtotal = static_cast<uint32_t>(P.wp_bytes / kChunk);
    const int wgrid = wp_packed ? 0 : static_cast<int>(wtotal / 256 + 1 < 4096 ? wtotal / 256 + 1 : 4096);
#define MIQ_PATCH_PREP(T, NP)                                                                                                  \
    patch_w8_prep_kernel<T, NP><<<pgrid + wgrid, 256, 0, st>>>(static_cast<const T*>(x), xp, (int)Cin, (int)H, (int)W, (int)P.Hp, \
                                                               (int)P.pitch, (int)pad, (uint32_t)P.ptot, (uint32_t)items, pgrid, w, \
                                                               wp, wtotal, (int)Cout)
    if (x_dtype == MIQ_F32)
        MIQ_PATCH_PREP(float, 2);
    else if (x_dtype == MIQ_F16)
        MIQ_PATCH_PREP(_Float16, 1);
    else
        MIQ_PATCH_PREP(__bf16, 1);
#undef MIQ_PATCH_PREP
    if (int rc = check_launch("patch_w8_prep_kernel")) return rc;

    PatchW8Args a{};
    a.xp = xp; a.wp = wp; a.bias = bias; a.alpha = alpha; a.out = out; a.beta = beta;
    a.Cout = (int)Cout; a.ncg = (int)P.ncg; a.tiles_n = (int)ceil_div(Cout, 64);
    a.pitch = (int)P.pitch; a.Hp = (int)P.Hp; a.Wo = (int)P.Wo; a.HoWo = (int)(P.Ho * P.Wo);
    a.ptot = (uint32_t)P.ptot;
    a.wo_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(P.Wo)) + 1u;
    a.xp_bytes = (uint32_t)P.xp_bytes; a.wp_bytes = (uint32_t)P.wp_bytes;
    a.ragged = (P.Ho * P.Wo) % 4 != 0;
    if (const char* ab = dev_switch("MIQ_W8_PATCH_ABLATE")) a.ablate = atoi(ab);
    set_patch_units(a, B, P.Ho * P.Wo, P.Hp, P.pitch, P.group);
    const int64_t esz = out_dtype == MIQ_BF16 ? 2 : 4;
    const bool stream_out = B * Cout * P.Ho * P.Wo * esz >= (192ll << 20) && out_dtype != MIQ_BF16;
#define MIQ_PATCH_RUN(NP, F16) \
    (out_dtype == MIQ_F32 ? launch_patch_w8<kOutF32, NP, F16>(a, stream_out, st) : launch_patch_w8<kOutBF16, NP, F16>(a, false, st))
    if (x_dtype == MIQ_F16) return MIQ_PATCH_RUN(1, true);
    return planes == 1 ? MIQ_PATCH_RUN(1, false) : MIQ_PATCH_RUN(2, false);
#undef MIQ_PATCH_RUN
}

// ---------------------------------------------------------------------------------------
// Stems: at most four input channels, five to eight filter columns, at most seven filter rows, stride 2 (ResNet's
// 3 -> 64 7 x 7 / 2).  In im2col order such a layer's K axis is mostly padding (KH rows of 64 elements, of which
// KW * Cin are real) and every activation is staged a dozen times; the tile kernel took 940 us (fp32) / 640 us (bf16) for
// ResNet-18's first layer at batch 256, a quarter of the whole stack.  Here (the int8 convolution's conv_stem_kernel,
// miq_conv2d_i8.hip, carried over) the padded copy is PIXEL-major, one 16-byte pixel of eight virtual channels:
// (c0 c1 c2 c3 | 0 0 0 0) for bf16 activations, (hi c0..c3 | lo c0..c3) for fp32 / fp16; a workgroup of eight waves owns
// 8 output rows x 64 pixels x 64 channels, stages the patch those pixels see once (14 + KH rows of 136 pixels) and the
// channel tile's filters once per workgroup, and the im2col expansion happens between LDS and registers: K-step
// j = 2 * kh + half of output pixel (y, x) is the four pixels (2y + kh, 2x + 4 * half + q), q = lane >> 4, one 16-byte
// read per lane.  The filters are widened to bf16 by the re-layout: [K-step][q][channel][8], the hi and the lo half of a
// pixel meeting the same four weights.  Wave w multiplies output row w (four fragments of 16 pixels) by all 64 channels;
// the activation fragment is the MFMA's first operand, so a lane ends with four consecutive pixels of one channel.
// ---------------------------------------------------------------------------------------
constexpr int kStemW8Pitch = 136;                         // patch pixels per row: 63 * 2 + 8 = 134, rounded up
constexpr int kStemW8PatchBytes = 21 * kStemW8Pitch * kChunk + 512;   // 14 + KH <= 21 rows, fetched as 45 wave-instructions
constexpr int kStemW8MaxSteps = 14;                       // 2 * KH
constexpr int kStemW8WeightBytes = kStemW8MaxSteps * 4 * 64 * kChunk;

struct StemW8Args {
    const void* xp;       // [b][hp][wp][8 bf16]
    const void* wp;       // [Cout / 64][K-step][q][64][8 bf16]
    const float* bias;
    const float* alpha;
    void* out;
    float beta;
    int Cout, KH, ksteps;
    int Hp, Wp, Ho, Wo;
    int tiles_x, tiles_y, tiles_n, units;
    uint32_t xp_bytes, wp_bytes;
};

// NCHW activation (Cin <= 4) -> padded pixel-major copy, and OIHW int8 filters -> the kernel's weight image, in one launch:
// workgroups [0, n_prep) take one padded pixel per thread, the rest one 16-byte weight chunk per thread.
template <typename T, int NP>
__global__ __launch_bounds__(256) void stem_w8_prep_kernel(const T* __restrict__ x, uint4* __restrict__ xp, int Cin, int H, int W,
                                                           int Hp, int Wp, int pad, uint32_t npix, int n_prep,
                                                           const int8_t* __restrict__ w, uint4* __restrict__ wp, uint32_t wchunks,
                                                           int Cout, int KH, int KW) {
    if (static_cast<int>(blockIdx.x) < n_prep) {
        // Four pixels a thread (256 apart, so that a wave's loads and stores stay contiguous), every load issued before the
        // first conversion: one pixel a thread left too few bytes in flight to fill the memory pipes.
        const size_t cs = static_cast<size_t>(H) * W;
        for (uint32_t base = blockIdx.x * 1024u + threadIdx.x; base < npix; base += 1024u * n_prep) {
            // (loads without branches: a padding pixel reads the nearest pixel of its image and drops it)
            T raw[4][4];
            bool real[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t idx = base + 256u * k, idc = min(idx, npix - 1u);
                const uint32_t rowi = idc / Wp, s = idc - rowi * Wp, b = rowi / Hp, hp = rowi - b * Hp;
                const int h = static_cast<int>(hp) - pad, ww = static_cast<int>(s) - pad;
                real[k] = idx < npix && h >= 0 && h < H && ww >= 0 && ww < W;
                const size_t first = (static_cast<size_t>(b) * Cin * H + min(max(h, 0), H - 1)) * W + min(max(ww, 0), W - 1);
#pragma unroll
                for (int c = 0; c < 4; ++c) raw[k][c] = x[first + min(c, Cin - 1) * cs];   // (channels past Cin: dropped below)
            }
            float v[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[k][c] = (real[k] && c < Cin) ? to_f32(raw[k][c]) : 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t idx = base + 256u * k;
                if (idx >= npix) break;
                float fh[4], fl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float e = v[k][c];
                    fh[c] = kPlaneF16<T, NP> ? e : bf16_round(e);   // fp16 activations as one plane keep their format
                    if constexpr (NP == 2) {
                        if (__builtin_isinf(fh[c]) && !__builtin_isinf(e))   // rounded past the largest bf16: truncate instead
                            fh[c] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, e) & 0xffff0000u);
                        fl[c] = (__builtin_isinf(fh[c]) || fh[c] != fh[c]) ? 0.0f : __fsub_rn(e, fh[c]);
                    }
                }
                if constexpr (kPlaneF16<T, NP>)   // (fh holds the fp16 values themselves: see below)
                    xp[idx] = uint4{pack2_f16(fh[0], fh[1]), pack2_f16(fh[2], fh[3]), 0u, 0u};
                else
                    xp[idx] = uint4{pack2_bf16(fh[0], fh[1]), pack2_bf16(fh[2], fh[3]), pack2_bf16(fl[0], fl[1]), pack2_bf16(fl[2], fl[3])};
            }
        }
        return;
    }
    const int n_pack = static_cast<int>(gridDim.x) - n_prep;
    for (uint32_t idx = (blockIdx.x - n_prep) * 256u + threadIdx.x; idx < wchunks; idx += 256u * n_pack) {
        const uint32_t nl = idx & 63, q = (idx >> 6) & 3;
        const uint32_t rest = idx >> 8, j = rest % (2 * KH), tn = rest / (2 * KH);
        const uint32_t kh = j >> 1, kw = 4 * (j & 1) + q, n = tn * 64 + nl;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (n < static_cast<uint32_t>(Cout) && kw < static_cast<uint32_t>(KW)) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < Cin) v[c] = static_cast<float>(w[((static_cast<size_t>(n) * Cin + c) * KH + kh) * KW + kw]);
        }
        const uint32_t lo = kPlaneF16<T, NP> ? pack2_f16(v[0], v[1]) : pack2_bf16(v[0], v[1]);
        const uint32_t hi = kPlaneF16<T, NP> ? pack2_f16(v[2], v[3]) : pack2_bf16(v[2], v[3]);
        wp[idx] = NP == 2 ? uint4{lo, hi, lo, hi} : uint4{lo, hi, 0u, 0u};
    }
}

template <int OUT, bool F16 = false>
__global__ __launch_bounds__(512, 2) void conv_stem_w8_kernel(StemW8Args g) {
    using frag_t = typename W8Frag<F16>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2] patches, the channel tile's filters
    char* const wst = smem + 2 * kStemW8PatchBytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int tn = static_cast<int>(blockIdx.x) % g.tiles_n;   // the grid is a multiple of tiles_n: one channel tile per workgroup

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.xp, g.xp_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.wp, g.wp_bytes);
    struct Block {
        int x0, y0;
        uint32_t b;
    };
    auto decode = [&](uint32_t unit) {
        Block k;
        uint32_t t = unit / g.tiles_n;
        k.x0 = (t % g.tiles_x) * 64;
        t /= g.tiles_x;
        k.y0 = (t % g.tiles_y) * 8;
        k.b = t / g.tiles_y;
        return k;
    };
    // patch: rows 2*y0 .. 2*y0 + 13 + KH of the padded image, 136 pixels from pixel 2*x0 (a run that leaves the row continues
    // into the next one, one that leaves the buffer reads zeros: both only under pixels that are never stored)
    const int pieces = (14 + g.KH) * kStemW8Pitch;
    auto fetch_patch = [&](const Block& k, char* patch) {
        for (int i = wave; i * 64 < pieces; i += 8) {
            const int id = i * 64 + lane, row = id / kStemW8Pitch, px = id - row * kStemW8Pitch;
            const uint32_t off = id < pieces ? ((k.b * g.Hp + k.y0 * 2 + row) * g.Wp + k.x0 * 2 + px) * kChunk : kOob;
            dma16(rx, patch + i * 1024, off);
        }
    };
    // filters of the channel tile: 4 wave-instructions a K-step, as they lie
    for (int c = wave; c < g.ksteps * 4; c += 8)
        dma16(rw, wst + c * 1024, (static_cast<uint32_t>(tn) * g.ksteps * 4 + c) * 1024 + lane * kChunk);
    // this lane's channel of each 16-channel fragment: epilogue parameters, once per workgroup
    float al[4], bi[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int n = tn * 64 + f * 16 + r;
        al[f] = 1.f;
        bi[f] = 0.f;
        if (n < g.Cout) {
            al[f] = g.alpha[n];
            if (g.bias) bi[f] = g.bias[n];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the counted waits below count from here)

    uint32_t unit = blockIdx.x;
    const uint32_t units = static_cast<uint32_t>(g.units);
    Block cur = decode(unit);
    fetch_patch(cur, smem);
    const char* w_lane = wst + (q * 64 + r) * kChunk;
    // this lane's patch offset for fragment 0 and K-step 0: output row `wave`, output column r, pixel q of the step's four
    const int a_off = ((wave * 2) * kStemW8Pitch + r * 2 + q) * kChunk;
    const int HoWo = g.Ho * g.Wo;
    constexpr int kEsz = OUT == kOutBF16 ? 2 : 4;
    int buf = 0;
    int stores = 0;   // store instructions this wave issued behind the fetch of the patch about to be read
    for (; unit < units; unit += gridDim.x) {
        // Vector memory operations retire in issue order: once all but the stores issued after it are done, this wave's
        // pieces of the patch have landed, and the stores go on draining under the next block's arithmetic.
        wait_vmcnt_upto16(stores);
        __syncthreads();   // every wave's pieces have landed; every wave has left the buffer the next fetch overwrites
        {
            // store instructions certain to issue (their lane 0 stores): fragments that start inside the row x channel
            // groups that start inside Cout, if the wave's row is inside the image
            const int fx = min(4, max(0, (g.Wo - cur.x0 + 15) >> 4)), fc = min(4, max(0, (g.Cout - tn * 64 + 15) >> 4));
            stores = cur.y0 + wave < g.Ho ? fx * fc : 0;
            // (fp32: instructions of four channels x the row's 64 pixels, those whose first channel is inside Cout)
            if constexpr (OUT == kOutF32) stores = cur.y0 + wave < g.Ho ? min(16, max(0, (g.Cout - tn * 64 + 3) >> 2)) : 0;
        }
        const uint32_t nunit = unit + gridDim.x;
        const Block nxt = decode(nunit < units ? nunit : unit);
        if (nunit < units) fetch_patch(nxt, smem + (buf ^ 1) * kStemW8PatchBytes);

        v4f acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[i][f] = v4f{0.f, 0.f, 0.f, 0.f};
        const char* a_lane = smem + buf * kStemW8PatchBytes + a_off;
        for (int j = 0; j < g.ksteps; ++j) {
            frag_t af[4], wf[4];
            const char* a = a_lane + ((j >> 1) * kStemW8Pitch + (j & 1) * 4) * kChunk;
#pragma unroll
            for (int f = 0; f < 4; ++f) wf[f] = __builtin_bit_cast(frag_t, lds_read16(w_lane + j * 4096 + f * 256));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(frag_t, lds_read16(a + i * 32 * kChunk));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[i][f] = mfma_w8<F16>(af[i], wf[f], acc[i][f]);
        }

        // D[R][c] of fragment (i, f): R = 4*(lane>>4) + reg -> pixel i*16 + R of the wave's row, c = lane&15 -> channel f*16 + c
        const int y = cur.y0 + wave;
        if constexpr (OUT == kOutF32) {
            // fp32 outputs are streamed past the caches, and a nontemporal store that covers half a cache line per channel
            // (the accumulator as it lies: sixteen channels x 64 bytes) is written out as partial lines at 3.2 TB/s against
            // 5.3 TB/s for whole lines (tools/nchw_write_probe.hip).  So a wave turns its row, sixteen channels at a time,
            // through an area of its own inside the patch every wave has just finished with, and stores four channels x
            // the row's 64 pixels (256 bytes each) per instruction.
            constexpr int kTrPitch = 272;   // one channel's 64 pixels + 16 bytes: the sixteen lanes of a 16-byte write meet 64 banks
            static_assert(8 * 16 * kTrPitch <= kStemW8PatchBytes, "transposition areas inside one patch buffer");
            __syncthreads();
            if (y < g.Ho) {
                char* const tr = smem + buf * kStemW8PatchBytes + wave * (16 * kTrPitch);
                const int xs = cur.x0 + r * 4;   // this lane's four pixels when storing
                float* const orow = static_cast<float*>(g.out) + static_cast<size_t>(cur.b) * g.Cout * HoWo + static_cast<size_t>(y) * g.Wo + xs;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const v4f a = acc[i][f];
                        *reinterpret_cast<v4f*>(tr + r * kTrPitch + (i * 16 + q * 4) * 4) =
                            v4f{dequant1f(a[0], bi[f], al[f], g.beta), dequant1f(a[1], bi[f], al[f], g.beta),
                                dequant1f(a[2], bi[f], al[f], g.beta), dequant1f(a[3], bi[f], al[f], g.beta)};
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int n = tn * 64 + f * 16 + k * 4 + q;
                        const v4f v = *reinterpret_cast<const v4f*>(tr + (k * 4 + q) * kTrPitch + r * 16);
                        if (n < g.Cout && xs < g.Wo) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(orow + static_cast<size_t>(n) * HoWo));
                    }
                }
            }
        } else if (y < g.Ho) {
            const int xq = cur.x0 + q * 4;   // this lane's first pixel of fragment 0
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int n = tn * 64 + f * 16 + r;
                if (n >= g.Cout) continue;
                char* const row = static_cast<char*>(g.out) +
                                  ((static_cast<size_t>(cur.b) * g.Cout + n) * HoWo + static_cast<size_t>(y) * g.Wo + xq) * kEsz;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (xq + i * 16 >= g.Wo) continue;
                    const v4f a = acc[i][f];
                    const float y0 = dequant1f(a[0], bi[f], al[f], g.beta), y1 = dequant1f(a[1], bi[f], al[f], g.beta);
                    const float y2 = dequant1f(a[2], bi[f], al[f], g.beta), y3 = dequant1f(a[3], bi[f], al[f], g.beta);
                    if constexpr (OUT == kOutF32) {
                        __builtin_nontemporal_store(v4f{y0, y1, y2, y3}, reinterpret_cast<v4f*>(row + i * 16 * kEsz));
                    } else {
                        const uint2 h = pack4_bf16(y0, y1, y2, y3);
                        *reinterpret_cast<uint2*>(row + i * 16 * kEsz) = h;
                    }
                }
            }
        }
        cur = nxt;
        buf ^= 1;
    }
}

// Whether a layer takes the stem kernel, and its workspace: the pixel-major copy (rounded up to 256 bytes) + the filters.
struct StemW8Geom {
    bool ok;
    int64_t Hp, Wp, Ho, Wo, xp_bytes, wp_bytes;
};

StemW8Geom make_stem_w8_geom(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
                             int64_t pad) {
    StemW8Geom p{};
    const char* sw = dev_switch("MIQ_CONV_W8_STEM");   // developer A/B switch: 0 = tile kernel
    if (Cin > 4 || KW < 5 || KW > 8 || KH > 7 || stride != 2 || Cout <= 0 || (sw && sw[0] == '0')) return p;
    p.Hp = H + 2 * pad;
    p.Wp = W + 2 * pad;
    if (p.Hp < KH || p.Wp < KW) return p;
    p.Ho = (p.Hp - KH) / 2 + 1;
    p.Wo = (p.Wp - KW) / 2 + 1;
    if (p.Wo % 4 != 0 || B * p.Ho * p.Wo < 1024) return p;
    p.xp_bytes = B * p.Hp * p.Wp * kChunk;
    p.wp_bytes = ceil_div(Cout, 64) * 2 * KH * 4 * 64 * kChunk;
    if (p.xp_bytes >= (1ll << 31) || B * ceil_div(p.Ho, 8) * ceil_div(p.Wo, 64) * ceil_div(Cout, 64) >= (1ll << 31)) return p;
    p.ok = true;
    return p;
}

int run_stem_w8(const StemW8Geom& P, const void* x, int x_dtype, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                int64_t Cout, int64_t KH, int64_t KW, int64_t pad, const float* bias, const float* alpha, float beta, void* out,
                int out_dtype, void* workspace, hipStream_t st, const void* wp_packed = nullptr) {
    uint4* xp = static_cast<uint4*>(workspace);
    uint4* wp = wp_packed ? static_cast<uint4*>(const_cast<void*>(wp_packed))
                          : reinterpret_cast<uint4*>(static_cast<char*>(workspace) + round_up(P.xp_bytes, 256));
    const uint32_t npix = static_cast<uint32_t>(P.xp_bytes / kChunk), wchunks = static_cast<uint32_t>(P.wp_bytes / kChunk);
    const int n_prep = static_cast<int>(std::min<int64_t>(ceil_div(npix, 1024), 1 << 20));   // four pixels a thread
    const int n_pack = wp_packed ? 0 : static_cast<int>(std::min<int64_t>(ceil_div(wchunks, 256), 4096));
#define MIQ_STEM_PREP(T, NP)                                                                                               \
    stem_w8_prep_kernel<T, NP><<<n_prep + n_pack, 256, 0, st>>>(static_cast<const T*>(x), xp, (int)Cin, (int)H, (int)W, (int)P.Hp, \
                                                                (int)P.Wp, (int)pad, npix, n_prep, w, wp, wchunks, (int)Cout, \
                                                                (int)KH, (int)KW)
    if (x_dtype == MIQ_F32)
        MIQ_STEM_PREP(float, 2);
    else if (x_dtype == MIQ_F16)
        MIQ_STEM_PREP(_Float16, 1);
    else
        MIQ_STEM_PREP(__bf16, 1);
#undef MIQ_STEM_PREP
    if (int rc = check_launch("stem_w8_prep_kernel")) return rc;

    StemW8Args a{};
    a.xp = xp; a.wp = wp; a.bias = bias; a.alpha = alpha; a.out = out; a.beta = beta;
    a.Cout = (int)Cout; a.KH = (int)KH; a.ksteps = (int)(2 * KH);
    a.Hp = (int)P.Hp; a.Wp = (int)P.Wp; a.Ho = (int)P.Ho; a.Wo = (int)P.Wo;
    a.tiles_x = (int)ceil_div(P.Wo, 64); a.tiles_y = (int)ceil_div(P.Ho, 8); a.tiles_n = (int)ceil_div(Cout, 64);
    a.units = (int)(B * a.tiles_y * a.tiles_x * a.tiles_n);
    a.xp_bytes = (uint32_t)P.xp_bytes; a.wp_bytes = (uint32_t)P.wp_bytes;
    constexpr int kLds = 2 * kStemW8PatchBytes + kStemW8WeightBytes;
    static_assert(kLds <= 160 * 1024, "LDS of one CU");
    void (*kern)(StemW8Args) = out_dtype == MIQ_F32 ? conv_stem_w8_kernel<kOutF32> : conv_stem_w8_kernel<kOutBF16>;
    if (x_dtype == MIQ_F16) kern = out_dtype == MIQ_F32 ? conv_stem_w8_kernel<kOutF32, true> : conv_stem_w8_kernel<kOutBF16, true>;
    if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
    int grid = a.units < 256 ? a.units : 256 - 256 % a.tiles_n;   // one persistent workgroup per CU, each keeping one channel tile
    if (grid < a.tiles_n) grid = a.units < a.tiles_n ? a.units : a.tiles_n;
    kern<<<grid, 512, kLds, st>>>(a);
    return check_launch("conv_stem_w8_kernel");
}

}  // namespace

extern "C" size_t miq_gemm_w8_workspace(int x_dtype, int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 16 != 0) return 0;
    // what miq_gemm_w8 accepts; anything else gets no workspace here and MIQ_ERANGE from the call (and no arithmetic is
    // done on sizes that may be any int64: a query must answer, not trap)
    if (!miq::all_below_2_31({M, N, K}) || M * K >= (1ll << 29) || N * K >= (1ll << 31)) return 0;
    return 2 * plane_bytes(x_dtype, M, K) + w8f_slab_bytes(M, N, K, x_dtype == MIQ_F32 ? 2 : 1);
}

extern "C" int miq_gemm_w8(const void* x, int x_dtype, const int8_t* W, int64_t M, int64_t N, int64_t K,
                           const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                           void* workspace, size_t workspace_bytes, void* stream) {
    MIQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, MIQ_EINVAL, "miq_gemm_w8: negative size (M=%lld N=%lld K=%lld)",
                (long long)M, (long long)N, (long long)K);
    MIQ_REQUIRE(x_dtype == MIQ_F32 || x_dtype == MIQ_F16 || x_dtype == MIQ_BF16, MIQ_EINVAL,
                "miq_gemm_w8: x_dtype must be MIQ_F32, MIQ_F16 or MIQ_BF16 (got %d)", x_dtype);
    MIQ_REQUIRE(out_dtype == MIQ_F32 || out_dtype == MIQ_BF16, MIQ_EINVAL,
                "miq_gemm_w8: out_dtype must be MIQ_F32 or MIQ_BF16 (got %d)", out_dtype);
    if (M == 0 || N == 0) return MIQ_OK;
    MIQ_REQUIRE(out && alpha, MIQ_EINVAL, "miq_gemm_w8: null output or alpha");
    MIQ_REQUIRE(K == 0 || (x && W), MIQ_EINVAL, "miq_gemm_w8: null operand");
    MIQ_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) && M * K < (1ll << 29) && N * K < (1ll << 31),
                MIQ_ERANGE, "miq_gemm_w8: operands are indexed with 31-bit byte offsets (M*K < 2^29, N*K < 2^31)");
    const size_t xsz = x_dtype == MIQ_F32 ? 4 : 2, esz = out_dtype == MIQ_BF16 ? 2 : 4;
    MIQ_REQUIRE(miq::aligned(x, xsz) && miq::aligned(out, esz) && miq::aligned(alpha, 4) && miq::aligned(bias, 4), MIQ_EALIGN,
                "miq_gemm_w8: x/out/alpha/bias not aligned to their element size");
    hipStream_t st = miq::as_stream(stream);

    W8fArgs g{x, W, bias, alpha, out, beta, static_cast<int>(M), static_cast<int>(N), static_cast<int>(K), 0u, 0, 0, 0};
    g.ldc = static_cast<int>(N);
    // The epilogues store 16-byte vectors (four fp32, or eight bf16 when N % 8 == 0) wherever N % 4 == 0.
    const bool out_vec_ok = (N % 4 != 0) || miq::aligned(out, 16);
    const bool mfma_ok = K > 0 && K % 16 == 0 && miq::aligned(x, 16) && miq::aligned(W, 16) && out_vec_ok;
    if (!mfma_ok) {
        if (out_dtype == MIQ_F32) {
            if (x_dtype == MIQ_F32) return launch_any<float, kOutF32>(x, g, st);
            if (x_dtype == MIQ_F16) return launch_any<_Float16, kOutF32>(x, g, st);
            return launch_any<__bf16, kOutF32>(x, g, st);
        }
        if (x_dtype == MIQ_F32) return launch_any<float, kOutBF16>(x, g, st);
        if (x_dtype == MIQ_F16) return launch_any<_Float16, kOutBF16>(x, g, st);
        return launch_any<__bf16, kOutBF16>(x, g, st);
    }

    const size_t need = miq_gemm_w8_workspace(x_dtype, M, N, K);
    if (need) {
        MIQ_REQUIRE(workspace && workspace_bytes >= need, MIQ_EWORKSPACE,
                    "miq_gemm_w8: this call needs %zu bytes of workspace (got %zu): see miq_gemm_w8_workspace", need,
                    workspace_bytes);
        MIQ_REQUIRE(miq::aligned(workspace, 256), MIQ_EALIGN, "miq_gemm_w8: workspace must be 256-byte aligned");
    }
    const size_t pb = plane_bytes(x_dtype, M, K);
    const int planes = x_dtype == MIQ_F32 ? 2 : 1;   // fp16 activations go to the fp16 MFMA as they are, bf16 ones to the bf16 MFMA
    float* slabs = w8f_slab_bytes(M, N, K, planes) ? reinterpret_cast<float*>(static_cast<char*>(workspace) + 2 * pb) : nullptr;
    const W8fPlan plan = plan_w8f(M, N, K, planes);
    if (x_dtype == MIQ_F32) {
        uint16_t* hi = static_cast<uint16_t*>(workspace);
        uint16_t* lo = reinterpret_cast<uint16_t*>(static_cast<char*>(workspace) + pb);
        const int64_t groups = M * K / 8;
        const int grid = static_cast<int>(groups / 256 + 1 < 8192 ? groups / 256 + 1 : 8192);
        split_planes_kernel<float><<<grid, 256, 0, st>>>(static_cast<const float*>(x), hi, lo, groups);
        if (int rc = check_launch("split_planes_kernel")) return rc;
        g.X = hi;
        g.plane_stride = static_cast<uint32_t>(pb);
    }
    auto run = [&](const W8fArgs& part, const W8fPlan& pl) {
        if (planes == 2)
            return out_dtype == MIQ_F32 ? launch_w8f<2, kOutF32>(part, pl, slabs, st) : launch_w8f<2, kOutBF16>(part, pl, slabs, st);
        if (x_dtype == MIQ_F16)
            return out_dtype == MIQ_F32 ? launch_w8f<1, kOutF32, true>(part, pl, slabs, st) : launch_w8f<1, kOutBF16, true>(part, pl, slabs, st);
        return out_dtype == MIQ_F32 ? launch_w8f<1, kOutF32>(part, pl, slabs, st) : launch_w8f<1, kOutBF16>(part, pl, slabs, st);
    };
    const int64_t m_main = w8f_rows_in_whole_rounds(M, N, K, planes);
    if (m_main < M) {   // whole rounds of 256 x 256 tiles, then the remaining rows as their own problem
        W8fArgs head = g, tail = g;
        head.M = static_cast<int>(m_main);
        if (int rc = run(head, plan_w8f(m_main, N, K, planes))) return rc;
        tail.X = static_cast<const char*>(g.X) + m_main * K * 2;
        tail.out = static_cast<char*>(out) + static_cast<size_t>(m_main) * N * esz;
        tail.M = static_cast<int>(M - m_main);
        return run(tail, plan_w8f(M - m_main, N, K, planes));
    }
    return run(g, plan);
}

// The padded channels-last bf16 planes of x (border `pad` all round) and the packed weight wp[Cout][rows][Lp] ((kw, c)
// order inside a run of kw_run taps), in one launch where the weight re-layout can ride along.  G.H / G.W / G.sub describe
// the (possibly subsampled) image; rows x kw_run = the filter's taps.
static int conv_w8_relayout(const void* x, int x_dtype, const int8_t* w, uint16_t* xp, int8_t* wp, const ConvW8Geom& G, int64_t B,
                     int64_t Cin, int64_t Cout, int64_t rows, int64_t kw_run, int64_t pad, hipStream_t st, bool skip_weight = false) {
    const int64_t H = G.H, W = G.W;
    const size_t xsz = x_dtype == MIQ_F32 ? 4 : 2;
    const uint32_t npix = static_cast<uint32_t>(B * G.Hp * G.Wp);
    const int cblocks = static_cast<int>(ceil_div(G.Cp, 64));
    const int n_interior = static_cast<int>(ceil_div(H * W, 64) * cblocks);
    const int n_border = static_cast<int>(ceil_div((2 * pad * G.Wp + 2 * pad * H) * (G.Cp / 8), 256));
    const uint32_t wtotal = static_cast<uint32_t>(Cout * G.K);
    // about four weight bytes per thread, spread over the images
    const bool pack_along = G.Cp > 16 && !skip_weight;
    ConvW8PackArgs pk{w, wp, wtotal, 0, (int)rows, (int)kw_run, (int)G.L, (int)G.Lp};
    if (pack_along) pk.n_wpack = static_cast<int>(std::min<int64_t>(std::max<int64_t>(ceil_div(ceil_div(wtotal, 1024), B), 1), 4096));
    const int64_t pblocks = B * (static_cast<int64_t>(n_interior) + n_border + pk.n_wpack);
    MIQ_REQUIRE(pblocks < (1ll << 31), MIQ_ERANGE, "miq_conv2d_w8: too many re-layout blocks");
    const int pgrid = static_cast<int>(pblocks);
    const uint32_t sitems = npix * static_cast<uint32_t>(G.Cp / 8);
    const int sgrid = static_cast<int>(sitems / 256 + 1 < 16384 ? sitems / 256 + 1 : 16384);
    const uint32_t plane_elems = static_cast<uint32_t>(G.plane_bytes / 2);
    // 16- or 8-byte source vectors: whole vectors per channel plane, every plane aligned to them, no subsampling
    const int64_t plane_b = H * W * static_cast<int64_t>(xsz);
    const bool vec16 = G.sub == 1 && plane_b % 16 == 0 && miq::aligned(x, 16);
    const char* v8 = dev_switch("MIQ_CONV_W8_VEC8");   // developer A/B switch: 0 = single elements instead of 8-byte vectors
    const bool vec8 = G.sub == 1 && plane_b % 8 == 0 && miq::aligned(x, 8) && !(v8 && v8[0] == '0');
#define MIQ_PREP(T, NP)                                                                                              \
    do {                                                                                                             \
        if (G.Cp <= 16)                                                                                              \
            conv_w8_prep_x_small_kernel<T, NP><<<sgrid, 256, 0, st>>>(static_cast<const T*>(x), xp, plane_elems, (int)Cin, (int)H, \
                                                                      (int)W, (int)G.Hp, (int)G.Wp, (int)G.Cp, (int)pad, npix, sitems, \
                                                                      (int)G.sub, (int)G.Hs, (int)G.Ws);     \
        else if (vec16)                                                                                              \
            conv_w8_prep_x_kernel<T, NP, 16><<<pgrid, 256, 0, st>>>(static_cast<const T*>(x), xp, plane_elems, (int)Cin, (int)H, \
                                                                    (int)W, (int)G.Hp, (int)G.Wp, (int)G.Cp, (int)pad, cblocks, \
                                                                    n_interior, n_border, 1, (int)G.Hs, (int)G.Ws, pk); \
        else if (vec8)                                                                                               \
            conv_w8_prep_x_kernel<T, NP, 8><<<pgrid, 256, 0, st>>>(static_cast<const T*>(x), xp, plane_elems, (int)Cin, (int)H, \
                                                                   (int)W, (int)G.Hp, (int)G.Wp, (int)G.Cp, (int)pad, cblocks, \
                                                                   n_interior, n_border, 1, (int)G.Hs, (int)G.Ws, pk); \
        else                                                                                                         \
            conv_w8_prep_x_kernel<T, NP, 0><<<pgrid, 256, 0, st>>>(static_cast<const T*>(x), xp, plane_elems, (int)Cin, (int)H, \
                                                                       (int)W, (int)G.Hp, (int)G.Wp, (int)G.Cp, (int)pad, cblocks, \
                                                                       n_interior, n_border, (int)G.sub, (int)G.Hs, (int)G.Ws, pk); \
    } while (0)
    if (x_dtype == MIQ_F32)
        MIQ_PREP(float, 2);
    else if (x_dtype == MIQ_F16)
        MIQ_PREP(_Float16, 1);
    else
        MIQ_PREP(__bf16, 1);
#undef MIQ_PREP
    if (int rc = check_launch("conv_w8_prep_x_kernel")) return rc;
    if (!pack_along && !skip_weight) {
        conv_w8_pack_w_kernel<<<static_cast<int>(wtotal / 256 + 1 < 4096 ? wtotal / 256 + 1 : 4096), 256, 0, st>>>(
            w, wp, wtotal, (int)Cin, (int)rows, (int)kw_run, (int)G.Cp, (int)G.L, (int)G.Lp);
        if (int rc = check_launch("conv_w8_pack_w_kernel")) return rc;
    }
    return MIQ_OK;
}

extern "C" size_t miq_conv2d_w8_workspace(int x_dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                                          int64_t KW, int64_t stride, int64_t pad) {
    ConvW8Geom G;
    if (!make_conv_w8_geom(B, Cin, H, W, Cout, KH, KW, stride, pad, G) || G.M == 0 || Cout == 0) return 0;
    if (!conv_w8_in_range(G, conv_w8_planes(x_dtype))) return 0;   // (the call answers MIQ_ERANGE)
    const PatchW8Geom P = make_patch_w8_geom(conv_w8_planes(x_dtype), B, Cin, H, W, Cout, KH, KW, stride, pad);
    if (P.ok) return static_cast<size_t>(round_up(P.xp_bytes, 256) + round_up(P.wp_bytes, 256));
    const StemW8Geom S = make_stem_w8_geom(B, Cin, H, W, Cout, KH, KW, stride, pad);
    if (S.ok) return static_cast<size_t>(round_up(S.xp_bytes, 256) + round_up(S.wp_bytes, 256));
    return static_cast<size_t>(conv_w8_planes(x_dtype) * G.plane_bytes + G.wp_bytes);
}

extern "C" size_t miq_conv2d_w8_prepacked_workspace(int x_dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                                    int64_t KH, int64_t KW, int64_t stride, int64_t pad) {
    ConvW8Geom G;
    if (!make_conv_w8_geom(B, Cin, H, W, Cout, KH, KW, stride, pad, G) || G.M == 0 || Cout == 0) return 0;
    if (!conv_w8_in_range(G, conv_w8_planes(x_dtype))) return 0;   // (the call answers MIQ_ERANGE)
    const PatchW8Geom P = make_patch_w8_geom(conv_w8_planes(x_dtype), B, Cin, H, W, Cout, KH, KW, stride, pad);
    if (P.ok) return static_cast<size_t>(round_up(P.xp_bytes, 256));
    const StemW8Geom S = make_stem_w8_geom(B, Cin, H, W, Cout, KH, KW, stride, pad);
    if (S.ok) return static_cast<size_t>(round_up(S.xp_bytes, 256));
    return static_cast<size_t>(conv_w8_planes(x_dtype) * G.plane_bytes);
}

// The packed weight of miq_conv2d_w8_pack_weight for activations of type x_dtype: the forms the three kernels stage from,
// each at a multiple of 256 bytes.  It depends on the filter's shape and the activation type only.
struct W8Packed {
    int64_t rows_bytes;               // wp[Cout][KH][Lp]: the im2col-order kernel
    int64_t patch_at, patch_bytes;    // conv3x3_patch_w8_kernel's widened filters (3 x 3 over whole 32-channel groups); 0 = none
    int64_t stem_at, stem_bytes;      // conv_stem_w8_kernel's (at most 4 input channels, 5-8 columns, at most 7 rows); 0 = none
    int64_t total;
};

static W8Packed w8_packed_layout(int x_dtype, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW) {
    W8Packed p{};
    // (sizes as make_conv_w8_geom bounds them; a rows form beyond 2 GiB is held at 2^60, which every caller refuses, and the
    // other forms, a small multiple of Cout * Cin, are computed only below that)
    if (!all_below_2_31({Cin, Cout}) || KH < 0 || KW < 0 || KH >= (1 << 15) || KW >= (1 << 15)) {
        p.rows_bytes = p.total = 1ll << 60;
        return p;
    }
    const int64_t Cp = round_up(Cin, 8), Lp = round_up(KW * Cp, 16);
    p.rows_bytes = round_up(sat_product({Cout, KH, Lp}), 256);
    p.total = p.rows_bytes;
    if (p.rows_bytes >= (1ll << 31)) return p;
    const int planes = conv_w8_planes(x_dtype);
    if (KH == 3 && KW == 3 && Cin % 32 == 0) {
        p.patch_at = p.total;
        p.patch_bytes = ceil_div(Cout, 64) * (Cin / (planes == 1 ? 32 : 16)) * (9 * (planes == 1 ? 4 : 2) * 64 * kChunk);
        p.total += round_up(p.patch_bytes, 256);
    }
    if (Cin <= 4 && KW >= 5 && KW <= 8 && KH <= 7) {
        p.stem_at = p.total;
        p.stem_bytes = ceil_div(Cout, 64) * 2 * KH * 4 * 64 * kChunk;
        p.total += round_up(p.stem_bytes, 256);
    }
    return p;
}

extern "C" size_t miq_conv2d_w8_packed_weight_bytes(int x_dtype, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW) {
    if (Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return 0;
    const int64_t total = w8_packed_layout(x_dtype, Cin, Cout, KH, KW).total;
    return total < (1ll << 31) ? static_cast<size_t>(total) : 0;   // (miq_conv2d_w8_pack_weight answers MIQ_ERANGE)
}

extern "C" int miq_conv2d_w8_pack_weight(const int8_t* w, int x_dtype, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, void* wp,
                                         void* stream) {
    MIQ_REQUIRE(Cin > 0 && Cout >= 0 && KH > 0 && KW > 0, MIQ_EINVAL, "miq_conv2d_w8_pack_weight: bad shape");
    MIQ_REQUIRE(x_dtype == MIQ_F32 || x_dtype == MIQ_F16 || x_dtype == MIQ_BF16, MIQ_EINVAL,
                "miq_conv2d_w8_pack_weight: x_dtype must be MIQ_F32, MIQ_F16 or MIQ_BF16 (got %d)", x_dtype);
    if (Cout == 0) return MIQ_OK;
    MIQ_REQUIRE(w && wp, MIQ_EINVAL, "miq_conv2d_w8_pack_weight: null pointer");
    MIQ_REQUIRE(miq::aligned(wp, 256), MIQ_EALIGN, "miq_conv2d_w8_pack_weight: wp must be 256-byte aligned");
    const W8Packed L = w8_packed_layout(x_dtype, Cin, Cout, KH, KW);
    MIQ_REQUIRE(L.total < (1ll << 31), MIQ_ERANGE, "miq_conv2d_w8_pack_weight: packed weight beyond 2 GiB");
    hipStream_t st = miq::as_stream(stream);
    char* base = static_cast<char*>(wp);
    {
        const int64_t Cp = round_up(Cin, 8), Lrun = KW * Cp, Lp = round_up(Lrun, 16);
        const uint32_t wtotal = static_cast<uint32_t>(Cout * KH * Lp);
        conv_w8_pack_w_kernel<<<static_cast<int>(wtotal / 256 + 1 < 4096 ? wtotal / 256 + 1 : 4096), 256, 0, st>>>(
            w, reinterpret_cast<int8_t*>(base), wtotal, (int)Cin, (int)KH, (int)KW, (int)Cp, (int)Lrun, (int)Lp);
        if (int rc = check_launch("conv_w8_pack_w_kernel")) return rc;
    }
    if (L.patch_at) {
        const uint32_t wtotal = static_cast<uint32_t>(L.patch_bytes / kChunk);
        const int wgrid = static_cast<int>(wtotal / 256 + 1 < 4096 ? wtotal / 256 + 1 : 4096);
        uint4* dst = reinterpret_cast<uint4*>(base + L.patch_at);
        // (the activation half of the kernel gets no workgroups: n_prep = 0)
        if (x_dtype == MIQ_F32)
            patch_w8_prep_kernel<float, 2><<<wgrid, 256, 0, st>>>(nullptr, nullptr, (int)Cin, 0, 0, 0, 1, 0, 1u, 0u, 0, w, dst, wtotal, (int)Cout);
        else if (x_dtype == MIQ_F16)
            patch_w8_prep_kernel<_Float16, 1><<<wgrid, 256, 0, st>>>(nullptr, nullptr, (int)Cin, 0, 0, 0, 1, 0, 1u, 0u, 0, w, dst, wtotal, (int)Cout);
        else
            patch_w8_prep_kernel<__bf16, 1><<<wgrid, 256, 0, st>>>(nullptr, nullptr, (int)Cin, 0, 0, 0, 1, 0, 1u, 0u, 0, w, dst, wtotal, (int)Cout);
        if (int rc = check_launch("patch_w8_prep_kernel (filters)")) return rc;
    }
    if (L.stem_at) {
        const uint32_t wchunks = static_cast<uint32_t>(L.stem_bytes / kChunk);
        const int n_pack = static_cast<int>(std::min<int64_t>(ceil_div(wchunks, 256), 4096));
        uint4* dst = reinterpret_cast<uint4*>(base + L.stem_at);
        if (x_dtype == MIQ_F32)
            stem_w8_prep_kernel<float, 2><<<n_pack, 256, 0, st>>>(nullptr, nullptr, (int)Cin, 0, 0, 1, 1, 0, 0u, 0, w, dst, wchunks, (int)Cout, (int)KH, (int)KW);
        else if (x_dtype == MIQ_F16)
            stem_w8_prep_kernel<_Float16, 1><<<n_pack, 256, 0, st>>>(nullptr, nullptr, (int)Cin, 0, 0, 1, 1, 0, 0u, 0, w, dst, wchunks, (int)Cout, (int)KH, (int)KW);
        else
            stem_w8_prep_kernel<__bf16, 1><<<n_pack, 256, 0, st>>>(nullptr, nullptr, (int)Cin, 0, 0, 1, 1, 0, 0u, 0, w, dst, wchunks, (int)Cout, (int)KH, (int)KW);
        if (int rc = check_launch("stem_w8_prep_kernel (filters)")) return rc;
    }
    return MIQ_OK;
}

// `w` is the OIHW filter (prepacked = false: re-laid by this call into the workspace, behind the activation planes) or the
// packed weight of miq_conv2d_w8_pack_weight for this x_dtype (prepacked = true: the workspace holds the planes only).
static int conv2d_w8_run(bool prepacked, const void* x, int x_dtype, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                         int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias,
                         const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                         size_t workspace_bytes, void* stream) {
    ConvW8Geom G;
    MIQ_REQUIRE(make_conv_w8_geom(B, Cin, H, W, Cout, KH, KW, stride, pad, G), MIQ_EINVAL,
                "miq_conv2d_w8: bad geometry (B=%lld Cin=%lld H=%lld W=%lld Cout=%lld KH=%lld KW=%lld stride=%lld pad=%lld)",
                (long long)B, (long long)Cin, (long long)H, (long long)W, (long long)Cout, (long long)KH, (long long)KW,
                (long long)stride, (long long)pad);
    MIQ_REQUIRE(x_dtype == MIQ_F32 || x_dtype == MIQ_F16 || x_dtype == MIQ_BF16, MIQ_EINVAL,
                "miq_conv2d_w8: x_dtype must be MIQ_F32, MIQ_F16 or MIQ_BF16 (got %d)", x_dtype);
    MIQ_REQUIRE(out_dtype == MIQ_F32 || out_dtype == MIQ_BF16, MIQ_EINVAL,
                "miq_conv2d_w8: out_dtype must be MIQ_F32 or MIQ_BF16 (got %d)", out_dtype);
    if (G.M == 0 || Cout == 0) return MIQ_OK;
    MIQ_REQUIRE(x && w && out && alpha, MIQ_EINVAL, "miq_conv2d_w8: null pointer");
    const int planes = conv_w8_planes(x_dtype);
    MIQ_REQUIRE(conv_w8_in_range(G, planes), MIQ_ERANGE,
                "miq_conv2d_w8: padded activation planes (%lld B) and packed weights (%lld B) must stay below 2 GiB",
                (long long)sat_product({planes, G.plane_bytes}), (long long)G.wp_bytes);
    const size_t xsz = x_dtype == MIQ_F32 ? 4 : 2, esz = out_dtype == MIQ_BF16 ? 2 : 4;
    MIQ_REQUIRE(miq::aligned(x, xsz) && miq::aligned(out, esz) && miq::aligned(alpha, 4) && miq::aligned(bias, 4), MIQ_EALIGN,
                "miq_conv2d_w8: x/out/alpha/bias not aligned to their element size");
    const size_t need = prepacked ? miq_conv2d_w8_prepacked_workspace(x_dtype, B, Cin, H, W, Cout, KH, KW, stride, pad)
                                  : miq_conv2d_w8_workspace(x_dtype, B, Cin, H, W, Cout, KH, KW, stride, pad);
    MIQ_REQUIRE(workspace && workspace_bytes >= need, MIQ_EWORKSPACE, "miq_conv2d_w8: needs %zu bytes of workspace (got %zu)",
                need, workspace_bytes);
    MIQ_REQUIRE(miq::aligned(workspace, 256), MIQ_EALIGN, "miq_conv2d_w8: workspace must be 256-byte aligned");
    hipStream_t st = miq::as_stream(stream);
    const W8Packed PK = w8_packed_layout(x_dtype, Cin, Cout, KH, KW);
    const char* packed = reinterpret_cast<const char*>(w);
    if (prepacked) MIQ_REQUIRE(miq::aligned(w, 256), MIQ_EALIGN, "miq_conv2d_w8_prepacked: the packed weight must be 256-byte aligned");
    // 3 x 3 / stride 1 over whole 32-channel groups: the kernel that keeps a tile's pixel patch in LDS (16-byte stores
    // of four fp32 pixels / 8-byte stores of four bf16 pixels need the output aligned to them)
    {
        const PatchW8Geom P = make_patch_w8_geom(planes, B, Cin, H, W, Cout, KH, KW, stride, pad);
        if (P.ok) {
            MIQ_REQUIRE((P.Ho * P.Wo) % 4 != 0 || miq::aligned(out, 4 * esz), MIQ_EALIGN,
                        "miq_conv2d_w8: out must be aligned to four of its elements");
            return run_patch_w8(P, x, x_dtype, w, B, Cin, H, W, Cout, pad, bias, alpha, beta, out, out_dtype, workspace, st,
                                prepacked ? packed + PK.patch_at : nullptr);
        }
    }
    // stems (at most four input channels, 5-8 filter columns, stride 2) whose output rows are whole 16-byte vectors
    {
        const StemW8Geom S = make_stem_w8_geom(B, Cin, H, W, Cout, KH, KW, stride, pad);
        if (S.ok) {
            MIQ_REQUIRE(miq::aligned(out, 4 * esz), MIQ_EALIGN, "miq_conv2d_w8: out must be aligned to four of its elements");
            return run_stem_w8(S, x, x_dtype, w, B, Cin, H, W, Cout, KH, KW, pad, bias, alpha, beta, out, out_dtype, workspace, st,
                               prepacked ? packed + PK.stem_at : nullptr);
        }
    }
    H = G.H; W = G.W; stride = G.stride;   // a subsampled 1 x 1 layer continues as its stride-1 equivalent

    uint16_t* xp = static_cast<uint16_t*>(workspace);
    int8_t* wp = prepacked ? reinterpret_cast<int8_t*>(const_cast<char*>(packed)) : static_cast<int8_t*>(workspace) + planes * G.plane_bytes;
    if (int rc = conv_w8_relayout(x, x_dtype, w, xp, wp, G, B, Cin, Cout, KH, KW, pad, st, prepacked)) return rc;

    W8fArgs g{xp, wp, bias, alpha, out, beta, static_cast<int>(G.M), static_cast<int>(Cout), static_cast<int>(G.K),
              static_cast<uint32_t>(G.plane_bytes), 0, 0, 0,
              static_cast<int>(G.Ho * G.Wo), static_cast<int>(G.Wo), static_cast<int>(stride), static_cast<int>(G.Lp),
              static_cast<int>(G.L), static_cast<uint32_t>(G.Wp * G.Cp * 2), static_cast<uint32_t>(G.Cp * 2),
              static_cast<uint32_t>(G.Hp * G.Wp * G.Cp * 2),
              static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(G.Lp)) + 1u};
    if (planes == 2) return out_dtype == MIQ_F32 ? launch_conv_w8f_np<2, kOutF32>(g, st) : launch_conv_w8f_np<2, kOutBF16>(g, st);
    if (x_dtype == MIQ_F16)
        return out_dtype == MIQ_F32 ? launch_conv_w8f_np<1, kOutF32, true>(g, st) : launch_conv_w8f_np<1, kOutBF16, true>(g, st);
    return out_dtype == MIQ_F32 ? launch_conv_w8f_np<1, kOutF32>(g, st) : launch_conv_w8f_np<1, kOutBF16>(g, st);
}

extern "C" int miq_conv2d_w8(const void* x, int x_dtype, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                             int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias,
                             const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                             size_t workspace_bytes, void* stream) {
    return conv2d_w8_run(false, x, x_dtype, w, B, Cin, H, W, Cout, KH, KW, stride, pad, bias, alpha, beta, out, out_dtype,
                         workspace, workspace_bytes, stream);
}

extern "C" int miq_conv2d_w8_prepacked(const void* x, int x_dtype, const void* wp, int64_t B, int64_t Cin, int64_t H, int64_t W,
                                       int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias,
                                       const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return conv2d_w8_run(true, x, x_dtype, static_cast<const int8_t*>(wp), B, Cin, H, W, Cout, KH, KW, stride, pad, bias, alpha,
                         beta, out, out_dtype, workspace, workspace_bytes, stream);
}

// ---- dense layers with a stride / padding per axis or a dilation ----------------------------------------
// The implicit-GEMM kernel addresses a window through byte strides, so these layers ride it: the padded copy gets one
// border width P = the largest of the four paddings and the windows start (P - pad_top) rows and (P - pad_left) pixels
// into it (W8fArgs::origin); the strides of the two axes and a dilation along y are strides of that addressing; under a
// dilation along x every tap becomes a run of its own (W8fArgs::kwt), Cp elements of one pixel.
namespace {

bool make_conv_w8_dense_geom(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW, int64_t sh,
                             int64_t sw, int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh, int64_t dw, ConvW8Geom& g,
                             int64_t& P, int64_t& rows, int64_t& kw_run) {
    if (B < 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout < 0 || KH <= 0 || KW <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 ||
        pt < 0 || pb < 0 || pl < 0 || pr < 0)
        return false;
    // (as make_conv_w8_geom: sizes below 2^31, filter / stride / dilation / padding below 2^15, long products held at 2^60)
    if (!all_below_2_31({B, Cin, H, W, Cout})) return false;
    for (int64_t v : {KH, KW, sh, sw, dh, dw, pt, pb, pl, pr})
        if (v >= (1 << 15)) return false;
    const int64_t eh = dh * (KH - 1) + 1, ew = dw * (KW - 1) + 1;
    if (H + pt + pb < eh || W + pl + pr < ew) return false;
    P = std::max(std::max(pt, pb), std::max(pl, pr));
    rows = dw == 1 ? KH : KH * KW;
    kw_run = dw == 1 ? KW : 1;
    g = ConvW8Geom{B, Cin, H, W, Cout, KH, KW, sh, P, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, H, W};
    g.Hp = H + 2 * P;
    g.Wp = W + 2 * P;
    g.Ho = (H + pt + pb - eh) / sh + 1;
    g.Wo = (W + pl + pr - ew) / sw + 1;
    g.Cp = round_up(Cin, 8);
    g.L = kw_run * g.Cp;
    g.Lp = round_up(g.L, 16);
    g.K = sat_product({rows, g.Lp});
    g.M = sat_product({B, g.Ho, g.Wo});
    g.plane_bytes = round_up(sat_product({B, g.Hp, g.Wp, g.Cp, 2}), 256);
    g.wp_bytes = round_up(sat_product({Cout, g.K}), 256);
    return true;
}

}  // namespace

namespace miq {

size_t conv2d_w8_dense_workspace(int x_dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW,
                                 int64_t sh, int64_t sw, int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh, int64_t dw) {
    ConvW8Geom G;
    int64_t P, rows, kw_run;
    if (!make_conv_w8_dense_geom(B, Cin, H, W, Cout, KH, KW, sh, sw, pt, pb, pl, pr, dh, dw, G, P, rows, kw_run) || G.M == 0 ||
        Cout == 0)
        return 0;
    // (0 as well for what the kernel's 32-bit offsets do not reach: the caller then takes the direct convolution)
    const int planes = conv_w8_planes(x_dtype);
    if (!conv_w8_in_range(G, planes) || sat_product({B, ceil_div(H * W, 64) * ceil_div(G.Cp, 64) + 8192}) >= (1ll << 31)) return 0;
    return static_cast<size_t>(planes * G.plane_bytes + G.wp_bytes);
}

// The caller (miq_conv2d_general) has checked pointers, dtypes and the alignment of x / out / bias / alpha.
int conv2d_w8_dense(const void* x, int x_dtype, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                    int64_t KH, int64_t KW, int64_t sh, int64_t sw, int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh,
                    int64_t dw, const float* bias, const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                    size_t workspace_bytes, hipStream_t st) {
    ConvW8Geom G;
    int64_t P, rows, kw_run;
    MIQ_REQUIRE(make_conv_w8_dense_geom(B, Cin, H, W, Cout, KH, KW, sh, sw, pt, pb, pl, pr, dh, dw, G, P, rows, kw_run), MIQ_EINVAL,
                "miq_conv2d_general: bad geometry");
    if (G.M == 0 || Cout == 0) return MIQ_OK;
    const int planes = conv_w8_planes(x_dtype);
    MIQ_REQUIRE(conv_w8_in_range(G, planes), MIQ_ERANGE,
                "miq_conv2d_general: padded activation planes (%lld B) and packed weights (%lld B) must stay below 2 GiB",
                (long long)sat_product({planes, G.plane_bytes}), (long long)G.wp_bytes);
    const size_t need = conv2d_w8_dense_workspace(x_dtype, B, Cin, H, W, Cout, KH, KW, sh, sw, pt, pb, pl, pr, dh, dw);
    MIQ_REQUIRE(workspace && workspace_bytes >= need, MIQ_EWORKSPACE, "miq_conv2d_general: needs %zu bytes of workspace (got %zu)",
                need, workspace_bytes);
    MIQ_REQUIRE(aligned(workspace, 256), MIQ_EALIGN, "miq_conv2d_general: workspace must be 256-byte aligned");
    uint16_t* xp = static_cast<uint16_t*>(workspace);
    int8_t* wp = static_cast<int8_t*>(workspace) + planes * G.plane_bytes;
    if (int rc = conv_w8_relayout(x, x_dtype, w, xp, wp, G, B, Cin, Cout, rows, kw_run, P, st)) return rc;

    W8fArgs g{xp, wp, bias, alpha, out, beta, static_cast<int>(G.M), static_cast<int>(Cout), static_cast<int>(G.K),
              static_cast<uint32_t>(G.plane_bytes), 0, 0, 0,
              static_cast<int>(G.Ho * G.Wo), static_cast<int>(G.Wo), static_cast<int>(sh), static_cast<int>(G.Lp),
              static_cast<int>(G.L), static_cast<uint32_t>(G.Wp * G.Cp * 2), static_cast<uint32_t>(G.Cp * 2),
              static_cast<uint32_t>(G.Hp * G.Wp * G.Cp * 2),
              static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(G.Lp)) + 1u};
    g.cstride_x = static_cast<int>(sw);
    g.krow_bytes = static_cast<uint32_t>(dh * G.Wp * G.Cp * 2);
    g.origin = static_cast<uint32_t>(((P - pt) * G.Wp + (P - pl)) * G.Cp * 2);
    if (dw > 1) {
        g.kwt = static_cast<int>(KW);
        g.kwt_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(KW)) + 1u;
        g.tap_bytes = static_cast<uint32_t>(dw * G.Cp * 2);
    }
    if (planes == 2) return out_dtype == MIQ_F32 ? launch_conv_w8f_np<2, kOutF32>(g, st) : launch_conv_w8f_np<2, kOutBF16>(g, st);
    if (x_dtype == MIQ_F16)
        return out_dtype == MIQ_F32 ? launch_conv_w8f_np<1, kOutF32, true>(g, st) : launch_conv_w8f_np<1, kOutBF16, true>(g, st);
    return out_dtype == MIQ_F32 ? launch_conv_w8f_np<1, kOutF32>(g, st) : launch_conv_w8f_np<1, kOutBF16>(g, st);
}

}  // namespace miq
