// Conv2d in the general form nn.Conv2d allows: dilation, groups, a stride and a padding per axis, a padding that
// differs before and after (what padding='same' needs for even filter extents).  int8 or floating-point NCHW
// activations, int8 OIHW weights, the per-channel epilogue of the other kernels.
//
// The reference's layers hand every one of these arguments to F.conv2d (quant/Weight_Only/Quantizer.py:73,
// quant/Weight_Activation/Quantizer.py:53-56); its own int8 kernel (cuda_kernels/conv2d_int8/conv2d_w8a8.cu:6-78)
// knows one stride and one padding only.  Dense layers (groups = 1) go to the implicit-GEMM MFMA kernels of
// miq_conv2d_i8 / miq_conv2d_w8 with generalised window strides (miq_conv_dense.h).  Grouped and depthwise layers (and
// whatever those kernels' 32-bit offsets do not reach) run the kernel below: a direct convolution on the vector ALUs,
// which is what such layers are bound by anyway (a depthwise layer has nine multiply-adds per output).  A lane owns
// one output pixel and kCo consecutive output channels of one group; the weights of the workgroup's channel block are
// staged through LDS kRT reduction elements (input channel, filter row, filter column) at a time, widened to the
// accumulator's type, and read back as broadcasts, so a lane issues one activation load per kCo multiply-adds.
// Depthwise and thin groups (at most 8 output channels a group) put all four waves on pixels instead, with 1, 4 or 8
// channels a lane.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <type_traits>

#include "miq_common.h"
#include "miq_conv_dense.h"
#include "miq_gemm_core.h"

namespace {

using namespace miq;

struct GeneralArgs {
    const void* x;
    const int8_t* w;
    const float* bias;
    const float* alpha;
    float beta;
    void* out;
    int B, Cin, H, W, Cout, KH, KW;
    int sh, sw, pt, pl, dh, dw;
    int cin_g, cout_g;   // channels per group
    int cblocks;         // channel blocks per group
    int Ho, Wo;
    int R;               // reduction length of one output: cin_g * KH * KW
    // Grid folding of conv_general_kernel: grid.y and grid.z hold at most 65535 entries each, so channel blocks and images
    // beyond that ride on grid.x, which is gx * fold_y * fold_z wide (x fastest).
    int gx, gy, fold_y, fold_z;
};

constexpr int kRT = 128;   // reduction elements staged per round

template <typename T>
__device__ __forceinline__ auto load_x(const T* p) {
    if constexpr (std::is_same_v<T, int8_t>)
        return static_cast<int>(*p);
    else if constexpr (std::is_same_v<T, float>)
        return *p;
    else if constexpr (std::is_same_v<T, __half>)
        return __half2float(*p);
    else   // bf16 bits
        return __uint_as_float(static_cast<uint32_t>(*p) << 16);
}

// T: int8_t | float | __half | uint16_t (bf16 bits).  PW: waves along the pixels (the other 4 / PW along channels).
// NPB: pixels per lane (64 apart, so that a wave's loads stay runs of one image row): a workgroup of a thin layer lives
// only a few microseconds, most of them waiting for its first loads, unless it has several pixels per lane to work on.
template <typename T, int OUT, int PW, int kCo, int NPB>
__global__ __launch_bounds__(256) void conv_general_kernel(GeneralArgs g) {
    using Acc = std::conditional_t<std::is_same_v<T, int8_t>, int, float>;
    constexpr int kCB = (4 / PW) * kCo;   // output channels of a workgroup
    __shared__ __attribute__((aligned(16))) Acc wl[kRT][kCB];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pw = wave % PW, cw = wave / PW;
    // (block indices unfolded: see GeneralArgs)
    const int bx = static_cast<int>(blockIdx.x % static_cast<uint32_t>(g.gx)), fold = static_cast<int>(blockIdx.x / static_cast<uint32_t>(g.gx));
    const int by = (fold % g.fold_y) * static_cast<int>(gridDim.y) + static_cast<int>(blockIdx.y);
    const int bz = (fold / g.fold_y) * static_cast<int>(gridDim.z) + static_cast<int>(blockIdx.z);
    if (by >= g.gy || bz >= g.B) return;   // (a whole workgroup: before any barrier)
    const int grp = by / g.cblocks, cb = by - grp * g.cblocks;
    const int co_wg = cb * kCB;   // the workgroup's first channel inside its group
    // Thin layers (PW == 4) deal the output pixels of the whole batch over the lanes, image after image: a small image alone
    // would not fill a workgroup, and all images share the channel block's weights.  Otherwise blockIdx.z is the image.
    constexpr bool kFlat = PW == 4;
    const int HoWo = g.Ho * g.Wo;
    int p[NPB], ih0[NPB], iw0[NPB], img[NPB];
    bool live[NPB];
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
        p[q] = ((bx * PW + pw) * NPB + q) * 64 + lane;
        if constexpr (kFlat) {
            live[q] = p[q] < g.B * HoWo;   // (B * HoWo and B * Cin * H * W fit 31 bits: checked by the host)
            img[q] = p[q] / HoWo;
            p[q] -= img[q] * HoWo;
        } else {
            live[q] = p[q] < HoWo;
            img[q] = bz;
        }
        const int ho = p[q] / g.Wo, wo = p[q] - ho * g.Wo;
        ih0[q] = ho * g.sh - g.pt;
        iw0[q] = wo * g.sw - g.pl;
    }
    const T* xg = static_cast<const T*>(g.x) + (static_cast<int64_t>(kFlat ? 0 : bz) * g.Cin + grp * g.cin_g) * g.H * g.W;
    const int img_elems = g.Cin * g.H * g.W;
    const int8_t* wg = g.w + static_cast<int64_t>(grp * g.cout_g + co_wg) * g.R;
    const int khw = g.KH * g.KW;

    Acc acc[NPB][kCo];
#pragma unroll
    for (int q = 0; q < NPB; ++q)
#pragma unroll
        for (int j = 0; j < kCo; ++j) acc[q][j] = 0;

    for (int r0 = 0; r0 < g.R; r0 += kRT) {
        const int nr = g.R - r0 < kRT ? g.R - r0 : kRT;
        int ci = r0 / khw, kh = (r0 - ci * khw) / g.KW, kw = r0 - ci * khw - kh * g.KW;   // (wave-uniform counters)
        // the activations of tap r + 1 are requested before those of tap r are used: two taps of loads in flight per lane
        auto load_tap = [&](Acc (&xv)[NPB]) {
#pragma unroll
            for (int q = 0; q < NPB; ++q) {
                const int ih = ih0[q] + kh * g.dh, iw = iw0[q] + kw * g.dw;
                xv[q] = 0;
                if (live[q] && static_cast<unsigned>(ih) < static_cast<unsigned>(g.H) && static_cast<unsigned>(iw) < static_cast<unsigned>(g.W))
                    xv[q] = load_x(xg + (kFlat ? img[q] * img_elems : 0) + (ci * g.H + ih) * g.W + iw);   // (31 bits: checked by the host)
            }
            if (++kw == g.KW) {
                kw = 0;
                if (++kh == g.KH) {
                    kh = 0;
                    ++ci;
                }
            }
        };
        Acc xv[NPB], xn[NPB];
        load_tap(xv);   // (ahead of the weight staging, whose latency it shares)
        __syncthreads();   // every wave has finished with the round before
        for (int e = threadIdx.x; e < kCB * nr; e += 256) {
            const int c = e / nr, r = e - c * nr;
            wl[r][c] = co_wg + c < g.cout_g ? static_cast<Acc>(wg[static_cast<int64_t>(c) * g.R + r0 + r]) : static_cast<Acc>(0);
        }
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            if (r + 1 < nr) load_tap(xn);
            const Acc* wr = &wl[r][cw * kCo];
            Acc wv[kCo];
            if constexpr (kCo < 4) {
#pragma unroll
                for (int j = 0; j < kCo; ++j) wv[j] = wr[j];
            } else {
#pragma unroll
                for (int j = 0; j < kCo; j += 4) {
                    const auto w4 = *reinterpret_cast<const std::conditional_t<std::is_same_v<Acc, int>, int4, float4>*>(wr + j);
                    wv[j] = w4.x;
                    wv[j + 1] = w4.y;
                    wv[j + 2] = w4.z;
                    wv[j + 3] = w4.w;
                }
            }
#pragma unroll
            for (int q = 0; q < NPB; ++q)
#pragma unroll
                for (int j = 0; j < kCo; ++j) {
                    if constexpr (std::is_same_v<Acc, int>)
                        acc[q][j] += xv[q] * wv[j];
                    else
                        acc[q][j] = __builtin_fmaf(xv[q], wv[j], acc[q][j]);
                }
#pragma unroll
            for (int q = 0; q < NPB; ++q) xv[q] = xn[q];
        }
    }

#pragma unroll
    for (int q = 0; q < NPB; ++q) {
        if (!live[q]) continue;
#pragma unroll
        for (int j = 0; j < kCo; ++j) {
            const int co = co_wg + cw * kCo + j;
            if (co >= g.cout_g) break;
            const int n = grp * g.cout_g + co;
            const int64_t idx = (static_cast<int64_t>(img[q]) * g.Cout + n) * HoWo + p[q];
            if constexpr (OUT == MIQ_I32) {
                static_cast<int*>(g.out)[idx] = static_cast<int>(acc[q][j]);
            } else {
                const float y = dequant1(acc[q][j], g.bias ? g.bias[n] : 0.f, g.alpha[n], g.beta);
                if constexpr (OUT == MIQ_F32)
                    static_cast<float*>(g.out)[idx] = y;
                else
                    static_cast<uint16_t*>(g.out)[idx] = f32_to_bf16_bits(y);
            }
        }
    }
}

// Depthwise layers (one input and one output channel a group) with stride 1 or 2 and dilation 1 along x and a filter 3, 5
// or 7 wide: a lane owns four consecutive outputs of one output row and, per filter row, loads the 3 * SW + KW activations
// under them once (the kernel above loads KW per output and checks every one against the image: four times the
// instructions of the arithmetic).  The units (image, channel, row, quad of pixels) are dealt over the lanes in that order,
// so small images fill workgroups too; a lane reads its channel's filter straight from memory (KH * KW cached bytes).
template <typename T, int OUT, int KW, int SW>
__global__ __launch_bounds__(256) void conv_depthwise_rows_kernel(GeneralArgs g, uint32_t units, int quads, int vec_in, int vec_out) {
    using Acc = std::conditional_t<std::is_same_v<T, int8_t>, int, float>;
    constexpr int NX = 3 * SW + KW;   // activations of one filter row under four outputs
    const uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= units) return;
    const uint32_t rowid = u / quads;                 // (image * C + channel) * Ho + ho
    const int wo0 = static_cast<int>(u - rowid * quads) * 4;
    const uint32_t plane = rowid / g.Ho;              // image * C + channel
    const int ho = static_cast<int>(rowid - plane * g.Ho);
    const int c = static_cast<int>(plane % g.Cout);
    const T* xp = static_cast<const T*>(g.x) + static_cast<size_t>(plane) * g.H * g.W;   // (Cin == Cout)
    const int8_t* wc = g.w + c * g.KH * KW;
    const int iw0 = wo0 * SW - g.pl;
    bool col_ok[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) col_ok[k] = static_cast<unsigned>(iw0 + k) < static_cast<unsigned>(g.W);
    // vec_in (wave-uniform: W % 4 == 0, aligned planes, pad_left <= KW - SW): the activations straight under the lane's
    // outputs, columns wo0 * SW .. + 4 * SW - 1 = k in [pl, pl + 4 * SW), are SW aligned vectors of four; each is taken as one
    // load where it lies inside the row (outputs may reach past it), and only the others are single loads
    struct alignas(4 * sizeof(T)) Vec4 {
        T e[4];
    };
    bool own[SW];
#pragma unroll
    for (int v = 0; v < SW; ++v) own[v] = vec_in && wo0 * SW + 4 * v + 4 <= g.W;
    Acc acc[4] = {0, 0, 0, 0};
    for (int kh = 0; kh < g.KH; ++kh) {
        const int ih = ho * g.sh - g.pt + kh * g.dh;
        if (static_cast<unsigned>(ih) >= static_cast<unsigned>(g.H)) continue;
        const T* xr = xp + ih * g.W + iw0;
        Vec4 vv[SW];
#pragma unroll
        for (int v = 0; v < SW; ++v)
            if (own[v]) vv[v] = *reinterpret_cast<const Vec4*>(xr + g.pl + 4 * v);
        Acc xv[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            xv[k] = 0;
            const int r = k - g.pl;   // position among the lane's own columns
            bool taken = false;
#pragma unroll
            for (int v = 0; v < SW; ++v) {
                if (own[v] && r >= 4 * v && r < 4 * v + 4) {
                    const int q = r - 4 * v;
                    const T e = q == 0 ? vv[v].e[0] : q == 1 ? vv[v].e[1] : q == 2 ? vv[v].e[2] : vv[v].e[3];
                    xv[k] = load_x(&e);
                    taken = true;
                }
            }
            if (!taken && col_ok[k]) xv[k] = load_x(xr + k);
        }
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            const Acc wv = static_cast<Acc>(wc[kh * KW + kw]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (std::is_same_v<Acc, int>)
                    acc[e] += xv[kw + e * SW] * wv;
                else
                    acc[e] = __builtin_fmaf(xv[kw + e * SW], wv, acc[e]);
            }
        }
    }
    const size_t idx = (static_cast<size_t>(plane) * g.Ho + ho) * g.Wo + wo0;
    float bias = 0.f, alpha = 1.f;
    if constexpr (OUT != MIQ_I32) {
        alpha = g.alpha[c];
        if (g.bias) bias = g.bias[c];
    }
    if (vec_out) {   // (wave-uniform) Wo % 4 == 0 and an aligned output: the lane's four outputs are one vector
        if constexpr (OUT == MIQ_I32) {
            *reinterpret_cast<int4*>(static_cast<int*>(g.out) + idx) = int4{(int)acc[0], (int)acc[1], (int)acc[2], (int)acc[3]};
        } else {
            const float y0 = dequant1(acc[0], bias, alpha, g.beta), y1 = dequant1(acc[1], bias, alpha, g.beta),
                        y2 = dequant1(acc[2], bias, alpha, g.beta), y3 = dequant1(acc[3], bias, alpha, g.beta);
            if constexpr (OUT == MIQ_F32)
                *reinterpret_cast<float4*>(static_cast<float*>(g.out) + idx) = float4{y0, y1, y2, y3};
            else
                *reinterpret_cast<uint2*>(static_cast<uint16_t*>(g.out) + idx) = pack4_bf16(y0, y1, y2, y3);
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (wo0 + e >= g.Wo) break;
        if constexpr (OUT == MIQ_I32) {
            static_cast<int*>(g.out)[idx + e] = static_cast<int>(acc[e]);
        } else {
            const float y = dequant1(acc[e], bias, alpha, g.beta);
            if constexpr (OUT == MIQ_F32)
                static_cast<float*>(g.out)[idx + e] = y;
            else
                static_cast<uint16_t*>(g.out)[idx + e] = f32_to_bf16_bits(y);
        }
    }
}

template <typename T, int OUT>
int launch_general(const GeneralArgs& a0, int64_t B, int64_t groups, hipStream_t st) {
    GeneralArgs a = a0;
    if (a.cin_g == 1 && a.cout_g == 1 && (a.sw == 1 || a.sw == 2) && a.dw == 1 && (a.KW == 3 || a.KW == 5 || a.KW == 7)) {
        const int64_t quads = ceil_div(a.Wo, 4), units = B * a.Cout * a.Ho * quads;
        if (units < (1ll << 31) && static_cast<int64_t>(a.H) * a.W < (1ll << 31)) {
            const unsigned grid = static_cast<unsigned>(ceil_div(units, 256));
            const uint32_t nu = static_cast<uint32_t>(units);
            const int nq = static_cast<int>(quads);
            // vectors of four: activations when every row of every plane starts on one and the lane's own columns lie inside
            // the filter's window (pad_left <= KW - stride); outputs when the output rows are whole vectors
            const int vec_in = a.W % 4 == 0 && aligned(a.x, 4 * sizeof(T)) && a.pl <= a.KW - a.sw;
            const int vec_out = a.Wo % 4 == 0 && aligned(a.out, OUT == MIQ_BF16 ? 8 : 16);
#define MIQ_DW_ROWS(KW_, SW_) conv_depthwise_rows_kernel<T, OUT, KW_, SW_><<<grid, 256, 0, st>>>(a, nu, nq, vec_in, vec_out)
            if (a.sw == 1) {
                if (a.KW == 3) MIQ_DW_ROWS(3, 1);
                else if (a.KW == 5) MIQ_DW_ROWS(5, 1);
                else MIQ_DW_ROWS(7, 1);
            } else {
                if (a.KW == 3) MIQ_DW_ROWS(3, 2);
                else if (a.KW == 5) MIQ_DW_ROWS(5, 2);
                else MIQ_DW_ROWS(7, 2);
            }
#undef MIQ_DW_ROWS
            return check_launch("conv_depthwise_rows_kernel");
        }
    }
    const int64_t HoWo = static_cast<int64_t>(a.Ho) * a.Wo;
    // depthwise and thin groups: every wave on pixels (of the whole batch), one block of 1, 4 or 8 channels ...
    const bool thin = a.cout_g <= 8 && B * HoWo < (1ll << 31) && B * a.Cin * a.H * a.W < (1ll << 31);
    const int cb = !thin ? 64 : a.cout_g == 1 ? 1 : a.cout_g <= 4 ? 4 : 8;
    a.cblocks = static_cast<int>(ceil_div(a.cout_g, cb));
    a.B = static_cast<int>(B);
    const int64_t gy = groups * a.cblocks;
    // ... and four pixels a lane once that still leaves every CU a few workgroups
    const bool many = thin && ceil_div(B * HoWo, 1024) * gy >= 1024;
    const int64_t gx = thin ? ceil_div(B * HoWo, many ? 1024 : 256) : ceil_div(HoWo, 64);
    const int64_t gz = thin ? 1 : B;
    const int64_t fold_y = ceil_div(gy, 65535), fold_z = ceil_div(gz, 65535);
    MIQ_REQUIRE(gx * fold_y * fold_z < (1ll << 31) && gy < (1ll << 31), MIQ_ERANGE,
                "miq_conv2d_general: too many workgroups (pixels blocks=%lld, channel blocks=%lld, images=%lld)", (long long)gx,
                (long long)gy, (long long)gz);
    a.gx = static_cast<int>(gx); a.gy = static_cast<int>(gy); a.fold_y = static_cast<int>(fold_y); a.fold_z = static_cast<int>(fold_z);
    const dim3 grid(static_cast<unsigned>(gx * fold_y * fold_z), static_cast<unsigned>(ceil_div(gy, fold_y)),
                    static_cast<unsigned>(ceil_div(gz, fold_z)));
    if (cb == 1 && many)
        conv_general_kernel<T, OUT, 4, 1, 4><<<grid, 256, 0, st>>>(a);
    else if (cb == 1)
        conv_general_kernel<T, OUT, 4, 1, 1><<<grid, 256, 0, st>>>(a);
    else if (cb == 4 && many)
        conv_general_kernel<T, OUT, 4, 4, 4><<<grid, 256, 0, st>>>(a);
    else if (cb == 4)
        conv_general_kernel<T, OUT, 4, 4, 1><<<grid, 256, 0, st>>>(a);
    else if (thin && many)
        conv_general_kernel<T, OUT, 4, 8, 4><<<grid, 256, 0, st>>>(a);
    else if (thin)
        conv_general_kernel<T, OUT, 4, 8, 1><<<grid, 256, 0, st>>>(a);
    else
        conv_general_kernel<T, OUT, 1, 16, 1><<<grid, 256, 0, st>>>(a);
    return check_launch("miq_conv2d_general");
}

template <typename T>
int launch_general_out(const GeneralArgs& a, int64_t B, int64_t groups, int out_dtype, hipStream_t st) {
    if constexpr (std::is_same_v<T, int8_t>) {
        if (out_dtype == MIQ_I32) return launch_general<T, MIQ_I32>(a, B, groups, st);
    }
    if (out_dtype == MIQ_F32) return launch_general<T, MIQ_F32>(a, B, groups, st);
    return launch_general<T, MIQ_BF16>(a, B, groups, st);
}

}  // namespace

// Workspace of the dense route; 0 = the direct kernel (grouped layers, or sizes past the MFMA kernels' offsets).
extern "C" size_t miq_conv2d_general_workspace(int x_dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                                               int64_t KW, int64_t stride_h, int64_t stride_w, int64_t pad_top, int64_t pad_bottom,
                                               int64_t pad_left, int64_t pad_right, int64_t dil_h, int64_t dil_w, int64_t groups) {
    if (groups != 1) return 0;
    if (!miq::all_below_2_31({B, Cin, H, W, Cout, KH, KW, stride_h, stride_w, pad_top, pad_bottom, pad_left, pad_right, dil_h, dil_w}))
        return 0;
    if (KH == 1) dil_h = 1;   // as in miq_conv2d_general: a filter one tap high / wide has no dilation
    if (KW == 1) dil_w = 1;
    if (x_dtype == MIQ_I8)
        return miq::conv2d_i8_dense_workspace(B, Cin, H, W, Cout, KH, KW, stride_h, stride_w, pad_top, pad_bottom, pad_left,
                                              pad_right, dil_h, dil_w);
    if (x_dtype == MIQ_F32 || x_dtype == MIQ_F16 || x_dtype == MIQ_BF16)
        return miq::conv2d_w8_dense_workspace(x_dtype, B, Cin, H, W, Cout, KH, KW, stride_h, stride_w, pad_top, pad_bottom,
                                              pad_left, pad_right, dil_h, dil_w);
    return 0;
}

extern "C" int miq_conv2d_general(const void* x, int x_dtype, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                                  int64_t Cout, int64_t KH, int64_t KW, int64_t stride_h, int64_t stride_w, int64_t pad_top,
                                  int64_t pad_bottom, int64_t pad_left, int64_t pad_right, int64_t dil_h, int64_t dil_w,
                                  int64_t groups, const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    MIQ_REQUIRE(B >= 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && KH > 0 && KW > 0, MIQ_EINVAL,
                "miq_conv2d_general: bad shape (B=%lld Cin=%lld H=%lld W=%lld Cout=%lld KH=%lld KW=%lld)", (long long)B,
                (long long)Cin, (long long)H, (long long)W, (long long)Cout, (long long)KH, (long long)KW);
    MIQ_REQUIRE(stride_h > 0 && stride_w > 0 && dil_h > 0 && dil_w > 0 && pad_top >= 0 && pad_bottom >= 0 && pad_left >= 0 &&
                    pad_right >= 0,
                MIQ_EINVAL, "miq_conv2d_general: stride and dilation must be positive, padding non-negative");
    MIQ_REQUIRE(miq::all_below_2_31({B, Cin, H, W, Cout, groups}), MIQ_ERANGE, "miq_conv2d_general: sizes must stay below 2^31");
    MIQ_REQUIRE(KH < (1 << 15) && KW < (1 << 15) && stride_h < (1 << 15) && stride_w < (1 << 15) && dil_h < (1 << 15) &&
                    dil_w < (1 << 15) && pad_top < (1 << 15) && pad_bottom < (1 << 15) && pad_left < (1 << 15) && pad_right < (1 << 15),
                MIQ_ERANGE, "miq_conv2d_general: filter size, stride, dilation and padding must stay below 32768");
    MIQ_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, MIQ_EINVAL,
                "miq_conv2d_general: groups (%lld) must divide Cin (%lld) and Cout (%lld)", (long long)groups, (long long)Cin,
                (long long)Cout);
    MIQ_REQUIRE(x_dtype == MIQ_I8 || x_dtype == MIQ_F32 || x_dtype == MIQ_F16 || x_dtype == MIQ_BF16, MIQ_EINVAL,
                "miq_conv2d_general: unknown x_dtype %d", x_dtype);
    MIQ_REQUIRE(out_dtype == MIQ_F32 || out_dtype == MIQ_BF16 || (out_dtype == MIQ_I32 && x_dtype == MIQ_I8), MIQ_EINVAL,
                "miq_conv2d_general: out_dtype must be MIQ_F32, MIQ_BF16 or, for int8 activations, MIQ_I32");
    // A filter one tap high (wide) has no dilation along that axis, whatever the caller names: the dense kernels' "a tap is
    // a row of its own" form of a dilation along x divides by the taps of a filter row with a 32-bit reciprocal, which one
    // tap a row does not have.
    if (KH == 1) dil_h = 1;
    if (KW == 1) dil_w = 1;
    const int64_t eh = dil_h * (KH - 1) + 1, ew = dil_w * (KW - 1) + 1;   // extent of the dilated filter
    MIQ_REQUIRE(H + pad_top + pad_bottom >= eh && W + pad_left + pad_right >= ew, MIQ_EINVAL,
                "miq_conv2d_general: dilated filter larger than the padded input");
    const int64_t Ho = (H + pad_top + pad_bottom - eh) / stride_h + 1, Wo = (W + pad_left + pad_right - ew) / stride_w + 1;
    const int64_t R = miq::sat_product({Cin / groups, KH, KW});
    MIQ_REQUIRE(miq::sat_product({Cin, H, W}) < (1ll << 31) && Ho * Wo < (1ll << 31) && miq::sat_product({Cout, R}) < (1ll << 31) &&
                    H + pad_top + pad_bottom < (1ll << 30) &&
                    W + pad_left + pad_right < (1ll << 30) && eh < (1ll << 30) && ew < (1ll << 30),
                MIQ_ERANGE, "miq_conv2d_general: shape too large");
    if (B == 0) return MIQ_OK;
    MIQ_REQUIRE(x && w && out, MIQ_EINVAL, "miq_conv2d_general: null pointer");
    MIQ_REQUIRE(out_dtype == MIQ_I32 || alpha, MIQ_EINVAL, "miq_conv2d_general: alpha is required unless out_dtype is MIQ_I32");
    const size_t xsz = x_dtype == MIQ_I8 ? 1 : x_dtype == MIQ_F32 ? 4 : 2, osz = out_dtype == MIQ_BF16 ? 2 : 4;
    MIQ_REQUIRE(miq::aligned(x, xsz) && miq::aligned(out, osz) && miq::aligned(bias, 4) && miq::aligned(alpha, 4), MIQ_EALIGN,
                "miq_conv2d_general: pointers must be aligned to their element size");
    hipStream_t st = miq::as_stream(stream);
    // dense layers: the MFMA kernels (their vector stores want four output elements' alignment)
    if (groups == 1 && miq::aligned(out, 4 * osz)) {
        const size_t need = miq_conv2d_general_workspace(x_dtype, B, Cin, H, W, Cout, KH, KW, stride_h, stride_w, pad_top,
                                                         pad_bottom, pad_left, pad_right, dil_h, dil_w, groups);
        if (need) {
            if (x_dtype == MIQ_I8)
                return miq::conv2d_i8_dense(static_cast<const int8_t*>(x), w, B, Cin, H, W, Cout, KH, KW, stride_h, stride_w, pad_top,
                                            pad_bottom, pad_left, pad_right, dil_h, dil_w, bias, alpha, beta, out, out_dtype,
                                            workspace, workspace_bytes, st);
            return miq::conv2d_w8_dense(x, x_dtype, w, B, Cin, H, W, Cout, KH, KW, stride_h, stride_w, pad_top, pad_bottom, pad_left,
                                        pad_right, dil_h, dil_w, bias, alpha, beta, out, out_dtype, workspace, workspace_bytes, st);
        }
    }
    GeneralArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.alpha = alpha; a.beta = beta; a.out = out;
    a.Cin = (int)Cin; a.H = (int)H; a.W = (int)W; a.Cout = (int)Cout; a.KH = (int)KH; a.KW = (int)KW;
    a.sh = (int)stride_h; a.sw = (int)stride_w; a.pt = (int)pad_top; a.pl = (int)pad_left; a.dh = (int)dil_h; a.dw = (int)dil_w;
    a.cin_g = (int)(Cin / groups); a.cout_g = (int)(Cout / groups);
    a.Ho = (int)Ho; a.Wo = (int)Wo; a.R = (int)R;
    switch (x_dtype) {
        case MIQ_I8: return launch_general_out<int8_t>(a, B, groups, out_dtype, st);
        case MIQ_F32: return launch_general_out<float>(a, B, groups, out_dtype, st);
        case MIQ_F16: return launch_general_out<__half>(a, B, groups, out_dtype, st);
        default: return launch_general_out<uint16_t>(a, B, groups, out_dtype, st);
    }
}
