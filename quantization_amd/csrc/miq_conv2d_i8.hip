// int8 Conv2d as an implicit GEMM on CDNA4 MFMA, NCHW in / OIHW weights / NCHW out.
//
//   M = B*Ho*Wo (output pixels), N = Cout, K = KH * KW * Cin.
//
// Two cheap re-layout passes make every MFMA operand K-contiguous:
//   * activations: NCHW -> zero-padded channels-last xp[B][H+2p][W+2p][Cp] (Cp = Cin rounded
//     up to 4).  For one output pixel and one filter row kh, the KW*Cp bytes it needs are then
//     one contiguous run starting at xp[b][ho*s+kh][wo*s][0]: no bounds checks, no gather.
//   * weights: OIHW -> wp[Cout][KH][Lp], Lp = KW*Cp rounded up to 16, zero where padded.  The
//     zero weights cancel whatever the activation run holds past KW*Cp.
// K is walked in 16-byte chunks; chunk q lies in filter row q / cps at byte (q % cps) * 16 of
// that row's run (cps = Lp / 16), relative to the pixel's base.  The main loop is the GEMM's: 128-byte K-tiles, LDS-DMA
// staging, double buffering, v_mfma_i32_16x16x64_i8.  Pixels are the MFMA's first operand,
// so a lane holds four consecutive pixels of one output channel: a 16-byte NCHW store.
//
// Replaces conv2d_kernel_int8 / conv2d_int8 (cuda_kernels/conv2d_int8/conv2d_w8a8.cu:6-114).
#include <algorithm>
#include <cstdlib>

#include <type_traits>

#include "miq_common.h"
#include "miq_conv_dense.h"
#include "miq_gemm_core.h"
#include "miq_patch.h"

namespace {

using namespace miq;

constexpr int kCinAlign = 4;

struct ConvGeom {
    int64_t B, Cin, H, W, Cout, KH, KW, stride, pad;
    int64_t Ho, Wo, Hp, Wp, Cp, L, Lp, Kp, M;
    int64_t xp_bytes, wp_bytes;
    // 3 x 3 / stride 1 layers (conv3x3_patch_kernel): chunk-plane layout of the padded copy
    bool planar;
    int64_t pitch;   // pixels per padded row (Wp rounded up to 4)
    int64_t ptot;    // flat padded pixels B * Hp * pitch
    int64_t group;   // small images: whole images per tile of the patch kernel (1 = tiles of one image)
    // 1 x 1 filters with a stride and no padding read only every stride-th pixel of every stride-th row: the
    // re-layout keeps just those (sub = the caller's stride, Hs x Ws = the caller's image) and the problem
    // continues as a stride-1 convolution over an Ho x Wo image (H, W, stride above are then the reduced ones).
    int64_t sub, Hs, Ws;
};

constexpr int kPatchBytes = 4 * kPlaneSlots * kChunk;   // 44 KiB
constexpr int kWStageBytes = 9 * 4 * 64 * kChunk;       // 36 KiB: [tap][chunk][channel][16 B]
constexpr int kPlanarStage = kPatchBytes + kWStageBytes;

bool make_geom(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
               int64_t pad, ConvGeom& g, bool relayout = true) {   // relayout = false: the caller's channels-last form is staged as it is
    g = ConvGeom{B, Cin, H, W, Cout, KH, KW, stride, pad, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, false, 0, 0, 1, 1, H, W};
    if (B < 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout < 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return false;
    // (sizes first: below 2^31 each, filter / stride / padding below 2^15, so that the sums and two-factor products below
    // stay inside int64; longer products are held at 2^60 and refused by the range checks of whoever asked)
    if (!all_below_2_31({B, Cin, H, W, Cout}) || KH >= (1 << 15) || KW >= (1 << 15) || stride >= (1 << 15) || pad >= (1 << 15))
        return false;
    const char* ss = dev_switch("MIQ_CONV_SUBSAMPLE");   // developer A/B switch: 0 = re-lay the whole image
    if (relayout && KH == 1 && KW == 1 && pad == 0 && stride > 1 && H * W < (1ll << 30) && !(ss && ss[0] == '0')) {
        g.sub = stride;
        g.H = H = (H - 1) / stride + 1;
        g.W = W = (W - 1) / stride + 1;
        g.stride = stride = 1;
    }
    g.Hp = H + 2 * pad;
    g.Wp = W + 2 * pad;
    if (g.Hp < KH || g.Wp < KW) return false;
    g.Ho = (g.Hp - KH) / stride + 1;  // conv2d_w8a8.cu:91-92
    g.Wo = (g.Wp - KW) / stride + 1;
    g.Cp = round_up(Cin, kCinAlign);
    g.L = KW * g.Cp;
    g.Lp = round_up(g.L, kChunk);
    g.Kp = KH * g.Lp;
    g.M = sat_product({B, g.Ho, g.Wo});
    g.xp_bytes = sat_product({B, g.Hp, g.Wp, g.Cp});
    g.wp_bytes = sat_product({Cout, g.Kp});
    // The 3 x 3 / stride 1 kernel: whole 64-channel groups, four output pixels per 16-byte store, and a
    // patch (the padded rows a tile touches, two more, two pixels) that fits the LDS planes.
    g.pitch = round_up(g.Wp, 4);
    g.ptot = sat_product({B, g.Hp, g.pitch});
    const char* sw = dev_switch("MIQ_CONV_PLANAR");   // developer A/B switch: 0 = im2col-order kernel for every layer
    g.planar = false;
    if (relayout && KH == 3 && KW == 3 && stride == 1 && Cin % 64 == 0 && g.M >= 4 * kTM && g.M < (1ll << 31) &&
        !(sw && sw[0] == '0')) {
        // Images whose pixel count is no multiple of four (7 x 7) leave pixel by pixel; their rows then need no whole
        // dwords of pixels either, and a tile holds more of them (eight 7 x 7 images instead of six).
        if ((g.Ho * g.Wo) % 4 != 0) {
            g.pitch = g.Wp;
            g.ptot = sat_product({B, g.Hp, g.pitch});
        }
        g.planar = patch_tiling(g.Ho * g.Wo, g.Wo, g.Hp, g.pitch, kPlaneSlots, g.group);
        if (!g.planar) {
            g.pitch = round_up(g.Wp, 4);
            g.ptot = sat_product({B, g.Hp, g.pitch});
        }
    }
    if (g.planar) {
        g.xp_bytes = sat_product({g.ptot, Cin});
        g.wp_bytes = sat_product({ceil_div(Cout, 64), Cin / 64, kWStageBytes});
    }
    return true;
}

// NCHW -> zero-padded NHWC.  One workgroup moves a 64-pixel x 64-channel block through LDS.
__global__ __launch_bounds__(256) void nhwc_pad_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ xp, int Cin,
                                                       int H, int W, int Hp, int Wp, int Cp, int pad, int cblocks) {
    __shared__ int8_t tile[64][68];
    const int w0 = (blockIdx.x / cblocks) * 64, c0 = (blockIdx.x % cblocks) * 64;
    const int hp = blockIdx.y, b = blockIdx.z;
    const int h = hp - pad;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = threadIdx.x + i * 256, c = idx >> 6, wl = idx & 63;
        const int w = w0 + wl - pad;
        int8_t v = 0;
        if (c0 + c < Cin && h >= 0 && h < H && w >= 0 && w < W)
            v = x[((static_cast<int64_t>(b) * Cin + c0 + c) * H + h) * W + w];
        tile[c][wl] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + i * 256, wl = idx >> 4, cw = (idx & 15) * 4;
        if (w0 + wl < Wp && c0 + cw < Cp) {
            const uint32_t v = (static_cast<uint32_t>(static_cast<uint8_t>(tile[cw][wl]))) |
                               (static_cast<uint32_t>(static_cast<uint8_t>(tile[cw + 1][wl])) << 8) |
                               (static_cast<uint32_t>(static_cast<uint8_t>(tile[cw + 2][wl])) << 16) |
                               (static_cast<uint32_t>(static_cast<uint8_t>(tile[cw + 3][wl])) << 24);
            *reinterpret_cast<uint32_t*>(xp + ((static_cast<int64_t>(b) * Hp + hp) * Wp + w0 + wl) * Cp + c0 + cw) = v;
        }
    }
}

// 4 x 4 byte transpose in registers: r[j] holds bytes (j, 0..3); d[i] gets bytes (0..3, i).
__device__ __forceinline__ void transpose4x4_bytes(const uint32_t (&r)[4], uint32_t (&d)[4]) {
    const uint32_t t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u);  // r0.b0 r1.b0 r0.b1 r1.b1
    const uint32_t t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);  // r0.b2 r1.b2 r0.b3 r1.b3
    const uint32_t u0 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u);
    const uint32_t u1 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    d[0] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
    d[1] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
    d[2] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
    d[3] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
}

// NCHW -> zero-padded NHWC, the image interior.  A workgroup moves 64 channels x 64 consecutive
// pixels (flattened h*W + w, so short rows still fill a tile) of one image: every thread gathers a
// 4-channel x 4-pixel byte block (one dword per channel plane when DWORD, i.e. H*W % 4 == 0 and an
// aligned base; single bytes otherwise), transposes it in registers, and the dwords go through a
// 4 KiB LDS tile laid out [pixel][channel] so that each thread can store 16 channel bytes of one
// pixel.  The 16-byte groups of a tile row are XOR-ed with the pixel's dword index to spread the
// writes over the banks.  The zero border is written by nhwc_border_items.
//
// plane_stride != 0 selects the chunk-plane layout instead: xp[c / 16][b][hp][wp][c % 16], planes
// plane_stride bytes apart, Wp then being the row pitch in pixels (Cp is a multiple of 64 there).
//
// sub > 1 (1 x 1 filters with a stride): the H x W image re-laid is every sub-th pixel of every sub-th row of
// a source image Ws pixels wide whose channel planes are HWs pixels apart.  MODE 0 gathers single bytes, MODE 1
// dwords (sub == 1), MODE 2 serves sub == 2 with W % 4 == 0 and Ws % 4 == 0: the four pixels of a thread are
// the even bytes of one aligned 8-byte run of the source row.
template <int MODE, int NT = 1>
__device__ __forceinline__ void nhwc_interior_block(const int8_t* __restrict__ x, int8_t* __restrict__ xp, int Cin, int H,
                                                    int W, int Hp, int Wp, int Cp, int pad, int cblocks, int bx,
                                                    int64_t b, int64_t plane_stride = 0, int sub = 1, int Ws = 0,
                                                    int HWs = 0) {
    // NT tiles of 64 pixels per workgroup, all their loads issued before the first is used: with one tile a CU's eight
    // workgroups keep 32 KB in flight, half of what the memory system needs to stay busy.
    __shared__ __attribute__((aligned(16))) uint32_t tile[NT][64 * 16];
    const int t = threadIdx.x;
    const int HW = H * W;
    const int p0 = (bx / cblocks) * (64 * NT), c0 = (bx % cblocks) * 64;
    const int cspan = Cp - c0 < 64 ? Cp - c0 : 64;   // channel bytes this block owns per pixel (multiple of 4)
    {
        const int p4 = t & 15, cg = t >> 4;
        uint32_t r[NT][4];
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int pf = p0 + s * 64 + p4 * 4;   // first of this thread's four pixels
            int so[4] = {0, 0, 0, 0};              // sub > 1: their offsets inside a source channel plane
            if (MODE != 1 && sub > 1) {
#pragma unroll
                for (int k = 0; k < (MODE == 2 ? 1 : 4); ++k) {
                    const int h = (pf + k) / W, w = pf + k - h * W;
                    so[k] = h * sub * Ws + w * sub;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + cg * 4 + j;
                r[s][j] = 0u;
                if (c < Cin) {
                    if constexpr (MODE == 1) {
                        const int8_t* src = x + (b * Cin + c) * HW + pf;
                        if (pf < HW) r[s][j] = *reinterpret_cast<const uint32_t*>(src);
                    } else if constexpr (MODE == 2) {
                        if (pf < HW) {
                            const uint2 v = *reinterpret_cast<const uint2*>(x + (b * Cin + c) * HWs + so[0]);
                            r[s][j] = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u);   // bytes 0, 2 of v.x, then of v.y
                        }
                    } else if (sub > 1) {
                        const int8_t* src = x + (b * Cin + c) * HWs;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (pf + k < HW) r[s][j] |= static_cast<uint32_t>(static_cast<uint8_t>(src[so[k]])) << (8 * k);
                    } else {
                        const int8_t* src = x + (b * Cin + c) * HW + pf;
                        if (pf + 3 < HW) {   // four pixels inside the plane: one dword load of any alignment
                            uint32_t v;
                            __builtin_memcpy(&v, src, 4);
                            r[s][j] = v;
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (pf + k < HW) r[s][j] |= static_cast<uint32_t>(static_cast<uint8_t>(src[k])) << (8 * k);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            uint32_t d[4];
            transpose4x4_bytes(r[s], d);
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[s][(p4 * 4 + i) * 16 + (cg ^ ((p4 & 3) << 2))] = d[i];
        }
    }
    __syncthreads();
    const int pl = t >> 2, part = t & 3;
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const int p = p0 + s * 64 + pl;
        if (p < HW && part * 16 < cspan) {
            const uint4 v = *reinterpret_cast<const uint4*>(&tile[s][pl * 16 + ((part ^ ((pl >> 2) & 3)) << 2)]);
            const int h = p / W, w = p - h * W;
            int8_t* o = xp + ((b * Hp + h + pad) * Wp + w + pad) * Cp + c0 + part * 16;
            if (plane_stride) o = xp + ((c0 >> 4) + part) * plane_stride + ((b * Hp + h + pad) * Wp + w + pad) * 16;
            if ((Cp & 15) == 0) {
                *reinterpret_cast<uint4*>(o) = v;
            } else {
                const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (part * 16 + k * 4 < cspan) *reinterpret_cast<uint32_t*>(o + k * 4) = vv[k];
            }
        }
    }
}

// Zeros in the `pad` rows above and below and the `pad` columns left and right of every image: one item per
// border pixel (all of its channel bytes), 32-bit index arithmetic (64-bit divisions are an order dearer).
__device__ __forceinline__ void nhwc_border_items(int8_t* __restrict__ xp, int H, int W, int Hp, int Wp, int Cp, int pad,
                                                  uint32_t total, uint32_t first, uint32_t step, int64_t plane_stride = 0) {
    const uint32_t per_image = 2 * pad * Wp + 2 * pad * H;   // border pixels of one image
    // An item is one 16-byte chunk of one border pixel where pixels are whole chunks (neighbouring lanes then write
    // neighbouring chunks: a pixel of 512 channels is 32 lanes' one store each, not one lane's 32), else a whole pixel.
    const bool chunked = plane_stride != 0 || (Cp & 15) == 0;
    const uint32_t cpp = chunked ? static_cast<uint32_t>(Cp >> 4) : 1u;   // items per pixel
    const uint64_t items = static_cast<uint64_t>(total) * cpp;
    for (uint64_t it = first; it < items; it += step) {
        const uint32_t idx = static_cast<uint32_t>(it / cpp), c = static_cast<uint32_t>(it - static_cast<uint64_t>(idx) * cpp);
        const uint32_t b = idx / per_image, e = idx - b * per_image;
        uint32_t hp, wp;
        if (e < static_cast<uint32_t>(2 * pad * Wp)) {   // full rows: the first `pad` and the last `pad`
            const uint32_t r = e / Wp;
            hp = r < static_cast<uint32_t>(pad) ? r : H + r;
            wp = e - r * Wp;
        } else {                                         // side columns of the H interior rows
            const uint32_t s = e - 2 * pad * Wp, r = s / (2 * pad), k = s - r * 2 * pad;
            hp = r + pad;
            wp = k < static_cast<uint32_t>(pad) ? k : W + k;
        }
        const int64_t pix = (static_cast<int64_t>(b) * Hp + hp) * Wp + wp;
        if (plane_stride) {
            *reinterpret_cast<uint4*>(xp + c * plane_stride + pix * 16) = uint4{0u, 0u, 0u, 0u};
        } else if (chunked) {
            reinterpret_cast<uint4*>(xp + pix * Cp)[c] = uint4{0u, 0u, 0u, 0u};
        } else {
            for (int k = 0; k < (Cp >> 2); ++k) reinterpret_cast<uint32_t*>(xp + pix * Cp)[k] = 0u;
        }
    }
}

// Same transform for Cin <= 4 (Cp == 4, e.g. an RGB stem): one thread per four pixels.
__global__ __launch_bounds__(256) void nhwc_pad_c4_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ xp, int Cin,
                                                          int H, int W, int Hp, int Wp, int pad, int64_t total) {
    const int W4 = W >> 2;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += 256ll * gridDim.x) {
        const int w4 = static_cast<int>(idx % W4);
        const int hp = static_cast<int>((idx / W4) % Hp);
        const int64_t b = idx / (static_cast<int64_t>(W4) * Hp);
        const int h = hp - pad;
        uint32_t* orow = reinterpret_cast<uint32_t*>(xp) + (b * Hp + hp) * Wp;
        uint32_t d[4] = {0u, 0u, 0u, 0u};
        if (h >= 0 && h < H) {
            uint32_t r[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < Cin) r[j] = *reinterpret_cast<const uint32_t*>(x + ((b * Cin + j) * H + h) * W + w4 * 4);
            transpose4x4_bytes(r, d);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) orow[w4 * 4 + i + pad] = d[i];
        if (w4 == 0)
            for (int i = 0; i < pad; ++i) orow[i] = 0u;
        if (w4 == W4 - 1)
            for (int i = W + pad; i < Wp; ++i) orow[i] = 0u;
    }
}

__device__ __forceinline__ uint32_t gather4_channels(const int8_t* __restrict__ w, int64_t at, int taps, int valid) {
    uint32_t v = 0u;   // w[at], w[at + taps], ...: the same tap of `valid` consecutive input channels
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < valid) v |= static_cast<uint32_t>(static_cast<uint8_t>(w[at + static_cast<int64_t>(j) * taps])) << (8 * j);
    return v;
}

// OIHW -> wp[Cout][KH][Lp] with (kw, c) order inside a filter row and zero fill.  One item per output dword:
// four consecutive input channels of one tap (Cp is a multiple of 4, so they share the tap).
__device__ __forceinline__ void pack_conv_weight_items(const int8_t* __restrict__ w, int8_t* __restrict__ wp,
                                                       uint32_t total_dwords, int Cin, int KH, int KW, int Cp, int L, int Lp,
                                                       uint32_t first, uint32_t step) {
    const uint32_t lp4 = Lp >> 2;
    for (uint32_t idx = first; idx < total_dwords; idx += step) {
        const uint32_t row = idx / lp4, l = (idx - row * lp4) * 4;   // row = n * KH + kh
        const uint32_t n = row / KH, kh = row - n * KH;
        uint32_t v = 0u;
        if (l < static_cast<uint32_t>(L)) {
            const uint32_t kw = l / Cp, c = l - kw * Cp;
            if (c < static_cast<uint32_t>(Cin))
                v = gather4_channels(w, ((static_cast<int64_t>(n) * Cin + c) * KH + kh) * KW + kw, KH * KW, Cin - static_cast<int>(c));
        }
        reinterpret_cast<uint32_t*>(wp)[idx] = v;
    }
}

// OIHW (3 x 3) -> wp[Cout / 64][Cin / 64][tap][chunk][channel % 64][16]: per 64-output-channel tile and
// 64-input-channel group one contiguous 36 KiB image of the LDS weight stage (zero past Cout).
__device__ __forceinline__ void pack_conv_weight_planar_items(const int8_t* __restrict__ w, int8_t* __restrict__ wp,
                                                              uint32_t total_dwords, int Cin, int Cout, uint32_t first,
                                                              uint32_t step) {
    const uint32_t ncg = Cin >> 6;
    for (uint32_t idx = first; idx < total_dwords; idx += step) {
        const uint32_t byte = (idx & 3) * 4, nl = (idx >> 2) & 63, kc = (idx >> 8) & 3;
        const uint32_t rest = idx >> 10;                   // (tn * ncg + cg) * 9 + tap
        const uint32_t tap = rest % 9, grp = rest / 9, cg = grp % ncg, n = (grp / ncg) * 64 + nl;
        const uint32_t c = cg * 64 + kc * 16 + byte;
        reinterpret_cast<uint32_t*>(wp)[idx] =
            n < static_cast<uint32_t>(Cout) ? gather4_channels(w, (static_cast<int64_t>(n) * Cin + c) * 9 + tap, 9, 4) : 0u;
    }
}

__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wp,
                                                               uint32_t total_dwords, int Cin, int KH, int KW, int Cp, int L,
                                                               int Lp) {
    pack_conv_weight_items(w, wp, total_dwords, Cin, KH, KW, Cp, L, Lp, blockIdx.x * 256u + threadIdx.x, 256u * gridDim.x);
}

__global__ __launch_bounds__(256) void pack_conv_weight_planar_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wp,
                                                                      uint32_t total_dwords, int Cin, int Cout) {
    pack_conv_weight_planar_items(w, wp, total_dwords, Cin, Cout, blockIdx.x * 256u + threadIdx.x, 256u * gridDim.x);
}

template <bool DWORD>
__global__ __launch_bounds__(256) void nhwc_interior_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ xp,
                                                            int Cin, int H, int W, int Hp, int Wp, int Cp, int pad,
                                                            int cblocks) {
    nhwc_interior_block<DWORD>(x, xp, Cin, H, W, Hp, Wp, Cp, pad, cblocks, blockIdx.x, blockIdx.y);
}

// Everything a convolution call does before its MFMA kernel, in one launch: workgroups
// [0, n_interior) of every image re-lay the interior (nhwc_interior_block); the n_extra workgroups
// after them share, across all images, the zero border of the padded copy and the weight re-layout.
struct ConvPrepArgs {
    const int8_t* x;
    int8_t* xp;
    const int8_t* w;
    int8_t* wp;
    int Cin, H, W, Hp, Wp, Cp, pad, cblocks;
    int KH, KW, L, Lp;
    int n_interior, n_extra;
    uint32_t border_total, weight_total;   // border pixels of all images; dwords of packed weights
    int64_t plane_stride;   // != 0: chunk-plane layout (Wp = row pitch), weights packed for conv3x3_patch_kernel
    int Cout;
    int sub, Ws, HWs;       // sub > 1: keep every sub-th pixel of every sub-th row of a Ws-wide source image
};

template <int MODE, int NT = 1>
__global__ __launch_bounds__(256) void conv_prep_kernel(ConvPrepArgs a) {
    if (static_cast<int>(blockIdx.x) < a.n_interior) {   // (n_interior = tiles of 64 * NT pixels x channel blocks)
        nhwc_interior_block<MODE, NT>(a.x, a.xp, a.Cin, a.H, a.W, a.Hp, a.Wp, a.Cp, a.pad, a.cblocks, blockIdx.x, blockIdx.y,
                                      a.plane_stride, a.sub, a.Ws, a.HWs);
        return;
    }
    const uint32_t first = (blockIdx.y * a.n_extra + (blockIdx.x - a.n_interior)) * 256u + threadIdx.x;
    const uint32_t step = gridDim.y * a.n_extra * 256u;
    nhwc_border_items(a.xp, a.H, a.W, a.Hp, a.Wp, a.Cp, a.pad, a.border_total, first, step, a.plane_stride);
    if (a.plane_stride)
        pack_conv_weight_planar_items(a.w, a.wp, a.weight_total, a.Cin, a.Cout, first, step);
    else
        pack_conv_weight_items(a.w, a.wp, a.weight_total, a.Cin, a.KH, a.KW, a.Cp, a.L, a.Lp, first, step);
}

// Tiles of 64 pixels a re-layout workgroup takes: two or four where the dword-load form applies and an image has the pixels.
inline int prep_tiles_per_workgroup(bool dword_loads, int64_t pixels) {
    const char* sw = dev_switch("MIQ_CONV_PREP_NT");   // developer A/B switch: tiles per workgroup (1, 2 or 4)
    if (sw && dword_loads && (sw[0] == '1' || sw[0] == '2' || sw[0] == '4')) return sw[0] - '0';
    // (whole call at batch 256, one / two / four tiles: 64->128 56x56 3x3/2 69.8 / 67.0 / 65.6 us, 128->128 28x28 61.1 / 59.1 /
    // 59.0, 128->256 28x28 3x3/2 42.8 / 41.2 / 40.8, 256->256 14x14 49.8 / 49.2 / 49.2)
    if (!dword_loads) return 1;
    return pixels >= 512 ? 4 : pixels >= 128 ? 2 : 1;
}

// four dwords that are only dword-aligned in memory (a global store of 16 bytes needs no more)
typedef int v4i_dw __attribute__((ext_vector_type(4), aligned(4)));
typedef float v4f_dw __attribute__((ext_vector_type(4), aligned(4)));

struct ConvArgs {
    const int8_t* xp;
    const int8_t* wp;
    const float* bias;
    const float* alpha;
    void* out;
    float beta;
    int M, Cout, Kp;          // GEMM dims
    int HoWo, Wo;             // pixel decomposition
    int img_stride;           // Hp*Wp*Cp, bytes per image in xp
    int row_stride;           // Wp*Cp
    int sx;                   // stride*Cp: bytes between horizontally adjacent output pixels' windows
    int sy;                   // stride*Wp*Cp
    int cps;                  // 16-byte chunks per filter row (Lp / 16)
    // A dilation along x (miq::conv2d_i8_dense): every filter tap is a "row" of its own, Cp bytes of one pixel; row t
    // then starts (t / kwt) * row_stride + (t % kwt) * tap_stride bytes into the window.  kwt = 0: rows are filter rows.
    int kwt, tap_stride;
    uint32_t kwt_magic;       // floor(2^32 / kwt) + 1
    uint32_t cps_magic;       // floor(2^32 / cps) + 1: q / cps == umulhi(q, cps_magic) for q < 2^16, cps > 1
    uint32_t xp_bytes, wp_bytes;
    int stream_out;           // 1: nontemporal 16-byte output stores (outputs far larger than the 256 MiB cache)
    int wide16;               // bf16 output: eight pixels per 16-byte store (HoWo % 8 == 0, 16-byte aligned output)
    int tiles_m, tiles_n;
    // channels-last output (CL kernels): element strides of out[b][y + border][x + border][channel]
    int o_img, o_row, o_pix, o_off;
};

// Byte offset of "row" t of a window's K axis (a filter row, or a single tap under a dilation along x).
__device__ __forceinline__ uint32_t row_offset(const ConvArgs& g, int t) {
    if (g.kwt == 0) return static_cast<uint32_t>(t * g.row_stride);
    const int kh = static_cast<int>(__umulhi(static_cast<uint32_t>(t), g.kwt_magic));
    return static_cast<uint32_t>(kh * g.row_stride + (t - kh * g.kwt) * g.tap_stride);
}

// Channels-last epilogue shared by the CL kernels: a lane holds four consecutive channels n .. n+3 (n % 4 == 0,
// Cout % 4 == 0) of output pixel m and stores them as one vector of the output type.
template <int OUT>
__device__ __forceinline__ void store_pixel_channels(const ConvArgs& g, int m, int n, const v4i& a, const float (&al)[4],
                                                     const float (&bi)[4]) {
    const int b = m / g.HoWo, p = m - b * g.HoWo;
    const int y = p / g.Wo, x = p - y * g.Wo;
    const size_t idx = static_cast<size_t>(b) * g.o_img + static_cast<size_t>(y) * g.o_row + x * g.o_pix + g.o_off + n;
    if constexpr (OUT == kOutI32) {
        *reinterpret_cast<v4i*>(static_cast<int*>(g.out) + idx) = a;
    } else {
        const float y0 = dequant1(a[0], bi[0], al[0], g.beta), y1 = dequant1(a[1], bi[1], al[1], g.beta),
                    y2 = dequant1(a[2], bi[2], al[2], g.beta), y3 = dequant1(a[3], bi[3], al[3], g.beta);
        if constexpr (OUT == kOutF32) {
            *reinterpret_cast<v4f*>(static_cast<float*>(g.out) + idx) = v4f{y0, y1, y2, y3};
        } else if constexpr (OUT == kOutBF16) {
            *reinterpret_cast<uint2*>(static_cast<uint16_t*>(g.out) + idx) = pack4_bf16(y0, y1, y2, y3);
        } else {
            *reinterpret_cast<uint32_t*>(static_cast<int8_t*>(g.out) + idx) = requant4_i8(y0, y1, y2, y3);
        }
    }
}

template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_rows(__amdgpu_buffer_rsrc_t rsrc, char* tile, int row0, int nrows, int K, int kt,
                                           int wave, int lane) {
    constexpr int kPieces = ROWS / 8;
    constexpr int kIters = (kPieces + NWAVES - 1) / NWAVES;
    const int r_in = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int i = 0; i < kIters; ++i) {
        const int piece = wave + i * NWAVES;
        if (kPieces % NWAVES != 0 && piece >= kPieces) break;
        const int r = piece * 8 + r_in;
        const int c = slot ^ ((r >> 1) & 7);
        const int k = kt * kTileK + c * kChunk;
        const int row = row0 + r;
        const uint32_t off =
            (row < nrows && k < K) ? static_cast<uint32_t>(row) * static_cast<uint32_t>(K) + k : kOob;
        dma16(rsrc, tile + piece * (8 * kTileK), off);
    }
}

// CL = channels-last output: the weight fragment is then the MFMA's first operand, so that a lane ends with four
// consecutive channels of one pixel (store_pixel_channels) instead of four consecutive pixels of one channel.
// STAGES = 2: double buffer, one __syncthreads per K-tile (which drains the DMA: the fetch of K-tile t + 1 overlaps only
// the MFMAs of K-tile t).  STAGES > 2: the GEMM's ring (gemm_i8_kernel): STAGES - 1 K-tiles in flight behind a counted
// vmcnt, one raw barrier per K-tile.
template <int BM, int BN, int WAVES_M, int WAVES_N, int OUT, bool STREAM, bool CL = false, int STAGES = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_i8_kernel(ConvArgs g) {
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int STAGE = (BM + BN) * kTileK;
    constexpr int kPiecesM = BM / 8;
    constexpr int kItersM = (kPiecesM + NWAVES - 1) / NWAVES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, 8, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int nchunks = g.Kp / kChunk;

    // Base offset in xp of the pixel each of this lane's staging rows belongs to.
    const int r_in = lane >> 3, slot = lane & 7;
    uint32_t pix[kItersM];
#pragma unroll
    for (int i = 0; i < kItersM; ++i) {
        const int m = m0 + (wave + i * NWAVES) * 8 + r_in;
        if (m < g.M) {
            const int b = m / g.HoWo, p = m % g.HoWo;
            pix[i] = static_cast<uint32_t>(b) * g.img_stride + (p / g.Wo) * g.sy + (p % g.Wo) * g.sx;
        } else {
            pix[i] = kOob;
        }
    }

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.xp, g.xp_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.wp, g.wp_bytes);

    v4i acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    auto stage_pixels = [&](char* tile, int kt) {
#pragma unroll
        for (int i = 0; i < kItersM; ++i) {
            const int piece = wave + i * NWAVES;
            if (kPiecesM % NWAVES != 0 && piece >= kPiecesM) break;
            const int r = piece * 8 + r_in;
            const int q = kt * kChunksPerRow + (slot ^ ((r >> 1) & 7));
            // chunk q of the K axis = filter row q / cps, byte (q % cps) * 16 of that row's run
            const int kh = g.cps == 1 ? q : static_cast<int>(__umulhi(static_cast<uint32_t>(q), g.cps_magic));
            uint32_t off = kOob;
            if (q < nchunks && pix[i] != kOob) off = pix[i] + row_offset(g, kh) + (q - kh * g.cps) * kChunk;
            dma16(rx, tile + piece * (8 * kTileK), off);
        }
    };

    const int nt = (g.Kp + kTileK - 1) / kTileK;
    const int lo0 = frag_lane_offset(lane, 0), lo1 = frag_lane_offset(lane, 1);

    auto stage = [&](char* buf, int kt) {   // (a K-tile past the end is requested past K: zero-filled, no traffic)
        stage_pixels(buf, kt);
        stage_rows<BN, NWAVES>(rw, buf + BM * kTileK, n0, g.Cout, g.Kp, kt, wave, lane);
    };
    auto mma_tile = [&](const char* cur) {
        const char* at = cur + wm * WM * kTileK;
        const char* wt = cur + BM * kTileK + wn * WN * kTileK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int lo = ks ? lo1 : lo0;
            v4i wf[FN], af[FM];
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[j] = lds_read16(wt + j * 16 * kTileK + lo);
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = lds_read16(at + i * 16 * kTileK + lo);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = CL ? __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], af[i], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], wf[j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (STAGES == 2) {
        stage(smem, 0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt) stage(smem + ((t + 1) & 1) * STAGE, t + 1);
            mma_tile(smem + (t & 1) * STAGE);
            __syncthreads();
        }
    } else {
        // every wave issues the same number of DMA instructions per K-tile, so one count serves all
        static_assert((BM / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0, "pieces must divide among the waves");
        constexpr int kLoads = (BM / 8 + BN / 8) / NWAVES;
        constexpr int kKeep = kLoads * (STAGES - 2);
        static_assert(kKeep <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
        for (int s0 = 0; s0 < STAGES - 1; ++s0) stage(smem + s0 * STAGE, s0);
        int slot_t = 0, fill = STAGES - 1;   // buffer of K-tile t; buffer K-tile t + STAGES - 1 goes into
        for (int t = 0; t < nt; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kKeep) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage(smem + fill * STAGE, t + STAGES - 1);
            mma_tile(smem + slot_t * STAGE);
            slot_t = slot_t + 1 == STAGES ? 0 : slot_t + 1;
            fill = fill + 1 == STAGES ? 0 : fill + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero fills requested for K-tiles past the end
        __syncthreads();                                    // every wave has read the last K-tile: an epilogue may reuse the LDS
    }

    if constexpr (CL) {   // D[r][c]: r = 4*(lane>>4) + reg -> channel, c = lane&15 -> pixel
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
            float al[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (OUT != kOutI32) {
                if (n < g.Cout) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        al[e] = g.alpha[n + e];
                        if (g.bias) bi[e] = g.bias[n + e];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * WM + i * 16 + (lane & 15);
                if (m < g.M && n < g.Cout) store_pixel_channels<OUT>(g, m, n, acc[i][j], al, bi);
            }
        }
        return;
    }

    // Epilogue.  D[i][j]: i = 4*(lane>>4) + reg -> pixel, j = lane&15 -> output channel.
    const int mbase = m0 + wm * WM + (lane >> 4) * 4;
    const int nbase = n0 + wn * WN + (lane & 15);
    const bool vec_ok = (g.HoWo & 3) == 0;
    if constexpr (STREAM && BM == 256 && OUT != kOutBF16) {
        // Streamed outputs leave through LDS, one channel's 256 pixels (1 KiB) per store instruction: a nontemporal
        // store that covers only half a cache line per channel (the accumulator as it lies: 16 channels x 64 bytes) is
        // written out as partial lines, and 822 MB of them take 250 us against 157 us for whole lines
        // (tools/nchw_write_probe.hip).  Sixteen channels per channel-wave at a time: rows of 256 + 4 values keep both the
        // 16-byte writes (sixteen rows at once) and the row-wise reads off each other's banks.
        if (vec_ok) {
            constexpr int kRowBytes = (BM + 4) * 4;
            constexpr int kPassRows = WAVES_N * 16;
            static_assert(kPassRows * kRowBytes <= 2 * STAGE && kPassRows % NWAVES == 0, "LDS of the transposed epilogue");
            const int ms = m0 + lane * 4;
            const bool row_ok = ms < g.M;
            size_t obase = 0;
            if (row_ok) obase = static_cast<size_t>(ms / g.HoWo) * g.Cout * g.HoWo + ms % g.HoWo;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = nbase + j * 16;
                float al = 1.f, bi = 0.f;
                if constexpr (OUT != kOutI32) {
                    if (n < g.Cout) {
                        al = g.alpha[n];
                        if (g.bias) bi = g.bias[n];
                    }
                }
                char* wr = smem + (wn * 16 + (lane & 15)) * kRowBytes + (wm * WM + (lane >> 4) * 4) * 4;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    v4i a = acc[i][j];
                    if constexpr (OUT == kOutF32)
                        a = v4i{__float_as_int(dequant1(a[0], bi, al, g.beta)), __float_as_int(dequant1(a[1], bi, al, g.beta)),
                                __float_as_int(dequant1(a[2], bi, al, g.beta)), __float_as_int(dequant1(a[3], bi, al, g.beta))};
                    *reinterpret_cast<v4i*>(wr + i * 64) = a;
                }
                __syncthreads();
#pragma unroll
                for (int r = wave; r < kPassRows; r += NWAVES) {
                    const int nn = n0 + (r >> 4) * WN + j * 16 + (r & 15);
                    const v4i v = *reinterpret_cast<const v4i*>(smem + r * kRowBytes + lane * 16);
                    if (row_ok && nn < g.Cout)
                        __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(static_cast<int*>(g.out) + obase + static_cast<size_t>(nn) * g.HoWo));
                }
                if (j + 1 < FN) __syncthreads();
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = nbase + j * 16;
        float al = 1.f, bi = 0.f;
        if constexpr (OUT != kOutI32) {
            if (n < g.Cout) {
                al = g.alpha[n];
                if (g.bias) bi = g.bias[n];
            }
        }
        if constexpr (OUT == kOutBF16 && FM % 2 == 0) {
            // Four bf16 pixels are only 8 bytes: two neighbouring pixel fragments trade 16-lane rows, and a
            // lane stores eight consecutive pixels of its channel as one 16-byte vector (a channel's 32
            // pixels of the pair leave as one 64-byte segment, in half the store instructions).
            if (g.wide16) {   // HoWo % 8 == 0 and a 16-byte aligned output: eight pixels share an image
#pragma unroll
                for (int i = 0; i < FM; i += 2) {
                    uint2 p[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const v4i a = acc[i + h][j];
                        p[h] = pack4_bf16(dequant1(a[0], bi, al, g.beta), dequant1(a[1], bi, al, g.beta),
                                          dequant1(a[2], bi, al, g.beta), dequant1(a[3], bi, al, g.beta));
                    }
                    const v4i v = pair_rows_bf16(p[0], p[1]);
                    const int m = m0 + wm * WM + i * 16 + paired_offset(lane);
                    if (m >= g.M || n >= g.Cout) continue;
                    const int b = m / g.HoWo, px = m % g.HoWo;
                    v4i* o = reinterpret_cast<v4i*>(static_cast<uint16_t*>(g.out) +
                                                    (static_cast<size_t>(b) * g.Cout + n) * g.HoWo + px);
                    if constexpr (STREAM) {
                        __builtin_nontemporal_store(v, o);
                    } else {
                        *o = v;
                    }
                }
                continue;
            }
        }
        if (n >= g.Cout) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mbase + i * 16;
            if (m >= g.M) continue;
            const v4i a = acc[i][j];
            if (vec_ok) {  // m % 4 == 0 and HoWo % 4 == 0: the four pixels share an image
                const int b = m / g.HoWo, p = m % g.HoWo;
                const size_t idx = (static_cast<size_t>(b) * g.Cout + n) * g.HoWo + p;
                if constexpr (OUT == kOutI32) {
                    v4i* o = reinterpret_cast<v4i*>(static_cast<int*>(g.out) + idx);
                    if constexpr (STREAM) {
                        __builtin_nontemporal_store(a, o);
                    } else {
                        *o = a;
                    }
                } else if constexpr (OUT == kOutF32) {
                    const v4f y{dequant1(a[0], bi, al, g.beta), dequant1(a[1], bi, al, g.beta),
                                dequant1(a[2], bi, al, g.beta), dequant1(a[3], bi, al, g.beta)};
                    v4f* o = reinterpret_cast<v4f*>(static_cast<float*>(g.out) + idx);
                    if constexpr (STREAM) {
                        __builtin_nontemporal_store(y, o);
                    } else {
                        *o = y;
                    }
                } else {
                    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(g.out) + idx) =
                        pack4_bf16(dequant1(a[0], bi, al, g.beta), dequant1(a[1], bi, al, g.beta),
                                   dequant1(a[2], bi, al, g.beta), dequant1(a[3], bi, al, g.beta));
                }
            } else {
                // Images whose pixel count is not a multiple of 4 (7 x 7): the four pixels are still consecutive in memory
                // when they share an image, only not 16-byte aligned: one vector store of dword alignment (global
                // stores need no more) instead of four scalars.
                const int b0 = m / g.HoWo, p0 = m - b0 * g.HoWo;
                if (OUT != kOutBF16 && m + 3 < g.M && p0 + 3 < g.HoWo) {
                    const size_t idx = (static_cast<size_t>(b0) * g.Cout + n) * g.HoWo + p0;
                    if constexpr (OUT == kOutI32) {
                        *reinterpret_cast<v4i_dw*>(static_cast<int*>(g.out) + idx) = a;
                    } else if constexpr (OUT == kOutF32) {
                        *reinterpret_cast<v4f_dw*>(static_cast<float*>(g.out) + idx) =
                            v4f{dequant1(a[0], bi, al, g.beta), dequant1(a[1], bi, al, g.beta), dequant1(a[2], bi, al, g.beta),
                                dequant1(a[3], bi, al, g.beta)};
                    }
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int me = m + e;
                    if (me >= g.M) break;
                    const int b = me / g.HoWo, p = me % g.HoWo;
                    const size_t idx = (static_cast<size_t>(b) * g.Cout + n) * g.HoWo + p;
                    if constexpr (OUT == kOutI32) {
                        static_cast<int*>(g.out)[idx] = a[e];
                    } else if constexpr (OUT == kOutF32) {
                        static_cast<float*>(g.out)[idx] = dequant1(a[e], bi, al, g.beta);
                    } else {
                        static_cast<uint16_t*>(g.out)[idx] = f32_to_bf16_bits(dequant1(a[e], bi, al, g.beta));
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// 256 pixels x 256 output channels per workgroup, software-pipelined: the int8 GEMM's four-phase schedule
// (quantization_amd/csrc/miq_gemm_i8.hip, gemm_i8_256_kernel) with the convolution's addressing on the pixel
// side.  Layers with at least 256 output channels and enough tiles to fill the chip run here: their K axis
// (KH * KW * Cin >= 1152) is long enough for a pipeline to pay, and the double-buffered kernel above waits a
// full LDS-DMA round trip per K-tile.
//
// 8 waves as 2 (pixels) x 4 (channels); a wave owns 128 x 64 outputs as four 64 x 32 quadrants.  A 128-byte
// K-tile lives in LDS as four 16 KiB half-tiles P0 P1 (pixels 0-127 / 128-255) and W0 W1 (channels likewise);
// two K-tile buffers.  Per K-tile four phases of 16 MFMAs, each LOAD (fragment reads, one half-tile of LDS-DMA
// for a later K-tile, a counted vmcnt) | barrier | MFMA | barrier, waves 4-7 one barrier behind waves 0-3:
//   P1: read W0(t)   | DMA W1(t+1) | MFMA (P0, W0)      P3: read P1(t)   | DMA P0(t+2) | MFMA (P1, W1)
//   P2: read W1(t)   | DMA P1(t+1) | MFMA (P0, W1)      P4: read P0(t+1) | DMA W0(t+2) | MFMA (P1, W0)
// vmcnt(8) after a phase's issue = "the half-tile issued four phases ago has landed"; a half-tile is read no
// earlier than one phase after that wait and re-staged three phases after its last read.
// Pixels are the MFMA's first operand: a lane ends with four consecutive pixels of one channel (NCHW stores).
// ---------------------------------------------------------------------------------------
#define MIQ_CONV_BARRIER()                  \
    do {                                    \
        asm volatile("" ::: "memory");      \
        __builtin_amdgcn_s_barrier();       \
        asm volatile("" ::: "memory");      \
    } while (0)
#define MIQ_CONV_WAIT_DMA() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

template <int OUT, bool STREAM, bool CL = false>
__global__ __launch_bounds__(512) void conv_i8_pipe_kernel(ConvArgs g) {
    constexpr int kHalf = 128 * kTileK;   // one half-tile: 128 rows x 128 B
    constexpr int kBuf = 4 * kHalf;       // P0 P1 W0 W1
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, 8, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    const int nchunks = g.Kp / kChunk;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.xp, g.xp_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.wp, g.wp_bytes);

    // Staging: a wave-instruction moves 8 rows x 128 B; a wave issues two per half-tile (row groups `wave` and
    // `wave + 8`).  The chunk slot a lane fills is fixed by the swizzle (the same for both of its rows).
    const int srow = wave * 8 + (lane >> 3);
    const int cslot = (lane & 7) ^ ((srow >> 1) & 7);
    uint32_t pix[2][2], wbase[2][2];   // [half][piece]: pixel base in xp / weight row offset at k = 0
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + h * 128 + srow + i * 64;
            if (m < g.M) {
                const int b = m / g.HoWo, p = m % g.HoWo;
                pix[h][i] = static_cast<uint32_t>(b) * g.img_stride + (p / g.Wo) * g.sy + (p % g.Wo) * g.sx;
            } else {
                pix[h][i] = kOob;
            }
            const int n = n0 + h * 128 + srow + i * 64;
            wbase[h][i] = n < g.Cout ? static_cast<uint32_t>(n) * g.Kp + cslot * kChunk : kOob;
        }
    auto stage_pix = [&](char* dst, const uint32_t (&base)[2], int kt) {
        // chunk q of the K axis = filter row q / cps, byte (q % cps) * 16 of that row's run
        const int q = kt * kChunksPerRow + cslot;
        const int kh = g.cps == 1 ? q : static_cast<int>(__umulhi(static_cast<uint32_t>(q), g.cps_magic));
        const uint32_t koff = row_offset(g, kh) + static_cast<uint32_t>((q - kh * g.cps) * kChunk);
        const bool ok = q < nchunks;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t off = (ok && base[i] != kOob) ? base[i] + koff : kOob;
            dma16(rx, dst + (wave + 8 * i) * (8 * kTileK), off);
        }
    };
    auto stage_w = [&](char* dst, const uint32_t (&base)[2], int kt) {
        const int k = kt * kTileK;
        const bool ok = k + cslot * kChunk < g.Kp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t off = (ok && base[i] != kOob) ? base[i] + k : kOob;
            dma16(rw, dst + (wave + 8 * i) * (8 * kTileK), off);
        }
    };

    v4i acc[2][2][4][2];   // [pixel half][channel half][pixel fragment][channel fragment]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = v4i{0, 0, 0, 0};

    const int nt = (g.Kp + kTileK - 1) / kTileK;
    const int lo[2] = {frag_lane_offset(lane, 0), frag_lane_offset(lane, 1)};
    const int p_frag = wm * 64 * kTileK;   // this wave's rows inside a pixel half-tile
    const int w_frag = wn * 32 * kTileK;   // and inside a weight half-tile

    // Prologue: all of tile 0 plus P0/W0 of tile 1; wait for P0(0), W0(0) only.
    stage_pix(smem + 0 * kHalf, pix[0], 0);
    stage_w(smem + 2 * kHalf, wbase[0], 0);
    stage_w(smem + 3 * kHalf, wbase[1], 0);
    stage_pix(smem + 1 * kHalf, pix[1], 0);
    stage_pix(smem + kBuf + 0 * kHalf, pix[0], 1);
    stage_w(smem + kBuf + 2 * kHalf, wbase[0], 1);
    MIQ_CONV_WAIT_DMA();
    MIQ_CONV_BARRIER();
    if (wm == 1) MIQ_CONV_BARRIER();   // waves 4-7 run one barrier behind

    v4i pf[2][4] = {}, pf2[2][4] = {}, wf0[2][2] = {}, wf1[2][2] = {};   // [k-step][fragment]
    auto read_w = [&](v4i (&wf)[2][2], const char* half) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[ks][j] = lds_read16(half + w_frag + j * 16 * kTileK + lo[ks]);
    };
    auto read_p = [&](v4i (&f)[2][4], const char* half) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) f[ks][i] = lds_read16(half + p_frag + i * 16 * kTileK + lo[ks]);
    };
    auto mma = [&](v4i (&c)[4][2], const v4i (&f)[2][4], const v4i (&wf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    c[i][j] = CL ? __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[ks][j], f[ks][i], c[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_i32_16x16x64_i8(f[ks][i], wf[ks][j], c[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    read_p(pf, smem + 0 * kHalf);   // P0(0)
    for (int t = 0; t < nt; ++t) {
        char* cur = smem + (t & 1) * kBuf;
        char* nxt = smem + ((t + 1) & 1) * kBuf;

        read_w(wf0, cur + 2 * kHalf);
        stage_w(nxt + 3 * kHalf, wbase[1], t + 1);
        MIQ_CONV_WAIT_DMA();   // W1(t) has landed
        MIQ_CONV_BARRIER();
        mma(acc[0][0], pf, wf0);
        MIQ_CONV_BARRIER();

        read_w(wf1, cur + 3 * kHalf);
        stage_pix(nxt + 1 * kHalf, pix[1], t + 1);
        MIQ_CONV_WAIT_DMA();   // P1(t)
        MIQ_CONV_BARRIER();
        mma(acc[0][1], pf, wf1);
        MIQ_CONV_BARRIER();

        read_p(pf2, cur + 1 * kHalf);
        stage_pix(cur + 0 * kHalf, pix[0], t + 2);
        MIQ_CONV_WAIT_DMA();   // P0(t+1)
        MIQ_CONV_BARRIER();
        mma(acc[1][1], pf2, wf1);
        MIQ_CONV_BARRIER();

        read_p(pf, nxt + 0 * kHalf);   // P0(t+1): retired by the wait one phase ago
        stage_w(cur + 2 * kHalf, wbase[0], t + 2);
        MIQ_CONV_WAIT_DMA();   // W0(t+1)
        MIQ_CONV_BARRIER();
        mma(acc[1][0], pf2, wf0);
        MIQ_CONV_BARRIER();
    }
    if (wm == 0) MIQ_CONV_BARRIER();   // every wave executes the same number of barriers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs issued for tiles past K

    if constexpr (CL) {   // D[r][c] of a fragment: r = 4*(lane>>4) + reg -> channel, c = lane&15 -> pixel
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + qn * 128 + wn * 32 + j * 16 + (lane >> 4) * 4;
                float al[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (OUT != kOutI32) {
                    if (n < g.Cout) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            al[e] = g.alpha[n + e];
                            if (g.bias) bi[e] = g.bias[n + e];
                        }
                    }
                }
#pragma unroll
                for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = m0 + qm * 128 + wm * 64 + i * 16 + (lane & 15);
                        if (m < g.M && n < g.Cout) store_pixel_channels<OUT>(g, m, n, acc[qm][qn][i][j], al, bi);
                    }
            }
        return;
    }

    // Epilogue.  D[r][c] of a fragment: r = 4*(lane>>4) + reg -> pixel, c = lane&15 -> output channel.
    const bool vec_ok = (g.HoWo & 3) == 0;
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + qn * 128 + wn * 32 + j * 16 + (lane & 15);
            float al = 1.f, bi = 0.f;
            if constexpr (OUT != kOutI32) {
                if (n < g.Cout) {
                    al = g.alpha[n];
                    if (g.bias) bi = g.bias[n];
                }
            }
#pragma unroll
            for (int qm = 0; qm < 2; ++qm) {
                const int mq = m0 + qm * 128 + wm * 64;
                if constexpr (OUT == kOutBF16) {
                    if (g.wide16) {   // HoWo % 8 == 0 and a 16-byte aligned output: eight pixels share an image
#pragma unroll
                        for (int i = 0; i < 4; i += 2) {
                            uint2 p[2];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const v4i a = acc[qm][qn][i + h][j];
                                p[h] = pack4_bf16(dequant1(a[0], bi, al, g.beta), dequant1(a[1], bi, al, g.beta),
                                                  dequant1(a[2], bi, al, g.beta), dequant1(a[3], bi, al, g.beta));
                            }
                            const v4i v = pair_rows_bf16(p[0], p[1]);
                            const int m = mq + i * 16 + paired_offset(lane);
                            if (m >= g.M || n >= g.Cout) continue;
                            const int b = m / g.HoWo, px = m % g.HoWo;
                            v4i* o = reinterpret_cast<v4i*>(static_cast<uint16_t*>(g.out) +
                                                            (static_cast<size_t>(b) * g.Cout + n) * g.HoWo + px);
                            if constexpr (STREAM) {
                                __builtin_nontemporal_store(v, o);
                            } else {
                                *o = v;
                            }
                        }
                        continue;
                    }
                }
                if (n >= g.Cout) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = mq + i * 16 + (lane >> 4) * 4;
                    if (m >= g.M) continue;
                    const v4i a = acc[qm][qn][i][j];
                    if (vec_ok) {   // m % 4 == 0 and HoWo % 4 == 0: the four pixels share an image
                        const int b = m / g.HoWo, p = m % g.HoWo;
                        const size_t idx = (static_cast<size_t>(b) * g.Cout + n) * g.HoWo + p;
                        if constexpr (OUT == kOutI32) {
                            v4i* o = reinterpret_cast<v4i*>(static_cast<int*>(g.out) + idx);
                            if constexpr (STREAM) {
                                __builtin_nontemporal_store(a, o);
                            } else {
                                *o = a;
                            }
                        } else if constexpr (OUT == kOutF32) {
                            const v4f y{dequant1(a[0], bi, al, g.beta), dequant1(a[1], bi, al, g.beta),
                                        dequant1(a[2], bi, al, g.beta), dequant1(a[3], bi, al, g.beta)};
                            v4f* o = reinterpret_cast<v4f*>(static_cast<float*>(g.out) + idx);
                            if constexpr (STREAM) {
                                __builtin_nontemporal_store(y, o);
                            } else {
                                *o = y;
                            }
                        } else {
                            *reinterpret_cast<uint2*>(static_cast<uint16_t*>(g.out) + idx) =
                                pack4_bf16(dequant1(a[0], bi, al, g.beta), dequant1(a[1], bi, al, g.beta),
                                           dequant1(a[2], bi, al, g.beta), dequant1(a[3], bi, al, g.beta));
                        }
                    } else {
                        // (as in conv_i8_kernel: a dword-aligned vector store where the four pixels share an image)
                        const int b0 = m / g.HoWo, p0 = m - b0 * g.HoWo;
                        if (OUT != kOutBF16 && m + 3 < g.M && p0 + 3 < g.HoWo) {
                            const size_t idx = (static_cast<size_t>(b0) * g.Cout + n) * g.HoWo + p0;
                            if constexpr (OUT == kOutI32) {
                                *reinterpret_cast<v4i_dw*>(static_cast<int*>(g.out) + idx) = a;
                            } else if constexpr (OUT == kOutF32) {
                                *reinterpret_cast<v4f_dw*>(static_cast<float*>(g.out) + idx) =
                                    v4f{dequant1(a[0], bi, al, g.beta), dequant1(a[1], bi, al, g.beta),
                                        dequant1(a[2], bi, al, g.beta), dequant1(a[3], bi, al, g.beta)};
                            }
                            continue;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int me = m + e;
                            if (me >= g.M) break;
                            const int b = me / g.HoWo, p = me % g.HoWo;
                            const size_t idx = (static_cast<size_t>(b) * g.Cout + n) * g.HoWo + p;
                            if constexpr (OUT == kOutI32) {
                                static_cast<int*>(g.out)[idx] = a[e];
                            } else if constexpr (OUT == kOutF32) {
                                static_cast<float*>(g.out)[idx] = dequant1(a[e], bi, al, g.beta);
                            } else {
                                static_cast<uint16_t*>(g.out)[idx] = f32_to_bf16_bits(dequant1(a[e], bi, al, g.beta));
                            }
                        }
                    }
                }
            }
        }
}

// ---------------------------------------------------------------------------------------
// 3 x 3, stride 1: the convolution as nine shifted GEMMs over a pixel patch that stays in LDS.
//
// The padded copy is kept in chunk planes, xp[c / 16][image][padded row][padded column][c % 16], rows at a
// pitch that is a multiple of 4 pixels.  A tile is kTM consecutive output pixels of one image (flat y * Wo +
// x, so output stores are the same aligned 64-byte runs as the im2col-order kernel's); its patch is every
// padded row those pixels' windows touch: per chunk plane ONE contiguous run in memory, copied by LDS-DMA as
// it lies.  The output pixel (y, x) reads, for filter tap (kh, kw), patch pixel (y - y0 + kh) * pitch + x +
// kw: per lane a base computed once per tile plus a shift that is the same for all lanes.  Every input byte
// is staged once per 64-channel tile instead of nine times, and a workgroup waits for one fetch per tile
// instead of one per 128-byte K-tile.  In LDS a fragment read takes plane lane>>4, pixel base(lane&15):
// consecutive 16-byte slots per plane (a step of pitch - Wo where the sixteen pixels cross a row end),
// planes a multiple of 256 bytes apart.  The weights of the 64-channel group sit beside the patch as
// [tap][chunk][channel].
//
// Two persistent workgroups per CU (80 KiB of LDS, 128 registers) walk (image, tile, channel tile) units: while
// one waits for its stage the other multiplies.  A unit's output stores are issued behind the next
// stage's fetch and are not waited for (counted vmcnt), so they drain under the following MFMAs.
// ---------------------------------------------------------------------------------------
struct PlanarArgs {
    const int8_t* xp;
    const int8_t* wp;
    const float* bias;
    const float* alpha;
    void* out;
    float beta;
    int Cout, ncg, tiles_n, tiles_img, units;
    int pitch, Hp, Wo, HoWo;
    uint32_t ptot;                // flat padded pixels of one chunk plane
    uint32_t wo_magic;            // floor(2^32 / Wo) + 1
    uint32_t xp_bytes, wp_bytes;
    int wide16;                   // bf16 output: paired 16-byte stores (HoWo % 8 == 0, 16-byte aligned output)
    int ragged;                   // NCHW output with HoWo % 4 != 0 (7 x 7 images): a lane's four pixels are stored one by one
    // channels-last form (CL kernels): xp is the caller's [image][padded row][padded column][cin] activation and
    // out[b][y + border][x + border][channel] is addressed with these element strides
    int cin;
    int o_img, o_row, o_pix, o_off;
    unsigned long long* stamps;   // developer library only: [workgroup][stage 0..7][8] shader-clock stamps, or null
    // small images: a unit is `group` consecutive whole images (tiles_img == 1); 1 = a unit is a tile of one image
    int group, B;
    int img_slots;                // Hp * pitch: patch slots of one image
    int pix_end;                  // group * HoWo: pixels of a unit that exist
    uint32_t hw_magic;            // floor(2^32 / HoWo) + 1
    // NCHW activation staged directly (DIRECT kernels): x[B][cin][H][W], W % 4 == 0
    const int8_t* x;
    uint32_t x_bytes;
    int H, W, pad;
    int rows;                     // patch rows staged per unit (the most a tile touches)
    int part_rows;                // ceil(rows / 12): rows one wave re-lays per piece
    int w4;                       // W / 4
    int cpc;                      // 16-byte chunks that cover a run of part_rows * W bytes starting anywhere in a chunk
    uint32_t w4_magic, cpc_magic; // floor(2^32 / d) + 1 for d = w4 and d = cpc
};

// DIRECT kernels: slots per chunk plane of the LDS patch, and the per-wave staging area for raw NCHW bytes
constexpr int kDirectSlots = 592;
constexpr int kDirectStageBytes = 1792;   // 16 channels x cpc chunks of one wave (cpc <= 7)

// developer library: shader-clock stamps of the first eight stages of every workgroup (wave 0, lane 0)
#ifdef MIQ_NO_DEV_SWITCHES
#define MIQ_PLANAR_STAMP(k)
#else
#define MIQ_PLANAR_STAMP(k)                                                                         \
    if (g.stamps && threadIdx.x == 0 && stamp_it < 8)                                               \
        g.stamps[(static_cast<size_t>(blockIdx.x) * 8 + stamp_it) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#endif

// One persistent workgroup per CU (eight waves, 256 registers each, all of the CU's LDS) walks (image, tile, channel
// tile) units (small images: a unit is `group` consecutive whole images, whose padded copies are one run in memory), with the phases of a stage overlapped in software:
//   * LDS holds two stages (patch + weights): the fetch of stage s + 1 is issued piece by piece among the filter taps
//     of stage s and lands under its MFMAs;
//   * the fragments of filter tap t + 1 are read from LDS while the MFMAs of tap t run;
//   * a finished unit's accumulators move to a second register set and leave, one fragment row per filter tap, among
//     the MFMAs of the next unit's first stage: their conversion is VALU work beside the other wave's matrix
//     instructions and their stores trickle out instead of queueing in one burst in front of the next fetch.
// One barrier per stage.  A workgroup keeps one channel tile (the grid is a multiple of tiles_n), so its per-channel
// epilogue parameters are loaded once, and with a single input-channel group its weights are too.
// (Measured against two workgroups per CU at 128 registers, whose waves spent 12,500 cycles of a stage on fetch issue,
// epilogue and barrier skew and 5,800 in the MFMA loop: 2-20 % faster for every output type and layout.)
//
// CL = channels-last activations in and out.  The patch is then gathered from pixel-major memory (a lane's 16 bytes are
// chunk `plane` of its pixel, cin bytes apart from its neighbour's) into the same chunk planes, the weight fragment is
// the MFMA's first operand, and its rows are read in an order that leaves a lane with eight consecutive output channels
// of one pixel: one 8-byte store per fragment row for an int8 output.
//
// DIRECT = the NCHW activation itself is the source (no re-layout launch, no padded copy; W % 4 == 0).  The raw bytes of
// a stage reach LDS by LDS-DMA in six pieces issued among taps 0-5: in a piece a wave fetches 16 channels (one chunk
// plane) x a twelfth of the patch rows into a staging area of its own, and in the following tap it reads that back
// four channels x four pixels at a time, transposes the 4 x 4 bytes in registers and writes each pixel's four channel
// bytes into the patch (chunk planes with a border that is zeroed once: interior writes never touch it; rows outside the
// image arrive as zeros from out-of-range DMA offsets).  A wave consumes only what it fetched itself, behind a counted
// vmcnt that leaves the younger output stores in flight, so no barrier is added.
//
// RAGGED = NCHW output whose images' pixel count is no multiple of four (7 x 7): a lane's four pixels are not one aligned
// 16-byte run (they may even lie in two images of a small-image unit) and leave one by one.  A kernel of its own: as a
// run-time branch of the others its code costs the bf16 form 40 spilled registers.
template <int OUT, bool STREAM, bool CL, bool DIRECT = false, bool RAGGED = false>
__global__ __launch_bounds__(512, 2) void conv3x3_patch_kernel(PlanarArgs g) {
    static_assert(!(CL && DIRECT), "DIRECT stages an NCHW activation");
    static_assert(!RAGGED || (!CL && !DIRECT && !STREAM && OUT != kOutI8), "RAGGED: NCHW outputs from the padded copy");
    constexpr int FM = 7, FN = 2;
    constexpr int kSlots = DIRECT ? kDirectSlots : kPlaneSlots;   // slots per chunk plane
    constexpr int kPatch = 4 * kSlots * kChunk;                   // bytes of one patch buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2] patches, [2] weight stages, (DIRECT) [8] staging areas
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.xp, g.xp_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.wp, g.wp_bytes);
    // A workgroup keeps one channel tile, so with at most two input-channel groups its filters stay where the first unit
    // put them (group cg in weight stage cg: the two stages are then not a double buffer) and later stages fetch patches only.
    const bool w_res = g.ncg <= 2;

    struct Unit {
        uint32_t b, p0, y0, tn;
    };
    auto decode = [&](int unit) {
        Unit u;
        u.tn = static_cast<uint32_t>(unit % g.tiles_n);
        const uint32_t t = static_cast<uint32_t>(unit / g.tiles_n);
        const uint32_t bi = t / g.tiles_img;
        u.b = bi * g.group;   // the unit's (first) image
        u.p0 = (t - bi * g.tiles_img) * kTM;
        u.y0 = u.p0 / g.Wo;
        return u;
    };
    // One stage's fetch is 44 wave-instructions of patch and 36 of weights, dealt over the eight waves: piece k of a wave
    // is patch instruction wave + 8k (k < 6) and weight instruction wave + 8k (k < 5).  Returns whether it issued any.
    auto issue_piece = [&](const Unit& u, int cg, int buf, bool weights, int k) -> bool {
        bool any = false;
        const int q = wave + 8 * k;
        if (!DIRECT && q < 44) {
            const uint32_t first = (u.b * g.Hp + u.y0) * g.pitch;
            // channels-last: the four planes of a 64-pixel segment are read from the same cache lines, so they go out
            // together (from four waves at once): issued plane by plane, a line has left the L1 before its next use
            const int plane = CL ? (q & 3) : q / 11, seg = CL ? (q >> 2) : q - plane * 11;
            uint32_t off;
            if constexpr (CL)
                off = (first + seg * 64 + lane) * g.cin + (cg * 4 + plane) * kChunk;
            else
                off = ((static_cast<uint32_t>(cg) * 4 + plane) * g.ptot + first + seg * 64 + lane) * kChunk;
            dma16(rx, smem + buf * kPatch + plane * (kSlots * kChunk) + seg * 1024, off);
            any = true;
        }
        if (weights && q < 36) {
            dma16(rw, smem + 2 * kPatch + (w_res ? cg : buf) * kWStageBytes + q * 1024,
                  ((u.tn * g.ncg + cg) * 36 + q) * 1024 + lane * kChunk);
            any = true;
        }
        return any;
    };

    // DIRECT: piece p (0..5) of a stage.  This wave's share: chunk plane wave & 3 (16 channels), patch rows
    // [part * part_rows, +part_rows) with part = 2p + (wave >> 2).  Per channel those rows are one run of part_rows * W
    // bytes in memory; it is fetched as the 16-byte chunks that cover it (the run starts `delta` bytes into the first).
    char* const staging = smem + 2 * kPatch + 2 * kWStageBytes + wave * kDirectStageBytes;
    const int cpc = g.cpc;   // 16-byte chunks fetched per channel
    auto direct_fetch = [&](const Unit& u, int cg, int p, char* staging) {
        const int part = 2 * p + (wave >> 2);
        const int y_img = static_cast<int>(u.y0) - g.pad + part * g.part_rows;   // image row of the part's first row (may lie outside)
        const int chan0 = static_cast<int>(u.b) * g.cin + cg * 64 + (wave & 3) * 16;
#pragma unroll 1
        for (int it = 0; it * 64 < 16 * cpc; ++it) {
            const uint32_t e = it * 64 + lane;
            const uint32_t ch = __umulhi(e, g.cpc_magic), k = e - ch * cpc;
            const int start = ((chan0 + static_cast<int>(ch)) * g.H + y_img) * g.W;   // byte offset of the run (negative above image 0)
            // a chunk outside the buffer reads zeros; whatever else it holds beside the run is never used.  Lanes past the
            // last chunk are switched off (an out-of-range offset would still write 16 zero bytes, into the next wave's area).
            if (e < static_cast<uint32_t>(16 * cpc))
                dma16(rx, staging + it * 1024, static_cast<uint32_t>((start & ~15) + static_cast<int>(k) * 16));
        }
    };
    auto direct_relay = [&](const Unit& u, int cg, int buf, int p, const char* staging) {
        const int part = 2 * p + (wave >> 2);
        const int row0 = part * g.part_rows;
        const int y_img = static_cast<int>(u.y0) - g.pad + row0;
        const int cq = lane & 3;
        const uint32_t pq = lane >> 2;   // pixel quad inside the part (at most 16 of them)
        const uint32_t row = __umulhi(pq, g.w4_magic), xq = pq - row * g.w4;
        if (pq < static_cast<uint32_t>(g.part_rows * g.w4) && row0 + static_cast<int>(row) < g.rows) {
            const int chan0 = static_cast<int>(u.b) * g.cin + cg * 64 + (wave & 3) * 16 + cq * 4;
            const bool inside = y_img + static_cast<int>(row) >= 0 && y_img + static_cast<int>(row) < g.H;
            uint32_t r[4] = {0u, 0u, 0u, 0u}, d[4];
            if (inside) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int delta = (((chan0 + i) * g.H + y_img) * g.W) & 15;
                    r[i] = *reinterpret_cast<const uint32_t*>(staging + (cq * 4 + i) * cpc * 16 + delta + pq * 4);
                }
            }
            transpose4x4_bytes(r, d);
            char* o = smem + buf * kPatch + (wave & 3) * (kSlots * kChunk) + cq * 4 +
                      ((row0 + row) * g.pitch + g.pad + xq * 4) * kChunk;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint32_t*>(o + i * kChunk) = d[i];
        }
    };

    // per-channel epilogue parameters of this workgroup's channel tile, once
    const int tn_mine = static_cast<int>(blockIdx.x) % g.tiles_n;
    constexpr int kParams = CL ? 8 : FN;
    float al[kParams], bi[kParams];
#pragma unroll
    for (int c = 0; c < kParams; ++c) {
        al[c] = 1.f;
        bi[c] = 0.f;
    }
    if constexpr (OUT != kOutI32) {
#pragma unroll
        for (int c = 0; c < kParams; ++c) {
            const int n = CL ? tn_mine * 64 + wn * 32 + (lane >> 4) * 8 + c : tn_mine * 64 + wn * 32 + c * 16 + (lane & 15);
            if (n < g.Cout) {
                al[c] = g.alpha[n];
                if (g.bias) bi[c] = g.bias[n];
            }
        }
    }

    v4i acc[FM][FN], done[FM][FN];   // the unit being accumulated; the finished one on its way out
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = done[i][j] = v4i{0, 0, 0, 0};

    // One fragment row (or, bf16 NCHW with 16-byte stores, one pair of rows) of the finished unit, as buffer stores
    // into the unit's image: a lane with nothing to store (pixels past the image, channels past Cout) gets an offset
    // past the buffer, which the hardware drops.  No branch, so the conversion shares a basic block with the tap's MFMAs
    // and is scheduled among them, and the number of store instructions is a constant (what the counted wait needs).
    constexpr int kEsz = OUT == kOutI8 ? 1 : OUT == kOutBF16 ? 2 : 4;
    constexpr int kNt = (STREAM && !CL) ? 2 : 0;   // cache policy of the output stores: 2 = nontemporal
    auto put16 = [&](__amdgpu_buffer_rsrc_t r, uint32_t off, v4i v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, kNt); };
    auto put8 = [&](__amdgpu_buffer_rsrc_t r, uint32_t off, uint2 v) {
        __builtin_amdgcn_raw_buffer_store_b64(v2u{v.x, v.y}, r, off, 0, kNt);
    };
    auto store_row_cl = [&](__amdgpu_buffer_rsrc_t r, const Unit& u, auto I) -> int {
        constexpr int i = decltype(I)::value;
        const uint32_t p = u.p0 + wm * FM * 16 + i * 16 + (lane & 15);
        const int n0 = u.tn * 64 + wn * 32 + (lane >> 4) * 8;
        v4i a0 = done[i][0], a1 = done[i][1];
        // (opaque: the conversion below then stays at this point of the instruction stream, among the MFMAs of the tap
        // it was written beside, instead of being computed for the whole unit ahead of the first tap)
        asm volatile("" : "+v"(a0), "+v"(a1));
        const uint32_t im = __umulhi(p, g.hw_magic), pp = p - im * g.HoWo;   // image of the unit, pixel inside it
        const uint32_t y = __umulhi(pp, g.wo_magic), x = pp - y * g.Wo;
        const uint32_t off = p < static_cast<uint32_t>(g.pix_end) ? (im * g.o_img + y * g.o_row + x * g.o_pix + g.o_off + n0) * kEsz : kOob;
        if constexpr (OUT == kOutI32) {
            put16(r, off, a0);
            put16(r, off + 16, a1);
            return 2;
        } else {
            float v[8];
            const v2f beta2{g.beta, g.beta};
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                const v2f lo = dequant2(a0[c], a0[c + 1], v2f{bi[c], bi[c + 1]}, v2f{al[c], al[c + 1]}, beta2);
                const v2f hi = dequant2(a1[c], a1[c + 1], v2f{bi[c + 4], bi[c + 5]}, v2f{al[c + 4], al[c + 5]}, beta2);
                v[c] = lo[0];
                v[c + 1] = lo[1];
                v[c + 4] = hi[0];
                v[c + 5] = hi[1];
            }
            if constexpr (OUT == kOutF32) {
                put16(r, off, v4i{__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])});
                put16(r, off + 16, v4i{__float_as_int(v[4]), __float_as_int(v[5]), __float_as_int(v[6]), __float_as_int(v[7])});
                return 2;
            } else if constexpr (OUT == kOutBF16) {
                const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
                put16(r, off, v4i{(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y});
                return 1;
            } else {
                put8(r, off, uint2{requant4_i8(v[0], v[1], v[2], v[3]), requant4_i8(v[4], v[5], v[6], v[7])});
                return 1;
            }
        }
    };
    auto store_row_nchw = [&](__amdgpu_buffer_rsrc_t r, const Unit& u, auto I) -> int {
        constexpr int i = decltype(I)::value;
        const uint32_t p = u.p0 + wm * FM * 16 + i * 16 + (lane >> 4) * 4;   // HoWo % 4 == 0: four pixels in or out together
        const uint32_t im = __umulhi(p, g.hw_magic), pp = p - im * g.HoWo;   // image of the unit, pixel inside it
        const uint32_t at = im * g.Cout * g.HoWo + pp;
        if constexpr (RAGGED) {
            {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = u.tn * 64 + wn * 32 + j * 16 + (lane & 15);
                    v4i a = done[i][j];
                    asm volatile("" : "+v"(a));   // (opaque, as above)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t pe = p + e, ime = __umulhi(pe, g.hw_magic), ppe = pe - ime * g.HoWo;
                        const uint32_t off = pe < static_cast<uint32_t>(g.pix_end) && n < g.Cout ? ((ime * g.Cout + n) * g.HoWo + ppe) * kEsz : kOob;
                        if constexpr (OUT == kOutI32) {
                            __builtin_amdgcn_raw_buffer_store_b32(a[e], r, off, 0, kNt);
                        } else {
                            const float y = dequant1(a[e], bi[j], al[j], g.beta);
                            if constexpr (OUT == kOutF32)
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(y), r, off, 0, kNt);
                            else
                                __builtin_amdgcn_raw_buffer_store_b16(static_cast<short>(f32_to_bf16_bits(y)), r, off, 0, kNt);
                        }
                    }
                }
                return 4 * FN;
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = u.tn * 64 + wn * 32 + j * 16 + (lane & 15);
            v4i a = done[i][j];
            asm volatile("" : "+v"(a));   // (opaque, as above)
            const uint32_t off = p < static_cast<uint32_t>(g.pix_end) && n < g.Cout ? (n * g.HoWo + at) * kEsz : kOob;
            if constexpr (OUT == kOutI32) {
                put16(r, off, a);
            } else {
                const v2f b2{bi[j], bi[j]}, a2{al[j], al[j]}, beta2{g.beta, g.beta};
                const v2f lo = dequant2(a[0], a[1], b2, a2, beta2), hi = dequant2(a[2], a[3], b2, a2, beta2);
                const float y0 = lo[0], y1 = lo[1], y2 = hi[0], y3 = hi[1];
                if constexpr (OUT == kOutF32)
                    put16(r, off, v4i{__float_as_int(y0), __float_as_int(y1), __float_as_int(y2), __float_as_int(y3)});
                else
                    put8(r, off, pack4_bf16(y0, y1, y2, y3));
            }
        }
        return FN;
    };
    // bf16 NCHW, 16-byte stores: fragment rows i - 1 and i trade 16-lane rows, a lane stores eight pixels of its channel
    auto store_pair_nchw_bf16 = [&](__amdgpu_buffer_rsrc_t r, const Unit& u, auto I) -> int {
        constexpr int i = decltype(I)::value;   // the second row of the pair
        const uint32_t p = u.p0 + wm * FM * 16 + (i - 1) * 16 + paired_offset(lane);   // HoWo % 8 == 0
        const uint32_t im = __umulhi(p, g.hw_magic), pp = p - im * g.HoWo;
        const uint32_t at = im * g.Cout * g.HoWo + pp;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = u.tn * 64 + wn * 32 + j * 16 + (lane & 15);
            uint2 h[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v4i a = done[i - 1 + e][j];
                asm volatile("" : "+v"(a));   // (opaque, as above)
                const v2f b2{bi[j], bi[j]}, a2{al[j], al[j]}, beta2{g.beta, g.beta};
                const v2f lo = dequant2(a[0], a[1], b2, a2, beta2), hi = dequant2(a[2], a[3], b2, a2, beta2);
                h[e] = pack4_bf16(lo[0], lo[1], hi[0], hi[1]);
            }
            const v4i v = pair_rows_bf16(h[0], h[1]);
            put16(r, p < static_cast<uint32_t>(g.pix_end) && n < g.Cout ? (n * g.HoWo + at) * kEsz : kOob, v);
        }
        return FN;
    };
    // what leaves beside filter tap T of the next unit's first stage
    auto store_step = [&](__amdgpu_buffer_rsrc_t r, const Unit& u, auto T) -> int {
        constexpr int t = decltype(T)::value;
        if constexpr (t >= FM) {
            return 0;
        } else if constexpr (CL) {
            return store_row_cl(r, u, T);
        } else if constexpr (RAGGED) {
            return store_row_nchw(r, u, T);
        } else if constexpr (OUT == kOutBF16) {
            if (g.wide16) {
                if constexpr (t % 2 == 1)
                    return store_pair_nchw_bf16(r, u, T);
                else if constexpr (t == FM - 1)
                    return store_row_nchw(r, u, T);
                else
                    return 0;
            }
            return store_row_nchw(r, u, T);
        } else {
            return store_row_nchw(r, u, T);
        }
    };
    // the finished unit's image as a buffer (offsets inside one image fit 31 bits: checked by the host)
    const uint32_t img_bytes = CL ? static_cast<uint32_t>(g.o_img) * kEsz : static_cast<uint32_t>(g.Cout) * g.HoWo * kEsz;
    auto image_of = [&](const Unit& u) {   // (the unit's images that exist: stores to one past the batch are dropped)
        const uint32_t n = static_cast<uint32_t>(g.B) - u.b < static_cast<uint32_t>(g.group) ? g.B - u.b : g.group;
        return make_rsrc(static_cast<char*>(g.out) + static_cast<size_t>(u.b) * img_bytes, n * img_bytes);
    };

    const int w_lane = CL ? ((lane >> 4) * 64 + wn * 32 + ((lane & 15) >> 2) * 8 + (lane & 3)) * kChunk
                          : ((lane >> 4) * 64 + wn * 32 + (lane & 15)) * kChunk;
    constexpr int kWFrag = CL ? 4 * kChunk : 16 * kChunk;

    // A workgroup's eight waves sit two to a SIMD (wave w with wave w + 4).  The second one goes first whenever both want
    // to issue: it runs a tap's MFMAs while the first waits, then converts its row while the first one's MFMAs run.
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#ifndef MIQ_NO_DEV_SWITCHES
    int stamp_it = 0;
#endif
    int unit = blockIdx.x;
    Unit cur = decode(unit);
    // The stage after a given one, decoded ahead of time: its divisions run among the taps of the stage before (behind the
    // stage's barrier they cost every wave 360-600 cycles with no MFMA in flight).
    struct Next {
        int cg, unit;
        bool fetch;
        Unit u;
    };
    auto successor = [&](int of_unit, int of_cg) {
        Next n;
        n.cg = of_cg + 1;
        n.unit = of_unit;
        if (n.cg == g.ncg) {
            n.cg = 0;
            n.unit = of_unit + static_cast<int>(gridDim.x);
        }
        n.fetch = n.unit < g.units;
        n.u = decode(n.fetch ? n.unit : of_unit);   // (past the last unit: arithmetic on a unit that exists, never used)
        return n;
    };
    Next nx = successor(unit, 0);
    Unit upcoming = cur;
    // DIRECT, before the loop: where piece p of this wave lands.  The LDS that is idle until the loop starts (the second
    // patch buffer, the second weight stage and the staging areas) holds 49 wave-pieces: the first stage's six pieces are
    // fetched at once and re-laid together.
    auto idle_of = [&](int p) -> char* {
        const int j = p * 8 + wave;
        return j < 21 ? smem + kPatch + j * kDirectStageBytes : smem + 2 * kPatch + kWStageBytes + (j - 21) * kDirectStageBytes;
    };
    if constexpr (DIRECT) {
        static_assert(21 * kDirectStageBytes <= kPatch && 27 * kDirectStageBytes <= kWStageBytes + 8 * kDirectStageBytes, "idle LDS");
        // both patch buffers start as zeros: the border columns and the slots behind the last row are never written again
        for (int i = threadIdx.x; i < kPatch / 16; i += 512) reinterpret_cast<v4i*>(smem)[i] = v4i{0, 0, 0, 0};
        __syncthreads();
#pragma unroll 1
        for (int p = 0; p < 6; ++p) direct_fetch(cur, 0, p, idle_of(p));
    }
#pragma unroll 1
    for (int k = 0; k < 6; ++k) (void)issue_piece(cur, 0, 0, true, k);
    if (w_res && g.ncg == 2) {   // the second group's filters, for good
#pragma unroll 1
        for (int q = wave; q < 36; q += 8)
            dma16(rw, smem + 2 * kPatch + kWStageBytes + q * 1024, ((cur.tn * 2 + 1) * 36 + q) * 1024 + lane * kChunk);
    }
    if constexpr (DIRECT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int p = 0; p < 6; ++p) direct_relay(cur, 0, 0, p, idle_of(p));
        __syncthreads();   // every wave has read its pieces: the second patch buffer can be cleared
        for (int i = threadIdx.x; i < kPatch / 16; i += 512) reinterpret_cast<v4i*>(smem + kPatch)[i] = v4i{0, 0, 0, 0};
    }
    int buf = 0;
    int stores = 0;         // store instructions this wave issued after its last piece of the next stage's fetch
    int after_patch = 0;    // DIRECT: vector memory instructions this wave issued after its last patch piece
    bool pending = false;   // `done` holds a unit that has not left yet
    Unit prev{};
    __amdgpu_buffer_rsrc_t rout = image_of(cur);
    for (; unit < g.units; unit += gridDim.x) {
        cur = upcoming;
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
                aoff[i] = (lane >> 4) * (kSlots * kChunk) + (im * g.img_slots + (y - cur.y0) * g.pitch + x) * kChunk;
            }
        }
        for (int cg = 0; cg < g.ncg; ++cg) {
            // This wave's pieces of the stage have landed once all but the stores it issued after the last of them are
            // done (vector memory operations retire in issue order); the barrier tells the same of the other waves'
            // pieces, and that every wave has left the stage before, whose buffers the next fetch overwrites.
            MIQ_PLANAR_STAMP(0)
            wait_vmcnt_upto16(stores);
            stores = 0;
            MIQ_PLANAR_STAMP(1)
            __syncthreads();
            MIQ_PLANAR_STAMP(2)
            // the next stage (decoded during the stage before), fetched piece by piece among this stage's taps
            const Next cn = nx;
            const int ncg = cn.cg;
            const bool fetch = cn.fetch;
            const Unit nxt = cn.u;
            if (ncg == 0) upcoming = nxt;
            MIQ_PLANAR_STAMP(3)
            const bool flush = pending && cg == 0;
            const char* patch = smem + buf * kPatch;
            const char* wst = smem + 2 * kPatch + (w_res ? cg : buf) * kWStageBytes + w_lane;
            {
                v4i af[2][FM], wf[2][FN];   // the fragments of tap t + 1 are read while the MFMAs of tap t run
                auto read_tap = [&](auto S, int kh, int kw) {
                    constexpr int s = decltype(S)::value;
                    const char* arow = patch + (kh * g.pitch + kw) * kChunk;
                    const char* wrow = wst + (kh * 3 + kw) * 4096;
#pragma unroll
                    for (int j = 0; j < FN; ++j) wf[s][j] = lds_read16(wrow + j * kWFrag);
#pragma unroll
                    for (int i = 0; i < FM; ++i) af[s][i] = lds_read16(arow + aoff[i]);
                };
                auto tap = [&](auto T) {
                    constexpr int t = decltype(T)::value, s = t & 1;
                    if constexpr (t == 6) nx = successor(cn.unit, cn.cg);   // scalar work beside this tap's MFMAs
                    if constexpr (DIRECT && t >= 1 && t <= 6) {
                        if (fetch) {   // piece t - 1 has landed once only what was issued after it is left in flight
                            wait_vmcnt_upto16(after_patch);
                            direct_relay(nxt, ncg, buf ^ 1, t - 1, staging);
                        }
                    }
                    if constexpr (DIRECT && t < 6) {
                        if (fetch) {
                            direct_fetch(nxt, ncg, t, staging);
                            stores = 0;
                            after_patch = 0;
                        }
                    }
                    if constexpr (t < 6) {
                        if (fetch && issue_piece(nxt, ncg, buf ^ 1, !w_res, t)) {
                            stores = 0;
                            if constexpr (DIRECT) after_patch += 1;
                        }
                    }
                    if constexpr (t + 1 < 9) read_tap(std::integral_constant<int, (t + 1) & 1>{}, (t + 1) / 3, (t + 1) % 3);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = CL ? __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[s][j], af[s][i], acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_i32_16x16x64_i8(af[s][i], wf[s][j], acc[i][j], 0, 0, 0);
                    // The finished unit's row t is converted and stored behind this tap's MFMAs, as a block of its own
                    // behind a branch (scheduled together with the MFMAs, or with the nine taps as one block, the compiler
                    // keeps so much in flight that it spills; left alone, it converts the whole unit ahead of the first tap).  The two waves of a SIMD run half a tap
                    // apart (priorities below), so one's conversion is covered by the other's MFMAs.
                    __builtin_amdgcn_sched_barrier(0);
                    if (flush) {
                        const int n = store_step(rout, prev, T);
                        stores += n;
                        if constexpr (DIRECT) after_patch += n;
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
            MIQ_PLANAR_STAMP(4)
#ifndef MIQ_NO_DEV_SWITCHES
            MIQ_PLANAR_STAMP(5)
            MIQ_PLANAR_STAMP(6)
            ++stamp_it;
#endif
            if (flush) pending = false;
            buf ^= 1;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                done[i][j] = acc[i][j];
                acc[i][j] = v4i{0, 0, 0, 0};
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

template <int OUT, bool CL = false, bool DIRECT = false>
int launch_planar(const PlanarArgs& g, bool stream_out, hipStream_t st) {
    void (*kern)(PlanarArgs) = conv3x3_patch_kernel<OUT, false, CL, DIRECT>;
    if constexpr (!CL) {
        if (stream_out) kern = conv3x3_patch_kernel<OUT, true, false, DIRECT>;
    }
    if constexpr (!CL && !DIRECT && OUT != kOutI8) {
        if (g.ragged) kern = conv3x3_patch_kernel<OUT, false, false, false, true>;
    } else {
        if (g.ragged) return fail(MIQ_EINVAL, "conv3x3_patch_kernel: pixel-by-pixel stores exist for NCHW outputs from the padded copy only");
    }
    // all of a CU's LDS (DIRECT: smaller patch planes make room for the eight staging areas)
    constexpr int kLds = DIRECT ? 2 * 4 * kDirectSlots * kChunk + 2 * kWStageBytes + 8 * kDirectStageBytes : 2 * kPlanarStage;
    static_assert(kLds <= 160 * 1024, "LDS of one CU");
    if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
    // one persistent workgroup per CU, each keeping one channel tile
    int grid = g.units < 256 ? g.units : 256 - 256 % g.tiles_n;
    if (grid < g.tiles_n) grid = g.units < g.tiles_n ? g.units : g.tiles_n;
    kern<<<grid, 512, kLds, st>>>(g);
    return check_launch("conv3x3_patch_kernel");
}


// ---------------------------------------------------------------------------------------
// Stems: at most four input channels (one pixel = 4 bytes of the padded channels-last copy), five to eight filter
// columns, stride 2 (ResNet's 3 -> 64 7 x 7 / 2).  In im2col order such a layer's K axis is short (KH rows of 32 bytes)
// and its M axis enormous, and the im2col-order kernel fetches every input byte about twelve times from the L2
// (1 GB for a 54 MB input).  Here a workgroup owns a 16 x 16 block of output pixels x 64 channels and stages the input
// patch those pixels see once (38 rows of 40 pixels at most, 6 KB) together with the channel tile's filters (16 KB);
// the im2col expansion happens on the way from LDS to registers: the K chunk c = 2 * kh + half of output pixel (y, x)
// is the 16 bytes at patch pixel (2y + kh, 2x + 4 * half), read as two 8-byte halves (the address is only 8-byte
// aligned).  Four waves, wave w multiplies output rows 4w .. 4w+3 (four fragments of 16 pixels) by all 64 channels.
// Chunks past the filter's last row meet zero weights.  Channels-last in and out; the weight fragment is the MFMA's
// first operand and the filters sit in LDS in an order that leaves a lane with sixteen consecutive channels of one
// pixel (one 16-byte store per pixel at int8).  Measured on ResNet-18's first layer at batch 256 (tools/conv_stem_ab.py):
// int8 out 235 -> 79 us, bf16 221 -> 117, fp32 329 -> 262; with int32 out (and for NCHW outputs, whose 16 x 16 blocks
// write 64-byte runs) the im2col-order kernel's long output runs win and it keeps those.
struct StemArgs {
    const int8_t* xp;
    const int8_t* wp;
    const float* bias;
    const float* alpha;
    void* out;
    float beta;
    int Cout, KH, Kp;   // Kp = KH * 32: bytes of one channel's packed filter
    int Hp, Wp, Ho, Wo;
    int tiles_x, tiles_y, tiles_n, units;
    uint32_t xp_bytes, wp_bytes;
    int o_img, o_row, o_pix, o_off;   // channels-last output: element strides of out[b][y + border][x + border][channel]
};
constexpr int kStemPitch = 40;                        // patch pixels per row: 15 * 2 + 8 = 38, rounded to 16 bytes
constexpr int kStemPatchBytes = 6 * 1024;             // 38 rows x 160 bytes = 6080, fetched as six wave-instructions
constexpr int kStemWeightBytes = 16 * 64 * kChunk;    // [chunk][channel][16 B]

template <int OUT>
__global__ __launch_bounds__(256) void conv_stem_kernel(StemArgs g) {
    // Persistent workgroups: the channel tile's filters are fetched once, the patches alternate between two buffers and
    // the next block's patch is on its way while this one is multiplied and stored.
    __shared__ __attribute__((aligned(16))) char smem[2 * kStemPatchBytes + kStemWeightBytes + 512];
    char* const wst = smem + 2 * kStemPatchBytes;
    float* const params = reinterpret_cast<float*>(wst + kStemWeightBytes);   // alpha[64], bias[64] of the channel tile
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
        k.x0 = (t % g.tiles_x) * 16;
        t /= g.tiles_x;
        k.y0 = (t % g.tiles_y) * 16;
        k.b = t / g.tiles_y;
        return k;
    };
    // patch: rows 2*y0 .. 2*y0 + 29 + KH of the image, 160 bytes from pixel 2*x0 (a run that leaves the row continues
    // into the next one, one that leaves the buffer reads zeros: both only under pixels that are never stored)
    const int pieces = (30 + g.KH) * 10;
    auto fetch_patch = [&](const Block& k, char* patch) {
        for (int i = wave; i * 64 < pieces; i += 4) {
            const int id = i * 64 + lane, row = id / 10, ch = id - row * 10;
            const uint32_t off = id < pieces ? ((k.b * g.Hp + k.y0 * 2 + row) * g.Wp + k.x0 * 2) * 4 + ch * kChunk : kOob;
            dma16(rx, patch + i * 1024, off);
        }
    };
    {
        // filters: chunk c of the 64 channels as one wave-instruction; position l of a chunk holds the channel that
        // leaves lane group l>>2 & 3 with sixteen consecutive channels (epilogue below)
        const int ch = ((lane & 15) >> 2) * 16 + (lane >> 4) * 4 + (lane & 3);
        const int n = tn * 64 + ch;
#pragma unroll
        for (int c = wave; c < 16; c += 4)
            dma16(rw, wst + c * 1024, (c < g.KH * 2 && n < g.Cout) ? static_cast<uint32_t>(n) * g.Kp + c * kChunk : kOob);
    }
    // per-channel epilogue parameters of the channel tile, once, in LDS (in registers they cost a third of the budget)
    if constexpr (OUT != kOutI32) {
        if (threadIdx.x < 128) {
            const int c = threadIdx.x & 63, n = tn * 64 + c;
            float v = threadIdx.x < 64 ? 1.f : 0.f;
            if (n < g.Cout) {
                if (threadIdx.x < 64)
                    v = g.alpha[n];
                else if (g.bias)
                    v = g.bias[n];
            }
            params[threadIdx.x] = v;
        }
    }
    const v2f beta2{g.beta, g.beta};

    uint32_t unit = blockIdx.x;
    const uint32_t units = static_cast<uint32_t>(g.units);
    Block cur = decode(unit);
    fetch_patch(cur, smem);
    const int ksteps = (g.KH * 2 + 3) >> 2;
    const char* w_lane = wst + (q * 64 + r) * kChunk;
    // this lane's patch offset for fragment row 0 and K step 0: filter row q>>1, half q&1, output column r
    const int a_off = (((wave * 4 * 2 + (q >> 1)) * kStemPitch) + (q & 1) * 4 + r * 2) * 4;
    int buf = 0;
    int stores = 0;   // store instructions this wave issued behind the fetch of the patch about to be read
    for (; unit < units; unit += gridDim.x) {
        // Vector memory operations retire in issue order: once all but the stores issued after it are done, this wave's
        // pieces of the patch have landed, and the stores go on draining under the next block's arithmetic.
        wait_vmcnt_upto16(stores);
        __syncthreads();   // every wave's pieces have landed; every wave has left the buffer the next fetch overwrites
        {
            // rows of this wave inside the image x store instructions per row (wave-uniform: lane 0 always stores)
            const int rows = min(4, max(0, g.Ho - (cur.y0 + wave * 4)));
            stores = rows * (OUT == kOutI8 ? 1 : OUT == kOutBF16 ? 2 : 4);
        }
        const uint32_t nunit = unit + gridDim.x;
        const Block nxt = decode(nunit < units ? nunit : unit);
        if (nunit < units) fetch_patch(nxt, smem + (buf ^ 1) * kStemPatchBytes);

        v4i acc[4][4];
        const char* a_lane = smem + buf * kStemPatchBytes + a_off;
        auto kstep = [&](int j, auto FIRST) {   // FIRST: the accumulators start from this step's products, not from zeros
            v4i af[4], wf[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) wf[f] = lds_read16(w_lane + j * 4096 + f * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char* a = a_lane + (i * 2 + j * 2) * (kStemPitch * 4);
                const v2u lo = *reinterpret_cast<const v2u*>(a), hi = *reinterpret_cast<const v2u*>(a + 8);
                af[i] = v4i{(int)lo[0], (int)lo[1], (int)hi[0], (int)hi[1]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const v4i c = decltype(FIRST)::value ? v4i{0, 0, 0, 0} : acc[i][f];
                    acc[i][f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[f], af[i], c, 0, 0, 0);
                }
        };
        kstep(0, std::true_type{});
        for (int j = 1; j < ksteps; ++j) kstep(j, std::false_type{});

        // D[R][c] of fragment f: R = 4*(lane>>4) + reg -> channel (lane>>4)*16 + f*4 + reg of the tile, c = lane&15 -> pixel
        const int n0 = tn * 64 + q * 16;
        const int x = cur.x0 + r;
        if (x < g.Wo && n0 < g.Cout) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = cur.y0 + wave * 4 + i;
                if (y >= g.Ho) continue;
                const size_t idx = static_cast<size_t>(cur.b) * g.o_img + static_cast<size_t>(y) * g.o_row + x * g.o_pix + g.o_off + n0;
                uint32_t packed[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const v4i a = acc[i][f];
                    if constexpr (OUT == kOutI32) {
                        reinterpret_cast<v4i*>(static_cast<int*>(g.out) + idx)[f] = a;
                    } else {
                        const v4f al = *reinterpret_cast<const v4f*>(params + q * 16 + f * 4);
                        const v4f bi = *reinterpret_cast<const v4f*>(params + 64 + q * 16 + f * 4);
                        const v2f lo = dequant2(a[0], a[1], v2f{bi[0], bi[1]}, v2f{al[0], al[1]}, beta2);
                        const v2f hi = dequant2(a[2], a[3], v2f{bi[2], bi[3]}, v2f{al[2], al[3]}, beta2);
                        if constexpr (OUT == kOutF32)
                            reinterpret_cast<v4f*>(static_cast<float*>(g.out) + idx)[f] = v4f{lo[0], lo[1], hi[0], hi[1]};
                        else if constexpr (OUT == kOutBF16) {
                            const uint2 h = pack4_bf16(lo[0], lo[1], hi[0], hi[1]);
                            if (f & 1)
                                reinterpret_cast<uint4*>(static_cast<uint16_t*>(g.out) + idx)[f >> 1] = uint4{packed[0], packed[1], h.x, h.y};
                            packed[0] = h.x;
                            packed[1] = h.y;
                        } else {
                            packed[f] = requant4_i8(lo[0], lo[1], hi[0], hi[1]);
                        }
                    }
                }
                if constexpr (OUT == kOutI8)
                    *reinterpret_cast<uint4*>(static_cast<int8_t*>(g.out) + idx) = uint4{packed[0], packed[1], packed[2], packed[3]};
            }
        }
        cur = nxt;
        buf ^= 1;
    }
}

// The stem kernel serves: one 4-byte pixel (Cin <= 4), 5 to 8 filter columns (32-byte filter rows), at most 8 filter rows,
// stride 2, no subsampling re-layout.
inline bool stem_shape(const ConvGeom& G) {
    return G.Cp == 4 && G.Lp == 32 && G.KH <= 8 && G.stride == 2 && G.sub == 1 && !G.planar && G.Ho > 0 && G.Wo > 0;
}

template <int OUT>
int launch_stem(StemArgs g, int64_t B, hipStream_t st) {
    g.tiles_x = (g.Wo + 15) / 16;
    g.tiles_y = (g.Ho + 15) / 16;
    g.tiles_n = (g.Cout + 63) / 64;
    const int64_t units = B * g.tiles_y * g.tiles_x * g.tiles_n;
    MIQ_REQUIRE(units < (1ll << 31), MIQ_ERANGE, "miq_conv2d_i8_nhwc: too many tiles");
    g.units = static_cast<int>(units);
    // persistent: up to five workgroups per CU (29 KB of LDS each), a multiple of the channel tiles
    int64_t grid = std::min<int64_t>(units, 1280);
    if (grid > g.tiles_n) grid -= grid % g.tiles_n;
    conv_stem_kernel<OUT><<<static_cast<unsigned>(grid), 256, 0, st>>>(g);
    return check_launch("conv_stem_kernel");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int OUT, bool CL = false, int STAGES = 2>
int launch_tile(ConvArgs g, hipStream_t st) {
    g.tiles_m = static_cast<int>(ceil_div(g.M, BM));
    g.tiles_n = static_cast<int>(ceil_div(g.Cout, BN));
    const int lds = STAGES * (BM + BN) * kTileK;
    void (*kern)(ConvArgs) = conv_i8_kernel<BM, BN, WAVES_M, WAVES_N, OUT, false, CL, STAGES>;
    if constexpr (!CL) {
        if (g.stream_out) kern = conv_i8_kernel<BM, BN, WAVES_M, WAVES_N, OUT, true, false, STAGES>;
    }
    if (int rc = allow_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    kern<<<g.tiles_m * g.tiles_n, 64 * WAVES_M * WAVES_N, lds, st>>>(g);
    return check_launch("conv_i8_kernel");
}

template <int OUT, bool CL = false>
int launch_pipe(ConvArgs g, hipStream_t st) {
    g.tiles_m = static_cast<int>(ceil_div(g.M, 256));
    g.tiles_n = static_cast<int>(ceil_div(g.Cout, 256));
    constexpr int lds = 2 * 4 * 128 * kTileK;
    void (*kern)(ConvArgs) = conv_i8_pipe_kernel<OUT, false, CL>;
    if constexpr (!CL) {
        if (g.stream_out) kern = conv_i8_pipe_kernel<OUT, true>;
    }
    if (int rc = allow_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    kern<<<g.tiles_m * g.tiles_n, 512, lds, st>>>(g);
    return check_launch("conv_i8_pipe_kernel");
}

template <int OUT>
int launch(const ConvArgs& g, hipStream_t st) {
    // Largest tile that still gives the 256 CUs enough workgroups; BN follows Cout.
    auto tiles = [&](int bm, int bn) { return ceil_div(g.M, bm) * ceil_div(g.Cout, bn); };
    if (const char* force = dev_switch("MIQ_CONV_TILE")) {   // developer A/B switch
        switch (force[0]) {
            case 'a': return launch_tile<256, 256, 2, 4, OUT>(g, st);
            case 'b': return launch_tile<256, 128, 2, 4, OUT>(g, st);
            case 'c': return launch_tile<256, 64, 4, 2, OUT>(g, st);
            case 'd': return launch_tile<128, 128, 2, 2, OUT>(g, st);
            case 'e': return launch_tile<128, 64, 2, 2, OUT>(g, st);
            case 'p': return launch_pipe<OUT>(g, st);
#ifndef MIQ_NO_DEV_SWITCHES
            case 'E': return launch_tile<128, 64, 2, 2, OUT, false, 3>(g, st);    // rings
            case 'F': return launch_tile<128, 64, 2, 2, OUT, false, 4>(g, st);
            case 'D': return launch_tile<128, 128, 2, 2, OUT, false, 3>(g, st);
            case 'G': return launch_tile<128, 128, 2, 4, OUT, false, 3>(g, st);
            case 'H': return launch_tile<64, 64, 2, 2, OUT, false, 4>(g, st);
            case 'I': return launch_tile<128, 128, 2, 4, OUT, false, 4>(g, st);
            case 'J': return launch_tile<128, 64, 4, 2, OUT, false, 4>(g, st);
#endif
            default: break;
        }
    }
    // Measured on the ResNet-18 layers at batch 256 (MIQ_CONV_TILE sweep): 256 x 256 wins once it gives most
    // CUs a tile (196 tiles: 55 vs 61-66 us on 256->256 14x14); up to 128 output channels the 128 x 64 tile
    // (three workgroups per CU) beats 256 x 128 and 128 x 128 (69 vs 81 / 76 us on 64->128 3x3/2).
    if (g.Cout > 128 && tiles(256, 256) >= 160) return launch_pipe<OUT>(g, st);
    if (g.Cout > 128) return launch_tile<128, 128, 2, 2, OUT>(g, st);
    if (g.Cout <= 64 && tiles(256, 64) >= 200) return launch_tile<256, 64, 4, 2, OUT>(g, st);
    return launch_tile<128, 64, 2, 2, OUT>(g, st);
}

// The same tile choice for the channels-last entry point.
template <int OUT>
int launch_cl(const ConvArgs& g, hipStream_t st) {
    auto tiles = [&](int bm, int bn) { return ceil_div(g.M, bm) * ceil_div(g.Cout, bn); };
    if (g.Cout > 128 && tiles(256, 256) >= 160) return launch_pipe<OUT, true>(g, st);
    if (g.Cout > 128) return launch_tile<128, 128, 2, 2, OUT, true>(g, st);
    if (g.Cout <= 64 && tiles(256, 64) >= 200) return launch_tile<256, 64, 4, 2, OUT, true>(g, st);
    return launch_tile<128, 64, 2, 2, OUT, true>(g, st);
}

// dst[c][r] = src[r][c], any shape, byte accesses through a padded LDS tile.
__global__ __launch_bounds__(256) void transpose_bytes_kernel(const int8_t* __restrict__ src, int64_t rows,
                                                              int64_t cols, int8_t* __restrict__ dst) {
    __shared__ int8_t tile[64][65];
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 64, c0 = static_cast<int64_t>(blockIdx.x) * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = threadIdx.x + i * 256, r = idx >> 6, c = idx & 63;
        if (r0 + r < rows && c0 + c < cols) tile[r][c] = src[(r0 + r) * cols + c0 + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = threadIdx.x + i * 256, r = idx >> 6, c = idx & 63;  // r: src column, c: src row
        if (c0 + r < cols && r0 + c < rows) dst[(c0 + r) * rows + r0 + c] = tile[c][r];
    }
}

}  // namespace

// A [rows, cols] -> [cols, rows] transpose is the channels-last re-layout of one "image" with
// `rows` channels and `cols` pixels and no border, so the register-transposing kernel serves both.
extern "C" int miq_transpose_i8(const int8_t* src, int64_t rows, int64_t cols, int8_t* dst, void* stream) {
    MIQ_REQUIRE(rows >= 0 && cols >= 0, MIQ_EINVAL, "miq_transpose_i8: negative size");
    if (rows == 0 || cols == 0) return MIQ_OK;
    MIQ_REQUIRE(src && dst, MIQ_EINVAL, "miq_transpose_i8: null pointer");
    MIQ_REQUIRE(rows < (1ll << 31) && cols < (1ll << 30), MIQ_ERANGE, "miq_transpose_i8: at most 2^31 x 2^30");
    hipStream_t st = miq::as_stream(stream);
    if (rows % 4 == 0 && miq::aligned(dst, 16) && miq::ceil_div(cols, 64) * miq::ceil_div(rows, 64) < (1ll << 31)) {
        const int cblocks = static_cast<int>(miq::ceil_div(rows, 64));
        dim3 grid(static_cast<unsigned>(miq::ceil_div(cols, 64) * cblocks), 1);
        if (cols % 4 == 0 && miq::aligned(src, 4))
            nhwc_interior_kernel<true><<<grid, 256, 0, st>>>(src, dst, (int)rows, 1, (int)cols, 1, (int)cols, (int)rows, 0,
                                                             cblocks);
        else
            nhwc_interior_kernel<false><<<grid, 256, 0, st>>>(src, dst, (int)rows, 1, (int)cols, 1, (int)cols, (int)rows, 0,
                                                              cblocks);
        return miq::check_launch("nhwc_interior_kernel (transpose)");
    }
    MIQ_REQUIRE(miq::ceil_div(rows, 64) < 65536, MIQ_ERANGE, "miq_transpose_i8: more than 4M rows with rows % 4 != 0");
    dim3 grid(static_cast<unsigned>(miq::ceil_div(cols, 64)), static_cast<unsigned>(miq::ceil_div(rows, 64)));
    transpose_bytes_kernel<<<grid, 256, 0, st>>>(src, rows, cols, dst);
    return miq::check_launch("transpose_bytes_kernel");
}

// The packed weight (miq_conv2d_i8_pack_weight), shared by the channels-last entry point and miq_conv2d_i8_prepacked: the
// im2col-order form wp[Cout][KH][Lp] and, for 3 x 3 filters over whole 64-channel groups, behind it (at the next multiple
// of 256 bytes) the form conv3x3_patch_kernel stages from (output channels rounded up to whole tiles of 64, zero filled).
// It depends on the filter's shape only, so a layer packs once, whatever images it is later given.
struct PackedWeight {
    int64_t rows_bytes;     // the im2col-order form
    int64_t planar_at;      // offset of the patch kernel's form; 0 = none
    int64_t planar_bytes;
    int64_t total;
};

// Developer A/B switch over the output-store policy: MIQ_CONV_STREAM=0 cached, =1 nontemporal, unset: the caller's choice.
static bool stream_choice(bool chosen) {
    if (const char* f = dev_switch("MIQ_CONV_STREAM")) return f[0] == '1';
    return chosen;
}

static PackedWeight packed_weight_layout(int64_t Cin, int64_t Cout, int64_t KH, int64_t KW) {
    PackedWeight pw{};
    // (sizes as make_geom bounds them; what does not fit is held at 2^60, beyond every caller's 2 GiB limit)
    if (!miq::all_below_2_31({Cin, Cout}) || KH < 0 || KW < 0 || KH >= (1 << 15) || KW >= (1 << 15)) {
        pw.rows_bytes = pw.total = 1ll << 60;
        return pw;
    }
    const int64_t Lp = miq::round_up(KW * miq::round_up(Cin, kCinAlign), kChunk);
    pw.rows_bytes = miq::sat_product({Cout, KH, Lp});
    pw.total = pw.rows_bytes;
    if (KH == 3 && KW == 3 && Cin % 64 == 0) {
        pw.planar_at = miq::round_up(pw.rows_bytes, 256);
        pw.planar_bytes = miq::sat_product({miq::ceil_div(Cout, 64), Cin / 64, kWStageBytes});
        pw.total = pw.planar_at + pw.planar_bytes;
    }
    return pw;
}

extern "C" size_t miq_conv2d_i8_workspace(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                                          int64_t KW, int64_t stride, int64_t pad) {
    ConvGeom g;
    if (!make_geom(B, Cin, H, W, Cout, KH, KW, stride, pad, g)) return 0;
    if (g.xp_bytes >= (1ll << 31) || g.wp_bytes >= (1ll << 31)) return 0;   // (the call answers MIQ_ERANGE)
    return static_cast<size_t>(miq::round_up(g.xp_bytes, 256) + miq::round_up(g.wp_bytes, 256));
}

// `w` is the OIHW filter (prepacked = false: it is re-laid into the workspace by this call, behind the padded copy) or the
// packed weight of miq_conv2d_i8_pack_weight (prepacked = true: the workspace holds the padded copy only).
static int conv2d_i8_run(bool prepacked, const int8_t* x, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                         int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias,
                         const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                         size_t workspace_bytes, void* stream) {
    ConvGeom G;
    MIQ_REQUIRE(make_geom(B, Cin, H, W, Cout, KH, KW, stride, pad, G), MIQ_EINVAL,
                "miq_conv2d_i8: bad geometry (B=%lld Cin=%lld H=%lld W=%lld Cout=%lld KH=%lld KW=%lld stride=%lld pad=%lld)",
                (long long)B, (long long)Cin, (long long)H, (long long)W, (long long)Cout, (long long)KH,
                (long long)KW, (long long)stride, (long long)pad);
    MIQ_REQUIRE(out_dtype == MIQ_F32 || out_dtype == MIQ_BF16 || out_dtype == MIQ_I32, MIQ_EINVAL,
                "miq_conv2d_i8: out_dtype must be MIQ_F32, MIQ_BF16 or MIQ_I32 (got %d)", out_dtype);
    if (G.M == 0 || Cout == 0) return MIQ_OK;
    MIQ_REQUIRE(x && w && out, MIQ_EINVAL, "miq_conv2d_i8: null pointer");
    MIQ_REQUIRE(out_dtype == MIQ_I32 || alpha, MIQ_EINVAL, "miq_conv2d_i8: alpha is required for a dequantised output");
    MIQ_REQUIRE(G.xp_bytes < (1ll << 31) && G.wp_bytes < (1ll << 31) && G.M < (1ll << 31) && B < 65536 && G.Hp < 65536,
                MIQ_ERANGE, "miq_conv2d_i8: padded activations (%lld B) and packed weights (%lld B) must stay below 2 GiB",
                (long long)G.xp_bytes, (long long)G.wp_bytes);
    MIQ_REQUIRE(G.Kp / kChunk <= 4096, MIQ_ERANGE, "miq_conv2d_i8: KH*KW*Cin too large (%lld > 65536)", (long long)G.Kp);
    const size_t esz = out_dtype == MIQ_BF16 ? 2 : 4;
    MIQ_REQUIRE(miq::aligned(out, 4 * esz) && miq::aligned(alpha, 4) && miq::aligned(bias, 4), MIQ_EALIGN,
                "miq_conv2d_i8: out must be 16-byte (8 for bf16) aligned, alpha/bias 4-byte aligned");
    const size_t need = prepacked ? static_cast<size_t>(miq::round_up(G.xp_bytes, 256))
                                  : miq_conv2d_i8_workspace(B, Cin, H, W, Cout, KH, KW, stride, pad);
    MIQ_REQUIRE(workspace && workspace_bytes >= need, MIQ_EWORKSPACE,
                "miq_conv2d_i8: needs %zu bytes of workspace (got %zu)", need, workspace_bytes);
    MIQ_REQUIRE(miq::aligned(workspace, 256), MIQ_EALIGN, "miq_conv2d_i8: workspace must be 256-byte aligned");
    hipStream_t st = miq::as_stream(stream);
    H = G.H; W = G.W; stride = G.stride;   // a subsampled 1 x 1 layer continues as its stride-1 equivalent

    int8_t* xp = static_cast<int8_t*>(workspace);
    int8_t* wp = xp + miq::round_up(G.xp_bytes, 256);
    if (prepacked) {   // the form this geometry's kernel stages from, inside the caller's packed weight
        MIQ_REQUIRE(miq::aligned(w, 16), MIQ_EALIGN, "miq_conv2d_i8_prepacked: the packed weight must be 16-byte aligned");
        const PackedWeight pw = packed_weight_layout(Cin, Cout, KH, KW);
        wp = const_cast<int8_t*>(w) + (G.planar ? pw.planar_at : 0);
    }

    // 3 x 3 / stride 1 layers whose rows are whole dwords: the patch kernel stages the NCHW activation itself (no padded
    // copy, no re-layout launch); only the filters are re-laid, into the workspace.
    if (G.planar && G.group == 1 && W % 4 == 0 && (G.Ho * G.Wo) % 4 == 0 && miq::aligned(x, 4) &&
        miq::sat_product({B, Cin, H, W}) < (1ll << 31) &&
        !(dev_switch("MIQ_CONV_DIRECT") && dev_switch("MIQ_CONV_DIRECT")[0] == '0')) {
        const int64_t HoWo = G.Ho * G.Wo, tiles_img = miq::ceil_div(HoWo, kTM);
        int64_t rows = 0;
        for (int64_t p0 = 0; p0 < HoWo; p0 += kTM)
            rows = std::max(rows, (std::min<int64_t>(p0 + kTM, HoWo) - 1) / G.Wo - p0 / G.Wo + 3);
        const int64_t part_rows = miq::ceil_div(rows, 12);
        const int64_t cpc = miq::ceil_div(part_rows * W + 12, 16);
        if (rows * G.Wp + 2 <= kDirectSlots && 16 * cpc * 16 <= kDirectStageBytes && part_rows * (W / 4) <= 16 && HoWo < 65536 &&
            Cin == 64) {   // (one input-channel group: with more, the staging work costs what the re-layout launch did)
            const uint32_t wtotal = static_cast<uint32_t>(G.wp_bytes / 4);
            const int wgrid = static_cast<int>(wtotal / 256 + 1 < 4096 ? wtotal / 256 + 1 : 4096);
            if (!prepacked && !dev_switch("MIQ_CONV_SKIP_WPACK"))   // (the switch: timing-only, the workspace still holds the last call's filters)
                pack_conv_weight_planar_kernel<<<wgrid, 256, 0, st>>>(w, wp, wtotal, (int)Cin, (int)Cout);
            if (int rc = miq::check_launch("pack_conv_weight_planar_kernel")) return rc;
            PlanarArgs pa{};
            pa.wp = wp; pa.bias = bias; pa.alpha = alpha; pa.out = out; pa.beta = beta;
            pa.Cout = (int)Cout; pa.ncg = (int)(Cin / 64); pa.tiles_n = (int)miq::ceil_div(Cout, 64);
            pa.tiles_img = (int)tiles_img;
            pa.pitch = (int)G.Wp; pa.Hp = (int)G.Hp; pa.Wo = (int)G.Wo; pa.HoWo = (int)HoWo;
            pa.wo_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(G.Wo)) + 1u;
            pa.wp_bytes = (uint32_t)G.wp_bytes;
            pa.wide16 = out_dtype == MIQ_BF16 && pa.HoWo % 8 == 0 && miq::aligned(out, 16);
            pa.x = x; pa.xp = x; pa.x_bytes = pa.xp_bytes = (uint32_t)(B * Cin * H * W);
            pa.cin = (int)Cin; pa.H = (int)H; pa.W = (int)W; pa.pad = (int)pad;
            pa.rows = (int)rows; pa.part_rows = (int)part_rows; pa.w4 = (int)(W / 4);
            pa.w4_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(pa.w4)) + 1u;
            pa.cpc = (int)cpc;
            pa.cpc_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(cpc)) + 1u;
            if (const char* sp = dev_switch("MIQ_PLANAR_STAMPS"))
                pa.stamps = reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 0));
            MIQ_REQUIRE(B * tiles_img * pa.tiles_n < (1ll << 31), MIQ_ERANGE, "miq_conv2d_i8: too many tiles");
            set_patch_units(pa, B, HoWo, G.Hp, G.Wp, 1);
            const bool stream_out = stream_choice(G.M * Cout * static_cast<int64_t>(esz) >= (192ll << 20) && out_dtype != MIQ_BF16);
            switch (out_dtype) {
                case MIQ_F32: return launch_planar<kOutF32, false, true>(pa, stream_out, st);
                case MIQ_BF16: return launch_planar<kOutBF16, false, true>(pa, stream_out, st);
                default: return launch_planar<kOutI32, false, true>(pa, stream_out, st);
            }
        }
    }
    {
        const int cblocks = static_cast<int>(miq::ceil_div(G.Cp, 64));
        bool packed = false;   // the weight re-layout rode along with the activation re-layout
        dim3 grid(static_cast<unsigned>(miq::ceil_div(G.Wp, 64) * cblocks), static_cast<unsigned>(G.Hp),
                  static_cast<unsigned>(B));
        if (G.planar) {
            ConvPrepArgs pa{};
            pa.x = x; pa.xp = xp; pa.w = w; pa.wp = wp;
            pa.Cin = (int)Cin; pa.H = (int)H; pa.W = (int)W; pa.Hp = (int)G.Hp; pa.Wp = (int)G.pitch; pa.Cp = (int)Cin;
            pa.pad = (int)pad; pa.cblocks = (int)(Cin / 64);
            pa.plane_stride = G.ptot * kChunk;
            pa.Cout = (int)Cout;
            const bool dwords = (H * W) % 4 == 0 && miq::aligned(x, 4);
            const int nt = prep_tiles_per_workgroup(dwords, H * W);
            pa.n_interior = static_cast<int>(miq::ceil_div(H * W, 64 * nt) * pa.cblocks);
            pa.border_total = static_cast<uint32_t>(B * (2 * pad * G.pitch + 2 * pad * H));
            pa.weight_total = (prepacked || dev_switch("MIQ_CONV_SKIP_WPACK")) ? 0u : static_cast<uint32_t>(G.wp_bytes / 4);
            const int64_t extra = miq::ceil_div(static_cast<int64_t>(pa.border_total) * (Cin / 16) + pa.weight_total, 1024);
            pa.n_extra = static_cast<int>(std::min<int64_t>(std::max<int64_t>(miq::ceil_div(extra, B), 1), 4096));
            dim3 pgrid(static_cast<unsigned>(pa.n_interior + pa.n_extra), static_cast<unsigned>(B));
            if (dwords && nt == 4)
                conv_prep_kernel<1, 4><<<pgrid, 256, 0, st>>>(pa);
            else if (dwords && nt == 2)
                conv_prep_kernel<1, 2><<<pgrid, 256, 0, st>>>(pa);
            else if (dwords)
                conv_prep_kernel<1><<<pgrid, 256, 0, st>>>(pa);
            else
                conv_prep_kernel<0><<<pgrid, 256, 0, st>>>(pa);
            packed = true;
        } else if (W % 4 == 0 && miq::aligned(x, 4) && G.Cp == 4 && G.sub == 1) {
            const int64_t total4 = B * G.Hp * (W / 4);
            const int g4 = static_cast<int>(total4 / 256 + 1 < (1 << 20) ? total4 / 256 + 1 : (1 << 20));
            nhwc_pad_c4_kernel<<<g4, 256, 0, st>>>(x, xp, (int)Cin, (int)H, (int)W, (int)G.Hp, (int)G.Wp, (int)pad, total4);
        } else if (H * W < (1ll << 30) && B < 65536) {
            ConvPrepArgs pa{};
            pa.x = x; pa.xp = xp; pa.w = w; pa.wp = wp;
            pa.Cin = (int)Cin; pa.H = (int)H; pa.W = (int)W; pa.Hp = (int)G.Hp; pa.Wp = (int)G.Wp; pa.Cp = (int)G.Cp;
            pa.pad = (int)pad; pa.cblocks = cblocks;
            pa.KH = (int)KH; pa.KW = (int)KW; pa.L = (int)G.L; pa.Lp = (int)G.Lp;
            const bool pairs = G.sub == 2 && W % 4 == 0 && G.Ws % 4 == 0 && (G.Hs * G.Ws) % 8 == 0 && miq::aligned(x, 8);
            const bool dwords = G.sub == 1 && (H * W) % 4 == 0 && miq::aligned(x, 4);
            const int nt = prep_tiles_per_workgroup(dwords, H * W);   // (two tiles of 8-byte pair loads: 40.7 -> 44.5 us on 64->128 1x1/2)
            pa.n_interior = static_cast<int>(miq::ceil_div(H * W, 64 * nt) * cblocks);
            pa.border_total = static_cast<uint32_t>(B * (2 * pad * G.Wp + 2 * pad * H));
            pa.weight_total = (prepacked || dev_switch("MIQ_CONV_SKIP_WPACK")) ? 0u : static_cast<uint32_t>(G.wp_bytes / 4);
            pa.sub = (int)G.sub; pa.Ws = (int)G.Ws; pa.HWs = (int)(G.Hs * G.Ws);
            // about four border chunks / weight dwords per extra thread, spread over the images' grid rows
            const int64_t extra = miq::ceil_div(static_cast<int64_t>(pa.border_total) * ((G.Cp & 15) ? 1 : G.Cp / 16) + pa.weight_total, 1024);
            pa.n_extra = static_cast<int>(std::min<int64_t>(std::max<int64_t>(miq::ceil_div(extra, B), 1), 4096));
            dim3 pgrid(static_cast<unsigned>(pa.n_interior + pa.n_extra), static_cast<unsigned>(B));
            if (pairs)
                conv_prep_kernel<2><<<pgrid, 256, 0, st>>>(pa);
            else if (dwords && nt == 4)
                conv_prep_kernel<1, 4><<<pgrid, 256, 0, st>>>(pa);
            else if (dwords && nt == 2)
                conv_prep_kernel<1, 2><<<pgrid, 256, 0, st>>>(pa);
            else if (dwords)
                conv_prep_kernel<1><<<pgrid, 256, 0, st>>>(pa);
            else
                conv_prep_kernel<0><<<pgrid, 256, 0, st>>>(pa);
            packed = true;
        } else {
            MIQ_REQUIRE(G.sub == 1, MIQ_ERANGE, "miq_conv2d_i8: image too large for the subsampling re-layout");
            nhwc_pad_kernel<<<grid, 256, 0, st>>>(x, xp, (int)Cin, (int)H, (int)W, (int)G.Hp, (int)G.Wp, (int)G.Cp,
                                                  (int)pad, cblocks);
        }
        if (int rc = miq::check_launch("nhwc re-layout kernel")) return rc;
        if (!packed && !prepacked) {
            const uint32_t total = static_cast<uint32_t>(G.wp_bytes / 4);   // dwords
            const int wgrid = static_cast<int>(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
            pack_conv_weight_kernel<<<wgrid, 256, 0, st>>>(w, wp, total, (int)Cin, (int)KH, (int)KW, (int)G.Cp, (int)G.L,
                                                           (int)G.Lp);
            if (int rc = miq::check_launch("pack_conv_weight_kernel")) return rc;
        }
    }

    if (G.planar) {
        PlanarArgs pa{};
        pa.xp = xp; pa.wp = wp; pa.bias = bias; pa.alpha = alpha; pa.out = out; pa.beta = beta;
        pa.Cout = (int)Cout; pa.ncg = (int)(Cin / 64); pa.tiles_n = (int)miq::ceil_div(Cout, 64);
        pa.tiles_img = (int)miq::ceil_div(G.Ho * G.Wo, kTM);
        pa.pitch = (int)G.pitch; pa.Hp = (int)G.Hp; pa.Wo = (int)G.Wo; pa.HoWo = (int)(G.Ho * G.Wo);
        pa.ptot = (uint32_t)G.ptot;
        pa.wo_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(G.Wo)) + 1u;
        pa.xp_bytes = (uint32_t)G.xp_bytes; pa.wp_bytes = (uint32_t)G.wp_bytes;
        pa.wide16 = out_dtype == MIQ_BF16 && pa.HoWo % 8 == 0 && miq::aligned(out, 16);
        pa.ragged = pa.HoWo % 4 != 0;
        if (const char* sp = dev_switch("MIQ_PLANAR_STAMPS"))   // address of a device buffer of 512 * 64 uint64
            pa.stamps = reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 0));
        MIQ_REQUIRE(B * miq::ceil_div(G.Ho * G.Wo, kTM) * pa.tiles_n < (1ll << 31), MIQ_ERANGE, "miq_conv2d_i8: too many tiles");
        set_patch_units(pa, B, G.Ho * G.Wo, G.Hp, G.pitch, G.group);
        // as below: outputs far beyond the Infinity Cache stream past it (16-byte stores only)
        const bool stream_out = stream_choice(G.M * Cout * static_cast<int64_t>(esz) >= (192ll << 20) && out_dtype != MIQ_BF16);
        switch (out_dtype) {
            case MIQ_F32:
                return launch_planar<kOutF32>(pa, stream_out, st);
            case MIQ_BF16:
                return launch_planar<kOutBF16>(pa, stream_out, st);
            default:
                return launch_planar<kOutI32>(pa, stream_out, st);
        }
    }

    ConvArgs a{};
    a.xp = xp;
    a.wp = wp;
    a.bias = bias;
    a.alpha = alpha;
    a.out = out;
    a.beta = beta;
    a.M = static_cast<int>(G.M);
    a.Cout = static_cast<int>(Cout);
    a.Kp = static_cast<int>(G.Kp);
    a.HoWo = static_cast<int>(G.Ho * G.Wo);
    a.Wo = static_cast<int>(G.Wo);
    a.img_stride = static_cast<int>(G.Hp * G.Wp * G.Cp);
    a.row_stride = static_cast<int>(G.Wp * G.Cp);
    a.sx = static_cast<int>(stride * G.Cp);
    a.sy = static_cast<int>(stride * G.Wp * G.Cp);
    a.cps = static_cast<int>(G.Lp / kChunk);
    a.cps_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(a.cps)) + 1u;
    a.xp_bytes = static_cast<uint32_t>(G.xp_bytes);
    a.wp_bytes = static_cast<uint32_t>(G.wp_bytes);
    // Measured at batch 256 with the output rotating over three buffers: nontemporal stores gain 14 % on the
    // 822 MB output of ResNet-18's first layer and 3 % at 205 MB, and lose 8-40 % on outputs of 26-103 MB,
    // which the Infinity Cache absorbs.
    a.wide16 = out_dtype == MIQ_BF16 && a.HoWo % 8 == 0 && miq::aligned(out, 16);
    // (8-byte nontemporal stores are slower than cached ones: bf16 outputs stream only as 16-byte vectors.)
    a.stream_out = stream_choice(G.M * Cout * static_cast<int64_t>(esz) >= (192ll << 20) && (out_dtype != MIQ_BF16 || a.wide16));
    switch (out_dtype) {
        case MIQ_F32:
            return launch<kOutF32>(a, st);
        case MIQ_BF16:
            return launch<kOutBF16>(a, st);
        default:
            return launch<kOutI32>(a, st);
    }
}

extern "C" int miq_conv2d_i8(const int8_t* x, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W,
                             int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias,
                             const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                             size_t workspace_bytes, void* stream) {
    return conv2d_i8_run(false, x, w, B, Cin, H, W, Cout, KH, KW, stride, pad, bias, alpha, beta, out, out_dtype, workspace,
                         workspace_bytes, stream);
}

extern "C" size_t miq_conv2d_i8_prepacked_workspace(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                                                    int64_t KW, int64_t stride, int64_t pad) {
    ConvGeom g;
    if (!make_geom(B, Cin, H, W, Cout, KH, KW, stride, pad, g)) return 0;
    if (g.xp_bytes >= (1ll << 31)) return 0;   // (the call answers MIQ_ERANGE)
    return static_cast<size_t>(miq::round_up(g.xp_bytes, 256));
}

extern "C" int miq_conv2d_i8_prepacked(const int8_t* x, const int8_t* wp, int64_t B, int64_t Cin, int64_t H, int64_t W,
                                       int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias,
                                       const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return conv2d_i8_run(true, x, wp, B, Cin, H, W, Cout, KH, KW, stride, pad, bias, alpha, beta, out, out_dtype, workspace,
                         workspace_bytes, stream);
}

// ---- dense layers with a stride / padding per axis or a dilation ----------------------------------------
// The im2col-order kernels above already address a window through byte strides, so these layers ride them as they are:
//   * the padded channels-last copy gets one border width P = the largest of the four paddings (the re-layout kernels
//     know a single one) and the windows start (P - pad_top) rows and (P - pad_left) pixels into it: the kernels are
//     handed the copy from that point on;
//   * the strides of the two axes go into ConvArgs::sy / sx, a dilation along y into row_stride;
//   * under a dilation along x a filter row is no longer one run of the copy, so every tap becomes a "row" of its own
//     (ConvArgs::kwt): Cp bytes of one pixel, padded to 16 in the packed weight as any other row.
namespace {

struct DenseGeom {
    int64_t P, Hp, Wp, Cp, rows, kw_run, L, Lp, Kp, Ho, Wo, M, xp_bytes, wp_bytes;
};

bool make_dense_geom(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW, int64_t sh, int64_t sw,
                     int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh, int64_t dw, DenseGeom& g) {
    if (B < 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout < 0 || KH <= 0 || KW <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 ||
        pt < 0 || pb < 0 || pl < 0 || pr < 0)
        return false;
    // (as make_geom: sizes below 2^31, filter / stride / dilation / padding below 2^15, long products held at 2^60)
    if (!all_below_2_31({B, Cin, H, W, Cout})) return false;
    for (int64_t v : {KH, KW, sh, sw, dh, dw, pt, pb, pl, pr})
        if (v >= (1 << 15)) return false;
    const int64_t eh = dh * (KH - 1) + 1, ew = dw * (KW - 1) + 1;
    if (H + pt + pb < eh || W + pl + pr < ew) return false;
    g.P = std::max(std::max(pt, pb), std::max(pl, pr));
    g.Hp = H + 2 * g.P;
    g.Wp = W + 2 * g.P;
    g.Cp = round_up(Cin, kCinAlign);
    g.rows = dw == 1 ? KH : KH * KW;
    g.kw_run = dw == 1 ? KW : 1;        // taps of one "row"
    g.L = g.kw_run * g.Cp;
    g.Lp = round_up(g.L, kChunk);
    g.Kp = sat_product({g.rows, g.Lp});
    g.Ho = (H + pt + pb - eh) / sh + 1;
    g.Wo = (W + pl + pr - ew) / sw + 1;
    g.M = sat_product({B, g.Ho, g.Wo});
    g.xp_bytes = sat_product({B, g.Hp, g.Wp, g.Cp});
    g.wp_bytes = sat_product({Cout, g.Kp});
    return true;
}

}  // namespace

namespace miq {

size_t conv2d_i8_dense_workspace(int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH, int64_t KW, int64_t sh,
                                 int64_t sw, int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh, int64_t dw) {
    DenseGeom g;
    if (!make_dense_geom(B, Cin, H, W, Cout, KH, KW, sh, sw, pt, pb, pl, pr, dh, dw, g) || g.M == 0 || Cout == 0) return 0;
    // (0 as well for what the kernels' 32-bit offsets do not reach: the caller then takes the direct convolution)
    if (g.xp_bytes >= (1ll << 31) || g.wp_bytes >= (1ll << 31) || g.M >= (1ll << 31) || B >= 65536 || g.Hp >= 65536 ||
        H * W >= (1ll << 30) || g.Kp / kChunk > 4096)
        return 0;
    return static_cast<size_t>(round_up(g.xp_bytes, 256) + round_up(g.wp_bytes, 256));
}

// The caller (miq_conv2d_general) has checked pointers, dtypes and alignment of bias / alpha.
int conv2d_i8_dense(const int8_t* x, const int8_t* w, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t KH,
                    int64_t KW, int64_t sh, int64_t sw, int64_t pt, int64_t pb, int64_t pl, int64_t pr, int64_t dh, int64_t dw,
                    const float* bias, const float* alpha, float beta, void* out, int out_dtype, void* workspace,
                    size_t workspace_bytes, hipStream_t st) {
    DenseGeom G;
    MIQ_REQUIRE(make_dense_geom(B, Cin, H, W, Cout, KH, KW, sh, sw, pt, pb, pl, pr, dh, dw, G), MIQ_EINVAL,
                "miq_conv2d_general: bad geometry");
    if (G.M == 0 || Cout == 0) return MIQ_OK;
    MIQ_REQUIRE(G.xp_bytes < (1ll << 31) && G.wp_bytes < (1ll << 31) && G.M < (1ll << 31) && B < 65536 && G.Hp < 65536 &&
                    H * W < (1ll << 30),
                MIQ_ERANGE, "miq_conv2d_general: padded activations (%lld B) and packed weights (%lld B) must stay below 2 GiB",
                (long long)G.xp_bytes, (long long)G.wp_bytes);
    MIQ_REQUIRE(G.Kp / kChunk <= 4096, MIQ_ERANGE, "miq_conv2d_general: KH*KW*Cin too large (%lld > 65536)", (long long)G.Kp);
    const size_t esz = out_dtype == MIQ_BF16 ? 2 : 4;
    MIQ_REQUIRE(aligned(out, 4 * esz), MIQ_EALIGN, "miq_conv2d_general: out must be 16-byte (8 for bf16) aligned");
    const size_t need = conv2d_i8_dense_workspace(B, Cin, H, W, Cout, KH, KW, sh, sw, pt, pb, pl, pr, dh, dw);
    MIQ_REQUIRE(workspace && workspace_bytes >= need, MIQ_EWORKSPACE, "miq_conv2d_general: needs %zu bytes of workspace (got %zu)",
                need, workspace_bytes);
    MIQ_REQUIRE(aligned(workspace, 256), MIQ_EALIGN, "miq_conv2d_general: workspace must be 256-byte aligned");
    int8_t* xp = static_cast<int8_t*>(workspace);
    int8_t* wp = xp + round_up(G.xp_bytes, 256);

    // the padded channels-last copy (border P all round) and the packed weight, one launch
    ConvPrepArgs pa{};
    pa.x = x; pa.xp = xp; pa.w = w; pa.wp = wp;
    pa.Cin = (int)Cin; pa.H = (int)H; pa.W = (int)W; pa.Hp = (int)G.Hp; pa.Wp = (int)G.Wp; pa.Cp = (int)G.Cp;
    pa.pad = (int)G.P; pa.cblocks = static_cast<int>(ceil_div(G.Cp, 64));
    pa.KH = (int)G.rows; pa.KW = (int)G.kw_run; pa.L = (int)G.L; pa.Lp = (int)G.Lp;   // (OIHW is [Cout][Cin][rows][kw_run] either way)
    pa.n_interior = static_cast<int>(ceil_div(H * W, 64) * pa.cblocks);
    pa.border_total = static_cast<uint32_t>(B * (2 * G.P * G.Wp + 2 * G.P * H));
    pa.weight_total = static_cast<uint32_t>(G.wp_bytes / 4);
    pa.sub = 1; pa.Ws = (int)W; pa.HWs = (int)(H * W);
    const int64_t extra = ceil_div(static_cast<int64_t>(pa.border_total) * ((G.Cp & 15) ? 1 : G.Cp / 16) + pa.weight_total, 1024);
    pa.n_extra = static_cast<int>(std::min<int64_t>(std::max<int64_t>(ceil_div(extra, B), 1), 4096));
    dim3 pgrid(static_cast<unsigned>(pa.n_interior + pa.n_extra), static_cast<unsigned>(B));
    if ((H * W) % 4 == 0 && aligned(x, 4))
        conv_prep_kernel<1><<<pgrid, 256, 0, st>>>(pa);
    else
        conv_prep_kernel<0><<<pgrid, 256, 0, st>>>(pa);
    if (int rc = check_launch("conv_prep_kernel")) return rc;

    const int64_t origin = ((G.P - pt) * G.Wp + (G.P - pl)) * G.Cp;   // where the first window starts in the copy
    ConvArgs a{};
    a.xp = xp + origin;
    a.wp = wp;
    a.bias = bias;
    a.alpha = alpha;
    a.out = out;
    a.beta = beta;
    a.M = static_cast<int>(G.M);
    a.Cout = static_cast<int>(Cout);
    a.Kp = static_cast<int>(G.Kp);
    a.HoWo = static_cast<int>(G.Ho * G.Wo);
    a.Wo = static_cast<int>(G.Wo);
    a.img_stride = static_cast<int>(G.Hp * G.Wp * G.Cp);
    a.row_stride = static_cast<int>(dh * G.Wp * G.Cp);
    a.sx = static_cast<int>(sw * G.Cp);
    a.sy = static_cast<int>(sh * G.Wp * G.Cp);
    a.cps = static_cast<int>(G.Lp / kChunk);
    a.cps_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(a.cps)) + 1u;
    if (dw > 1) {
        a.kwt = static_cast<int>(KW);
        a.tap_stride = static_cast<int>(dw * G.Cp);
        a.kwt_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(KW)) + 1u;
    }
    a.xp_bytes = static_cast<uint32_t>(G.xp_bytes - origin);
    a.wp_bytes = static_cast<uint32_t>(G.wp_bytes);
    a.wide16 = out_dtype == MIQ_BF16 && a.HoWo % 8 == 0 && aligned(out, 16);
    a.stream_out = stream_choice(G.M * Cout * static_cast<int64_t>(esz) >= (192ll << 20) && (out_dtype != MIQ_BF16 || a.wide16));
    switch (out_dtype) {
        case MIQ_F32: return launch<kOutF32>(a, st);
        case MIQ_BF16: return launch<kOutBF16>(a, st);
        default: return launch<kOutI32>(a, st);
    }
}

}  // namespace miq

// ---- channels-last entry points ---------------------------------------------------------------------
extern "C" size_t miq_conv2d_i8_packed_weight_bytes(int64_t Cin, int64_t Cout, int64_t KH, int64_t KW) {
    if (Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return 0;
    const int64_t total = packed_weight_layout(Cin, Cout, KH, KW).total;
    return total < (1ll << 31) ? static_cast<size_t>(total) : 0;   // (miq_conv2d_i8_pack_weight answers MIQ_ERANGE)
}

extern "C" int miq_conv2d_i8_pack_weight(const int8_t* w, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, int8_t* wp,
                                         void* stream) {
    MIQ_REQUIRE(Cin > 0 && Cout >= 0 && KH > 0 && KW > 0, MIQ_EINVAL, "miq_conv2d_i8_pack_weight: bad shape");
    if (Cout == 0) return MIQ_OK;
    MIQ_REQUIRE(w && wp, MIQ_EINVAL, "miq_conv2d_i8_pack_weight: null pointer");
    MIQ_REQUIRE(miq::aligned(wp, 16), MIQ_EALIGN, "miq_conv2d_i8_pack_weight: wp must be 16-byte aligned");
    const PackedWeight pw = packed_weight_layout(Cin, Cout, KH, KW);
    MIQ_REQUIRE(pw.total < (1ll << 31), MIQ_ERANGE, "miq_conv2d_i8_pack_weight: packed weight beyond 2 GiB");
    const int64_t Cp = miq::round_up(Cin, kCinAlign), L = KW * Cp, Lp = miq::round_up(L, kChunk);
    const uint32_t total = static_cast<uint32_t>(pw.rows_bytes / 4);
    const int wgrid = static_cast<int>(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
    pack_conv_weight_kernel<<<wgrid, 256, 0, miq::as_stream(stream)>>>(w, wp, total, (int)Cin, (int)KH, (int)KW, (int)Cp, (int)L,
                                                                       (int)Lp);
    if (int rc = miq::check_launch("pack_conv_weight_kernel")) return rc;
    if (pw.planar_at) {
        const uint32_t ptotal = static_cast<uint32_t>(pw.planar_bytes / 4);
        const int pgrid = static_cast<int>(ptotal / 256 + 1 < 4096 ? ptotal / 256 + 1 : 4096);
        pack_conv_weight_planar_kernel<<<pgrid, 256, 0, miq::as_stream(stream)>>>(w, wp + pw.planar_at, ptotal, (int)Cin, (int)Cout);
        return miq::check_launch("pack_conv_weight_planar_kernel");
    }
    return MIQ_OK;
}

extern "C" int miq_conv2d_i8_nhwc(const int8_t* x, const int8_t* wp, int64_t B, int64_t Cin, int64_t H, int64_t W,
                                  int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, const float* bias,
                                  const float* alpha, float beta, void* out, int out_dtype, int64_t out_pad, void* stream) {
    ConvGeom G;
    MIQ_REQUIRE(make_geom(B, Cin, H, W, Cout, KH, KW, stride, pad, G, false) && out_pad >= 0, MIQ_EINVAL,
                "miq_conv2d_i8_nhwc: bad geometry (B=%lld Cin=%lld H=%lld W=%lld Cout=%lld KH=%lld KW=%lld stride=%lld pad=%lld)",
                (long long)B, (long long)Cin, (long long)H, (long long)W, (long long)Cout, (long long)KH,
                (long long)KW, (long long)stride, (long long)pad);
    MIQ_REQUIRE(out_dtype == MIQ_F32 || out_dtype == MIQ_BF16 || out_dtype == MIQ_I32 || out_dtype == MIQ_I8, MIQ_EINVAL,
                "miq_conv2d_i8_nhwc: out_dtype must be MIQ_I8, MIQ_F32, MIQ_BF16 or MIQ_I32 (got %d)", out_dtype);
    MIQ_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, MIQ_EINVAL,
                "miq_conv2d_i8_nhwc: channel counts must be multiples of 4 (Cin=%lld Cout=%lld)", (long long)Cin, (long long)Cout);
    if (G.M == 0 || Cout == 0) return MIQ_OK;
    MIQ_REQUIRE(x && wp && out, MIQ_EINVAL, "miq_conv2d_i8_nhwc: null pointer");
    MIQ_REQUIRE(out_dtype == MIQ_I32 || alpha, MIQ_EINVAL, "miq_conv2d_i8_nhwc: alpha is required for a dequantised output");
    MIQ_REQUIRE(out_pad < (1ll << 15), MIQ_ERANGE, "miq_conv2d_i8_nhwc: out_pad must stay below 32768");
    const int64_t Hop = G.Ho + 2 * out_pad, Wop = G.Wo + 2 * out_pad;
    MIQ_REQUIRE(G.xp_bytes < (1ll << 31) && G.wp_bytes < (1ll << 31) && G.M < (1ll << 31) && B < 65536 && G.Hp < 65536 &&
                    miq::sat_product({B, Hop, Wop, Cout}) < (1ll << 31),
                MIQ_ERANGE, "miq_conv2d_i8_nhwc: activations, packed weights and output are indexed with 31 bits");
    MIQ_REQUIRE(G.Kp / kChunk <= 4096, MIQ_ERANGE, "miq_conv2d_i8_nhwc: KH*KW*Cin too large (%lld > 65536)", (long long)G.Kp);
    const size_t esz = out_dtype == MIQ_I8 ? 1 : out_dtype == MIQ_BF16 ? 2 : 4;
    MIQ_REQUIRE(miq::aligned(x, 16) && miq::aligned(wp, 16) && miq::aligned(out, 4 * esz) && miq::aligned(alpha, 4) &&
                    miq::aligned(bias, 4),
                MIQ_EALIGN, "miq_conv2d_i8_nhwc: x and wp must be 16-byte aligned, out aligned to four of its elements");

    hipStream_t st = miq::as_stream(stream);
    // 3 x 3 / stride 1 over whole 64-channel groups: the kernel that keeps a tile's pixel patch in LDS, where its
    // tiles are mostly pixels and the patch fits (the conditions of the NCHW entry point, with the caller's row
    // pitch Wp in place of the workspace's).
    const PackedWeight pw = packed_weight_layout(Cin, Cout, KH, KW);
    if (pw.planar_at && Cout % 64 == 0 && stride == 1 && G.M >= 4 * kTM && miq::aligned(out, 16) && miq::aligned(alpha, 16) &&
        miq::aligned(bias, 16) && !(dev_switch("MIQ_CONV_PLANAR") && dev_switch("MIQ_CONV_PLANAR")[0] == '0')) {
        const int64_t HoWo = G.Ho * G.Wo;
        int64_t group = 1;
        if (patch_tiling(HoWo, G.Wo, G.Hp, G.Wp, kPlaneSlots, group) && HoWo < 65536) {
            PlanarArgs pa{};
            pa.xp = x; pa.wp = wp + pw.planar_at; pa.bias = bias; pa.alpha = alpha; pa.out = out; pa.beta = beta;
            pa.Cout = (int)Cout; pa.ncg = (int)(Cin / 64); pa.tiles_n = (int)(Cout / 64);
            pa.pitch = (int)G.Wp; pa.Hp = (int)G.Hp; pa.Wo = (int)G.Wo; pa.HoWo = (int)HoWo;
            pa.wo_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(G.Wo)) + 1u;
            pa.xp_bytes = (uint32_t)G.xp_bytes; pa.wp_bytes = (uint32_t)pw.planar_bytes;
            pa.cin = (int)Cin;
            pa.o_pix = (int)Cout; pa.o_row = (int)(Wop * Cout); pa.o_img = (int)(Hop * Wop * Cout);
            pa.o_off = (int)((out_pad * Wop + out_pad) * Cout);
            if (const char* sp = dev_switch("MIQ_PLANAR_STAMPS"))   // address of a device buffer of 512 * 64 uint64
                pa.stamps = reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 0));
            MIQ_REQUIRE(B * miq::ceil_div(HoWo, kTM) * pa.tiles_n < (1ll << 31), MIQ_ERANGE, "miq_conv2d_i8_nhwc: too many tiles");
            set_patch_units(pa, B, HoWo, G.Hp, G.Wp, group);
            switch (out_dtype) {
                case MIQ_F32: return launch_planar<kOutF32, true>(pa, false, st);
                case MIQ_BF16: return launch_planar<kOutBF16, true>(pa, false, st);
                case MIQ_I32: return launch_planar<kOutI32, true>(pa, false, st);
                default: return launch_planar<kOutI8, true>(pa, false, st);
            }
        }
    }

    if (stem_shape(G) && out_dtype != MIQ_I32 && Cout % 16 == 0 && miq::aligned(out, 16) && miq::aligned(alpha, 16) && miq::aligned(bias, 16) &&
        !(dev_switch("MIQ_CONV_STEM") && dev_switch("MIQ_CONV_STEM")[0] == '0')) {
        StemArgs sa{};
        sa.xp = x; sa.wp = wp; sa.bias = bias; sa.alpha = alpha; sa.out = out; sa.beta = beta;
        sa.Cout = (int)Cout; sa.KH = (int)KH; sa.Kp = (int)G.Kp;
        sa.Hp = (int)G.Hp; sa.Wp = (int)G.Wp; sa.Ho = (int)G.Ho; sa.Wo = (int)G.Wo;
        sa.xp_bytes = (uint32_t)G.xp_bytes; sa.wp_bytes = (uint32_t)G.wp_bytes;
        sa.o_pix = (int)Cout; sa.o_row = (int)(Wop * Cout); sa.o_img = (int)(Hop * Wop * Cout);
        sa.o_off = (int)((out_pad * Wop + out_pad) * Cout);
        switch (out_dtype) {
            case MIQ_F32: return launch_stem<kOutF32>(sa, B, st);
            case MIQ_BF16: return launch_stem<kOutBF16>(sa, B, st);
            default: return launch_stem<kOutI8>(sa, B, st);
        }
    }

    ConvArgs a{};
    a.xp = x; a.wp = wp; a.bias = bias; a.alpha = alpha; a.out = out; a.beta = beta;
    a.M = static_cast<int>(G.M);
    a.Cout = static_cast<int>(Cout);
    a.Kp = static_cast<int>(G.Kp);
    a.HoWo = static_cast<int>(G.Ho * G.Wo);
    a.Wo = static_cast<int>(G.Wo);
    a.img_stride = static_cast<int>(G.Hp * G.Wp * G.Cp);
    a.row_stride = static_cast<int>(G.Wp * G.Cp);
    a.sx = static_cast<int>(stride * G.Cp);
    a.sy = static_cast<int>(stride * G.Wp * G.Cp);
    a.cps = static_cast<int>(G.Lp / kChunk);
    a.cps_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(a.cps)) + 1u;
    a.xp_bytes = static_cast<uint32_t>(G.xp_bytes);
    a.wp_bytes = static_cast<uint32_t>(G.wp_bytes);
    a.o_pix = static_cast<int>(Cout);
    a.o_row = static_cast<int>(Wop * Cout);
    a.o_img = static_cast<int>(Hop * Wop * Cout);
    a.o_off = static_cast<int>((out_pad * Wop + out_pad) * Cout);
    switch (out_dtype) {
        case MIQ_F32: return launch_cl<kOutF32>(a, st);
        case MIQ_BF16: return launch_cl<kOutBF16>(a, st);
        case MIQ_I32: return launch_cl<kOutI32>(a, st);
        default: return launch_cl<kOutI8>(a, st);
    }
}
