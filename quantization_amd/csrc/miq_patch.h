// What the LDS-resident-patch convolution kernels share (the int8 one in miq_conv2d_i8.hip, the weight-only one on
// floating-point activations in miq_gemm_w8f.hip): the tile geometry, how an image is cut into tiles, and the counted wait.
#pragma once
#include <algorithm>
#include <cstdint>

#include "miq_common.h"
#include "miq_gemm_core.h"

namespace miq {

// conv3x3_patch_kernel's tile: kTM output pixels of one image x 64 output channels per workgroup; one stage =
// the pixel patch of a 64-channel group (4 chunk planes of kPlaneSlots 16-byte slots) + that group's weights.
constexpr int kTM = 448;                                // 28 fragments: 4 wave rows x 7
constexpr int kPlaneSlots = 704;                        // >= (rows a tile touches + 2) * pitch + 2

// How the patch kernel tiles an image of HoWo output pixels whose padded copy has Hp rows of `pitch` slots: tiles of kTM
// consecutive pixels of one image (group = 1) if the padded rows a tile touches (+2) fit `slots` and the tiles are mostly
// pixels, or, for images of at most half a tile, `group` whole images per tile (their padded copies are one run in memory).
// False if neither fits.
inline bool patch_tiling(int64_t HoWo, int64_t Wo, int64_t Hp, int64_t pitch, int64_t slots, int64_t& group) {
    group = 1;
    if (2 * HoWo <= kTM) {
        group = std::min<int64_t>(kTM / HoWo, (slots - 2) / (Hp * pitch));
        const bool ok = group >= 2 && 5 * group * HoWo >= 4 * kTM;   // ... and still mostly pixels
        if (!ok) group = 1;
        return ok;
    }
    int64_t rows = 0;
    for (int64_t p0 = 0; p0 < HoWo; p0 += kTM) {
        const int64_t p1 = std::min<int64_t>(p0 + kTM, HoWo) - 1;
        rows = std::max(rows, p1 / Wo - p0 / Wo + 3);
    }
    const int64_t tiles_img = ceil_div(HoWo, kTM);
    return rows * pitch + 2 <= slots && 5 * HoWo >= 4 * tiles_img * kTM;
}

// The fields of PlanarArgs that describe the units, from the tiling above.
template <typename Args>
void set_patch_units(Args& pa, int64_t B, int64_t HoWo, int64_t Hp, int64_t pitch, int64_t group) {
    pa.group = (int)group;
    pa.B = (int)B;
    pa.img_slots = (int)(Hp * pitch);
    pa.pix_end = (int)(group * HoWo);
    pa.hw_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(HoWo)) + 1u;
    pa.tiles_img = group > 1 ? 1 : (int)ceil_div(HoWo, kTM);
    pa.units = (int)(ceil_div(B, group) * pa.tiles_img * pa.tiles_n);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, 16] (the instruction takes an immediate).
__device__ __forceinline__ void wait_vmcnt_upto16(int n) {
    switch (__builtin_amdgcn_readfirstlane(n)) {
#define MIQ_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        MIQ_VMCNT_CASE(1) MIQ_VMCNT_CASE(2) MIQ_VMCNT_CASE(3) MIQ_VMCNT_CASE(4) MIQ_VMCNT_CASE(5) MIQ_VMCNT_CASE(6)
        MIQ_VMCNT_CASE(7) MIQ_VMCNT_CASE(8) MIQ_VMCNT_CASE(9) MIQ_VMCNT_CASE(10) MIQ_VMCNT_CASE(11) MIQ_VMCNT_CASE(12)
        MIQ_VMCNT_CASE(13) MIQ_VMCNT_CASE(14) MIQ_VMCNT_CASE(15) MIQ_VMCNT_CASE(16)
#undef MIQ_VMCNT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

}  // namespace miq
