// Device-side building blocks shared by the int8 GEMM and the implicit-GEMM conv:
// LDS tile geometry, the bank-conflict-free swizzle, LDS-DMA staging, MFMA fragment reads
// and the per-channel dequant epilogue.
//
// Geometry.  Both operands are K-contiguous int8 rows.  A K-tile is 128 bytes deep, so an
// LDS tile is `rows` x 128 B, cut in 16-byte chunks (8 per row).  `v_mfma_i32_16x16x64_i8`
// takes 16 bytes per lane per operand: lane l supplies row (l & 15) and the 16-byte chunk
// (l >> 4) of a 64-deep k-step, for both operands alike, so the k order inside the
// instruction never has to be known: both operands use the same chunk -> lane map and an
// integer sum does not depend on its order.
//
// Swizzle.  `ds_read_b128` is served in 16-lane groups over a 256-byte bank row, i.e. 16
// chunk slots shared by two 128-byte tile rows.  Chunk c of row r is stored at slot
// c ^ ((r >> 1) & 7) of its row, which makes every group of a fragment read hit 16
// different slots.  LDS-DMA writes LDS linearly (wave base + lane * 16), so the swizzle is
// applied to the per-lane SOURCE address when staging and again to the read address.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace miq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

constexpr int kTileK = 128;                 // bytes (= int8 elements) per K-tile
constexpr int kChunk = 16;                  // bytes per LDS-DMA piece / MFMA operand per lane
constexpr int kChunksPerRow = kTileK / kChunk;
constexpr uint32_t kOob = 0x80000000u;      // buffer offset that is always out of range (-> reads 0)

using lds_ptr = __attribute__((address_space(3))) void*;

// Raw buffer resource over `bytes` bytes at p (bytes < 2^31).  Loads past the end return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// 16 bytes global -> LDS, no register staging.  `dst` is the wave's base (lane l lands at
// dst + 16*l); `voffset` is this lane's byte offset into the buffer (kOob -> zeros).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, uint32_t voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)dst, 16, voffset, 0, 0, 0);
}

// The same with a wave-uniform byte offset added in the scalar operand of the instruction: a lane's
// `voffset` can then stay fixed for a whole kernel while the K-tile and the row group advance in
// scalar registers.  The range check covers `voffset` only (the scalar offset is not part of it), so
// this form is for tiles that lie wholly inside their matrix.
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, char* dst, uint32_t voffset, uint32_t soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)dst, 16, voffset, soffset, 0, 0);
}

// Byte offset, inside a rows x 128 B tile, of logical chunk `c` (0..7) of row `r`.
__device__ __forceinline__ int tile_offset(int r, int c) {
    return r * kTileK + ((c ^ ((r >> 1) & 7)) * kChunk);
}

// Per-lane part of a fragment read address for k-step ks (0 or 1) of a K-tile.  The fragment's
// first row must be a multiple of 16, so the swizzle term depends on the lane only.
__device__ __forceinline__ int frag_lane_offset(int lane, int ks) {
    const int r = lane & 15;
    const int c = ks * 4 + (lane >> 4);
    return r * kTileK + ((c ^ (r >> 1)) * kChunk);
}

__device__ __forceinline__ v4i lds_read16(const char* p) { return *reinterpret_cast<const v4i*>(p); }

// Data that one workgroup hands to another inside a launch (partial results behind an arrival counter).  The eight XCDs
// have an L2 each, so ordinary stores may sit in the writer's L2 and ordinary loads may hit stale lines of the reader's:
// making them visible with device-scope FENCES works but writes back and invalidates a whole L2 per fence (measured: a
// split-K GEMM ran 4-5x slower).  Instead the hand-over data itself moves with device-scope relaxed atomic stores and
// loads (written through to / read from the level all XCDs share), the writer waits for its stores to complete
// (s_waitcnt vmcnt(0)) before it takes its ticket with a device-scope atomic add, and the reader loads after it has seen
// the last ticket.  Nothing else is flushed.
__device__ __forceinline__ void store_shared(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_shared(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int load_shared(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_shared(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// four int32 at an 8-byte aligned address, as two 64-bit accesses
__device__ __forceinline__ void store_shared4(int* p, v4i v) {
    typedef unsigned long long u64;
    u64* q = reinterpret_cast<u64*>(p);
    __hip_atomic_store(q, static_cast<u64>(static_cast<uint32_t>(v[0])) | (static_cast<u64>(static_cast<uint32_t>(v[1])) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, static_cast<u64>(static_cast<uint32_t>(v[2])) | (static_cast<u64>(static_cast<uint32_t>(v[3])) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ v4i load_shared4(const int* p) {
    typedef unsigned long long u64;
    const u64* q = reinterpret_cast<const u64*>(p);
    const u64 a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v4i{static_cast<int>(a), static_cast<int>(a >> 32), static_cast<int>(b), static_cast<int>(b >> 32)};
}
// Every store this wave issued has completed; then the workgroup's.  Returns the ticket thread 0 drew from `counter`.
__device__ __forceinline__ unsigned take_ticket(unsigned* counter) {
    __shared__ unsigned ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return ticket;
}

// XCD-aware tile order: the dispatcher deals workgroups round-robin over the 8 XCDs, so
// ids b, b+8, ... share an L2.  Give each XCD a contiguous range of tile ids (bijective for
// any grid size), then walk that range in bands of `group` tile-rows so that the tiles in
// flight on one XCD share operand panels.
__device__ __forceinline__ void tile_coords(int pid, int nwg, int tiles_m, int tiles_n, int group, int& tm,
                                            int& tn) {
    const int q = nwg / 8, r = nwg % 8, xcd = pid % 8;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pid / 8;
    const int band = group * tiles_n;
    const int first_m = (id / band) * group;
    const int gsz = tiles_m - first_m < group ? tiles_m - first_m : group;
    const int in_band = id % band;
    tm = first_m + in_band % gsz;
    tn = in_band / gsz;
}

enum : int { kOutF32 = 0, kOutBF16 = 2, kOutI32 = 3, kOutI8 = 4 };  // == MIQ_F32 / MIQ_BF16 / MIQ_I32 / MIQ_I8

// fp32 -> bf16, round-half-even, NaN stays NaN: the hardware conversion (v_cvt_pk_bf16_f32, two values
// per instruction) instead of a dozen integer operations per value.
typedef __bf16 miq_bf16x2 __attribute__((ext_vector_type(2)));
typedef float miq_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(miq_f32x2{a, b}, miq_bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float x) {
    return __builtin_bit_cast(uint16_t, static_cast<__bf16>(x));
}
__device__ __forceinline__ uint2 pack4_bf16(float a, float b, float c, float d) {
    uint2 p;
    p.x = pack2_bf16(a, b);
    p.y = pack2_bf16(c, d);
    return p;
}

// Requantisation of a dequantised value to int8, the per-tensor form of Qop.quantize with the scale folded into
// alpha (quant/Weight_Only/Qop.py:154-157): clamp(rint(y), -128, 127), round-half-even; four of them as one dword.
// Clamping first (the bounds are integers, so the order does not matter) lets one addition do the rounding and the
// conversion: y + 1.5 * 2^23 has an ulp of 1, so the sum is rounded half-even to an integer whose two's complement low
// byte sits in the low byte of the float's mantissa.  Three byte permutes then pack four of them.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t requant4_i8(float a, float b, float c, float d) {
#pragma clang fp contract(off)
    auto clamp = [](float y) { return __builtin_amdgcn_fmed3f(y, -128.0f, 127.0f); };
    const v2f magic{12582912.0f, 12582912.0f};
    const v2f lo2 = v2f{clamp(a), clamp(b)} + magic, hi2 = v2f{clamp(c), clamp(d)} + magic;   // two packed additions
    const uint32_t lo = __builtin_amdgcn_perm(__float_as_uint(lo2[1]), __float_as_uint(lo2[0]), 0x0c0c0400u);   // a.b0, b.b0, 0, 0
    const uint32_t hi = __builtin_amdgcn_perm(__float_as_uint(hi2[1]), __float_as_uint(hi2[0]), 0x04000c0cu);   // 0, 0, c.b0, d.b0
    return lo | hi;
}

// One value requantised: the low byte of requant4_i8.
__device__ __forceinline__ int8_t requant1_i8(float y) {
#pragma clang fp contract(off)
    const float r = __builtin_amdgcn_fmed3f(y, -128.0f, 127.0f) + 12582912.0f;
    return static_cast<int8_t>(__float_as_uint(r) & 0xffu);
}

// The division of Qop.quantize's per-tensor form, round(y / scale) (quant/Weight_Only/Qop.py:155): IEEE, as the
// reference's tensor / tensor division; a scale of exactly 1 leaves y as it is.
__device__ __forceinline__ float requant_div(float y, float scale) { return scale == 1.0f ? y : __fdiv_rn(y, scale); }

// dequant1 on two values at once, as packed fp32 instructions (one rounding per step, no contraction).
__device__ __forceinline__ v2f dequant2(int a, int b, v2f bias, v2f alpha, v2f beta) {
#pragma clang fp contract(off)
    v2f t{static_cast<float>(a), static_cast<float>(b)};
    t = t + bias;
    t = t * alpha;
    t = t + beta;
    return t;
}

// Two MFMA result fragments X and Y, each packed as four bf16 (8 bytes) per lane, where the 16-lane row
// c = lane >> 4 of a fragment holds elements 4c .. 4c+3 of its 16 along the register axis.  Rows trade
// places (v_permlane16_swap: the odd rows of one operand with the even rows of the other) so that every
// lane ends up with eight consecutive elements of ONE fragment, 16 bytes: rows 0 and 2 hold elements 0-7
// and 8-15 of X, rows 1 and 3 those of Y.  paired_offset() is the element offset of a lane's eight, with
// Y's elements counted from 16 (Y follows X along the register axis).
__device__ __forceinline__ v4i pair_rows_bf16(uint2 x, uint2 y) {
    const auto lo = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
    return v4i{static_cast<int>(lo[0]), static_cast<int>(hi[0]), static_cast<int>(lo[1]), static_cast<int>(hi[1])};
}
__device__ __forceinline__ int paired_offset(int lane) {
    const int c = lane >> 4;
    return (c & 1) * 16 + (c >> 1) * 8;
}

// (float(acc) + bias) * alpha + beta with one rounding per step (no contraction), the order
// the reference's Python applies: Weight_Only/Quantizer.py:79-86, Weight_Activation/Quantizer.py:58-60.
__device__ __forceinline__ float dequant1(int acc, float bias, float alpha, float beta) {
    return __fadd_rn(__fmul_rn(__fadd_rn(static_cast<float>(acc), bias), alpha), beta);
}

__device__ __forceinline__ float dequant1(float acc, float bias, float alpha, float beta) {
    return __fadd_rn(__fmul_rn(__fadd_rn(acc, bias), alpha), beta);
}

// Epilogue of a 256 x (QN * 128) tile held as acc[qm][qn][i][j] by 8 waves (2 x 4), int32 or fp32 accumulators
// (Args: a kernel argument block with M, N, ldc, bias, alpha, beta, out; ldc = elements between two output
// rows).  D[i][j] of a fragment: i = 4*(lane>>4) + reg -> output channel, j = lane&15 -> row.
//
// bf16 outputs: a lane's four channels of one fragment are only 8 bytes, so the fragments j = 0, 1 of a
// row group (channels 0-15 and 16-31 of the wave's 32) first trade 16-lane rows (pair_rows_bf16); every
// lane then stores eight consecutive channels as one 16-byte vector: half the store instructions, and a
// row's 32 channels leave as one 64-byte segment.
//
// EXT (the int8 GEMM's argument block): g.by_row != 0 takes bias / alpha per output ROW m instead of per column n (the
// transposed product of a column-sharded layer, whose rows are the output channels); OUT == kOutI8 requantises the
// dequantised value, clamp(rint(y / g.out_div)), four consecutive columns as one dword.
//
// ADJ (QN == 2): the kernel staged the column operand so that a wave's two quadrants are NEIGHBOURS in the output, columns
// wn * 64 + qn * 32 .. + 32 of the tile instead of qn * 128 + wn * 32 .. + 32: a row of a wave is then 64 consecutive columns,
// 128 bytes at bf16, and the bf16 epilogue stores whole cache lines as the fp32 one does.
template <int OUT, bool NT = true, bool WIDE16 = true, int QN = 2, bool EXT = false, bool ADJ = false, typename Args, typename V4>
__device__ __forceinline__ void store_tile_256(const Args& g, const V4 (&acc)[2][QN][4][2], int m0, int n0, int wm,
                                               int wn, int lane) {
    static_assert(!ADJ || QN == 2, "neighbouring quadrants: two of them");
    const bool vec_ok = (g.ldc & 3) == 0;
    bool by_row = false;
    if constexpr (EXT) by_row = g.by_row != 0;
    if constexpr (ADJ && OUT == kOutBF16 && WIDE16 && NT) {
        const int nb0 = n0 + wn * 64;
        if ((g.ldc & 7) == 0 && nb0 + 64 <= g.N) {   // wave-uniform
            // Per quadrant as below (the fragments j = 0, 1 trade 16-lane rows: eight consecutive columns a lane); then
            // the two quadrants trade halves between lanes r and r ^ 8 of every 16-lane row (one DPP rotation), so that a
            // store instruction writes eight rows x 128 bytes instead of sixteen rows x 64: a nontemporal store of half a
            // cache line is written out as a partial line (tools/nchw_write_probe.hip).
            float al[2][2][4], bi[2][2][4];
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = nb0 + qn * 32 + j * 16 + (lane >> 4) * 4 + e;
                        al[qn][j][e] = by_row ? 1.f : g.alpha[n];
                        bi[qn][j][e] = (by_row || !g.bias) ? 0.f : g.bias[n];
                    }
            const int r = lane & 15;
            const bool lo = r < 8;
            const int n = nb0 + (lo ? 0 : 32) + paired_offset(lane);
#pragma unroll
            for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int mb = m0 + qm * 128 + wm * 64 + i * 16;
                    const int m = mb + r, mc = m < g.M ? m : g.M - 1;   // (rows past the edge compute and trade, and store nothing)
                    float ra = 1.f, rb = 0.f;
                    if constexpr (EXT) {
                        if (by_row) {
                            ra = g.alpha[mc];
                            rb = g.bias ? g.bias[mc] : 0.f;
                        }
                    }
                    v4i v[2];
#pragma unroll
                    for (int qn = 0; qn < 2; ++qn) {
                        uint2 p[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const V4 a = acc[qm][qn][i][j];
                            float y[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                y[e] = by_row ? dequant1(a[e], rb, ra, g.beta) : dequant1(a[e], bi[qn][j][e], al[qn][j][e], g.beta);
                            p[j] = pack4_bf16(y[0], y[1], y[2], y[3]);
                        }
                        v[qn] = pair_rows_bf16(p[0], p[1]);
                    }
                    v4i first, second;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int send = lo ? v[1][e] : v[0][e];
                        const int recv = __builtin_amdgcn_update_dpp(0, send, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
                        first[e] = lo ? v[0][e] : recv;
                        second[e] = lo ? recv : v[1][e];
                    }
                    const int m1 = mb + (r & 7), m2 = m1 + 8;
                    uint16_t* o1 = static_cast<uint16_t*>(g.out) + static_cast<size_t>(m1) * g.ldc + n;
                    if (m1 < g.M) __builtin_nontemporal_store(first, reinterpret_cast<v4i*>(o1));
                    if (m2 < g.M) __builtin_nontemporal_store(second, reinterpret_cast<v4i*>(o1 + static_cast<size_t>(8) * g.ldc));
                }
            return;
        }
    }
#pragma unroll
    for (int qn = 0; qn < QN; ++qn) {
        const int nb = ADJ ? n0 + wn * 64 + qn * 32 : n0 + qn * 128 + wn * 32;   // first of this wave's 32 channels in the quadrant
        float al[2][4], bi[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nb + j * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                al[j][e] = 1.f;
                bi[j][e] = 0.f;
                if constexpr (OUT != kOutI32) {
                    if (!by_row && n + e < g.N) {
                        al[j][e] = g.alpha[n + e];
                        if (g.bias) bi[j][e] = g.bias[n + e];
                    }
                }
            }
        }
        // the parameters of output row m (by_row): one value for all of a lane's columns
        auto row_params = [&](int m, float (&a4)[4], float (&b4)[4], int j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a4[e] = al[j][e];
                b4[e] = bi[j][e];
            }
            if constexpr (EXT && OUT != kOutI32) {
                if (by_row) {
                    const float a = g.alpha[m], b = g.bias ? g.bias[m] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a4[e] = a;
                        b4[e] = b;
                    }
                }
            }
        };
        if constexpr (OUT == kOutBF16 && WIDE16) {
            if ((g.ldc & 7) == 0 && nb + 32 <= g.N) {   // wave-uniform
                const int n = nb + paired_offset(lane);
#pragma unroll
                for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = m0 + qm * 128 + wm * 64 + i * 16 + (lane & 15);
                        const int mc = m < g.M ? m : g.M - 1;   // (rows past the edge compute and trade, and store nothing)
                        uint2 p[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const V4 a = acc[qm][qn][i][j];
                            float a4[4], b4[4];
                            row_params(mc, a4, b4, j);
                            p[j] = pack4_bf16(dequant1(a[0], b4[0], a4[0], g.beta), dequant1(a[1], b4[1], a4[1], g.beta),
                                              dequant1(a[2], b4[2], a4[2], g.beta), dequant1(a[3], b4[3], a4[3], g.beta));
                        }
                        const v4i v = pair_rows_bf16(p[0], p[1]);
                        if (m >= g.M) continue;
                        v4i* o = reinterpret_cast<v4i*>(static_cast<uint16_t*>(g.out) + static_cast<size_t>(m) * g.ldc + n);
                        if constexpr (NT) {
                            __builtin_nontemporal_store(v, o);
                        } else {
                            *o = v;
                        }
                    }
                continue;
            }
        }
        if constexpr ((OUT == kOutF32 || OUT == kOutI32) && NT) {
            // Whole cache lines per store instruction.  As the accumulators lie, an instruction writes sixteen rows x 64
            // bytes, and a nontemporal store of half a line is written out as a partial line (3.2 TB/s against 5.3 TB/s for
            // whole lines, tools/nchw_write_probe.hip).  The two 16-column fragments of a row trade halves between lanes r
            // and r ^ 8 of each 16-lane row (one DPP rotation): an instruction then writes eight rows x 128 bytes.
            if (vec_ok && nb + 32 <= g.N) {   // wave-uniform
                const int r = lane & 15;
                const bool lo = r < 8;
                const int n = nb + (lo ? 0 : 16) + (lane >> 4) * 4;
#pragma unroll
                for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int mb = m0 + qm * 128 + wm * 64 + i * 16;
                        const int m = mb + r, mc = m < g.M ? m : g.M - 1;   // (rows past the edge compute and trade, and store nothing)
                        int y[2][4];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const V4 a = acc[qm][qn][i][j];
                            if constexpr (OUT == kOutI32) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) y[j][e] = a[e];
                            } else {
                                float a4[4], b4[4];
                                row_params(mc, a4, b4, j);
#pragma unroll
                                for (int e = 0; e < 4; ++e) y[j][e] = __float_as_int(dequant1(a[e], b4[e], a4[e], g.beta));
                            }
                        }
                        v4i first, second;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int send = lo ? y[1][e] : y[0][e];
                            const int recv = __builtin_amdgcn_update_dpp(0, send, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
                            first[e] = lo ? y[0][e] : recv;
                            second[e] = lo ? recv : y[1][e];
                        }
                        const int m1 = mb + (r & 7), m2 = m1 + 8;
                        int* o1 = static_cast<int*>(g.out) + static_cast<size_t>(m1) * g.ldc + n;
                        if (m1 < g.M) __builtin_nontemporal_store(first, reinterpret_cast<v4i*>(o1));
                        if (m2 < g.M) __builtin_nontemporal_store(second, reinterpret_cast<v4i*>(o1 + static_cast<size_t>(8) * g.ldc));
                    }
                continue;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nb + j * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + qm * 128 + wm * 64 + i * 16 + (lane & 15);
                    if (m >= g.M || n >= g.N) continue;
                    const V4 a = acc[qm][qn][i][j];
                    const size_t idx = static_cast<size_t>(m) * g.ldc + n;
                    if constexpr (OUT == kOutI32) {
                        int* o = static_cast<int*>(g.out) + idx;
                        if (vec_ok && n + 3 < g.N) {
                            if constexpr (NT) {
                                __builtin_nontemporal_store(a, reinterpret_cast<v4i*>(o));
                            } else {
                                *reinterpret_cast<v4i*>(o) = a;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < g.N) o[e] = a[e];
                        }
                    } else {
                        float y[4], a4[4], b4[4];
                        row_params(m, a4, b4, j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = dequant1(a[e], b4[e], a4[e], g.beta);
                        if constexpr (OUT == kOutF32) {
                            float* o = static_cast<float*>(g.out) + idx;
                            if (vec_ok && n + 3 < g.N) {
                                // Nontemporal: the output is written once and read by a later kernel; kept
                                // out of the caches it costs the operand stream nothing (8 % on 8192^3
                                // when successive layers write different buffers).
                                if constexpr (NT) {
                                    __builtin_nontemporal_store(v4f{y[0], y[1], y[2], y[3]}, reinterpret_cast<v4f*>(o));
                                } else {
                                    *reinterpret_cast<v4f*>(o) = v4f{y[0], y[1], y[2], y[3]};
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (n + e < g.N) o[e] = y[e];
                            }
                        } else if constexpr (OUT == kOutI8) {
                            int8_t* o = static_cast<int8_t*>(g.out) + idx;
                            float d = 1.0f;
                            if constexpr (EXT) d = g.out_div;
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = requant_div(y[e], d);
                            if (vec_ok && n + 3 < g.N) {
                                *reinterpret_cast<uint32_t*>(o) = requant4_i8(y[0], y[1], y[2], y[3]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (n + e < g.N) o[e] = requant1_i8(y[e]);
                            }
                        } else {
                            uint16_t* o = static_cast<uint16_t*>(g.out) + idx;
                            if (vec_ok && n + 3 < g.N) {
                                // cached: as 8-byte nontemporal stores this epilogue ran 23 % slower (measured)
                                *reinterpret_cast<uint2*>(o) = pack4_bf16(y[0], y[1], y[2], y[3]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (n + e < g.N) o[e] = f32_to_bf16_bits(y[e]);
                            }
                        }
                    }
                }
        }
    }
}

}  // namespace miq
