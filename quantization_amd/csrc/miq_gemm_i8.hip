// int8 x int8 -> int32 GEMM on CDNA4 MFMA (v_mfma_i32_16x16x64_i8) with a fused per-channel
// dequant epilogue:  out[m,n] = (float(sum_k A[m,k] * W[n,k]) + bias[n]) * alpha[n] + beta.
//
// Both operands are K-contiguous (A [M,K], W [N,K]).  A workgroup owns a BM x BN output
// tile and walks K in 128-byte tiles, double-buffered in LDS and filled by LDS-DMA
// (buffer_load ... lds, 16 B per lane, zero-filled past the matrix edges by the buffer
// range check).  The weight tile is the MFMA's first operand and the activation tile the
// second, so a lane's four accumulator registers are four consecutive output channels of
// one output row: the epilogue loads alpha/bias as one 16-byte vector and stores 16 bytes.
//
// Replaces linear_w8a8_kernel (cuda_kernels/linear_int8/linear_w8a8_kernel.cu:5-22) and the
// dequantisation the reference's Python applies after its matmul (see include/miq.h).
#include <mutex>

#include "miq_common.h"
#include "miq_gemm_core.h"

// Scratch of the allocation-free entry point miq_gemm_i8_noalloc: one array per GPU, part of the loaded code
// object, held until the process ends.  Host code never passes the array's name to a kernel or a memset (on the
// host the name is a stand-in, not a device address): scratch_for_current_device() asks the runtime for the
// address on the current device once and hands that pointer on.
constexpr size_t kScratchBytes = 32u << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {

using namespace miq;

struct GemmArgs {
    const int8_t* A;  // [M,K]
    const int8_t* W;  // [N,K]
    const float* bias;
    const float* alpha;
    void* out;
    float beta;
    int M, N, K;
    int ldc;           // elements between two rows of `out` (N, or the full width when this is a column block)
    int tiles_m, tiles_n;
    int kt_per_split;  // K-tiles each blockIdx.y slice sums (split-K: raw int32 slabs out); 0 = all of K
    unsigned* stamps;  // timing-only builds: per-wave cycle totals of the main loop's segments
    int by_row;        // != 0: bias / alpha are indexed by the output ROW m (the transposed product: its rows are the channels)
    float out_div;     // int8 outputs: clamp(rint(y / out_div)); 1 = the dequantised value itself
};

// One output element from its int32 sum: the epilogue of the kernels that finish one element per thread.
template <int OUT>
__device__ __forceinline__ void store_one(const GemmArgs& g, int m, int n, int sum) {
    const size_t o = static_cast<size_t>(m) * g.ldc + n;
    if constexpr (OUT == kOutI32) {
        static_cast<int*>(g.out)[o] = sum;
    } else {
        const int pi = g.by_row ? m : n;
        const float y = dequant1(sum, g.bias ? g.bias[pi] : 0.0f, g.alpha[pi], g.beta);
        if constexpr (OUT == kOutF32) {
            static_cast<float*>(g.out)[o] = y;
        } else if constexpr (OUT == kOutI8) {
            static_cast<int8_t*>(g.out)[o] = requant1_i8(requant_div(y, g.out_div));
        } else {
            static_cast<uint16_t*>(g.out)[o] = f32_to_bf16_bits(y);
        }
    }
}

// Fills one LDS stage: rows [row0, row0 + ROWS) x K-tile kt of a [nrows, K] int8 matrix.
// Each wave-instruction moves 8 rows x 128 B; the chunk a lane fetches is the one its LDS
// slot must hold under the swizzle.
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* tile, int row0, int nrows, int K,
                                           int kt, int wave, int lane) {
    static_assert(ROWS % 8 == 0, "a piece is 8 rows");
    constexpr int kPieces = ROWS / 8;
    constexpr int kIters = (kPieces + NWAVES - 1) / NWAVES;
    const int r_in = lane >> 3;      // row within the 8-row piece
    const int slot = lane & 7;       // chunk slot within the row
#pragma unroll
    for (int i = 0; i < kIters; ++i) {
        const int piece = wave + i * NWAVES;  // wave-uniform
        if (kPieces % NWAVES != 0 && piece >= kPieces) break;
        const int r = piece * 8 + r_in;
        const int c = slot ^ ((r >> 1) & 7);
        const int k = kt * kTileK + c * kChunk;
        const int row = row0 + r;
        const uint32_t off = (row < nrows && k < K) ? static_cast<uint32_t>(row) * static_cast<uint32_t>(K) + k
                                                    : kOob;
        dma16(rsrc, tile + piece * (8 * kTileK), off);
    }
}

// STAGES = 2: double buffer, one __syncthreads per K-tile (which also drains the DMA: the fetch of
// tile t+1 only overlaps the MFMAs of tile t).  STAGES > 2: a ring of K-tile buffers with STAGES-1
// tiles in flight and a counted vmcnt, for the thin problems (M <= 128) whose K-tile of MFMAs is far
// shorter than a fetch: per K-tile, wait until all but the newest STAGES-2 tiles have landed, one raw
// barrier (tile t visible to every wave, tile t-1 read by every wave), re-stage tile t-1's buffer with
// tile t+STAGES-1, then the MFMAs of tile t.
template <int BM, int BN, int WAVES_M, int WAVES_N, int OUT, int STAGES = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_i8_kernel(GemmArgs g) {
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // per-wave output tile
    constexpr int FM = WM / 16, FN = WN / 16;            // 16x16 fragments per wave
    constexpr int STAGE = (BM + BN) * kTileK;            // bytes per pipeline stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, 8, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, static_cast<uint32_t>(g.M) * g.K);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.W, static_cast<uint32_t>(g.N) * g.K);

    v4i acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    // Split-K: slice blockIdx.y sums K-tiles [kt0, kt0 + nt) into its own int32 slab of the output.
    const int nt_all = (g.K + kTileK - 1) / kTileK;
    const int kt0 = g.kt_per_split ? static_cast<int>(blockIdx.y) * g.kt_per_split : 0;
    const int nt = g.kt_per_split ? (nt_all - kt0 < g.kt_per_split ? nt_all - kt0 : g.kt_per_split) : nt_all;
    if (g.kt_per_split) {   // slabs are dense [splits][M][N]
        g.out = static_cast<int*>(g.out) + static_cast<size_t>(blockIdx.y) * g.M * g.N;
        g.ldc = g.N;
    }
    const int lo0 = frag_lane_offset(lane, 0), lo1 = frag_lane_offset(lane, 1);

    auto stage = [&](char* buf, int kt) {
        stage_tile<BM, NWAVES>(ra, buf, m0, g.M, g.K, kt, wave, lane);
        stage_tile<BN, NWAVES>(rw, buf + BM * kTileK, n0, g.N, g.K, kt, wave, lane);
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
                    acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (STAGES == 2) {
        stage(smem, kt0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt) stage(smem + ((t + 1) & 1) * STAGE, kt0 + t + 1);
            mma_tile(smem + (t & 1) * STAGE);
            __syncthreads();  // stage t+1 landed (LDS-DMA drained) and stage t fully read
        }
    } else {
        // every wave issues the same number of DMA instructions per K-tile, so one count serves all
        static_assert((BM / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0, "pieces must divide among the waves");
        constexpr int kLoads = (BM / 8 + BN / 8) / NWAVES;
        constexpr int kKeep = kLoads * (STAGES - 2);
        static_assert(kKeep <= 63, "vmcnt is a 6-bit counter");
        const int kt_end = kt0 + nt;   // tiles past the slice are requested past K: zero-filled, no traffic
        auto stage_or_skip = [&](char* buf, int kt) { stage(buf, kt < kt_end ? kt : nt_all); };
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s) stage_or_skip(smem + s * STAGE, kt0 + s);
        int slot = 0, fill = STAGES - 1;   // buffer of tile t; buffer tile t + STAGES - 1 goes into
        for (int t = 0; t < nt; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kKeep) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage_or_skip(smem + fill * STAGE, kt0 + t + STAGES - 1);
            mma_tile(smem + slot * STAGE);
            slot = slot + 1 == STAGES ? 0 : slot + 1;
            fill = fill + 1 == STAGES ? 0 : fill + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // Epilogue.  D[i][j] of a fragment: i = 4*(lane>>4) + reg -> output channel, j = lane&15 -> row.
    const int mrow = m0 + wm * WM + (lane & 15);
    const int ncol = n0 + wn * WN + (lane >> 4) * 4;
    const bool vec_ok = (g.ldc & 3) == 0;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = ncol + j * 16;
        float al[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (OUT != kOutI32) {
            if (!g.by_row) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < g.N) {
                        al[e] = g.alpha[n + e];
                        if (g.bias) bi[e] = g.bias[n + e];
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mrow + i * 16;
            if (m >= g.M || n >= g.N) continue;
            const v4i a = acc[i][j];
            const size_t idx = static_cast<size_t>(m) * g.ldc + n;
            if constexpr (OUT == kOutI32) {
                int* o = static_cast<int*>(g.out) + idx;
                if (vec_ok && n + 3 < g.N) {
                    *reinterpret_cast<v4i*>(o) = a;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < g.N) o[e] = a[e];
                }
            } else {
                if (g.by_row) {   // the parameters of output row m, for all of this lane's columns
                    const float ar = g.alpha[m], br = g.bias ? g.bias[m] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        al[e] = ar;
                        bi[e] = br;
                    }
                }
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = dequant1(a[e], bi[e], al[e], g.beta);
                if constexpr (OUT == kOutF32) {
                    float* o = static_cast<float*>(g.out) + idx;
                    if (vec_ok && n + 3 < g.N) {
                        *reinterpret_cast<v4f*>(o) = v4f{y[0], y[1], y[2], y[3]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < g.N) o[e] = y[e];
                    }
                } else if constexpr (OUT == kOutI8) {
                    int8_t* o = static_cast<int8_t*>(g.out) + idx;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = requant_div(y[e], g.out_div);
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

// ---------------------------------------------------------------------------------------
// 256 x 256 tile, software-pipelined: the kernel large GEMMs run on.
//
// 8 waves as 2 (m) x 4 (n); a wave owns 128 x 64 outputs as four 64 x 32 quadrants.  Each
// 128-byte K-tile is kept in LDS as four 16 KiB half-tiles: A0/A1 = activation rows 0-127 /
// 128-255 of the block, W0/W1 likewise for the weight rows.  Quadrant (qm, qn) of wave
// (wm, wn) covers rows qm*128 + wm*64 .. +64 and columns qn*128 + wn*32 .. +32, so quadrant q
// reads only half-tile q.  Two K-tile buffers (128 KiB).
//
// A K-tile is four phases of 16 MFMAs (one quadrant each).  A phase is a LOAD segment (LDS
// fragment reads + one half-tile of LDS-DMA for a later K-tile + a counted vmcnt), a raw barrier,
// an MFMA segment, a raw barrier:
//   P1: read W0 of tile t    | DMA W1(t+1) | vmcnt(8) | barrier | MFMA quadrant (0,0) | barrier
//   P2: read W1              | DMA A1(t+1) | vmcnt(8) | barrier | MFMA (0,1)          | barrier
//   P3: read A1              | DMA A0(t+2) | vmcnt(8) | barrier | MFMA (1,1)          | barrier
//   P4: read A0 of tile t+1  | DMA W0(t+2) | vmcnt(8) | barrier | MFMA (1,0)          | barrier
// (A0 and A1 live in two register sets, so LDS reads per phase are 4, 4, 8, 8; VARIANT 0, kept for
// timing experiments, reads A0 in P1 instead: 12, 4, 8, 0 with one register set.)
// That is VARIANT 2, the form of the developer library's A/B runs.  The product runs VARIANT 5: the
// same half-tiles, registers and wave groups with TWO segments of 32 MFMAs per K-tile (half the
// barriers; see the loop body) and an MFMA issue order that costs the matrix pipes less power:
// 4 % faster at 8192^3 on random int8 operands, bit-equal.
// Waves 4-7 (the second wave of every SIMD) run one barrier behind waves 0-3, so while one group
// is in an MFMA segment the other is in a LOAD segment and the matrix pipe is not left waiting.
// Every wave issues exactly two DMA instructions per phase (tiles past K are fetched as
// out-of-range: zero-filled, no memory traffic), so vmcnt(8) after a phase's issue means "the
// half-tile issued four phases ago has landed": each half-tile has four phases of flight, four
// half-tiles (64 KiB) are in flight per CU, and the wait that retires a half-tile sits one
// barrier before its first read.  A half-tile is re-staged at least two phases after its last
// read, whose data the MFMAs of that phase have consumed before the barrier that ends it.
// ---------------------------------------------------------------------------------------
#define MIQ_BARRIER()                       \
    do {                                    \
        asm volatile("" ::: "memory");      \
        __builtin_amdgcn_s_barrier();       \
        asm volatile("" ::: "memory");      \
    } while (0)

// All but the 8 youngest DMA instructions (four half-tiles) have landed.  (ABL bit 32: timing-only
// build that keeps just two half-tiles in flight, to measure what the depth is worth.)
#define MIQ_WAIT_DMA()                                            \
    do {                                                          \
        if constexpr ((ABL & 32) != 0 || VARIANT >= 4)            \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      \
        else                                                      \
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      \
    } while (0)

template <int OUT, int VARIANT, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_i8_256_kernel(GemmArgs g) {
    constexpr int kHalf = 128 * kTileK;   // one half-tile: 128 rows x 128 B
    constexpr int kBuf = 4 * kHalf;       // A0 A1 W0 W1
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, 8, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, static_cast<uint32_t>(g.M) * g.K);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.W, static_cast<uint32_t>(g.N) * g.K);

    // Staging: a wave-instruction moves 8 rows x 128 B; a wave issues two per half-tile (row
    // groups `wave` and `wave + 8`).  The chunk a lane fetches is fixed by the swizzle.
    const int srow = wave * 8 + (lane >> 3);                    // row inside the half-tile, first piece
    const int cbyte = ((lane & 7) ^ ((srow >> 1) & 7)) * kChunk;
    uint32_t abase[2][2], wbase[2][2];                          // [half][piece] byte offset at k = 0
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // Activation rows as they come; the column operand permuted so that the 32 rows wave column wn reads from
            // half-tile h are output columns wn * 64 + h * 32 .. + 32: a wave's two quadrants are neighbours in the output
            // (store_tile_256, ADJ) and its rows leave as whole cache lines whatever the output type.
            const int rho = srow + i * 64;
            const int ra_ = m0 + h * 128 + rho, rw_ = n0 + (rho >> 5) * 64 + h * 32 + (rho & 31);
            abase[h][i] = ra_ < g.M ? static_cast<uint32_t>(ra_) * g.K + cbyte : kOob;
            wbase[h][i] = rw_ < g.N ? static_cast<uint32_t>(rw_) * g.K + cbyte : kOob;
        }
    auto stage_half = [&](const __amdgpu_buffer_rsrc_t& rs, char* dst, const uint32_t (&base)[2], int kt) {
        int k = kt * kTileK;
        bool k_ok = k + cbyte < g.K;
        if ((ABL & 1) && kt > 1) return;   // timing-only builds: no DMA in the main loop,
        if ((ABL & 4) && kt > 1) k_ok = false;   // DMA issued but out of range (no memory traffic),
        if ((ABL & 8) && kt > 1) k = (kt & 1) * kTileK;   // or always the same two K-tiles (cache-hot)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t off = (k_ok && base[i] != kOob) ? base[i] + k : kOob;
            dma16(rs, dst + (wave + 8 * i) * (8 * kTileK), off);
        }
    };

    v4i acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = v4i{0, 0, 0, 0};

    const int nt = (g.K + kTileK - 1) / kTileK;
    const int lo[2] = {frag_lane_offset(lane, 0), frag_lane_offset(lane, 1)};
    const int a_frag = wm * 64 * kTileK;   // this wave's rows inside an A half-tile
    const int w_frag = wn * 32 * kTileK;   // and inside a W half-tile

    // Prologue: all of tile 0 plus A0/W0 of tile 1; wait for tile 0 only.
    stage_half(ra, smem + 0 * kHalf, abase[0], 0);
    stage_half(rw, smem + 2 * kHalf, wbase[0], 0);
    stage_half(rw, smem + 3 * kHalf, wbase[1], 0);
    stage_half(ra, smem + 1 * kHalf, abase[1], 0);
    stage_half(ra, smem + kBuf + 0 * kHalf, abase[0], 1);
    if constexpr (VARIANT < 4) stage_half(rw, smem + kBuf + 2 * kHalf, wbase[0], 1);
    MIQ_WAIT_DMA();   // A0(0), W0(0) (VARIANT 4, 5: and W1(0); A1(0), A0(1) stay in flight)
    MIQ_BARRIER();
    // Waves 4-7 (the second wave on each SIMD) run one barrier behind waves 0-3: while one group
    // issues MFMAs the other does its LDS reads and DMA, so the matrix pipe never waits for loads.
    if constexpr (VARIANT != 3) {
        if (wm == 1) MIQ_BARRIER();
    }

    v4i af[2][4] = {}, af2[2][4] = {}, wf0[2][2] = {}, wf1[2][2] = {};   // [k-step][fragment]
    auto read_w = [&](v4i (&wf)[2][2], const char* half) {
        if (ABL & 2) {   // timing-only build: no LDS reads, operands made opaque
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(wf[ks][j]));
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[ks][j] = lds_read16(half + w_frag + j * 16 * kTileK + lo[ks]);
    };
    auto read_a = [&](v4i (&af)[2][4], const char* half) {
        if (ABL & 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(af[ks][i]));
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = lds_read16(half + a_frag + i * 16 * kTileK + lo[ks]);
    };
    auto mma = [&](v4i (&c)[4][2], const v4i (&af)[2][4], const v4i (&wf)[2][2]) {
        if (ABL & 16) {   // timing-only build: no MFMAs, operands kept live
            asm volatile("" :: "v"(wf[0][0]), "v"(wf[1][1]), "v"(af[0][0]), "v"(af[1][3]));
            return;
        }
        // Issue order of a quadrant's sixteen MFMAs (the sums are the same whatever the order).  VARIANT 5 issues
        // the two k-steps of an accumulator back to back (pinned with scheduling barriers: left alone the compiler
        // restores the k-step-major order) and lets the weight fragment alternate 0 1 1 0 over consecutive
        // activation fragments, so that consecutive MFMAs always share an accumulator or an operand: the matrix
        // pipes then draw less power per instruction and the chip holds a higher clock under them (2-4 % on
        // random int8 operands, nothing on zeros).  ABL bits 2048 / 1024 switch the two on separately for the
        // other variants, bit 4096 keeps the weight fragment for four MFMAs instead.
        constexpr bool kPaired = VARIANT == 5 || (ABL & 2048) != 0, kSnake = VARIANT == 5 || (ABL & 1024) != 0;
        if constexpr (kPaired) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = (kSnake && (i & 1)) ? 1 - jj : jj;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        c[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[ks][j], af[ks][i], c[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        } else if constexpr ((ABL & 4096) != 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int i = (j & 1) ? 3 - ii : ii;
                        c[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[ks][j], af[ks][i], c[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = (kSnake && (i & 1)) ? 1 - jj : jj;
                        c[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[ks][j], af[ks][i], c[i][j], 0, 0, 0);
                    }
        }
    };

    // Timing-only build (ABL bit 64): cycles between consecutive points of a phase, summed per
    // phase and segment: [issue reads + DMA | wait DMA | barrier | MFMA issue | barrier].
    unsigned seg[4][5] = {};
    unsigned long long last = 0;
    if constexpr ((ABL & 64) != 0) last = __builtin_amdgcn_s_memtime();
#define MIQ_STAMP(P, S)                                               \
    do {                                                              \
        if constexpr ((ABL & 64) != 0) {                              \
            const unsigned long long now = __builtin_amdgcn_s_memtime(); \
            seg[P][S] += static_cast<unsigned>(now - last);           \
            last = now;                                               \
        }                                                             \
    } while (0)
    if constexpr (VARIANT >= 2) read_a(af, smem + 0 * kHalf);   // A0(0)
    if constexpr ((ABL & 8192) != 0) {   // static priority for the second wave of every SIMD, no per-segment flips
        if (wm == 1) __builtin_amdgcn_s_setprio(1);
    }
    for (int t = 0; t < nt; ++t) {
        char* cur = smem + (t & 1) * kBuf;
        char* nxt = smem + ((t + 1) & 1) * kBuf;
        if constexpr (VARIANT == 0) {
            // four phases of 16 MFMAs, two barriers each
            read_w(wf0, cur + 2 * kHalf);
            read_a(af, cur + 0 * kHalf);
            stage_half(rw, nxt + 3 * kHalf, wbase[1], t + 1);
            MIQ_STAMP(0, 0);
            MIQ_WAIT_DMA();   // W1(t) has landed
            MIQ_STAMP(0, 1);
            MIQ_BARRIER();
            MIQ_STAMP(0, 2);
            __builtin_amdgcn_s_setprio(1);
            mma(acc[0][0], af, wf0);
            __builtin_amdgcn_s_setprio(0);
            MIQ_STAMP(0, 3);
            MIQ_BARRIER();
            MIQ_STAMP(0, 4);

            read_w(wf1, cur + 3 * kHalf);
            stage_half(ra, nxt + 1 * kHalf, abase[1], t + 1);
            MIQ_STAMP(1, 0);
            MIQ_WAIT_DMA();   // A1(t)
            MIQ_STAMP(1, 1);
            MIQ_BARRIER();
            MIQ_STAMP(1, 2);
            __builtin_amdgcn_s_setprio(1);
            mma(acc[0][1], af, wf1);
            __builtin_amdgcn_s_setprio(0);
            MIQ_STAMP(1, 3);
            MIQ_BARRIER();
            MIQ_STAMP(1, 4);

            read_a(af, cur + 1 * kHalf);
            stage_half(ra, cur + 0 * kHalf, abase[0], t + 2);
            MIQ_STAMP(2, 0);
            MIQ_WAIT_DMA();   // A0(t+1)
            MIQ_STAMP(2, 1);
            MIQ_BARRIER();
            MIQ_STAMP(2, 2);
            __builtin_amdgcn_s_setprio(1);
            mma(acc[1][1], af, wf1);
            __builtin_amdgcn_s_setprio(0);
            MIQ_STAMP(2, 3);
            MIQ_BARRIER();
            MIQ_STAMP(2, 4);

            stage_half(rw, cur + 2 * kHalf, wbase[0], t + 2);
            MIQ_STAMP(3, 0);
            MIQ_WAIT_DMA();   // W0(t+1)
            MIQ_STAMP(3, 1);
            MIQ_BARRIER();
            MIQ_STAMP(3, 2);
            __builtin_amdgcn_s_setprio(1);
            mma(acc[1][0], af, wf0);
            __builtin_amdgcn_s_setprio(0);
            MIQ_STAMP(3, 3);
            MIQ_BARRIER();
            MIQ_STAMP(3, 4);
        } else if constexpr (VARIANT == 2) {
            // As variant 0, with the A0 fragments of the next K-tile read one phase early into a
            // second register set: LDS reads per phase are 4, 4, 8, 8 instead of 12, 4, 8, 0.
            read_w(wf0, cur + 2 * kHalf);
            stage_half(rw, nxt + 3 * kHalf, wbase[1], t + 1);
            MIQ_WAIT_DMA();   // W1(t) has landed
            MIQ_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            mma(acc[0][0], af, wf0);
            __builtin_amdgcn_s_setprio(0);
            MIQ_BARRIER();

            read_w(wf1, cur + 3 * kHalf);
            stage_half(ra, nxt + 1 * kHalf, abase[1], t + 1);
            MIQ_WAIT_DMA();   // A1(t)
            MIQ_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            mma(acc[0][1], af, wf1);
            __builtin_amdgcn_s_setprio(0);
            MIQ_BARRIER();

            read_a(af2, cur + 1 * kHalf);
            stage_half(ra, cur + 0 * kHalf, abase[0], t + 2);
            MIQ_WAIT_DMA();   // A0(t+1)
            MIQ_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            mma(acc[1][1], af2, wf1);
            __builtin_amdgcn_s_setprio(0);
            MIQ_BARRIER();

            read_a(af, nxt + 0 * kHalf);   // A0(t+1): retired by the wait one phase ago
            stage_half(rw, cur + 2 * kHalf, wbase[0], t + 2);
            MIQ_WAIT_DMA();   // W0(t+1)
            MIQ_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            mma(acc[1][0], af2, wf0);
            __builtin_amdgcn_s_setprio(0);
            MIQ_BARRIER();
        } else if constexpr (VARIANT >= 4) {
            // Two segments of 32 MFMAs per K-tile (half the barriers of the four-phase forms): quadrants (0,0) (0,1),
            // then (1,1) (1,0).  Per wave and segment four DMA instructions; vmcnt(4) after a segment's issue retires
            // the two half-tiles issued one segment earlier, which are first read one segment later:
            //   S1: read W0(t) W1(t)    | DMA W0(t+1) W1(t+1) | vmcnt(4): A1(t) A0(t+1) landed   | barrier | 32 MFMAs | barrier
            //   S2: read A1(t) A0(t+1)  | DMA A1(t+1) A0(t+2) | vmcnt(4): W0(t+1) W1(t+1) landed | barrier | 32 MFMAs | barrier
            // Every half-tile is re-staged two segments after its last read.
            read_w(wf0, cur + 2 * kHalf);
            read_w(wf1, cur + 3 * kHalf);
            stage_half(rw, nxt + 2 * kHalf, wbase[0], t + 1);
            stage_half(rw, nxt + 3 * kHalf, wbase[1], t + 1);
            MIQ_STAMP(0, 0);
            MIQ_WAIT_DMA();
            MIQ_STAMP(0, 1);
            MIQ_BARRIER();
            MIQ_STAMP(0, 2);
            if constexpr ((ABL & 8192) == 0) __builtin_amdgcn_s_setprio(1);
            mma(acc[0][0], af, wf0);
            mma(acc[0][1], af, wf1);
            if constexpr ((ABL & 8192) == 0) __builtin_amdgcn_s_setprio(0);
            MIQ_STAMP(0, 3);
            MIQ_BARRIER();
            MIQ_STAMP(0, 4);

            read_a(af2, cur + 1 * kHalf);
            read_a(af, nxt + 0 * kHalf);
            stage_half(ra, nxt + 1 * kHalf, abase[1], t + 1);
            stage_half(ra, cur + 0 * kHalf, abase[0], t + 2);
            MIQ_STAMP(1, 0);
            MIQ_WAIT_DMA();
            MIQ_STAMP(1, 1);
            MIQ_BARRIER();
            MIQ_STAMP(1, 2);
            if constexpr ((ABL & 8192) == 0) __builtin_amdgcn_s_setprio(1);
            mma(acc[1][1], af2, wf1);
            mma(acc[1][0], af2, wf0);
            if constexpr ((ABL & 8192) == 0) __builtin_amdgcn_s_setprio(0);
            MIQ_STAMP(1, 3);
            MIQ_BARRIER();
            MIQ_STAMP(1, 4);
        } else if constexpr (VARIANT == 3) {
            // Variant 2's instruction stream with ONE barrier per phase and wave: waves 0-3 run LOAD, MFMA,
            // wait, barrier; waves 4-7 run LOAD, wait, barrier, MFMA, so that after every barrier one wave of
            // a SIMD starts on its MFMAs while the other starts on its loads, and nothing forces the two to
            // alternate in between.  The waits and distances are variant 2's: a half-tile is read one phase
            // after the wait that retires it and re-staged two phases after its last read.
#define MIQ_PHASE3(LOADS, ACC, AF, WF)                \
            LOADS;                                    \
            if (wm == 1) {                            \
                MIQ_WAIT_DMA();                       \
                MIQ_BARRIER();                        \
            }                                         \
            __builtin_amdgcn_s_setprio(1);            \
            mma(ACC, AF, WF);                         \
            __builtin_amdgcn_s_setprio(0);            \
            if (wm == 0) {                            \
                MIQ_WAIT_DMA();                       \
                MIQ_BARRIER();                        \
            }
            MIQ_PHASE3((read_w(wf0, cur + 2 * kHalf), stage_half(rw, nxt + 3 * kHalf, wbase[1], t + 1)), acc[0][0], af, wf0);
            MIQ_PHASE3((read_w(wf1, cur + 3 * kHalf), stage_half(ra, nxt + 1 * kHalf, abase[1], t + 1)), acc[0][1], af, wf1);
            MIQ_PHASE3((read_a(af2, cur + 1 * kHalf), stage_half(ra, cur + 0 * kHalf, abase[0], t + 2)), acc[1][1], af2, wf1);
            MIQ_PHASE3((read_a(af, nxt + 0 * kHalf), stage_half(rw, cur + 2 * kHalf, wbase[0], t + 2)), acc[1][0], af2, wf0);
#undef MIQ_PHASE3
        }
    }
    if constexpr ((ABL & 64) != 0) {
        if (g.stamps != nullptr && lane < 20)
            g.stamps[(static_cast<size_t>(blockIdx.x) * 8 + wave) * 20 + lane] = seg[lane / 5][lane % 5];
    }
    if constexpr (VARIANT != 3) {
        if (wm == 0) MIQ_BARRIER();   // every wave executes the same number of barriers
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs issued for tiles past K

    store_tile_256<OUT, (ABL & 256) == 0, (ABL & 512) == 0, 2, true, true>(g, acc, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------
// 256 x 384 tile: the kernel the interior of a large GEMM runs on.
//
// The 256 x 256 kernel above is bound by the rate at which a CU pulls operand bytes L2 -> LDS, not by
// its MFMAs; a 256 x 384 tile moves (256 + 384) / (256 * 384) = 5/6 of the bytes per multiply-add.
// Eight waves as 2 (m) x 4 (n); a wave owns 128 x 96 outputs as 2 x 3 blocks of 64 x 32 (192
// accumulator registers).  A 128-byte K-tile is five 16 KiB sub-tiles in LDS, A0 A1 (activation rows
// 0-127 / 128-255) and W0 W1 W2 (weight rows 0-127 / 128-255 / 256-383); block (h, q) of a wave reads
// its 64 rows of A<h> and its 32 rows of W<q>.  Two K-tile buffers: 160 KiB, all of the CU's LDS.
//
// A K-tile is four phases of 24 MFMAs, (k-step, h) = (0,0) (0,1) (1,0) (1,1): the six weight
// fragments of a k-step stay in registers for two phases, the four activation fragments are read per
// phase (40 operand registers).  A phase is LOAD (fragment reads, LDS-DMA for the NEXT K-tile, a
// counted vmcnt) | barrier | 24 MFMAs | barrier, and waves 4-7 run one barrier behind waves 0-3, so
// one wave of every SIMD computes while the other loads.
//   P1: read W*(k0) A0(k0) | DMA A0 W0 of t+1 | vmcnt(4): A1(t) landed     | MFMA (0, q)
//   P2: read A1(k0)        | DMA W1 W2 of t+1 |                            | MFMA (1, q)
//   P3: read W*(k1) A0(k1) | DMA A1 of t+1    |                            | MFMA (0, q)
//   P4: read A1(k1)        |                  | vmcnt(2): A0 W* of t+1 landed | MFMA (1, q)
// A sub-tile is read no earlier than one phase after the wait that retires it and re-staged no
// earlier than two phases after its last read (the MFMAs that consume a read sit before the barrier
// that ends its phase; with the two wave groups one barrier apart that is the distance at which every
// wave has consumed its reads / passed its wait).
//
// Only tiles that lie wholly inside the matrices come here (M % 256 == 0, N % 384 == 0, K % 128 == 0
// for the block this launch covers): a lane's buffer offset is then fixed for the whole kernel and
// the K-tile and row-group advance in the instruction's scalar offset, which leaves the main loop
// without vector address arithmetic.  K-tiles past the end are requested through a descriptor of
// zero records (no memory traffic) so that every wave issues the same count of DMA instructions.
// ---------------------------------------------------------------------------------------
#ifndef MIQ_NO_DEV_SWITCHES
template <int OUT, int SCHED = 0>
__global__ __launch_bounds__(512) void gemm_i8_wide_kernel(GemmArgs g) {
    constexpr int kSub = 128 * kTileK;   // one sub-tile: 128 rows x 128 B
    constexpr int kStage = 5 * kSub;     // A0 A1 W0 W1 W2
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, 8, tm, tn);
    const int m0 = tm * 256, n0 = tn * 384;
    const uint32_t K = static_cast<uint32_t>(g.K);
    const int nt = g.K / kTileK;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, static_cast<uint32_t>(g.M) * K);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.W, static_cast<uint32_t>(g.N) * K);
    const __amdgpu_buffer_rsrc_t ra_none = make_rsrc(g.A, 0), rw_none = make_rsrc(g.W, 0);

    // Staging: a wave-instruction (a "piece") moves 8 rows x 128 B; a wave issues two per sub-tile (row
    // groups `wave` and `wave + 8`, 64 rows apart).  The chunk a lane fetches is fixed by the swizzle.
    const int srow = wave * 8 + (lane >> 3);
    const int cbyte = ((lane & 7) ^ ((srow >> 1) & 7)) * kChunk;
    const uint32_t va = static_cast<uint32_t>(m0 + srow) * K + cbyte;
    const uint32_t vw = static_cast<uint32_t>(n0 + srow) * K + cbyte;
    // sub: 0, 1 = A0, A1; 2, 3, 4 = W0, W1, W2.  pieces: bit 0 = rows 0-63 of the sub-tile, bit 1 = rows 64-127.
    auto stage = [&](char* stg, int sub, int pieces, int kt) {
        const bool is_w = sub >= 2;
        const __amdgpu_buffer_rsrc_t rs = is_w ? (kt < nt ? rw : rw_none) : (kt < nt ? ra : ra_none);
        const uint32_t so = static_cast<uint32_t>(kt) * kTileK + static_cast<uint32_t>(is_w ? sub - 2 : sub) * 128u * K;
        char* dst = stg + sub * kSub + wave * (8 * kTileK);
        const uint32_t v = is_w ? vw : va;
        if (pieces & 1) dma16s(rs, dst, v, so);
        if (pieces & 2) dma16s(rs, dst + 8 * (8 * kTileK), v, so + 64u * K);
    };

    v4i acc[2][3][4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][q][i][j] = v4i{0, 0, 0, 0};

    const int lo[2] = {frag_lane_offset(lane, 0), frag_lane_offset(lane, 1)};
    const int a_frag = wm * 64 * kTileK;   // this wave's rows inside an A sub-tile
    const int w_frag = wn * 32 * kTileK;   // and inside a W sub-tile

    v4i wf[3][2] = {}, af[4] = {};
    auto read_w = [&](const char* stg, int ks) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[q][j] = lds_read16(stg + (2 + q) * kSub + w_frag + j * 16 * kTileK + lo[ks]);
    };
    auto read_a = [&](const char* stg, int h, int ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = lds_read16(stg + h * kSub + a_frag + i * 16 * kTileK + lo[ks]);
    };
    // 24 MFMAs; `mid(u)` runs after the MFMAs of weight fragment pair q = u (u = 0, 1): the place SCHED 3 issues DMA from
    auto mma = [&](v4i (&c)[3][4][2], auto&& mid) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    c[q][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[q][j], af[i], c[q][i][j], 0, 0, 0);
            if (q < 2) mid(q);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto none = [](int) {};

#define MIQ_WIDE_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    // Prologue: all of K-tile 0, A1 last (the loop's first wait expects it to be the youngest but for what SCHED 3
    // stages ahead).
    stage(smem, 0, 3, 0);
    stage(smem, 2, 3, 0);
    stage(smem, 3, 3, 0);
    stage(smem, 4, 3, 0);
    stage(smem, 1, 3, 0);
    if constexpr (SCHED == 3) {
        stage(smem + kStage, 0, 3, 1);
        stage(smem + kStage, 2, 1, 1);
        MIQ_WIDE_WAIT(5);
    } else {
        MIQ_WIDE_WAIT(2);   // A0 W0 W1 W2 of tile 0
    }
    MIQ_BARRIER();
    if (wm == 1) MIQ_BARRIER();   // waves 4-7 run one barrier behind

    for (int t = 0; t < nt; ++t) {
        char* cur = smem + (t & 1) * kStage;
        char* oth = smem + ((t + 1) & 1) * kStage;
        if constexpr (SCHED == 0 || SCHED == 1) {
            // DMA per phase 4 / 4 / 2 / 0
            if constexpr (SCHED == 1) { stage(oth, 0, 3, t + 1); stage(oth, 2, 3, t + 1); }
            read_w(cur, 0);
            read_a(cur, 0, 0);
            if constexpr (SCHED == 0) { stage(oth, 0, 3, t + 1); stage(oth, 2, 3, t + 1); }
            MIQ_WIDE_WAIT(4);   // A1(t)
            MIQ_BARRIER();
            mma(acc[0], none);
            MIQ_BARRIER();

            if constexpr (SCHED == 1) { stage(oth, 3, 3, t + 1); stage(oth, 4, 3, t + 1); }
            read_a(cur, 1, 0);
            if constexpr (SCHED == 0) { stage(oth, 3, 3, t + 1); stage(oth, 4, 3, t + 1); }
            MIQ_BARRIER();
            mma(acc[1], none);
            MIQ_BARRIER();

            if constexpr (SCHED == 1) stage(oth, 1, 3, t + 1);
            read_w(cur, 1);
            read_a(cur, 0, 1);
            if constexpr (SCHED == 0) stage(oth, 1, 3, t + 1);
            MIQ_BARRIER();
            mma(acc[0], none);
            MIQ_BARRIER();

            read_a(cur, 1, 1);
            MIQ_WIDE_WAIT(2);   // A0 W0 W1 W2 of tile t+1
            MIQ_BARRIER();
            mma(acc[1], none);
            MIQ_BARRIER();
        } else if constexpr (SCHED == 2 || SCHED == 4) {
            // DMA per phase 3 / 3 / 2 / 2: A0 W0a | W0b W1 | W2 | A1   (SCHED 4: issued before the reads)
            if constexpr (SCHED == 4) { stage(oth, 0, 3, t + 1); stage(oth, 2, 1, t + 1); }
            read_w(cur, 0);
            read_a(cur, 0, 0);
            if constexpr (SCHED == 2) { stage(oth, 0, 3, t + 1); stage(oth, 2, 1, t + 1); }
            MIQ_WIDE_WAIT(3);   // A1(t), issued one phase ago
            MIQ_BARRIER();
            mma(acc[0], none);
            MIQ_BARRIER();

            if constexpr (SCHED == 4) { stage(oth, 2, 2, t + 1); stage(oth, 3, 3, t + 1); }
            read_a(cur, 1, 0);
            if constexpr (SCHED == 2) { stage(oth, 2, 2, t + 1); stage(oth, 3, 3, t + 1); }
            MIQ_BARRIER();
            mma(acc[1], none);
            MIQ_BARRIER();

            if constexpr (SCHED == 4) stage(oth, 4, 3, t + 1);
            read_w(cur, 1);
            read_a(cur, 0, 1);
            if constexpr (SCHED == 2) stage(oth, 4, 3, t + 1);
            MIQ_BARRIER();
            mma(acc[0], none);
            MIQ_BARRIER();

            if constexpr (SCHED == 4) stage(oth, 1, 3, t + 1);
            read_a(cur, 1, 1);
            if constexpr (SCHED == 2) stage(oth, 1, 3, t + 1);
            MIQ_WIDE_WAIT(2);   // A0 W0 W1 W2 of tile t+1
            MIQ_BARRIER();
            mma(acc[1], none);
            MIQ_BARRIER();
        } else {
            // SCHED 3: DMA issued between the MFMAs, 3 / 2 / 2 / 3 per phase: W0b W1 | W2 | A1 of tile t+1, then
            // A0 W0a of tile t+2 (their buffers were last read in P3 and are consumed by the barrier before P4's MFMAs)
            read_w(cur, 0);
            read_a(cur, 0, 0);
            MIQ_WIDE_WAIT(3);   // A1(t)
            MIQ_BARRIER();
            mma(acc[0], [&](int u) { if (u == 0) { stage(oth, 2, 2, t + 1); stage(oth, 3, 1, t + 1); } else stage(oth, 3, 2, t + 1); });
            MIQ_BARRIER();

            read_a(cur, 1, 0);
            MIQ_BARRIER();
            mma(acc[1], [&](int u) { stage(oth, 4, u == 0 ? 1 : 2, t + 1); });
            MIQ_BARRIER();

            read_w(cur, 1);
            read_a(cur, 0, 1);
            MIQ_BARRIER();
            mma(acc[0], [&](int u) { stage(oth, 1, u == 0 ? 1 : 2, t + 1); });
            MIQ_BARRIER();

            read_a(cur, 1, 1);
            MIQ_WIDE_WAIT(2);   // A0 W0 W1 W2 of tile t+1 (all but A1's two)
            MIQ_BARRIER();
            mma(acc[1], [&](int u) { if (u == 0) stage(cur, 0, 3, t + 2); else stage(cur, 2, 1, t + 2); });
            MIQ_BARRIER();
        }
    }
#undef MIQ_WIDE_WAIT
    if (wm == 0) MIQ_BARRIER();   // every wave executes the same number of barriers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-record DMAs issued for the tiles past K

    store_tile_256<OUT, true, true, 3, true>(g, acc, m0, n0, wm, wn, lane);
}
#endif   // MIQ_NO_DEV_SWITCHES

// Any K and any alignment: one thread per output element.  Only shapes the MFMA path cannot
// take (K not a multiple of 16, or misaligned pointers) come here; they are tiny layers.
template <int OUT>
__global__ __launch_bounds__(256) void gemm_i8_any_kernel(GemmArgs g) {
    const int64_t total = static_cast<int64_t>(g.M) * g.N;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += 256ll * gridDim.x) {
        const int m = static_cast<int>(idx / g.N), n = static_cast<int>(idx % g.N);
        const int8_t* a = g.A + static_cast<int64_t>(m) * g.K;
        const int8_t* w = g.W + static_cast<int64_t>(n) * g.K;
        int sum = 0;
        for (int k = 0; k < g.K; ++k) sum += static_cast<int>(a[k]) * static_cast<int>(w[k]);
        store_one<OUT>(g, m, n, sum);
    }
}

// Sums the split-K slabs and applies the epilogue: out[m,n] = ((sum_s slab[s][m][n]) + bias[n]) * alpha[n] + beta.
template <int OUT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const int* __restrict__ slabs, int splits, GemmArgs g) {
    const int64_t total = static_cast<int64_t>(g.M) * g.N;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += 256ll * gridDim.x) {
        int sum = 0;
        for (int s = 0; s < splits; ++s) sum += slabs[s * total + idx];
        store_one<OUT>(g, static_cast<int>(idx / g.N), static_cast<int>(idx % g.N), sum);
    }
}

// ---------------------------------------------------------------------------------------
// M <= 16 (decode): the weight is streamed once, HBM -> MFMA operand registers.  (The kernel is
// written for MT groups of 16 activation rows, but past one group every wave re-reads MT times its
// weight bytes of activations from L2 and the tiled kernels with split-K are faster: measured
// 24 vs 20 us at M = 32 and 33 vs 22 us at M = 64 on [14336, 4096].)
//
// Nothing is shared between waves, so there is no LDS staging: a lane loads the 16 bytes the MFMA
// wants from it (row lane&15 of a 16-row group, chunk lane>>4 of a 64-byte K-step) straight from
// global memory, for the weight and for the (tiny, L2-resident) activation alike.  A workgroup of
// 8 waves owns WN*16 output channels; wave s sums the 128-byte K-tiles s, s+8, s+16, ..., so the
// workgroup reads each of its weight rows as consecutive 1 KiB runs.  Every wave keeps kDepth
// K-tiles of loads in flight (a ring of register slots: consume one, refill it with the tile
// kDepth rounds ahead; tiles past K load as zeros and add nothing).  The eight partial sums meet
// in LDS, and one thread per output finishes the sum and applies the epilogue.
// ---------------------------------------------------------------------------------------
template <int MT, int WN, int OUT, int DEPTH = 4, bool NT = false>
__global__ __launch_bounds__(512) void gemm_i8_skinny_kernel(GemmArgs g) {
    constexpr int kWaves = 8, kDepth = DEPTH;
    __shared__ int red[kWaves * MT * WN * 256];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * (WN * 16);
    const int r = lane & 15, cb = (lane >> 4) * kChunk;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, static_cast<uint32_t>(g.M) * g.K);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.W, static_cast<uint32_t>(g.N) * g.K);
    uint32_t wbase[WN], abase[MT];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + j * 16 + r;
        wbase[j] = n < g.N ? static_cast<uint32_t>(n) * g.K + cb : kOob;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = i * 16 + r;
        abase[i] = m < g.M ? static_cast<uint32_t>(m) * g.K + cb : kOob;
    }

    struct Slot {
        v4i w[2][WN], a[2][MT];   // [k-step][fragment]
    };
    // (NT: the weight is read once by one workgroup, so its loads are marked non-temporal and do not displace
    // the activations, which every workgroup re-reads, from the caches)
    auto load16 = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t base, int k, bool weight) -> v4i {
        const uint32_t off = (base != kOob && k + cb < g.K) ? base + k : kOob;
        if (NT && weight) return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2));
        return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    };
    auto fill = [&](Slot& s, int kt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k = kt * kTileK + ks * 64;
#pragma unroll
            for (int j = 0; j < WN; ++j) s.w[ks][j] = load16(rw, wbase[j], k, true);
#pragma unroll
            for (int i = 0; i < MT; ++i) s.a[ks][i] = load16(ra, abase[i], k, false);
        }
    };

    v4i acc[MT][WN];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    const int nt = (g.K + kTileK - 1) / kTileK;
    Slot ring[kDepth];
#pragma unroll
    for (int u = 0; u < kDepth; ++u) fill(ring[u], wave + u * kWaves);
    for (int t = wave; t < nt; t += kWaves * kDepth) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[u].w[ks][j], ring[u].a[ks][i], acc[i][j], 0, 0, 0);
            fill(ring[u], t + (u + kDepth) * kWaves);
        }
    }

    // D of a fragment: 4*(lane>>4) + reg -> output channel, lane&15 -> activation row.
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
            *reinterpret_cast<v4i*>(&red[(((wave * MT + i) * WN + j) * 64 + lane) * 4]) = acc[i][j];
    __syncthreads();
    for (int e = threadIdx.x; e < MT * 16 * WN * 16; e += 512) {
        const int ml = e / (WN * 16), nl = e % (WN * 16);
        const int m = ml, n = n0 + nl;
        if (m >= g.M || n >= g.N) continue;
        const int i = ml >> 4, col = ml & 15, j = nl >> 4, ch = nl & 15;
        const int src = (((i * WN + j) * 64) + (ch >> 2) * 16 + col) * 4 + (ch & 3);
        int sum = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) sum += red[w * MT * WN * 256 + src];
        store_one<OUT>(g, m, n, sum);
    }
}

template <int MT, int OUT>
int launch_skinny(const GemmArgs& g, hipStream_t st) {
#ifndef MIQ_NO_DEV_SWITCHES
    // developer sweep: MIQ_SKINNY = "<channel groups per workgroup><K-tiles in flight per wave><n|c>", e.g. "28n"
    if (const char* sk = dev_switch("MIQ_SKINNY")) {
        if constexpr (OUT == kOutF32) {
            const int wn = sk[0] - '0', depth = sk[1] - '0';
            const bool nt = sk[2] == 'n';
#define MIQ_SK(W, D)                                                                                              \
    if (wn == W && depth == D) {                                                                                  \
        if (nt) gemm_i8_skinny_kernel<MT, W, OUT, D, true><<<static_cast<int>(ceil_div(g.N, 16 * W)), 512, 0, st>>>(g);  \
        else gemm_i8_skinny_kernel<MT, W, OUT, D, false><<<static_cast<int>(ceil_div(g.N, 16 * W)), 512, 0, st>>>(g);    \
        return check_launch("gemm_i8_skinny_kernel");                                                             \
    }
            MIQ_SK(1, 2) MIQ_SK(1, 4) MIQ_SK(1, 8) MIQ_SK(2, 2) MIQ_SK(2, 4) MIQ_SK(2, 8) MIQ_SK(4, 2) MIQ_SK(4, 4)
#undef MIQ_SK
        }
    }
#endif
    // 32 channels per workgroup when that still gives every CU a workgroup, else 16.
    if (ceil_div(g.N, 32) >= 256) {
        gemm_i8_skinny_kernel<MT, 2, OUT><<<static_cast<int>(ceil_div(g.N, 32)), 512, 0, st>>>(g);
    } else {
        gemm_i8_skinny_kernel<MT, 1, OUT><<<static_cast<int>(ceil_div(g.N, 16)), 512, 0, st>>>(g);
    }
    return check_launch("gemm_i8_skinny_kernel");
}

// Tile and split-K choice for shapes that do not fill the chip with 256 x 256 tiles.  Thin
// problems (few output tiles, deep K) are cut along K so that a few hundred workgroups stream the
// weight concurrently; the slices write int32 slabs that splitk_reduce_kernel sums.
struct GemmPlan {
    int bm;      // activation rows per tile: 256 (pipelined kernel), 128, 64, 32, 16, or 0 = weight-streaming kernel
    int bn;
    int splits;  // 1 = no split
    int kt_per_split;
    bool ring;   // 128-row tiles: eight-wave three-deep ring kernel (every workgroup resident at once)
};

GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K) {
    GemmPlan p{256, 256, 1, 0, false};
    const int64_t big_tiles = ceil_div(M, 256) * ceil_div(N, 256);
    // (a weight narrower than 3/4 of its 256-column tiles would leave the tile's columns empty)
    const bool wide_enough = 4 * N >= 3 * ceil_div(N, 256) * 256;
    const char* p256 = dev_switch("MIQ_GEMM_PLAN256");   // developer A/B switch: 0 = keep large problems off the 256 x 256 kernel
    if (M > 128 && big_tiles >= 192 && wide_enough && !(p256 && p256[0] == '0')) return p;
    if (M <= 16) {   // one 16-row MFMA group: beyond it the activation re-reads cost more than tiles + split-K
        const char* sk = dev_switch("MIQ_GEMM_SKINNY");   // developer A/B switch: 0 = tiled kernels for small M too
        // ... unless the weight has enough 64-channel tiles to fill the chip without split-K: the 32-row ring
        // kernel then streams it faster (15.7-16.6 vs 17.0-19.2 us on [14336,4096]; with split-K it loses,
        // 21-22 vs 16-18 us on [4096,14336])
        const bool wide = ceil_div(N, 64) >= 160 && !(sk && sk[0] == '1');
        if (!(sk && sk[0] == '0') && !wide) {
            p.bm = 0;
            return p;
        }
    }
    const char* mid = dev_switch("MIQ_GEMM_MID");   // developer A/B switch: largest M that takes 64-row tiles
    const int64_t mid_m = mid ? atoll(mid) : 128;
    if (M > mid_m) {
        p.bm = 128; p.bn = N <= 64 ? 64 : 128;   // 137.7 vs 151.7 us at 802816 x 64 x 576
    } else if (M > 32) {
        p.bm = 64; p.bn = 64;
    } else if (M > 16 || ceil_div(N, 64) >= 160) {
        p.bm = 32; p.bn = 64;
    } else {
        p.bm = 16; p.bn = 64;
    }
    const int64_t tiles = ceil_div(M, p.bm) * ceil_div(N, p.bn);
    const int64_t nt = ceil_div(K, kTileK);
    int64_t splits = 1;
    if (tiles < 160 && nt >= 16) {
        splits = ceil_div(512, tiles);                       // aim at ~2 workgroups per CU
        if (splits > nt / 8) splits = nt / 8;                // at least 8 K-tiles (1 KiB of K) per slice
        if (splits > K / (8 * M)) splits = K / (8 * M);      // slab traffic (2 * 4*splits*M*N) <= weight bytes N*K
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
    }
    // 128-row tiles: cut so that all slices are resident at once, one workgroup per CU, and run on the
    // eight-wave ring kernel (measured against ~2 double-buffered workgroups per CU: 29.8 vs 31.8 us at
    // 256 x 4096 x 14336, 74.2 vs 88.2 us at 256 x 8192 x 28672, 18.7 vs 21.2 us at 256 x 4096 x 4096)
    const char* sr = dev_switch("MIQ_GEMM_SPLIT_RING");   // developer A/B switch: 0 = double-buffered slices
    if (p.bm == 128 && p.bn == 128 && splits > 1 && !(sr && sr[0] == '0') && tiles * 2 <= 256) {
        splits = 256 / tiles;
        if (splits > nt / 8) splits = nt / 8;
        p.ring = true;
    }
    if (splits > 1) {
        p.kt_per_split = static_cast<int>(ceil_div(nt, splits));
        p.splits = static_cast<int>(ceil_div(nt, p.kt_per_split));
    }
    return p;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int OUT, int STAGES = 2>
int launch_tile(GemmArgs g, hipStream_t st, int splits = 1, int kt_per_split = 0, int* slabs = nullptr) {
    g.tiles_m = static_cast<int>(ceil_div(g.M, BM));
    g.tiles_n = static_cast<int>(ceil_div(g.N, BN));
    constexpr int kLds = STAGES * (BM + BN) * kTileK;
    const int64_t nwg = static_cast<int64_t>(g.tiles_m) * g.tiles_n;
    if (nwg >= (1ll << 31)) return fail(MIQ_ERANGE, "miq_gemm_i8: too many tiles");
    if (splits > 1) {
        GemmArgs part = g;
        part.out = slabs;
        part.kt_per_split = kt_per_split;
        auto kern = gemm_i8_kernel<BM, BN, WAVES_M, WAVES_N, kOutI32, STAGES>;
        if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
        kern<<<dim3(static_cast<unsigned>(nwg), static_cast<unsigned>(splits)), 64 * WAVES_M * WAVES_N, kLds, st>>>(part);
        if (int rc = check_launch("gemm_i8_kernel (split-K)")) return rc;
        const int64_t total = static_cast<int64_t>(g.M) * g.N;
        const int rgrid = static_cast<int>(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
        splitk_reduce_kernel<OUT><<<rgrid, 256, 0, st>>>(slabs, splits, g);
        return check_launch("splitk_reduce_kernel");
    }
    auto kern = gemm_i8_kernel<BM, BN, WAVES_M, WAVES_N, OUT, STAGES>;
    if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
    kern<<<static_cast<int>(nwg), 64 * WAVES_M * WAVES_N, kLds, st>>>(g);
    return check_launch("gemm_i8_kernel");
}

// The 256 x 256 kernel of a product build is gemm_i8_256_kernel<OUT, 5, 0> and nothing else.  The other
// instantiations (an earlier schedule, and timing-only builds that leave out DMA, LDS reads, MFMAs or stores and
// therefore return WRONG results) exist only in the developer library, which is built without
// MIQ_NO_DEV_SWITCHES and loaded by tools/ alone.
template <int OUT>
auto pick_256_kernel(GemmArgs& g) -> void (*)(GemmArgs) {
    void (*kern)(GemmArgs) = gemm_i8_256_kernel<OUT, 5>;
#ifndef MIQ_NO_DEV_SWITCHES
    const char* var = dev_switch("MIQ_GEMM_VARIANT");
    if (var && var[0] == '0') kern = gemm_i8_256_kernel<OUT, 0>;
    if (var && var[0] == '3') kern = gemm_i8_256_kernel<OUT, 3>;
    if (var && var[0] == '2') kern = gemm_i8_256_kernel<OUT, 2>;
    if (var && var[0] == '4') kern = gemm_i8_256_kernel<OUT, 4>;
    if (var && var[0] == '6') kern = gemm_i8_256_kernel<OUT, 5, 8192>;   // static wave priorities
    const char* abl = dev_switch("MIQ_GEMM_ABL");
    if (abl && abl[0] == 'n') kern = gemm_i8_256_kernel<OUT, 5, 256>;   // plain (cached) epilogue stores
    if (abl && abl[0] == 'h') kern = gemm_i8_256_kernel<OUT, 5, 512>;   // bf16: 8-byte stores, fragments unpaired
    if (abl && abl[0] == 'H') kern = gemm_i8_256_kernel<OUT, 5, 512 + 256>;
    if (abl && abl[0] == 'o') kern = gemm_i8_256_kernel<OUT, 2, 1024>;          // MFMA issue orders (results unchanged)
    if (abl && abl[0] == 'k') kern = gemm_i8_256_kernel<OUT, 2, 2048>;
    if (abl && abl[0] == 'K') kern = gemm_i8_256_kernel<OUT, 2, 1024 + 2048>;
    if (abl && abl[0] == 'j') kern = gemm_i8_256_kernel<OUT, 2, 4096>;
    if constexpr (OUT == kOutF32) {
        if (abl) {
            if (abl[0] == '1') kern = gemm_i8_256_kernel<OUT, 0, 1>;
            if (abl[0] == '7') kern = gemm_i8_256_kernel<OUT, 0, 16>;
            if (abl[0] == '8') kern = gemm_i8_256_kernel<OUT, 0, 16 + 2>;
            if (abl[0] == '9') kern = gemm_i8_256_kernel<OUT, 0, 16 + 2 + 8>;
            if (abl[0] == 'a') kern = gemm_i8_256_kernel<OUT, 0, 32 + 16 + 2>;
            if (abl[0] == 'b') kern = gemm_i8_256_kernel<OUT, 0, 32>;
            if (abl[0] == 'c') kern = gemm_i8_256_kernel<OUT, 0, 32 + 2>;
            if (abl[0] == 's') {   // segment stamps into the buffer whose address MIQ_GEMM_STAMPS holds (hex)
                kern = gemm_i8_256_kernel<OUT, 0, 64>;
                const char* sp = dev_switch("MIQ_GEMM_STAMPS");
                g.stamps = sp ? reinterpret_cast<unsigned*>(strtoull(sp, nullptr, 16)) : nullptr;
            }
            if (abl[0] == 't') {   // the same stamps for the two-segment schedule (rows 0 and 1 of the table)
                kern = gemm_i8_256_kernel<OUT, 5, 64>;
                const char* sp = dev_switch("MIQ_GEMM_STAMPS");
                g.stamps = sp ? reinterpret_cast<unsigned*>(strtoull(sp, nullptr, 16)) : nullptr;
            }
            if (abl[0] == '3') kern = gemm_i8_256_kernel<OUT, 0, 3>;
            if (abl[0] == '4') kern = gemm_i8_256_kernel<OUT, 0, 4 + 2>;
            if (abl[0] == '5') kern = gemm_i8_256_kernel<OUT, 0, 8 + 2>;
            if (abl[0] == '6') kern = gemm_i8_256_kernel<OUT, 0, 8>;
        }
    }
#endif
    return kern;
}

// One sub-problem on the kernel its plan names.
template <int OUT>
int launch(GemmArgs g, const GemmPlan& plan, int* slabs, hipStream_t st) {
    if (plan.bm == 0) return launch_skinny<1, OUT>(g, st);
    if (plan.bm == 256) {
        const char* simple = dev_switch("MIQ_GEMM_SIMPLE");   // developer A/B switch
        if (simple && simple[0] == '1') return launch_tile<256, 256, 2, 4, OUT>(g, st);
        g.tiles_m = static_cast<int>(ceil_div(g.M, 256));
        g.tiles_n = static_cast<int>(ceil_div(g.N, 256));
        constexpr int kLds = 2 * 4 * 128 * kTileK;
        auto kern = pick_256_kernel<OUT>(g);
        if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
        kern<<<g.tiles_m * g.tiles_n, 512, kLds, st>>>(g);
        return check_launch("gemm_i8_256_kernel");
    }
    if (const char* force = dev_switch("MIQ_GEMM_TILE")) {   // developer A/B switch
        switch (force[0]) {
            case 'a': return launch_tile<128, 128, 2, 2, OUT>(g, st);
            case 'b': return launch_tile<128, 64, 2, 2, OUT>(g, st);
            case 'c': return launch_tile<64, 128, 2, 2, OUT>(g, st);
            case 'd': return launch_tile<128, 128, 2, 4, OUT>(g, st);
            case 'e': return launch_tile<256, 128, 2, 4, OUT>(g, st);
            case 'f': return launch_tile<256, 64, 4, 2, OUT>(g, st);
            case 'g': return launch_tile<64, 64, 2, 2, OUT>(g, st);
            case 'h': return launch_tile<128, 64, 4, 2, OUT>(g, st);
            case 'i': return launch_tile<128, 64, 2, 2, OUT, 3>(g, st);
            case 'j': return launch_tile<128, 64, 4, 2, OUT, 3>(g, st);
            case 'k': return launch_tile<64, 128, 2, 2, OUT, 3>(g, st);
            case 'l': return launch_tile<64, 64, 2, 2, OUT, 4>(g, st);
            case 'm': return launch_tile<128, 128, 2, 2, OUT, 3>(g, st);
            case 'n': return launch_tile<128, 128, 2, 4, OUT, 4>(g, st);
            case 'o': return launch_tile<128, 128, 2, 4, OUT, 3>(g, st);
            case 'p': return launch_tile<128, 256, 2, 4, OUT, 3>(g, st);
            default: break;
        }
    }
    const char* stg = dev_switch("MIQ_GEMM_STAGES");   // developer A/B switch: 2 = double-buffered kernels
    if (stg && stg[0] == '2') {
        if (plan.bm == 64) return launch_tile<64, 64, 2, 2, OUT>(g, st, plan.splits, plan.kt_per_split, slabs);
        if (plan.bm == 32) return launch_tile<32, 64, 1, 4, OUT>(g, st, plan.splits, plan.kt_per_split, slabs);
    }
    // 128 x 128: double-buffered (two workgroups per CU at 64 KiB each beat one with a 96 KiB ring: 18 vs 24 us
    // at M = 128 on [14336, 4096]); 64- and 32-row tiles: four-deep ring (12-14 vs 18-20 us unsplit).
    if (plan.bm == 128 && plan.bn == 128 && plan.splits == 1 && !(stg && stg[0] == '2')) {
        // Every tile resident at once (one workgroup per CU): nothing to gain from a second workgroup per CU,
        // so the LDS goes to a three-deep ring and the tile to eight waves (measured: 24.9 vs 30.4 us at
        // 256 x 14336 x 4096, 25.7 vs 29.8 us at 1024 x 4096 x 4096); with a deep K the same holds for
        // 128 x 256 tiles when those fit one round (67.8 vs 82.2 us at 256 x 28672 x 8192; no gain at K = 4096).
        const int64_t t128 = ceil_div(g.M, 128) * ceil_div(g.N, 128), t256 = ceil_div(g.M, 128) * ceil_div(g.N, 256);
        if (t128 <= 256) return launch_tile<128, 128, 2, 4, OUT, 3>(g, st);
        if (t256 <= 256 && g.K >= 8192) return launch_tile<128, 256, 2, 4, OUT, 3>(g, st);
    }
    if (plan.bm == 128 && plan.bn == 64) return launch_tile<128, 64, 2, 2, OUT>(g, st, plan.splits, plan.kt_per_split, slabs);
    if (plan.bm == 128 && plan.ring) return launch_tile<128, 128, 2, 4, OUT, 3>(g, st, plan.splits, plan.kt_per_split, slabs);
    if (plan.bm == 128) return launch_tile<128, 128, 2, 2, OUT>(g, st, plan.splits, plan.kt_per_split, slabs);
    if (plan.bm == 64) return launch_tile<64, 64, 2, 2, OUT, 4>(g, st, plan.splits, plan.kt_per_split, slabs);
    if (plan.bm == 32) return launch_tile<32, 64, 1, 4, OUT, 4>(g, st, plan.splits, plan.kt_per_split, slabs);
    return launch_tile<16, 64, 1, 4, OUT>(g, st, plan.splits, plan.kt_per_split, slabs);
}

template <int OUT>
int launch_any(const GemmArgs& g, hipStream_t st) {
    const int64_t total = static_cast<int64_t>(g.M) * g.N;
    const int grid = static_cast<int>(total / 256 + 1 < 65536 ? total / 256 + 1 : 65536);
    gemm_i8_any_kernel<OUT><<<grid, 256, 0, st>>>(g);
    return check_launch("gemm_i8_any_kernel");
}

// The 256 x 384 kernel on a block of whole tiles (g.M % 256 == 0, g.N % 384 == 0, g.K % 128 == 0).
template <int OUT>
int launch_wide(GemmArgs g, hipStream_t st) {
#ifdef MIQ_NO_DEV_SWITCHES
    (void)g; (void)st;
    return fail(MIQ_EINVAL, "the 256 x 384 kernel exists only in the developer library");
#else
    g.tiles_m = g.M / 256;
    g.tiles_n = g.N / 384;
    constexpr int kLds = 2 * 5 * 128 * kTileK;   // 160 KiB: the whole LDS of a CU
    void (*kern)(GemmArgs) = gemm_i8_wide_kernel<OUT>;
    const char* sc = dev_switch("MIQ_GEMM_WIDE_SCHED");   // developer A/B switch: where the wide kernel issues its DMA
    if (sc && sc[0] == '1') kern = gemm_i8_wide_kernel<OUT, 1>;
    if (sc && sc[0] == '2') kern = gemm_i8_wide_kernel<OUT, 2>;
    if (sc && sc[0] == '3') kern = gemm_i8_wide_kernel<OUT, 3>;
    if (sc && sc[0] == '4') kern = gemm_i8_wide_kernel<OUT, 4>;
    if (int rc = allow_lds(reinterpret_cast<const void*>(kern), kLds)) return rc;
    kern<<<g.tiles_m * g.tiles_n, 512, kLds, st>>>(g);
    return check_launch("gemm_i8_wide_kernel");
#endif
}

// A tile count a little over whole rounds of the 256 CUs costs the 256 x 256 kernel a whole extra round (288 tiles
// take as long as 384: 169 against 104 us for 256).  Such problems are cut along M: the tile rows that make whole
// rounds run on the 256 x 256 kernel, the few remaining rows as a second, small problem on the thin-tile kernels
// (2304 x 8192 x 8192: 169 -> 136 us).  Returns the rows of the first part, M when the problem stays whole.
int64_t rows_in_whole_rounds(int64_t M, int64_t N, const GemmPlan& whole) {
    if (whole.bm != 256) return M;
    const char* sw = dev_switch("MIQ_GEMM_ROWSPLIT");   // developer A/B switch: 0 = never cut
    if (sw && sw[0] == '0') return M;
    const int64_t tn = ceil_div(N, 256), tm = ceil_div(M, 256), tiles = tm * tn;
    if (tiles <= 256 || tiles % 256 == 0) return M;
    const int64_t tm_main = (tiles / 256) * 256 / tn;   // tile rows inside the whole rounds
    if (tm_main < 1 || tm_main >= tm) return M;
    const int64_t main_tiles = tm_main * tn, tail_tiles = ceil_div(M - tm_main * 256, 256) * tn;
    if (main_tiles % 256 != 0 && main_tiles % 256 < 200) return M;   // the rounds kept must be (nearly) full
    if (tail_tiles > 96) return M;                                   // ... and the rest small against a round
    return tm_main * 256;
}

// How one call is cut into launches.  Made once per call (the workspace query makes it again for the same
// sizes): up to three blocks of the output, each with the kernel plan it runs on.
//   wide:  rows [0, M) x channels [0, wide_n) on the 256 x 384 kernel (wide_n = 0: none)
//   head:  rows [0, head_m) x channels [wide_n, N) on `head`
//   tail:  rows [head_m, M) x channels [wide_n, N) on `tail` (head_m = M: none)
struct Schedule {
    int64_t wide_n, head_m;
    GemmPlan head, tail;
    size_t slab_bytes;   // int32 split-K slabs of the one block that needs them (the blocks run one after another)
};

// Channels for the 256 x 384 kernel: none in the product library.  Measured on MI355X (tools/gemm_wide_ab.py,
// profiles/r2_gemm_wide.txt) the wide tile moves 17 % fewer bytes L2 -> LDS and issues 22 % fewer LDS reads per
// multiply-add, holds the matrix pipes busy for the same share of cycles at the same clock, and ends within 1 % of
// the 256 x 256 kernel in time: the chip is power-limited on random int8 operands, not limited by that stream.  The
// kernel stays in the developer library as the measured alternative (MIQ_GEMM_WIDE = count of 384-channel tile
// columns to give it).
int64_t wide_columns(int64_t M, int64_t N, int64_t K) {
#ifdef MIQ_NO_DEV_SWITCHES
    (void)M; (void)N; (void)K;
    return 0;
#else
    const char* sw = dev_switch("MIQ_GEMM_WIDE");
    if (!sw || sw[0] == '0') return 0;
    if (M % 256 != 0 || K % 128 != 0 || M < 256 || N < 384) return 0;
    const int64_t forced = atoll(sw);
    return forced * 384 <= N ? forced * 384 : 0;
#endif
}

Schedule make_schedule(int64_t M, int64_t N, int64_t K, bool mfma_ok) {
    Schedule s{};
    s.head_m = M;
    if (!mfma_ok) return s;
    s.wide_n = wide_columns(M, N, K);
    const int64_t n_rest = N - s.wide_n;
    if (n_rest == 0) return s;
    s.head = plan_gemm(M, n_rest, K);
    s.head_m = rows_in_whole_rounds(M, n_rest, s.head);
    auto slabs = [&](const GemmPlan& p, int64_t rows) {
        return p.splits > 1 ? static_cast<size_t>(round_up(static_cast<int64_t>(p.splits) * rows * n_rest * 4, 256)) : size_t{0};
    };
    if (s.head_m < M) {
        s.head = plan_gemm(s.head_m, n_rest, K);   // whole rounds of 256 x 256 tiles: never split
        s.tail = plan_gemm(M - s.head_m, n_rest, K);
        s.slab_bytes = slabs(s.tail, M - s.head_m);
    } else {
        s.slab_bytes = slabs(s.head, M);
    }
    return s;
}

// Workspace = [re-laid [N,K] weight, only for MIQ_W_KN][split-K int32 slabs, only for thin problems].
size_t transposed_bytes(int w_layout, int64_t N, int64_t K) {
    return w_layout == MIQ_W_KN ? static_cast<size_t>(round_up(N * K, 256)) : 0;
}

// The address of `scratch` (top of this file) on the current device, asked of the runtime once per device.

int scratch_for_current_device(void** out) {
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static void* ptrs[kMaxDevices] = {};
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(MIQ_ELAUNCH, "miq_gemm_i8_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail(MIQ_ERANGE, "miq_gemm_i8_noalloc: device index %d is beyond %d", dev, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(MIQ_ELAUNCH, "miq_gemm_i8_noalloc: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
        ptrs[dev] = p;
    }
    *out = ptrs[dev];
    return MIQ_OK;
}

bool mfma_path_ok(const int8_t* A, const int8_t* W, int w_layout, int64_t ldo, int64_t K, const void* out) {
    // 16-byte vector stores in the epilogue (four fp32 / int32, or eight bf16 when ldo % 8 == 0; four int8 as a dword)
    // need 16-byte aligned output rows whenever ldo % 4 == 0.  A [K,N] weight is re-laid into the 256-byte aligned
    // workspace.
    const bool out_vec_ok = (ldo % 4 != 0) || aligned(out, 16);
    return K > 0 && K % 16 == 0 && aligned(A, 16) && (w_layout == MIQ_W_KN || aligned(W, 16)) && out_vec_ok;
}

size_t out_esz(int out_dtype) { return out_dtype == MIQ_I8 ? 1 : out_dtype == MIQ_BF16 ? 2 : 4; }

// Output placement and requantisation of one call (miq_gemm_i8_ex; the plain entry points use the defaults).
struct OutSpec {
    int layout;       // MIQ_OUT_MN: out[m * ldo + n]; MIQ_OUT_NM: out[n * ldo + m]
    int64_t ldo;      // elements between two rows of `out`
    float out_div;    // MIQ_I8: clamp(rint(y / out_div))
};

int validate(const int8_t* A, const int8_t* W, int w_layout, int64_t M, int64_t N, int64_t K, const float* bias,
             const float* alpha, const void* out, int out_dtype, const OutSpec& os, bool extended, const char* who) {
    MIQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, MIQ_EINVAL, "%s: negative size (M=%lld N=%lld K=%lld)", who, (long long)M,
                (long long)N, (long long)K);
    MIQ_REQUIRE(w_layout == MIQ_W_NK || w_layout == MIQ_W_KN, MIQ_EINVAL, "%s: unknown w_layout %d", who, w_layout);
    if (extended) {
        MIQ_REQUIRE(out_dtype == MIQ_F32 || out_dtype == MIQ_BF16 || out_dtype == MIQ_I32 || out_dtype == MIQ_I8, MIQ_EINVAL,
                    "%s: out_dtype must be MIQ_F32, MIQ_BF16, MIQ_I32 or MIQ_I8 (got %d)", who, out_dtype);
        MIQ_REQUIRE(os.layout == MIQ_OUT_MN || os.layout == MIQ_OUT_NM, MIQ_EINVAL, "%s: unknown out_layout %d", who, os.layout);
        MIQ_REQUIRE(os.ldo >= (os.layout == MIQ_OUT_MN ? N : M), MIQ_EINVAL,
                    "%s: ldo (%lld) is smaller than a row of the output (%lld)", who, (long long)os.ldo,
                    (long long)(os.layout == MIQ_OUT_MN ? N : M));
        MIQ_REQUIRE(out_dtype != MIQ_I8 || (os.out_div > 0.0f && os.out_div < __builtin_inff()), MIQ_EINVAL,
                    "%s: requant_scale must be positive and finite (got %g)", who, (double)os.out_div);
    } else {
        MIQ_REQUIRE(out_dtype == MIQ_F32 || out_dtype == MIQ_BF16 || out_dtype == MIQ_I32, MIQ_EINVAL,
                    "%s: out_dtype must be MIQ_F32, MIQ_BF16 or MIQ_I32 (got %d)", who, out_dtype);
    }
    if (M == 0 || N == 0) return MIQ_OK;
    MIQ_REQUIRE(out, MIQ_EINVAL, "%s: null output", who);
    MIQ_REQUIRE(K == 0 || (A && W), MIQ_EINVAL, "%s: null operand", who);
    MIQ_REQUIRE(out_dtype == MIQ_I32 || alpha, MIQ_EINVAL, "%s: alpha is required for a dequantised output", who);
    MIQ_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) && M * K < (1ll << 31) && N * K < (1ll << 31) &&
                    os.ldo < (1ll << 31),
                MIQ_ERANGE, "%s: operands are indexed with 31-bit byte offsets (M*K and N*K < 2^31)", who);
    MIQ_REQUIRE(aligned(out, out_esz(out_dtype)) && aligned(alpha, 4) && aligned(bias, 4), MIQ_EALIGN,
                "%s: out/alpha/bias not aligned to their element size", who);
    return MIQ_OK;
}

// Everything after validation and the workspace check: the optional weight re-layout and the launches of the
// schedule.  `workspace` holds transposed_bytes() + sched.slab_bytes bytes.  (M, N) are the caller's; a transposed
// output swaps the operands' roles: the kernels then see the weight as their row operand (Mk = N rows), the activation
// as their column operand (Nk = M), write out[n * ldo + m] as an ordinary row-major result and take bias / alpha by row.
int run_schedule(const Schedule& sched, const int8_t* A, const int8_t* W, int w_layout, int64_t M, int64_t N, int64_t K,
                 const float* bias, const float* alpha, float beta, void* out, int out_dtype, const OutSpec& os, bool mfma_ok,
                 void* workspace, void* stream) {
    hipStream_t st = as_stream(stream);
    const size_t esz = out_esz(out_dtype);
    int* slabs = nullptr;
    if (K > 0) {
        if (w_layout == MIQ_W_KN) {
            int rc = miq_transpose_i8(W, K, N, static_cast<int8_t*>(workspace), stream);
            if (rc != MIQ_OK) return rc;
            W = static_cast<const int8_t*>(workspace);
        }
        if (sched.slab_bytes) slabs = reinterpret_cast<int*>(static_cast<char*>(workspace) + transposed_bytes(w_layout, N, K));
    }
    const bool tr = os.layout == MIQ_OUT_NM;
    const int64_t Mk = tr ? N : M, Nk = tr ? M : N;
    GemmArgs g{tr ? W : A, tr ? A : W, bias, alpha, out, beta, static_cast<int>(Mk), static_cast<int>(Nk), static_cast<int>(K),
               static_cast<int>(os.ldo), 0, 0, 0, nullptr, tr ? 1 : 0, os.out_div};
    if (!mfma_ok) {
        switch (out_dtype) {
            case MIQ_F32: return launch_any<kOutF32>(g, st);
            case MIQ_BF16: return launch_any<kOutBF16>(g, st);
            case MIQ_I8: return launch_any<kOutI8>(g, st);
            default: return launch_any<kOutI32>(g, st);
        }
    }
    auto run = [&](const GemmArgs& part, const GemmPlan& plan, int* part_slabs) {
        switch (out_dtype) {
            case MIQ_F32: return launch<kOutF32>(part, plan, part_slabs, st);
            case MIQ_BF16: return launch<kOutBF16>(part, plan, part_slabs, st);
            case MIQ_I8: return launch<kOutI8>(part, plan, part_slabs, st);
            default: return launch<kOutI32>(part, plan, part_slabs, st);
        }
    };
    if (sched.wide_n) {   // (developer library only; plain row-major outputs)
        GemmArgs wide = g;
        wide.N = static_cast<int>(sched.wide_n);
        int rc;
        switch (out_dtype) {
            case MIQ_F32: rc = launch_wide<kOutF32>(wide, st); break;
            case MIQ_BF16: rc = launch_wide<kOutBF16>(wide, st); break;
            case MIQ_I8: rc = launch_wide<kOutI8>(wide, st); break;
            default: rc = launch_wide<kOutI32>(wide, st); break;
        }
        if (rc != MIQ_OK || sched.wide_n == Nk) return rc;
        // the remaining channels: a column block of the same output
        g.W = g.W + sched.wide_n * K;
        if (bias && !tr) g.bias = bias + sched.wide_n;
        if (alpha && !tr) g.alpha = alpha + sched.wide_n;
        g.out = static_cast<char*>(out) + static_cast<size_t>(sched.wide_n) * esz;
        g.N = static_cast<int>(Nk - sched.wide_n);
    }
    if (sched.head_m < Mk) {   // whole rounds of 256 x 256 tiles, then the remaining rows as their own problem
        GemmArgs head = g, tail = g;
        head.M = static_cast<int>(sched.head_m);
        if (int rc = run(head, sched.head, nullptr)) return rc;
        tail.A = g.A + sched.head_m * K;
        tail.out = static_cast<char*>(g.out) + static_cast<size_t>(sched.head_m) * os.ldo * esz;
        tail.M = static_cast<int>(Mk - sched.head_m);
        if (tr) {   // row parameters move with the rows
            if (bias) tail.bias = bias + sched.head_m;
            if (alpha) tail.alpha = alpha + sched.head_m;
        }
        return run(tail, sched.tail, slabs);
    }
    return run(g, sched.head, slabs);
}

// The common body of the entry points.  scratch_mode: the workspace is the library's own buffer (miq_gemm_i8_noalloc).
int gemm_i8_call(const char* who, bool extended, bool scratch_mode, const int8_t* A, const int8_t* W, int w_layout, int64_t M,
                 int64_t N, int64_t K, const float* bias, const float* alpha, float beta, void* out, int out_dtype, OutSpec os,
                 void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = validate(A, W, w_layout, M, N, K, bias, alpha, out, out_dtype, os, extended, who)) return rc;
    if (M == 0 || N == 0) return MIQ_OK;
    const bool tr = os.layout == MIQ_OUT_NM;
    const bool mfma_ok = mfma_path_ok(A, W, w_layout, os.ldo, K, out);
    const Schedule sched = make_schedule(tr ? N : M, tr ? M : N, K, mfma_ok);
    const size_t need = K > 0 ? transposed_bytes(w_layout, N, K) + sched.slab_bytes : 0;
    if (scratch_mode) {
        MIQ_REQUIRE(need <= kScratchBytes, MIQ_EWORKSPACE,
                    "%s: this call needs %zu bytes of scratch, the built-in buffer holds %zu: use miq_gemm_i8 "
                    "with a caller workspace", who, need, kScratchBytes);
        workspace = nullptr;
        if (need) {
            if (int rc = scratch_for_current_device(&workspace)) return rc;
        }
    } else if (need) {
        MIQ_REQUIRE(workspace && workspace_bytes >= need, MIQ_EWORKSPACE,
                    "%s: this call needs %zu bytes of workspace (got %zu): see %s_workspace", who, need, workspace_bytes, who);
        MIQ_REQUIRE(miq::aligned(workspace, 256), MIQ_EALIGN, "%s: workspace must be 256-byte aligned", who);
    }
    return run_schedule(sched, A, W, w_layout, M, N, K, bias, alpha, beta, out, out_dtype, os, mfma_ok, workspace, stream);
}

}  // namespace

// What validate() accepts (M*K and N*K below 2^31): anything else gets no workspace here and MIQ_ERANGE from the call.
bool sizes_in_range(int64_t M, int64_t N, int64_t K) {
    return miq::all_below_2_31({M, N, K}) && M * K < (1ll << 31) && N * K < (1ll << 31);
}

extern "C" size_t miq_gemm_i8_workspace(int w_layout, int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || !sizes_in_range(M, N, K)) return 0;
    // (alignment of the caller's pointers is unknown here: a problem that ends on the one-thread-per-output
    // kernel needs no slabs, so this is an upper bound for it)
    return transposed_bytes(w_layout, N, K) + make_schedule(M, N, K, K % 16 == 0).slab_bytes;
}

extern "C" int miq_gemm_i8(const int8_t* A, const int8_t* W, int w_layout, int64_t M, int64_t N, int64_t K,
                           const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                           void* workspace, size_t workspace_bytes, void* stream) {
    return gemm_i8_call("miq_gemm_i8", false, false, A, W, w_layout, M, N, K, bias, alpha, beta, out, out_dtype,
                        OutSpec{MIQ_OUT_MN, N, 1.0f}, workspace, workspace_bytes, stream);
}

extern "C" size_t miq_gemm_i8_ex_workspace(int w_layout, int out_layout, int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || !sizes_in_range(M, N, K)) return 0;
    const bool tr = out_layout == MIQ_OUT_NM;
    return transposed_bytes(w_layout, N, K) + make_schedule(tr ? N : M, tr ? M : N, K, K % 16 == 0).slab_bytes;
}

extern "C" int miq_gemm_i8_ex(const int8_t* A, const int8_t* W, int w_layout, int64_t M, int64_t N, int64_t K,
                              const float* bias, const float* alpha, float beta, void* out, int out_dtype, int out_layout,
                              int64_t ldo, float requant_scale, void* workspace, size_t workspace_bytes, void* stream) {
    return gemm_i8_call("miq_gemm_i8_ex", true, false, A, W, w_layout, M, N, K, bias, alpha, beta, out, out_dtype,
                        OutSpec{out_layout, ldo, requant_scale}, workspace, workspace_bytes, stream);
}

extern "C" size_t miq_gemm_i8_noalloc_max_workspace(void) { return kScratchBytes; }

extern "C" int miq_gemm_i8_noalloc(const int8_t* A, const int8_t* W, int w_layout, int64_t M, int64_t N, int64_t K,
                                   const float* bias, const float* alpha, float beta, void* out, int out_dtype,
                                   void* stream) {
    return gemm_i8_call("miq_gemm_i8_noalloc", false, true, A, W, w_layout, M, N, K, bias, alpha, beta, out, out_dtype,
                        OutSpec{MIQ_OUT_MN, N, 1.0f}, nullptr, 0, stream);
}
