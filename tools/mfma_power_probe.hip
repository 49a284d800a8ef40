// What the matrix pipes of an MI355X sustain on int8 data with nothing else going on: MFMAs over operands held in
// registers (no LDS, no memory traffic in the loop), one workgroup of eight waves per CU, on all-zero and on random int8 operands, for
// several instruction shapes and issue orders.  The chip lowers its clock under MFMAs on random data, so the rate on
// random operands is the ceiling of any int8 GEMM kernel on this part, whatever its schedule.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power_probe tools/mfma_power_probe.hip && ./mfma_power_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

// MODE 0: 16x16x64, 4 x 2 accumulators, two k-steps, k-step-major order (the GEMM kernel's phase)
// MODE 1: the same, weight fragment alternating 0 1 1 0
// MODE 2: the two k-steps of an accumulator back to back
// MODE 3: MODE 2 with the alternating weight fragment
// MODE 4: four k-steps of an accumulator back to back (operands of k-steps 2, 3 = those of 0, 1 swapped around)
// MODE 8: MODE 3 with the k-step order alternating 0 1 / 1 0 from one accumulator to the next (every MFMA then shares its
//         accumulator or one operand with the one before it)
// MODE 5: 32x32x32, 2 x 1 accumulators of 16 registers, four k-steps, k-step-major
// MODE 6: 32x32x32, the four k-steps of an accumulator back to back
// MODE 7: 32x32x32, 2 x 2 accumulators, two k-steps, k-step-major
template <int MODE>
__global__ __launch_bounds__(512) void probe(const v4i* __restrict__ src, int iters, int* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    v4i a[2][4], w[2][2];   // [k-step][fragment]: what one phase of the GEMM kernel holds
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[ks][i] = src[(ks * 6 + i) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 2; ++j) w[ks][j] = src[(ks * 6 + 4 + j) * 64 + lane];
    }
    if constexpr (MODE <= 4 || MODE == 8) {
        v4i c[4][2] = {};
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = (MODE == 1 && (i & 1)) ? 1 - jj : jj;
                            c[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[ks][j], a[ks][i], c[i][j], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = ((MODE == 3 || MODE == 8) && (i & 1)) ? 1 - jj : jj;
#pragma unroll
                        for (int kk = 0; kk < (MODE == 4 ? 4 : 2); ++kk) {
                            const int ks = (MODE == 8 && ((i * 2 + jj) & 1)) ? 1 - kk : kk;
                            c[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[ks & 1][j ^ (ks >> 1)], a[ks & 1][i ^ (ks >> 1)], c[i][j], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
        }
        int s = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s += c[i][j][0] ^ c[i][j][1] ^ c[i][j][2] ^ c[i][j][3];
        if (s == 0x7fffffff) sink[0] = s;
    } else {
        constexpr int NJ = MODE == 7 ? 2 : 1;
        v16i c[2][NJ] = {};
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 5) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        c[i][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[ks & 1][ks >> 1], a[ks & 1][i * 2 + (ks >> 1)], c[i][0], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else if constexpr (MODE == 6) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        c[i][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[ks & 1][ks >> 1], a[ks & 1][i * 2 + (ks >> 1)], c[i][0], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            c[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[ks][j], a[ks][i], c[i][j], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
            }
        }
        int s = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += c[i][j][e];
        if (s == 0x7fffffff) sink[0] = s;
    }
}

template <int MODE>
double run(const v4i* src, int* sink, int iters, const char* what, const char* data) {
    // int8 multiply-adds per iteration and wave: MODE 0-3: 16 MFMAs x 16K; MODE 4: 32 x 16K; 5, 6, 7: 8 x 32K
    const double macs = (MODE == 4 ? 32.0 : 16.0) * 16384.0;
    const int grid = 256;       // one workgroup of 8 waves per CU (100 KiB of unused LDS keeps a second one out): two waves per SIMD, as in the GEMM kernel
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 40; ++i) probe<MODE><<<grid, 512, 100 * 1024>>>(src, iters, sink);   // clock ramp
    CHECK(hipDeviceSynchronize());
    const int launches = 40;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < launches; ++i) probe<MODE><<<grid, 512, 100 * 1024>>>(src, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tops = 2.0 * macs * iters * 8.0 * grid * launches / (ms * 1e-3) / 1e12;
    printf("%-62s %-7s %7.1f us/launch  %6.0f TOPS  (%.1f %% of 5000)\n", what, data, ms / launches * 1e3, tops, tops / 50.0);
    fflush(stdout);
    return tops;
}

int main() {
    const size_t n = 12 * 64;   // twelve fragments of 64 lanes x 16 bytes
    std::vector<v4i> zero(n, v4i{0, 0, 0, 0}), rnd(n), small(n);
    uint64_t state = 0x9e3779b97f4a7c15ull;
    auto next = [&]() {
        state ^= state << 13;
        state ^= state >> 7;
        state ^= state << 17;
        return state;
    };
    for (size_t i = 0; i < n; ++i) {
        for (int e = 0; e < 4; ++e) {
            rnd[i][e] = static_cast<int>(next() >> 16);
            uint32_t v = 0;
            for (int b = 0; b < 4; ++b) v |= static_cast<uint32_t>(static_cast<uint8_t>(static_cast<int8_t>(static_cast<int>(next() % 7) - 3))) << (8 * b);
            small[i][e] = static_cast<int>(v);
        }
    }
    v4i *dz, *dr, *ds;
    int* sink;
    CHECK(hipMalloc(&dz, n * sizeof(v4i)));
    CHECK(hipMalloc(&dr, n * sizeof(v4i)));
    CHECK(hipMalloc(&ds, n * sizeof(v4i)));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemcpy(dz, zero.data(), n * sizeof(v4i), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dr, rnd.data(), n * sizeof(v4i), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(ds, small.data(), n * sizeof(v4i), hipMemcpyHostToDevice));
    const int iters = 2000;
    for (int round = 0; round < 2; ++round) {
        struct {
            const v4i* p;
            const char* name;
        } data[3] = {{dz, "zeros"}, {ds, "|v|<=3"}, {dr, "random"}};
        for (auto& d : data) {
            run<0>(d.p, sink, iters, "16x16x64, 4x2 accumulators, k-step-major", d.name);
            run<1>(d.p, sink, iters, "16x16x64, k-step-major, weight fragment 0 1 1 0", d.name);
            run<2>(d.p, sink, iters, "16x16x64, two k-steps of an accumulator back to back", d.name);
            run<3>(d.p, sink, iters, "16x16x64, back to back + weight fragment 0 1 1 0", d.name);
            run<4>(d.p, sink, iters, "16x16x64, four k-steps of an accumulator back to back", d.name);
            run<8>(d.p, sink, iters, "16x16x64, back to back, 0 1 1 0, k-steps 0 1 / 1 0", d.name);
            run<5>(d.p, sink, iters, "32x32x32, 2x1 accumulators, k-step-major", d.name);
            run<6>(d.p, sink, iters, "32x32x32, four k-steps of an accumulator back to back", d.name);
            run<7>(d.p, sink, iters, "32x32x32, 2x2 accumulators, k-step-major", d.name);
        }
    }
    return 0;
}
