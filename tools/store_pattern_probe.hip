// What shape of store instruction an MI355X writes fast: every wave-instruction writes 1 KiB as 1024 / S segments of S contiguous
// bytes, the segments one channel plane (50176 bytes) apart, consecutive instructions of a wave extending each segment -- the
// NCHW output pattern of the convolution kernels (S = 64: sixteen planes x 64 bytes) against whole-line and fill-like ones.
// 822 MB ([256][64][112 x 112] fp32) per launch.
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern_probe tools/store_pattern_probe.hip && ./store_pattern_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                    \
        }                                                               \
    } while (0)

constexpr int kPlane = 112 * 112 * 4;   // bytes of a channel plane
constexpr int kPlanes = 64, kImages = 256, kChunks = 7;

// S = bytes per segment; NT = nontemporal stores; a wave owns (image, group of 1024 / S planes, a seventh of the plane)
template <int S, bool NT>
__global__ __launch_bounds__(256) void probe(char* __restrict__ out) {
    constexpr int nseg = 1024 / S, lanes_per_seg = S / 16, groups = kPlanes / nseg, ipw = kPlane / (S * kChunks);
    const int lane = threadIdx.x & 63;
    const long wave = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) >> 6;
    const long waves_per_image = static_cast<long>(groups) * kChunks;
    const long b = wave / waves_per_image, rest = wave % waves_per_image;
    if (b >= kImages) return;
    const int pg = static_cast<int>(rest / kChunks), c = static_cast<int>(rest % kChunks);
    const int seg = lane / lanes_per_seg, within = lane % lanes_per_seg;
    char* p = out + ((b * kPlanes + pg * nseg + seg) * static_cast<long>(kPlane)) + static_cast<long>(c) * ipw * S + within * 16;
    const v4f v{1.f, 2.f, 3.f, static_cast<float>(lane)};
    for (int i = 0; i < ipw; ++i) {
        if constexpr (NT)
            __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p + static_cast<long>(i) * S));
        else
            *reinterpret_cast<v4f*>(p + static_cast<long>(i) * S) = v;
    }
}

template <int S, bool NT>
void run(char* bufs[3]) {
    constexpr int nseg = 1024 / S, groups = kPlanes / nseg;
    const long waves = static_cast<long>(kImages) * groups * kChunks;
    const int grid = static_cast<int>((waves + 3) / 4);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 6; ++i) probe<S, NT><<<grid, 256>>>(bufs[i % 3]);
    CHECK(hipEventRecord(e0));
    const int n = 12;
    for (int i = 0; i < n; ++i) probe<S, NT><<<grid, 256>>>(bufs[i % 3]);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = static_cast<double>(kImages) * kPlanes * kPlane;
    printf("%4d-byte segments (%2d planes an instruction), %s stores: %7.1f us  %5.2f TB/s\n", S, nseg, NT ? "nontemporal" : "cached     ",
           ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    char* bufs[3];
    for (auto& b : bufs) CHECK(hipMalloc(&b, static_cast<size_t>(kImages) * kPlanes * kPlane));
    for (int round = 0; round < 2; ++round) {
        run<64, false>(bufs);
        run<64, true>(bufs);
        run<128, false>(bufs);
        run<128, true>(bufs);
        run<256, false>(bufs);
        run<256, true>(bufs);
        run<1024, false>(bufs);
        run<1024, true>(bufs);
    }
    return 0;
}
