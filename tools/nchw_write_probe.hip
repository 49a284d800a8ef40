// Write-rate probe for NCHW int32 convolution outputs: what HBM takes from store patterns shaped like a convolution
// epilogue, with no arithmetic in the way.  out[B][C][P] int32; a workgroup of 256 threads owns TP consecutive pixels of
// TC channels of one image and stores them in one of three instruction shapes:
//   shape 0  a wave's instruction = 64 lanes x 16 B = 1 KiB of ONE channel (what an LDS-transposed epilogue would issue)
//   shape 1  a wave's instruction = 16 channels x 64 B (an MFMA accumulator stored as it lies: lane&15 -> channel,
//            lane>>4 -> four pixels)
//   shape 2  a wave's instruction = 4 channels x 256 B
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/nchw_write_probe tools/nchw_write_probe.hip
// Run:   tools/nchw_write_probe            (prints one line per pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

template <int SHAPE, bool NT>
__global__ __launch_bounds__(256) void probe(int* __restrict__ out, int C, int P, int TP, int TC, int tiles_p, int tiles_c,
                                              int order) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int t = blockIdx.x;
    int b, tp, tc;
    if (order == 0) {          // channel tile innermost, then pixel tile, then image
        tc = t % tiles_c; t /= tiles_c; tp = t % tiles_p; b = t / tiles_p;
    } else {                   // pixel tile innermost, then channel tile, then image
        tp = t % tiles_p; t /= tiles_p; tc = t % tiles_c; b = t / tiles_c;
    }
    const int p0 = tp * TP, c0 = tc * TC;
    const v4i v{lane, wave, b, tp};
    auto put = [&](int c, int p) {
        if (c < C && p + 3 < P + 0 && p < p0 + TP) {
            v4i* o = reinterpret_cast<v4i*>(out + (static_cast<size_t>(b) * C + c) * P + p);
            if constexpr (NT) __builtin_nontemporal_store(v, o); else *o = v;
        }
    };
    if constexpr (SHAPE == 0) {
        for (int c = wave; c < TC; c += 4)
            for (int p = lane * 4; p < TP; p += 256) put(c0 + c, p0 + p);
    } else if constexpr (SHAPE == 1) {
        // a wave walks 16-channel x 16-pixel blocks: 64 pixels of the tile per wave and pass, as an MFMA tiling would
        for (int cb = 0; cb < TC; cb += 16)
            for (int pb = wave * 64; pb < TP; pb += 256)
                for (int i = 0; i < 4; ++i) put(c0 + cb + (lane & 15), p0 + pb + i * 16 + (lane >> 4) * 4);
    } else if constexpr (SHAPE == 2) {
        for (int cb = wave * 4; cb < TC; cb += 16)
            for (int pb = 0; pb < TP; pb += 64) put(c0 + cb + (lane >> 4), p0 + pb + (lane & 15) * 4);
    } else {
        // shape 3: a wave owns 112 consecutive pixels (TP = 448): per four channels one instruction of 256 B runs and one of 192 B runs
        for (int cb = 0; cb < TC; cb += 4) {
            put(c0 + cb + (lane >> 4), p0 + wave * 112 + (lane & 15) * 4);
            if ((lane & 15) < 12) put(c0 + cb + (lane >> 4), p0 + wave * 112 + 64 + (lane & 15) * 4);
        }
    }
}

template <int SHAPE, bool NT>
static float run(int* out, int B, int C, int P, int TP, int TC, int order, int reps) {
    const int tiles_p = (P + TP - 1) / TP, tiles_c = (C + TC - 1) / TC;
    const int grid = B * tiles_p * tiles_c;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<SHAPE, NT><<<grid, 256>>>(out, C, P, TP, TC, tiles_p, tiles_c, order);
    CK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        probe<SHAPE, NT><<<grid, 256>>>(out + (r % 2) * (static_cast<size_t>(B) * C * P), C, P, TP, TC, tiles_p, tiles_c, order);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[r], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    return ms[reps / 2];
}

int main() {
    struct Layer { const char* name; int B, C, P; } layers[] = {
        {"stem 64ch 112x112", 256, 64, 12544}, {"64ch 56x56", 256, 64, 3136}, {"128ch 28x28", 256, 128, 784},
        {"256ch 14x14", 256, 256, 196}};
    for (const Layer& L : layers) {
        const size_t n = static_cast<size_t>(L.B) * L.C * L.P;
        int* out;
        CK(hipMalloc(&out, 2 * n * 4));
        const double mb = n * 4 / 1e6;
        printf("%s: %.0f MB\n", L.name, mb);
        const int tps[] = {256};
        const int tcs[] = {16, 64, 256};
        for (int TP : tps) {
            if (TP > L.P && TP != 256) continue;
            for (int TC : tcs) {
                if (TC > L.C) continue;
                for (int order = 0; order < 2; ++order) {
                    float a = run<0, false>(out, L.B, L.C, L.P, TP, TC, order, 9);
                    float an = run<0, true>(out, L.B, L.C, L.P, TP, TC, order, 9);
                    float b = TP >= 256 ? run<1, false>(out, L.B, L.C, L.P, TP, TC, order, 9) : 0.f;
                    float bn = TP >= 256 ? run<1, true>(out, L.B, L.C, L.P, TP, TC, order, 9) : 0.f;
                    float c = run<2, false>(out, L.B, L.C, L.P, TP, TC, order, 9);
                    printf("  TP %5d TC %3d order %d | 1KiB/instr %7.1f us %5.2f TB/s (nt %7.1f) | 16ch x 64B %7.1f us %5.2f TB/s (nt %7.1f) | 4ch x 256B %7.1f us\n",
                           TP, TC, order, a * 1e3, mb / a / 1e3 * 1e-3 * 1e3, an * 1e3, b * 1e3, b > 0 ? mb / b / 1e3 : 0.0, bn * 1e3, c * 1e3);
                    fflush(stdout);
                }
            }
        }
        if (L.P % 448 == 0) {
            for (int TC : tcs) {
                if (TC > L.C) continue;
                float a = run<3, false>(out, L.B, L.C, L.P, 448, TC, 0, 9), an = run<3, true>(out, L.B, L.C, L.P, 448, TC, 0, 9);
                float c = run<2, true>(out, L.B, L.C, L.P, 256, TC, 0, 9);
                printf("  TC %3d | 112-pixel waves (4ch x 256B + 4ch x 192B): cached %7.1f us, nt %7.1f us | 4ch x 256B TP 256 nt %7.1f us\n", TC, a * 1e3, an * 1e3, c * 1e3);
            }
        }
        CK(hipFree(out));
    }
    return 0;
}
