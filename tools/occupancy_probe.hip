// How many workgroups of a given LDS size and thread count share a CU: every workgroup records the wall
// clock (100 MHz) when it starts and ends, after spinning for about 20 us; the host counts the overlap.
//   hipcc --offload-arch=gfx950 -O2 tools/occupancy_probe.hip -o tools/bin/occupancy_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned long long* st, int pad_vgprs) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) st[2 * blockIdx.x] = t0;
    lds[threadIdx.x] = static_cast<char>(pad_vgprs);
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) st[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() + lds[threadIdx.x + 1] * 0;
}

int main() {
    unsigned long long* st;
    const int grid = 1024;
    hipMalloc(&st, grid * 16);
    for (int threads : {256, 512})
        for (int kb : {32, 40, 48, 52, 56, 60, 64, 72, 76, 78, 80, 96, 128, 160}) {
            const int bytes = kb * 1024;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
                printf("LDS %d KiB: attribute refused\n", kb);
                continue;
            }
            hipMemset(st, 0, grid * 16);
            spin<<<grid, threads, bytes>>>(st, 0);
            if (hipDeviceSynchronize() != hipSuccess) {
                printf("LDS %d KiB: launch failed\n", kb);
                continue;
            }
            std::vector<unsigned long long> h(grid * 2);
            hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost);
            std::vector<std::pair<unsigned long long, int>> ev;
            for (int i = 0; i < grid; ++i) {
                ev.push_back({h[2 * i], 1});
                ev.push_back({h[2 * i + 1], -1});
            }
            std::sort(ev.begin(), ev.end());
            int cur = 0, mx = 0;
            for (auto& e : ev) {
                cur += e.second;
                mx = std::max(mx, cur);
            }
            printf("threads %d LDS %3d KiB: max concurrent workgroups %d (%.2f per CU of 256)\n", threads, kb, mx, mx / 256.0);
        }
    return 0;
}
