// tools/census_probe.hip -- which workgroups share a CU?  512 workgroups of 256 threads with 66 KB of LDS each (two fit a CU),
// every one records HW_REG_HW_ID / XCC_ID and its start clock.  Speed-only knowledge (placement is not a contract).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void census(unsigned *out, int spin) {
    __shared__ char lds[66000];
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x * 4 + 0] = hw, out[blockIdx.x * 4 + 1] = xcc, out[blockIdx.x * 4 + 2] = (unsigned)t0, out[blockIdx.x * 4 + 3] = lds[5];
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}
int main() {
    for (int grid : {512, 768}) {
        unsigned *d;
        hipMalloc(&d, grid * 16);
        hipLaunchKernelGGL(census, dim3(grid), dim3(256), 0, 0, d, 200);
        hipDeviceSynchronize();
        std::vector<unsigned> h(grid * 4);
        hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> by_cu;
        for (int b = 0; b < grid; ++b) {
            const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        }
        printf("grid %d: %zu distinct (xcc, se, sh, cu) places\n", grid, by_cu.size());
        int shown = 0;
        std::map<int, int> delta_hist, count_hist;
        for (auto &kv : by_cu) {
            count_hist[(int)kv.second.size()]++;
            if (kv.second.size() >= 2) delta_hist[kv.second[1] - kv.second[0]]++;
            if (shown++ < 12) {
                printf("  place %05x:", kv.first);
                for (int b : kv.second) printf(" %d(hw %08x t %u)", b, h[b * 4], h[b * 4 + 2]);
                printf("\n");
            }
        }
        printf("  workgroups per place:");
        for (auto &kv : count_hist) printf(" %d x%d", kv.first, kv.second);
        printf("\n  index distance between the first two workgroups of a place:");
        for (auto &kv : delta_hist) printf(" %d x%d", kv.first, kv.second);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
