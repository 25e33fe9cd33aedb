// dense_probe.hip -- where the time of dense_planes_pipe_kernel goes on the LSTM2 projection shape (M = 33 * 1024, N = 1280,
// K = 256): both kernels of c3_dense.h with parts switched off (ABL bits).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc tools/dense_probe.hip -o tools/bin/dense_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
static std::string g_err;
static int fail(const char *, ...) { return -1; }
#include "c3_dense.h"
using namespace c3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class F>
static float time_us(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024;
    const int M = 33 * B, N = 1280, K = 256;
    std::vector<uint16_t> ha((size_t)M * K * 2), hw((size_t)N * K * 2);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    // fp16 values in roughly (-1, 1): sign, exponent 8..14, random mantissa (hi and lo pieces alike: power realism)
    for (auto &x : ha) { const uint32_t r = rnd(); x = (uint16_t)(((r >> 31) << 15) | ((8 + (r >> 8) % 7) << 10) | (r & 0x3ff)); }
    for (auto &x : hw) { const uint32_t r = rnd(); x = (uint16_t)(((r >> 31) << 15) | ((6 + (r >> 8) % 7) << 10) | (r & 0x3ff)); }
    void *da, *dw;
    float *dc, *dbias;
    CK(hipMalloc(&da, ha.size() * 2));
    CK(hipMalloc(&dw, hw.size() * 2));
    CK(hipMalloc((void **)&dc, (size_t)M * N * 4));
    CK(hipMalloc((void **)&dbias, N * 4));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dbias, 0, N * 4));
    DensePlanesParams dp;
    dp.a = da, dp.w = dw, dp.bias = dbias, dp.c = dc, dp.post = dbias;
    dp.M = M, dp.N = N, dp.K = K, dp.tiles_n = N / kDnBN, dp.tiles = ((M + kDnBM - 1) / kDnBM) * dp.tiles_n;
    const int grid = dp.tiles < 256 ? dp.tiles : 256;
    const double gflop = 2.0 * M * N * K * 3 * 1e-9;
    printf("proj2 shape M=%d N=%d K=%d: %d tiles on %d workgroups, %.1f GFLOP executed (three piece products)\n", M, N, K, dp.tiles, grid, gflop);
    auto report = [&](const char *name, float us) { printf("  %-58s %7.1f us  %6.0f TF executed\n", name, us, gflop / us * 1e-3); };
#define RUN(abl, name) report(name, time_us([&] { hipLaunchKernelGGL((dense_planes_pipe_kernel<false, false, abl>), dim3(grid), dim3(kDnThreads), 0, 0, dp); }, 20))
    RUN(0, "pipe kernel");
    RUN(16, "  - result stores");
    RUN(1, "  - operand loads");
    RUN(2, "  - LDS staging writes");
    RUN(3, "  - loads - staging");
    RUN(8, "  - fragment reads");
    RUN(4, "  - matrix instructions");
    RUN(12, "  - fragment reads - matrix instructions");
    RUN(19, "  - loads - staging - stores");
    RUN(27, "  - loads - staging - stores - fragment reads (matrix only)");
    RUN(59, "  matrix instructions only, no barriers");
    RUN(23, "  fragment reads + barriers only");
    RUN(28, "  loads + staging + barriers only");
    RUN(15, "  stores + barriers only");
    {   // weights resident in registers (fragment order: any bytes do for timing)
        DenseWresParams wp;
        wp.a = da, wp.w = dw, wp.bias = dbias, wp.c = dc, wp.post_scale = 1.f / 256.f;
        wp.M = M, wp.N = N, wp.tiles_m = (M + kWrBM - 1) / kWrBM, wp.tiles_n = N / kWrBN, wp.lanes_per_xcd = 6;
        const int gw = 8 * wp.lanes_per_xcd * wp.tiles_n;
#define RUNR(abl, name) report(name, time_us([&] { hipLaunchKernelGGL((dense_planes_wres_kernel<abl>), dim3(gw), dim3(kDnThreads), 0, 0, wp); }, 20))
        RUNR(0, "weights-resident kernel (240 workgroups)");
        RUNR(16, "  - result stores");
        RUNR(1, "  - activation loads");
        RUNR(2, "  - LDS staging writes");
        RUNR(8, "  - fragment reads");
        RUNR(4, "  - matrix instructions");
        RUNR(32, "  - barriers");
        RUNR(19, "  - loads - staging - stores");
        RUNR(27, "  - loads - staging - stores - fragment reads (matrix + barriers)");
        RUNR(59, "  matrix instructions only");
        RUNR(15, "  stores + barriers only");
        for (int l : {4, 5, 6}) {
            wp.lanes_per_xcd = l;
            char nm[64];
            snprintf(nm, sizeof nm, "  full kernel, %d lanes per XCD (%d workgroups)", l, 8 * l * wp.tiles_n);
            report(nm, time_us([&] { hipLaunchKernelGGL((dense_planes_wres_kernel<0>), dim3(8 * l * wp.tiles_n), dim3(kDnThreads), 0, 0, wp); }, 20));
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
