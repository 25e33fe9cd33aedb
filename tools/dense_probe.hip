// dense_probe.hip -- where the time of dense_planes_pipe_kernel goes on the LSTM2 projection shape (M = 33 * 1024, N = 1280,
// K = 256): both kernels of c3_dense.h with parts switched off (ABL bits).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc tools/dense_probe.hip -o tools/bin/dense_probe
#include <hip/hip_runtime.h>
#include <cstring>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
static std::string g_err;
static int fail(const char *, ...) { return -1; }
#include "c3_dense.h"
#include "c3_conv3s2.h"
#include "c3_conv3s2q.h"
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
    auto report = [&](const char *name, float us) { printf("  %-58s %7.1f us  %6.0f TF executed\n", name, us, gflop / us * 1e3); };
#define RUN(abl, name) report(name, time_us([&] { hipLaunchKernelGGL((dense_planes_pipe_kernel<abl>), dim3(grid), dim3(kDnThreads), 0, 0, dp); }, 20))
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
        if (argc > 2 && !strcmp(argv[2], "gx2")) {  // round 6: the upper bound of a narrower gx2, three alternating repeats
            for (int rep = 0; rep < 3; ++rep) {
                RUNR(0, "weights-resident kernel, fp32 pre-activations (173 MB at B = 1024)");
                RUNR(256, "  three bytes per value (12-byte stores, 130 MB)");
                RUNR(128, "  two bytes per value (fp16, 8-byte stores, 86 MB)");
                RUNR(16, "  no result stores");
            }
            return 0;
        }
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
    {   // ---- the stride-2 convolutions (conv3: 45 x 17 x 64 -> 23 x 9 x 128; conv5: 23 x 9 x 128 -> 12 x 5 x 256), B = 256 windows
        for (int Bc : {256, 1000}) {
        const int shapes[2][6] = {{45, 17, 64, 23, 9, 128}, {23, 9, 128, 12, 5, 256}};
        for (int si = 0; si < 2; ++si) {
            const int Hin = shapes[si][0], Win = shapes[si][1], Cin = shapes[si][2], Ho = shapes[si][3], Wo = shapes[si][4], Co = shapes[si][5];
            const int Mc = Bc * Ho * Wo, Kc = 9 * Cin;
            const size_t abytes = (size_t)Bc * Hin * Win * Cin * 4, wbytes = (size_t)Co * Kc * 4, cbytes = (size_t)Mc * Co * 4;
            void *ca, *cw, *cc;
            CK(hipMalloc(&ca, abytes));
            CK(hipMalloc(&cw, wbytes));
            CK(hipMalloc(&cc, cbytes));
            // any fp16-looking bytes will do for timing: reuse the random pieces generated above
            for (size_t off = 0; off < abytes; off += ha.size() * 2) CK(hipMemcpy((char *)ca + off, ha.data(), std::min(ha.size() * 2, abytes - off), hipMemcpyHostToDevice));
            for (size_t off = 0; off < wbytes; off += hw.size() * 2) CK(hipMemcpy((char *)cw + off, hw.data(), std::min(hw.size() * 2, wbytes - off), hipMemcpyHostToDevice));
            auto magic = [](int d) { return d <= 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) / (uint64_t)d) + 1); };
            const double gf = 2.0 * Mc * Co * (double)Kc * 3 * 1e-9;
            printf("stride-2 convolution %dx%dx%d -> %dx%dx%d, B = %d: M=%d N=%d K=%d, %.1f GFLOP executed\n", Hin, Win, Cin, Ho, Wo, Co, Bc, Mc, Co, Kc, gf);
            auto rep = [&](const char *name, float us) { printf("  %-66s %7.1f us  %6.0f TF executed\n", name, us, gf / us * 1e3); };
            if (Co % kS2BN == 0) {
                S2ConvParams sp;
                sp.a = ca, sp.wf = cw, sp.bias = dbias, sp.post = dbias, sp.c = cc, sp.range_flag = nullptr;
                sp.M = Mc, sp.N = Co, sp.NK = Kc / 64, sp.tiles_n = Co / kS2BN, sp.tiles = (Mc + kS2BM - 1) / kS2BM * sp.tiles_n;
                sp.Hin = Hin, sp.Win = Win, sp.Cin = Cin, sp.Ho = Ho, sp.Wo = Wo, sp.mg_hw = magic(Ho * Wo), sp.mg_w = magic(Wo);
                const int unit = 8 * sp.tiles_n;
                const int g1 = sp.tiles > 256 ? 256 / unit * unit : sp.tiles, g2 = sp.tiles > 512 ? 512 / unit * unit : sp.tiles;
                printf(" weights straight into registers (c3_conv3s2.h): %d tiles of 128 x 128, workgroups of 512 threads\n", sp.tiles);
#define RUNS(abl, pair, gs, name) rep(name, time_us([&] { hipLaunchKernelGGL((conv3x3_s2_planes_kernel<abl, pair>), dim3(gs), dim3(kS2Threads), 0, 0, sp); }, 20))
                RUNS(0, false, g1, "one workgroup per CU, two sets of fragment registers");
                RUNS(0, true, g1, "one workgroup per CU, one set of fragment registers");
                RUNS(0, true, g2, "two workgroups per CU (one set of fragment registers)");
                RUNS(0, false, g1, "one workgroup per CU, two sets of fragment registers (again)");
                RUNS(1, false, g1, "  - the pixel requests in the loop");
                RUNS(2, false, g1, "  - the weight loads");
                RUNS(16, false, g1, "  - epilogue");
                RUNS(8, false, g1, "  - fragment reads");
                RUNS(4, false, g1, "  - matrix instructions");
                RUNS(11, false, g1, "  matrix instructions + barriers only");
                RUNS(14, false, g1, "  pixel requests + barriers only");
                RUNS(13, false, g1, "  weight loads + barriers only");
                RUNS(31, false, g1, "  barriers only (prologue + launch)");
#undef RUNS
                // the four-wave form (a wave = 128 x 32 outputs): bit-identical output, times, ablations
                {
                    void *cc2;
                    CK(hipMalloc(&cc2, cbytes));
                    CK(hipMemset(cc, 0xff, cbytes));
                    CK(hipMemset(cc2, 0xff, cbytes));
                    hipLaunchKernelGGL((conv3x3_s2_planes_kernel<0, false>), dim3(g1), dim3(kS2Threads), 0, 0, sp);
                    S2ConvParams sq = sp;
                    sq.c = cc2;
                    hipLaunchKernelGGL((conv3x3_s2q_planes_kernel<0>), dim3(g2), dim3(kS2QThreads), 0, 0, sq);
                    CK(hipGetLastError());
                    CK(hipDeviceSynchronize());
                    std::vector<uint32_t> h1(cbytes / 4), h2(cbytes / 4);
                    CK(hipMemcpy(h1.data(), cc, cbytes, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(h2.data(), cc2, cbytes, hipMemcpyDeviceToHost));
                    size_t nd = 0, unw = 0;
                    for (size_t i = 0; i < h1.size(); ++i) nd += h1[i] != h2[i], unw += h2[i] == 0xffffffffu;
                    printf(" four waves of 128 x 32 (conv3x3_s2q_planes_kernel): %zu of %zu words differ from the eight-wave form, %zu unwritten\n", nd, h1.size(), unw);
                    const int g3 = sp.tiles > 768 ? 768 / unit * unit : sp.tiles;
#define RUNQ(abl, gs, name) rep(name, time_us([&] { hipLaunchKernelGGL((conv3x3_s2q_planes_kernel<abl>), dim3(gs), dim3(kS2QThreads), 0, 0, sq); }, 20))
                    RUNQ(0, g2, "four waves, two workgroups per CU");
                    RUNQ(0, g1, "four waves, one workgroup per CU");
                    RUNQ(1, g2, "  - the pixel requests in the loop");
                    RUNQ(2, g2, "  - the weight loads");
                    RUNQ(16, g2, "  - epilogue");
                    RUNQ(8, g2, "  - fragment reads");
                    RUNQ(11, g2, "  matrix instructions + barriers only");
                    RUNQ(0, g2, "four waves, two workgroups per CU (again)");
                    (void)g3;
#undef RUNQ
                    (void)hipFree(cc2);
                }
            }
            (void)hipFree(ca), (void)hipFree(cw), (void)hipFree(cc);
        }
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
