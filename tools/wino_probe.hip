// tools/wino_probe.hip -- phase ablation of the 32x64 Winograd workgroups (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/wino_probe.hip -o /tmp/wino_probe && /tmp/wino_probe
// Times wino_conv_kernel_n64 and the persistent wino_conv_kernel_p, whole and with parts switched off (ABL bits:
// 1 no patch loads, 2 no transform/LDS writes, 4 no V loads, 8 no epilogue, 16 no MFMAs), on the three residual-block
// shapes of a B=256 full-alignment batch.  Numbers only -- correctness is the parity tests' job.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../clair3_amd/csrc/c3_wino_p.h"
using namespace c3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class K>
static float time_it(K launch, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}

static WinoParams make(const float *x, const float *zeros, const float *v, const float *bias, const float *res, float *out, int B, int H, int W, int C) {
    WinoParams wp;
    wp.x = x, wp.zeros = zeros, wp.v = v, wp.bias = bias, wp.res = res, wp.out = out;
    wp.B = B, wp.H = H, wp.W = W, wp.Cin = C, wp.Cout = C;
    wp.th = (H + 1) / 2, wp.tw = (W + 1) / 2, wp.P = B * wp.th * wp.tw;
    wp.tiles_n = C / 64, wp.tiles = ((wp.P + 31) / 32) * wp.tiles_n;
    return wp;
}
template <int ABL> static float run_n64(const WinoParams &wp) {
    return time_it([&] { hipLaunchKernelGGL((wino_conv_kernel_n64<true, ABL>), dim3(wp.tiles), dim3(256), 0, 0, wp); });
}
template <int ABL> static float run_p(const WinoParams &wp, int slots) {
    const int grid = std::min(wp.tiles, slots / wp.tiles_n * wp.tiles_n);
    return time_it([&] { hipLaunchKernelGGL((wino_conv_kernel_p<true, ABL>), dim3(grid), dim3(256), 0, 0, wp); });
}
template <int ABL> static float run_p16(const WinoParams &wp, int slots) {
    const int grid = std::min(wp.tiles, slots / wp.tiles_n * wp.tiles_n);
    return time_it([&] { hipLaunchKernelGGL((wino_conv_kernel_p<true, ABL, 0, true>), dim3(grid), dim3(256), 0, 0, wp); });
}
int main() {
    const int B = 256;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int slots = 2 * prop.multiProcessorCount;
    struct Shape { const char *name; int H, W, C; } shapes[3] = {{"res1", 45, 17, 64}, {"res2", 23, 9, 128}, {"res3", 12, 5, 256}};
    for (const Shape &sh : shapes) {
        const size_t n = (size_t)B * sh.H * sh.W * sh.C;
        float *x, *y, *r, *v, *bias, *zeros;
        CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&r, n * 4));
        CK(hipMalloc(&v, (size_t)16 * sh.C * sh.C * 4)); CK(hipMalloc(&bias, sh.C * 4)); CK(hipMalloc(&zeros, 256));
        std::vector<float> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 24) / 256.f - 0.5f;
        CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(r, h.data(), n * 4, hipMemcpyHostToDevice));
        std::vector<float> hv((size_t)16 * sh.C * sh.C);
        for (size_t i = 0; i < hv.size(); ++i) hv[i] = (float)((i * 40503u) & 1023) / 65536.f;
        CK(hipMemcpy(v, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(bias, 0, sh.C * 4)); CK(hipMemset(zeros, 0, 256));
        const WinoParams wp = make(x, zeros, v, bias, r, y, B, sh.H, sh.W, sh.C);
        const double fl = 2.0 * B * sh.H * sh.W * 9.0 * sh.C * sh.C;
        const double mf = 2.0 * 16.0 * wp.P * sh.C * sh.C;  // real MFMA flops (tiles padded to whole 2x2)
        float t[2][8];
        t[0][0] = run_n64<0>(wp), t[0][1] = run_n64<1>(wp), t[0][2] = run_n64<3>(wp), t[0][3] = run_n64<4>(wp);
        t[0][4] = run_n64<8>(wp), t[0][5] = run_n64<16>(wp), t[0][6] = run_n64<15>(wp), t[0][7] = run_n64<28>(wp);
        t[1][0] = run_p<0>(wp, slots), t[1][1] = run_p<1>(wp, slots), t[1][2] = run_p<3>(wp, slots), t[1][3] = run_p<4>(wp, slots);
        t[1][4] = run_p<8>(wp, slots), t[1][5] = run_p<16>(wp, slots), t[1][6] = run_p<15>(wp, slots), t[1][7] = run_p<28>(wp, slots);
        const char *wn[8] = {"full", "no patch loads", "no patch loads/transform", "no V loads", "no epilogue", "no MFMA", "MFMA + LDS reads only", "patch+transform only"};
        printf("== %s: %d workgroups of 32 tiles x 64 couts, %d slots; pure MFMA time at 157.3 TF = %.1f us\n", sh.name, wp.tiles, slots, mf / 157.3e6);
        for (int i = 0; i < 8; ++i)
            printf("  %-26s n64 %6.1f us (%5.1f TF-eq)   persistent %6.1f us (%5.1f TF-eq)\n", wn[i], t[0][i], fl / t[0][i] / 1e6, t[1][i], fl / t[1][i] / 1e6);
        {   // the F16 form (fp16x3 products): same ablations, then its phase trace
            float tf[8] = {run_p16<0>(wp, slots), run_p16<1>(wp, slots), run_p16<3>(wp, slots), run_p16<4>(wp, slots),
                           run_p16<8>(wp, slots), run_p16<16>(wp, slots), run_p16<15>(wp, slots), run_p16<28>(wp, slots)};
            for (int i = 0; i < 8; ++i) printf("  F16 %-26s persistent %6.1f us\n", wn[i], tf[i]);
            long long *tb; CK(hipMalloc(&tb, 4 * 256 * 16)); CK(hipMemset(tb, 0, 4 * 256 * 16));
            WinoParams wt = wp; wt.zeros = reinterpret_cast<const float *>(tb);
            const int grid = std::min(wt.tiles, slots / wt.tiles_n * wt.tiles_n);
            hipLaunchKernelGGL((wino_conv_kernel_p<true, 0, 2, true>), dim3(grid), dim3(256), 0, 0, wt);
            CK(hipDeviceSynchronize());
            std::vector<long long> ht(4 * 256 * 2);
            CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
            for (int w = 0; w < 1; ++w) {
                printf("  F16 trace wave %d (tag:+cycles):", w);
                for (int i = 1; i < 120 && ht[(w * 256 + i) * 2] != 0; ++i)
                    printf(" %lld:+%lld", ht[(w * 256 + i) * 2], ht[(w * 256 + i) * 2 + 1] - ht[(w * 256 + i - 1) * 2 + 1]);
                printf("\n");
            }
            hipFree(tb);
        }
        {   // phase trace of workgroup 0 (OPT bit1): shader-clock deltas between phase boundaries, wave 0
            long long *tb; CK(hipMalloc(&tb, 4 * 256 * 16)); CK(hipMemset(tb, 0, 4 * 256 * 16));
            WinoParams wt = wp; wt.zeros = reinterpret_cast<const float *>(tb);
            const int grid = std::min(wt.tiles, slots / wt.tiles_n * wt.tiles_n);
            hipLaunchKernelGGL((wino_conv_kernel_p<true, 0, 2>), dim3(grid), dim3(256), 0, 0, wt);
            CK(hipDeviceSynchronize());
            std::vector<long long> ht(4 * 256 * 2);
            CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
            for (int w = 0; w < 2; ++w) {
                printf("  trace wave %d (tag:+cycles):", w);
                for (int i = 1; i < 250 && ht[(w * 256 + i) * 2] != 0; ++i)
                    printf(" %lld:+%lld", ht[(w * 256 + i) * 2], ht[(w * 256 + i) * 2 + 1] - ht[(w * 256 + i - 1) * 2 + 1]);
                printf("\n");
            }
            hipFree(tb);
        }
        for (int s2 : {slots / 2}) {
            const int grid = std::min(wp.tiles, s2 / wp.tiles_n * wp.tiles_n);
            float tt = time_it([&] { hipLaunchKernelGGL((wino_conv_kernel_p<true, 0>), dim3(grid), dim3(256), 0, 0, wp); });
            printf("  persistent with grid %4d: %6.1f us\n", grid, tt);
        }
        hipFree(x); hipFree(y); hipFree(r); hipFree(v); hipFree(bias); hipFree(zeros);
    }
    return 0;
}
