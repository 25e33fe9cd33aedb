// tools/conv_probe.hip -- phase ablation of conv3x3_planes_kernel (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I clair3_amd/csrc tools/conv_probe.hip -o /tmp/conv_probe && /tmp/conv_probe
// Times the kernel on the three residual-block shapes of a B = 256 full-alignment batch, whole and with parts switched off (ABL
// bits: 1 no weight loads, 2 no halo loads after the first tile, 4 no epilogue, 8 no matrix instructions), at one and two
// workgroups per CU, with the second workgroup of a CU started later (skew), and prints a shader-clock trace of workgroups 0 and
// 256 (the two that share a CU: tools/census_probe.hip).  Numbers only -- correctness is the parity tests' job.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../clair3_amd/csrc/c3_conv3.h"
using namespace c3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class K>
static float time_it(K launch, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
template <int C, int ABL> static float run(const PlaneConvParams &cp, int grid) {
    return time_it([&] { hipLaunchKernelGGL((conv3x3_planes_kernel<C, true, ABL>), dim3(grid), dim3(kPlThreads), 0, 0, cp); });
}
template <int C> static int shape(const char *name, int B, int H, int W, int cus) {
    constexpr int NS = C / 64;
    const size_t bytes = (size_t)B * H * W * C * 4;
    void *x, *y, *r, *wf; float *bias, *post; uint32_t *flag;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&r, bytes));
    CK(hipMalloc(&wf, (size_t)9 * C * C * 4)); CK(hipMalloc(&bias, C * 4)); CK(hipMalloc(&post, C * 4)); CK(hipMalloc(&flag, 256));
    // realistic plane activations: post-ReLU values (a third of them zero) as genuine hi / lo fp16 pairs -- the power draw of
    // a matrix instruction (and with it the clock the chip sustains) depends on how many operand bits toggle
    std::vector<_Float16> h(bytes / 2);
    for (size_t g = 0; g < h.size() / 128; ++g)  // one (pixel, slab) group: [hi 64][lo 64]
        for (int c = 0; c < 64; ++c) {
            const uint32_t u = (uint32_t)((g * 64 + c) * 2654435761u);
            const float xv = (u % 3 == 0) ? 0.f : (float)(u >> 8) / 16777216.f * 3.f;
            const _Float16 hi = (_Float16)xv;
            h[g * 128 + c] = hi, h[g * 128 + 64 + c] = (_Float16)(xv - (float)hi);
        }
    CK(hipMemcpy(x, h.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(r, h.data(), bytes, hipMemcpyHostToDevice));
    std::vector<uint16_t> hw((size_t)9 * C * C * 2);  // any fragment order will do for timing: genuine hi / lo pairs per 16-byte piece pair
    for (size_t g = 0; g < hw.size() / 1024; ++g)   // [hi: 64 lanes x 8][lo: 64 lanes x 8]
        for (int c = 0; c < 512; ++c) {
            const uint32_t u = (uint32_t)((g * 512 + c) * 40503u + 12345u) * 2654435761u;
            const float xv = ((float)(u >> 8) / 16777216.f - 0.5f) * 8.f;  // folded weights times the packing scale
            const _Float16 hi = (_Float16)xv, lo = (_Float16)(xv - (float)hi);
            memcpy(&hw[g * 1024 + c], &hi, 2), memcpy(&hw[g * 1024 + 512 + c], &lo, 2);
        }
    CK(hipMemcpy(wf, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hb(C, 0.f), hp(C, 1.f / 4096.f);
    CK(hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(post, hp.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemset(flag, 0, 256));
    PlaneConvParams cp;
    cp.x = x, cp.wf = wf, cp.bias = bias, cp.res = r, cp.out = y, cp.range_flag = flag, cp.post = post, cp.pre = post;
    cp.M = B * H * W, cp.H = H, cp.W = W;
    cp.mg_hw = (uint32_t)((1ull << 32) / (uint64_t)(H * W) + 1), cp.mg_w = (uint32_t)((1ull << 32) / (uint64_t)W + 1);  // fast_div magics
    const int tiles_m = (cp.M + kPlBM - 1) / kPlBM;
    cp.tiles = tiles_m * NS;
    const int unit = 8 * NS;
    auto grid_for = [&](int s) { return cp.tiles <= s ? cp.tiles : s / unit * unit; };
    const int g = grid_for(2 * cus);
    const double mf = 2.0 * tiles_m * kPlBM * (double)C * 9.0 * C * 3;
    printf("== %s: M = %d, %d tiles of %d pixels, grid %d; executed matrix work %.1f GFLOP = %.1f us at 2500 TF; in+out+res %.0f MB\n", name, cp.M, cp.tiles, kPlBM, g,
           mf / 1e9, mf / 2500e6, 3 * bytes / 1e6);
    printf("  full                         %6.1f us\n", run<C, 0>(cp, g));
    printf("  no weight loads              %6.1f us\n", run<C, 1>(cp, g));
    printf("  no halo loads                %6.1f us\n", run<C, 2>(cp, g));
    printf("  no epilogue                  %6.1f us\n", run<C, 4>(cp, g));
    printf("  no loads, no epilogue        %6.1f us\n", run<C, 7>(cp, g));
    printf("  no MFMA                      %6.1f us\n", run<C, 8>(cp, g));
    printf("  full, one workgroup per CU   %6.1f us (grid %d)\n", run<C, 0>(cp, grid_for(cus)), grid_for(cus));
    for (int skew : {4, 8, 16}) {
        cp.skew = skew;
        printf("  full, skew %2d x 1024 cycles  %6.1f us\n", skew, run<C, 0>(cp, g));
    }
    cp.skew = 0;
    {   // shader-clock trace of workgroups 0 and 256, wave 0 (RES = false: p.res is the trace buffer)
        long long *tb; CK(hipMalloc(&tb, 2 * 256 * 16)); CK(hipMemset(tb, 0, 2 * 256 * 16));
        PlaneConvParams ct = cp;
        ct.res = tb;
        hipLaunchKernelGGL((conv3x3_planes_kernel<C, false, 64>), dim3(g), dim3(kPlThreads), 0, 0, ct);
        hipLaunchKernelGGL((conv3x3_planes_kernel<C, false, 64>), dim3(g), dim3(kPlThreads), 0, 0, ct);
        CK(hipDeviceSynchronize());
        std::vector<long long> ht(2 * 256 * 2);
        CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
        for (int wg = 0; wg < 2; ++wg) {
            printf("  trace workgroup %d (tag:+cycles; 2 first loads issued 3 prologue done 10+tap chunk done 30 tile done 31 staged 32 stored 33 next halo in):\n   ", wg ? 256 : 0);
            for (int i = 1; i < 250 && ht[(wg * 256 + i) * 2] != 0; ++i)
                printf(" %lld:+%lld", ht[(wg * 256 + i) * 2], ht[(wg * 256 + i) * 2 + 1] - ht[(wg * 256 + i - 1) * 2 + 1]);
            printf("\n");
        }
        hipFree(tb);
    }
    hipFree(x); hipFree(y); hipFree(r); hipFree(wf); hipFree(bias); hipFree(post); hipFree(flag);
    return 0;
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    if (shape<64>("res1", 256, 45, 17, cus)) return 1;
    if (shape<128>("res2", 256, 23, 9, cus)) return 1;
    if (shape<256>("res3", 256, 12, 5, cus)) return 1;
    return 0;
}
