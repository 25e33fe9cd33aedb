// tools/duo_probe.hip -- conv3x3_duo_kernel (c3_conv3d.h: two workgroups per CU, weights straight into registers) against
// conv3x3_planes_kernel (c3_conv3.h) on the three residual-block shapes of a B = 256 full-alignment batch: bit-identical output
// planes, time per launch (whole and with parts switched off), start skew sweep, phase trace.  Run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I clair3_amd/csrc tools/duo_probe.hip -o /tmp/duo_probe && /tmp/duo_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../clair3_amd/csrc/c3_conv3d.h"
using namespace c3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class K>
static float time_it(K launch, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
template <int C, int ABL> static float run_old(const PlaneConvParams &cp, int grid) {
    return time_it([&] { hipLaunchKernelGGL((conv3x3_planes_kernel<C, true, ABL>), dim3(grid), dim3(kPlThreads), 0, 0, cp); });
}
template <int C, int ABL> static float run_duo(const PlaneConvParams &cp, int grid) {
    return time_it([&] { hipLaunchKernelGGL((conv3x3_duo_kernel<C, true, ABL>), dim3(grid), dim3(kDuThreads), 0, 0, cp); });
}
template <int C> static int shape(const char *name, int B, int H, int W, int cus) {
    constexpr int NS = C / 64;
    const size_t bytes = (size_t)B * H * W * C * 4;
    void *x, *y, *y2, *r, *w, *wf; float *bias, *post; uint32_t *flag;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&y2, bytes)); CK(hipMalloc(&r, bytes));
    CK(hipMalloc(&w, (size_t)9 * C * C * 4)); CK(hipMalloc(&wf, (size_t)9 * C * C * 4)); CK(hipMalloc(&bias, C * 4)); CK(hipMalloc(&post, C * 4)); CK(hipMalloc(&flag, 256));
    std::vector<_Float16> h(bytes / 2);
    for (size_t g = 0; g < h.size() / 128; ++g)
        for (int c = 0; c < 64; ++c) {
            const uint32_t u = (uint32_t)((g * 64 + c) * 2654435761u);
            const float xv = (u % 3 == 0) ? 0.f : (float)(u >> 8) / 16777216.f * 3.f;
            const _Float16 hi = (_Float16)xv;
            h[g * 128 + c] = hi, h[g * 128 + 64 + c] = (_Float16)(xv - (float)hi);
        }
    CK(hipMemcpy(x, h.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(r, h.data(), bytes, hipMemcpyHostToDevice));
    std::vector<uint16_t> hw((size_t)9 * C * C * 2), hf(hw.size());  // chunks [tn][slab][tap][64 couts][16 pieces][8]
    for (size_t g = 0; g < hw.size() / 128; ++g)
        for (int c = 0; c < 64; ++c) {
            const uint32_t u = (uint32_t)((g * 64 + c) * 40503u + 12345u) * 2654435761u;
            const float xv = ((float)(u >> 8) / 16777216.f - 0.5f) * 0.08f;
            const _Float16 hi = (_Float16)xv, lo = (_Float16)(xv - (float)hi);
            memcpy(&hw[g * 128 + c], &hi, 2), memcpy(&hw[g * 128 + 64 + c], &lo, 2);
        }
    for (int tn = 0; tn < NS; ++tn) for (int slab = 0; slab < NS; ++slab) for (int tap = 0; tap < 9; ++tap)
        for (int wn = 0; wn < 2; ++wn) for (int ks = 0; ks < 4; ++ks) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int n = 32 * wn + (lane & 31), g = 2 * ks + (lane >> 5);
            const size_t src = ((((((size_t)tn * NS + slab) * 9 + tap) * 64 + n) * 16 + g) * 8) + j;
            const size_t dst = (((((((size_t)tn * NS + slab) * 9 + tap) * 2 + wn) * 4 + ks) * 2) * 64 + lane) * 8 + j;
            hf[dst] = hw[src], hf[dst + 512] = hw[src + 64];
        }
    CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(wf, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hb(C), hp(C);
    for (int c = 0; c < C; ++c) hb[c] = 0.01f * (c % 7) - 0.02f, hp[c] = 1.0f;
    CK(hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(post, hp.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemset(flag, 0, 256));
    PlaneConvParams cp;
    cp.x = x, cp.w = w, cp.wf = wf, cp.bias = bias, cp.res = r, cp.out = y, cp.range_flag = flag, cp.post = post, cp.pre = post;
    cp.M = B * H * W, cp.H = H, cp.W = W;
    cp.mg_hw = (uint32_t)((1ull << 32) / (uint64_t)(H * W) + 1), cp.mg_w = (uint32_t)((1ull << 32) / (uint64_t)W + 1);
    // ---- the one-workgroup form
    const int tiles_old = (cp.M + kPlBM - 1) / kPlBM * NS;
    auto grid_old = [&](int s) { return tiles_old <= s ? tiles_old : s / (8 * NS) * (8 * NS); };
    cp.tiles = tiles_old;
    CK(hipMemset(y, 0, bytes));
    hipLaunchKernelGGL((conv3x3_planes_kernel<C, true, 0>), dim3(grid_old(cus)), dim3(kPlThreads), 0, 0, cp);
    CK(hipDeviceSynchronize());
    const float t_old = run_old<C, 0>(cp, grid_old(cus));
    // ---- duo
    PlaneConvParams cd = cp;
    cd.out = y2;
    const int tiles_duo = (cd.M + kDuBM - 1) / kDuBM * NS;
    cd.tiles = tiles_duo;
    auto grid_duo = [&](int s) { return tiles_duo <= s ? tiles_duo : s / (8 * NS) * (8 * NS); };
    const int g2 = grid_duo(2 * cus);
    CK(hipMemset(y2, 0, bytes));
    hipLaunchKernelGGL((conv3x3_duo_kernel<C, true, 0>), dim3(g2), dim3(kDuThreads), 0, 0, cd);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> a(bytes / 4), b(bytes / 4);
    CK(hipMemcpy(a.data(), y, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y2, bytes, hipMemcpyDeviceToHost));
    size_t diff = 0, nz = 0;
    for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i], nz += a[i] != 0;
    printf("== %s: M = %d; old %d tiles of 256 on grid %d, duo %d tiles of 128 on grid %d; outputs differing: %zu of %zu words (%zu non-zero)\n", name, cp.M,
           tiles_old, grid_old(cus), tiles_duo, g2, diff, a.size(), nz);
    printf("  one workgroup per CU (c3_conv3.h)    %6.1f us\n", t_old);
    for (int skew : {0, 2, 4, 6, 8, 12, 16, 24}) {
        cd.skew = skew;
        printf("  duo, skew %2d x 1024 cycles           %6.1f us\n", skew, run_duo<C, 0>(cd, g2));
    }
    cd.skew = 6;
    printf("  duo on %d workgroups (one per CU)     %6.1f us\n", grid_duo(cus), run_duo<C, 0>(cd, grid_duo(cus)));
    printf("  duo, no weight loads                 %6.1f us\n", run_duo<C, 1>(cd, g2));
    printf("  duo, no halo loads                   %6.1f us\n", run_duo<C, 2>(cd, g2));
    printf("  duo, no epilogue                     %6.1f us\n", run_duo<C, 4>(cd, g2));
    printf("  duo, no loads, no epilogue           %6.1f us\n", run_duo<C, 7>(cd, g2));
    printf("  duo, no MFMA                         %6.1f us\n", run_duo<C, 8>(cd, g2));
    {
        long long *tb; CK(hipMalloc(&tb, 2 * 256 * 16)); CK(hipMemset(tb, 0, 2 * 256 * 16));
        PlaneConvParams ct = cd;
        ct.res = tb;
        hipLaunchKernelGGL((conv3x3_duo_kernel<C, false, 64>), dim3(g2), dim3(kDuThreads), 0, 0, ct);
        hipLaunchKernelGGL((conv3x3_duo_kernel<C, false, 64>), dim3(g2), dim3(kDuThreads), 0, 0, ct);
        CK(hipDeviceSynchronize());
        std::vector<long long> ht(2 * 256 * 2);
        CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
        for (int wg = 0; wg < 2; ++wg) {
            printf("  trace workgroup %d (tag:+cycles; 2 loads issued (+ skew) 3 prologue done 10+tap chunk done 30 tile done 31 staged 32 stored 33 next halo in):\n   ", wg ? 256 : 0);
            for (int i = 1; i < 250 && ht[(wg * 256 + i) * 2] != 0; ++i)
                printf(" %lld:+%lld", ht[(wg * 256 + i) * 2], ht[(wg * 256 + i) * 2 + 1] - ht[(wg * 256 + i - 1) * 2 + 1]);
            printf("\n");
        }
        hipFree(tb);
    }
    hipFree(x); hipFree(y); hipFree(y2); hipFree(r); hipFree(w); hipFree(wf); hipFree(bias); hipFree(post); hipFree(flag);
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
