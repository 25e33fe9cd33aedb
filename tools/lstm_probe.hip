// tools/lstm_probe.hip -- where a step of the pileup recurrences goes (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I clair3_amd/csrc tools/lstm_probe.hip -o /tmp/lstm_probe && /tmp/lstm_probe
// lstm_recurrent_kernel_v2<160> (LSTM2; half and full tiles) and lstm1_fused_kernel (LSTM1, int8 windows, planes out) on B = 1024
// windows x 33 steps of random data: time per launch and the shader-clock stamps of workgroup (0, 0) -- per step: 0 top of the step,
// 1 matrix instructions issued, 2 gate exchange done (LSTM2) / cell done (LSTM1), 3 cell done / barrier passed.  Numbers only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../clair3_amd/csrc/c3_lstm_fused.h"
using namespace c3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class K>
static float time_it(K launch, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
static void fill(std::vector<uint16_t> &v, float scale) {
    for (size_t i = 0; i < v.size(); ++i) {
        const uint32_t u = (uint32_t)(i * 2654435761u + 12345u) * 40503u;
        const _Float16 h = (_Float16)(((float)(u >> 8) / 16777216.f - 0.5f) * scale);
        memcpy(&v[i], &h, 2);
    }
}
static void show(const char *what, const unsigned long long *t, int T) {
    for (int w : {0, 7}) {
        printf("  %s wave %d, steps 2..6 and %d (cycles: matrix | exchange+barrier | cell | to next step):", what, w, T - 1);
        for (int s : {2, 3, 4, 5, 6, T - 1}) {
            const unsigned long long *p = t + (w * 64 + s) * 4, *n = t + (w * 64 + s + 1) * 4;
            printf("  %llu|%llu|%llu|%llu", p[1] - p[0], p[2] - p[1], p[3] - p[2], s + 1 < T ? n[0] - p[3] : 0ull);
        }
        printf("\n");
    }
    const unsigned long long *a = t + (0 * 64 + 1) * 4, *b = t + (0 * 64 + T - 1) * 4;
    printf("  %s: steps 1..%d of wave 0: %.0f cycles per step\n", what, T - 1, (double)(b[0] - a[0]) / (T - 2));
}
int main() {
    const int B = 1024, T = 33;
    // ---- LSTM2
    {
        float *gx, *hout; void *whh; unsigned long long *tr;
        CK(hipMalloc(&gx, (size_t)B * T * 1280 * 4)); CK(hipMalloc(&hout, (size_t)B * T * 320 * 4));
        CK(hipMalloc(&whh, (size_t)2 * 40 * 10 * 64 * 16)); CK(hipMalloc(&tr, 8 * 64 * 4 * 8)); CK(hipMemset(tr, 0, 8 * 64 * 4 * 8));
        std::vector<float> hg((size_t)B * T * 1280);
        for (size_t i = 0; i < hg.size(); ++i) hg[i] = ((float)((uint32_t)(i * 2654435761u) >> 8) / 16777216.f - 0.5f) * 2.f;
        CK(hipMemcpy(gx, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint16_t> hw((size_t)2 * 40 * 10 * 64 * 8); fill(hw, 0.3f);
        CK(hipMemcpy(whh, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        Lstm2Params lp{gx, (const float *)whh, hout, B, T, 1280};
        printf("== LSTM2 (H = 160), B = %d\n", B);
        printf("  half tiles (8 windows, 256 workgroups)   %6.1f us\n", time_it([&] { hipLaunchKernelGGL((lstm_recurrent_kernel_v2<160, true, 4>), dim3(B / 8, 2), dim3(512), 0, 0, lp); }));
        printf("  full tiles (16 windows, 128 workgroups)  %6.1f us\n", time_it([&] { hipLaunchKernelGGL((lstm_recurrent_kernel_v2<160, true, 0>), dim3(B / 16, 2), dim3(512), 0, 0, lp); }));
        lp.trace = tr;
        std::vector<unsigned long long> ht(8 * 64 * 4);
        hipLaunchKernelGGL((lstm_recurrent_kernel_v2<160, true, 12>), dim3(B / 8, 2), dim3(512), 0, 0, lp);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost));
        show("half tiles", ht.data(), T);
        hipLaunchKernelGGL((lstm_recurrent_kernel_v2<160, true, 8>), dim3(B / 16, 2), dim3(512), 0, 0, lp);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost));
        show("full tiles", ht.data(), T);
        hipFree(gx); hipFree(hout); hipFree(whh); hipFree(tr);
    }
    return 0;
}
