// tools/gemm_trace.hip -- shader-clock phase trace of the tiled implicit-GEMM kernel on the stride-2 convolutions
// conv3 (45x17x64 -> 23x9x128) and conv5 (23x9x128 -> 12x5x256) of a B=256 full-alignment batch (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/gemm_trace.hip -o /tmp/gemm_trace && /tmp/gemm_trace
// Workgroup 0 records the clock at: 10 start, 11 prologue done, per K chunk 1 global loads issued, 2 staging writes
// issued, 3 MFMAs issued, 4 barrier passed; 12 epilogue stores issued (gemm_mfma_kernel ABL bit 6).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../clair3_amd/csrc/c3_gemm.h"
using namespace c3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int BN>
static int run(const char *name, int B, int H, int W, int Cin, int Cout) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, M = B * Ho * Wo;
    float *x, *w, *bias, *zeros, *out; long long *tb;
    CK(hipMalloc(&x, (size_t)B * H * W * Cin * 4)); CK(hipMalloc(&w, (size_t)Cout * 9 * Cin * 4)); CK(hipMalloc(&bias, Cout * 4));
    CK(hipMalloc(&zeros, 256)); CK(hipMalloc(&out, (size_t)M * Cout * 4)); CK(hipMalloc(&tb, 4 * 256 * 16));
    CK(hipMemset(x, 0, (size_t)B * H * W * Cin * 4)); CK(hipMemset(w, 0, (size_t)Cout * 9 * Cin * 4)); CK(hipMemset(bias, 0, Cout * 4));
    CK(hipMemset(zeros, 0, 256)); CK(hipMemset(tb, 0, 4 * 256 * 16));
    ConvLoaderParams lp{x, zeros, H, W, Cin, Ho, Wo, 2, Cin / 32};
    GemmParams gp; gp.bt = w; gp.ldb = 9 * Cin; gp.M = M; gp.N = Cout; gp.nk = 9 * Cin / 32; gp.tiles_n = Cout / BN;
    gp.tiles = ((M + 127) / 128) * gp.tiles_n;
    EpilogueParams ep{out, bias, reinterpret_cast<const float *>(tb), Cout, 0};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_mfma_kernel<ConvLoader<4>, EPI_BIAS_RELU, 128, BN, 0>), dim3(gp.tiles), dim3(256), 0, 0, lp, gp, ep);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((gemm_mfma_kernel<ConvLoader<4>, EPI_BIAS_RELU, 128, BN, 0>), dim3(gp.tiles), dim3(256), 0, 0, lp, gp, ep);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipLaunchKernelGGL((gemm_mfma_kernel<ConvLoader<4>, EPI_BIAS_RELU, 128, BN, 64>), dim3(gp.tiles), dim3(256), 0, 0, lp, gp, ep);
    CK(hipDeviceSynchronize());
    std::vector<long long> ht(4 * 256 * 2);
    CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
    printf("== %s: M=%d N=%d K=%d, %d tiles of 128x%d, %.1f us per launch; MFMA time of one chunk = %d cycles\n", name, M, Cout, 9 * Cin, gp.tiles, BN,
           ms * 100.f, 64 * (BN / 64) * 2 * 16);
    for (int wv = 0; wv < 2; ++wv) {
        printf("  wave %d (tag:+cycles):", wv);
        for (int i = 1; i < 250 && ht[(wv * 256 + i) * 2] != 0; ++i)
            printf(" %lld:+%lld", ht[(wv * 256 + i) * 2], ht[(wv * 256 + i) * 2 + 1] - ht[(wv * 256 + i - 1) * 2 + 1]);
        printf("\n");
    }
    return 0;
}

int main() {
    if (run<128>("conv3 (128x128 tiles)", 256, 45, 17, 64, 128)) return 1;
    if (run<64>("conv5 (128x64 tiles)", 256, 23, 9, 128, 256)) return 1;
    return 0;
}
