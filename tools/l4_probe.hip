// l4_probe.hip -- where the time of l4_stream_kernel (c3_l4.h) goes on the two L4 shapes: pileup 1024 x 128 x 10 560, full alignment
// 256 x 256 x 3 584.  Parts switched off (ABL bits), split factors, and a plain read of the activation matrix (what the memory system gives
// a kernel that does nothing else) -- each both back to back (the matrix stays wherever the previous launch left it) and behind a kernel
// that rewrites the matrix first (as LSTM2 does in the step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc tools/l4_probe.hip -o /tmp/l4_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
static std::string g_err;
static int fail(const char *, ...) { return -1; }
#include "c3_l4.h"
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

// every thread adds up 16-byte pieces of the matrix: rows of `rowf` floats, a workgroup walks `rows` rows x `cols` floats at column c0
__global__ __launch_bounds__(512) void read_kernel(const float *a, float *out, int64_t total4) {
    const int64_t stride = (int64_t)gridDim.x * 512;
    float4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < total4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(a)[i];
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(512) void write_kernel(float *a, int64_t total4, float v) {
    const int64_t stride = (int64_t)gridDim.x * 512;
    for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < total4; i += stride) reinterpret_cast<float4 *>(a)[i] = float4{v, v, v, v};
}

int main() {
    const int shapes[2][3] = {{1024, 128, 10560}, {256, 256, 3584}};
    const int splits[2][4] = {{15, 5, 11, 33}, {28, 14, 7, 56}};
    for (int si = 0; si < 2; ++si) {
        const int n = shapes[si][0], FC = shapes[si][1], K4 = shapes[si][2];
        float *da, *dpart;
        void *dw;
        CK(hipMalloc((void **)&da, (size_t)n * K4 * 4));
        CK(hipMalloc(&dw, (size_t)FC * K4 * 4));
        CK(hipMalloc((void **)&dpart, (size_t)66 * n * FC * 4));
        std::vector<float> ha((size_t)n * K4);
        uint32_t s = 777u;
        for (auto &x : ha) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 9) % 2001 - 1000) * 1e-3f; }
        CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint16_t> hw((size_t)FC * K4 * 2);
        for (auto &x : hw) { s = s * 1664525u + 1013904223u; x = (uint16_t)(((s >> 31) << 15) | ((6 + (s >> 8) % 7) << 10) | (s & 0x3ff)); }
        CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        const int64_t total4 = (int64_t)n * K4 / 4;
        printf("L4 %d windows x %d features x %d inputs: %.1f MB of activations, %.1f MB of weight pieces\n", n, FC, K4, n * (double)K4 * 4e-6, FC * (double)K4 * 4e-6);
        auto rep = [&](const char *name, float us) { printf("  %-76s %7.1f us  %6.2f TB/s of activations\n", name, us, n * (double)K4 * 4e-6 / us); };
        const float wr = time_us([&] { hipLaunchKernelGGL(write_kernel, dim3(1024), dim3(512), 0, 0, da, total4, 0.25f); }, 20);
        rep("rewriting the matrix (write_kernel alone)", wr);
        rep("plain read of the matrix, back to back", time_us([&] { hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(512), 0, 0, da, dpart, total4); }, 20));
        rep("plain read behind a rewrite (minus the rewrite)",
            time_us([&] { hipLaunchKernelGGL(write_kernel, dim3(1024), dim3(512), 0, 0, da, total4, 0.25f); hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(512), 0, 0, da, dpart, total4); }, 20) - wr);
        for (int k = 0; k < 4; ++k) {
            const int S = splits[si][k];
            if ((K4 / 64) % S) continue;
            L4Params lp{da, K4, dw, dpart, n, FC, K4 / 64, S, (n + kL4BM - 1) / kL4BM, FC / kL4BN};
            const int grid = lp.m_tiles * lp.n_tiles * S;
            char nm[128];
#define RUNL(abl, what)                                                                                                          \
    snprintf(nm, sizeof nm, "S = %d (%d workgroups, %d chunks each): %s", S, grid, lp.nk / S, what);                               \
    rep(nm, time_us([&] { hipLaunchKernelGGL((l4_stream_kernel<abl>), dim3(grid), dim3(kL4Threads), 0, 0, lp); }, 20))
            RUNL(0, "the kernel");
            if (k == 0) {
                RUNL(1, "- activation loads");
                RUNL(2, "- weight requests");
                RUNL(4, "- matrix instructions");
                RUNL(3, "- both load streams");
                snprintf(nm, sizeof nm, "S = %d: the kernel behind a rewrite of the matrix (minus the rewrite)", S);
                rep(nm, time_us([&] { hipLaunchKernelGGL(write_kernel, dim3(1024), dim3(512), 0, 0, da, total4, 0.25f); hipLaunchKernelGGL((l4_stream_kernel<0>), dim3(grid), dim3(kL4Threads), 0, 0, lp); }, 20) - wr);
            }
        }
        (void)hipFree(da), (void)hipFree(dw), (void)hipFree(dpart);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
