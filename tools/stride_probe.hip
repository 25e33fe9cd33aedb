// stride_probe.hip -- how fast can 43 MB of fp32 rows (1024 x 10560, the LSTM2 output L4 reads) be streamed once, as a function of
// the contiguous run a wave instruction covers per row?  Workgroup = 128 rows x one K split (as l4_stream_kernel); run = 128 B
// (8 lanes per row, 8 rows per instruction) ... 1024 B (64 lanes on one row).  Cold (buffer > caches flushed by a 1 GB memset) and warm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int LPR>  // lanes per row run: 8 -> 128 B, 16 -> 256 B, 32 -> 512 B, 64 -> 1024 B
__global__ __launch_bounds__(512) void stream(const float *a, int lda, int kfloats, float *out) {
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * 128, k0 = blockIdx.y * kfloats;
    constexpr int ROWS_PER_PASS = 512 / LPR;           // rows covered by the workgroup per pass
    constexpr int PASSES = 128 / ROWS_PER_PASS;         // passes to cover 128 rows for one run of LPR * 4 floats
    f32x4 acc = {0, 0, 0, 0};
    for (int k = 0; k < kfloats; k += LPR * 4) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = m0 + ps * ROWS_PER_PASS + tid / LPR;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(a + (size_t)row * lda + k0 + k + (tid % LPR) * 4);
            acc += v;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
int main() {
    const int M = 1024, K = 10560;
    float *a, *out, *junk;
    hipMalloc(&a, (size_t)M * K * 4); hipMalloc(&out, 256); hipMalloc(&junk, (size_t)1 << 30);
    hipMemset(a, 0, (size_t)M * K * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name, int S, bool cold) {
        const int kf = K / S;
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            if (cold) hipMemsetAsync(junk, r, (size_t)1 << 30, 0);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(M / 128, S), dim3(512), 0, 0, a, K, kf, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-26s S=%2d %s: %7.1f us  %6.2f TB/s\n", name, S, cold ? "cold" : "warm", best * 1e3f, (double)M * K * 4 / (best * 1e-3) / 1e12);
    };
    for (int S : {15, 30}) for (bool cold : {true, false}) {
        if (K / S % 32 == 0) run(stream<8>, "run 128 B (8 rows/instr)", S, cold);
        if (K / S % 64 == 0) run(stream<16>, "run 256 B", S, cold);
        if (K / S % 128 == 0) run(stream<32>, "run 512 B", S, cold);
        if (K / S % 256 == 0) run(stream<64>, "run 1024 B", S, cold);
    }
    // K splits whose slices are multiples of 256 floats do not exist for 10560; use S = 11 (960 floats) and S = 33 (320)
    for (int S : {11, 33}) for (bool cold : {true, false}) {
        run(stream<8>, "run 128 B (8 rows/instr)", S, cold);
        run(stream<16>, "run 256 B", S, cold);
        if (K / S % 128 == 0) run(stream<32>, "run 512 B", S, cold);
    }
    return 0;
}
