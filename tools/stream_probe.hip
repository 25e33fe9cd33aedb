// stream_probe.hip -- what a back-to-back v_mfma_f32_32x32x2_f32 stream achieves per SIMD as a function of
// (a) dependent chains per wave, (b) distinct vs repeated operand registers, (c) waves per SIMD, (d) vector-memory
// instructions issued inside the stream (the proj_stream_kernel pattern: 32 x 16-byte loads + 16 dword stores per
// 128 MFMAs).  Prints cycles per MFMA per SIMD at 2.4 GHz (64 = the pipe's limit).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MEM: 0 none, 1 loads, 2 stores, 3 both
template <int NACC, int NA, int NB, int MEM>
__global__ __launch_bounds__(256, 2) void stream(const float *src, float *out, int iters, unsigned long long *clk) {
    const int lane = threadIdx.x & 63;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float a[NA], b[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = src[lane + 64 * i];
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = src[lane + 64 * (i + NA)];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1 << 22, 0x00020000);
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1u << 30, 0x00020000);
    const uint32_t wo = (uint32_t)((blockIdx.x * 256 + threadIdx.x) * 4);
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    f32x4 ring[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ring[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 128; ++j) {
            acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j % NA], b[j % NB], acc[j % NACC], 0, 0, 0);
            if ((MEM & 1) && (j % 4) == 3) {
                ring[(j / 4) % 8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)lane * 16u, ((it * 32 + j / 4) & 1023) * 1024, 0));
            }
            if ((MEM & 2) && (j % 8) == 7) {
                const float val = acc[(j + 1) % NACC][j / 8];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), ws, wo + (uint32_t)((j / 8) * 1048576), 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[i][v];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += ring[i][0] + ring[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = __builtin_readcyclecounter() - c0, clk[1] = wall_clock64() - w0;
}

template <class F>
static float time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

template <int NACC, int NA, int NB, int MEM>
static void run(const char *name, const float *src, float *out) {
    const int iters = 400;
    static unsigned long long *clk = nullptr;
    if (!clk) hipHostMalloc(&clk, 16);
    for (int wg : {256, 512}) {
        const float us = time_us([&] { hipLaunchKernelGGL((stream<NACC, NA, NB, MEM>), dim3(wg), dim3(256), 0, 0, src, out, iters, clk); }, 3);
        const double mfma_per_simd = (double)wg / 256.0 * iters * 128;
        hipDeviceSynchronize();
        printf("%-44s %d waves/SIMD: %8.1f us  %.1f cycles per MFMA per SIMD at 2.4 GHz | wave 0: %.1f shader cycles per MFMA, shader clock %.0f MHz\n", name, wg / 256,
               us, us * 2400.0 / mfma_per_simd, (double)clk[0] / (iters * 128.0), (double)clk[0] / ((double)clk[1] / 100.0));
    }
}

int main() {
    float *src, *out;
    CK(hipMalloc(&src, 1 << 22));
    CK(hipMalloc(&out, 1u << 30));
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    CK(hipMemcpy(src, h.data(), 1 << 22, hipMemcpyHostToDevice));
    run<1, 1, 1, 0>("1 chain, same A/B registers", src, out);
    run<2, 1, 1, 0>("2 chains, same A/B registers", src, out);
    run<1, 128, 64, 0>("1 chain, 128 A x 64 B registers", src, out);
    run<2, 128, 64, 0>("2 chains, 128 A x 64 B registers", src, out);
    run<1, 128, 64, 1>("1 chain, distinct regs, + 32 b128 loads", src, out);
    run<1, 128, 64, 2>("1 chain, distinct regs, + 16 stores", src, out);
    run<1, 128, 64, 3>("1 chain, distinct regs, + loads + stores", src, out);
    run<2, 128, 64, 3>("2 chains, distinct regs, + loads + stores", src, out);
    return 0;
}
