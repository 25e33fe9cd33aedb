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

// bf16x6: one fp32-equivalent 32x32x16 block = six v_mfma_f32_32x32x16_bf16 (x0w0 x0w1 x1w0 x0w2 x1w1 x2w0).
// SPLIT 0: A pieces resident (pre-split); 1: the wave splits its 8 fp32 A values per block into 3 bf16 pieces
// (v_cvt_pk_bf16_f32 + shift/and + subtract, twice) inside the stream; NBLK = column blocks sharing one split A.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 (&p)[3]) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = x[i];
#pragma unroll
    for (int lvl = 0; lvl < 3; ++lvl) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const bf16x2 h = __builtin_convertvector(f32x2{r[i], r[i + 1]}, bf16x2);
            p[lvl][i] = h[0], p[lvl][i + 1] = h[1];
            if (lvl < 2) {
                const unsigned u = __builtin_bit_cast(unsigned, h);
                r[i] -= __uint_as_float(u << 16);
                r[i + 1] -= __uint_as_float(u & 0xffff0000u);
            }
        }
    }
}
template <int SPLIT, int NBLK>
__global__ __launch_bounds__(256, 2) void stream_bf16x6(const float *src, float *out, int iters, unsigned long long *clk) {
    const int lane = threadIdx.x & 63;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    bf16x8 wp[NBLK][3];
#pragma unroll
    for (int n = 0; n < NBLK; ++n)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) wp[n][q][e] = (__bf16)src[lane + 64 * (e + 8 * q + 24 * n)];
    float xa[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xa[e] = src[4096 + lane * 8 + e];
    bf16x8 ap[3];
    split3(xa, ap);
    f32x16 acc[NBLK];
#pragma unroll
    for (int n = 0; n < NBLK; ++n)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[n][v] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {  // 16 blocks of k = 16  (= 128 fp32 32x32x2 MFMAs per column block)
            if (SPLIT) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xa[e] = xa[e] * 1.0001f + acc[0][e & 3];  // fresh fp32 values every block
                split3(xa, ap);
            }
#pragma unroll
            for (int n = 0; n < NBLK; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], wp[n][0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], wp[n][1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], wp[n][2], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], wp[n][0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], wp[n][1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], wp[n][0], acc[n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < NBLK; ++n)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[n][v];
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

template <int SPLIT, int NBLK>
static void run6(const char *name, const float *src, float *out) {
    const int iters = 400;
    static unsigned long long *clk = nullptr;
    if (!clk) hipHostMalloc(&clk, 16);
    for (int wg : {256, 512}) {
        const float us = time_us([&] { hipLaunchKernelGGL((stream_bf16x6<SPLIT, NBLK>), dim3(wg), dim3(256), 0, 0, src, out, iters, clk); }, 3);
        const double blocks_per_simd = (double)wg / 256.0 * iters * 16 * NBLK;  // fp32-equivalent 32x32x16 blocks
        hipDeviceSynchronize();
        printf("%-44s %d waves/SIMD: %8.1f us  %.1f cycles per 32x32x16 block per SIMD at 2.4 GHz (fp32 MFMA: 512) | wave 0: %.1f shader cycles per block\n",
               name, wg / 256, us, us * 2400.0 / blocks_per_simd, (double)clk[0] / (iters * 16.0 * NBLK));
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
    run6<0, 2>("bf16x6, pieces resident, 2 column blocks", src, out);
    run6<1, 1>("bf16x6, A split in the stream, 1 column block", src, out);
    run6<1, 2>("bf16x6, A split in the stream, 2 column blocks", src, out);
    run6<1, 4>("bf16x6, A split in the stream, 4 column blocks", src, out);
    return 0;
}
