// tools/coissue_probe.hip -- do a partner wave's VALU / LDS / VMEM instructions issue under fp32 MFMAs on one SIMD?
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/coissue_probe.hip -o /tmp/coissue_probe && /tmp/coissue_probe
// 512-thread workgroups, one per CU: waves 0-3 (older) and 4-7 (younger) sit pairwise on the 4 SIMDs.  One half issues
// v_mfma_f32_32x32x2_f32 back to back (8 independent accumulators), the other half a stream of VALU / LDS / buffer-load
// instructions.  Reported: time of each half alone and of both together (ideal overlap = max, none = sum).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PART: 0 VALU (v_pk_add_f32), 1 LDS write+read, 2 buffer loads (L2 hits), 3 mix of the three
// ACC: 0 accumulators in arch VGPRs (builtin), 1 in AGPRs (inline asm)
template <int PART, int ACC, int NOPS = 0, int NACC = 8>
__global__ __launch_bounds__(512) void coissue(float *out, const float *src, int iters, int mode, int swap, int prio, int nwork) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane((wave >> 2) ^ swap);  // 0: MFMA half, 1: partner half
    float res = 0.f;
    if (role == 0) {
        if (!(mode & 1)) return;
        if (prio & 1) __builtin_amdgcn_s_setprio(1);
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
        float a = tid * 1e-3f, b = blockIdx.x * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if constexpr (ACC == 0) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i % NACC], 0, 0, 0);
                    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                    if constexpr (NOPS >= 16) asm volatile("s_nop 15");
                    if constexpr (NOPS >= 32) asm volatile("s_nop 15");
                    if constexpr (NOPS >= 48) asm volatile("s_nop 15");
                    if constexpr (NOPS % 16) asm volatile("s_nop %0" ::"i"(NOPS % 16 - 1));
                }
        }
        if constexpr (ACC == 1) asm volatile("s_nop 15\n\ts_nop 15");
        for (int i = 0; i < 8; ++i) for (int v = 0; v < 16; ++v) res += acc[i][v];
    } else {
        if (!(mode & 2)) return;
        if (prio & 2) __builtin_amdgcn_s_setprio(1);
        f32x2 r[8];
        for (int i = 0; i < 8; ++i) r[i] = f32x2{tid * 1.f, i * 1.f};
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1 << 20, 0x00020000);
        for (int it = 0; it < iters; ++it) {
            for (int k = 0; k < nwork; ++k) {  // one unit = 8 instructions
                if constexpr (PART == 0 || PART == 3) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) r[i] = r[i] + r[(i + 1) & 7];
                }
                if constexpr (PART == 1 || PART == 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x2 *>(&lds[(tid * 2 + i * 1024) & 8191]) = r[i];
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i + 4] += *reinterpret_cast<f32x2 *>(&lds[(tid * 2 + i * 1024 + 2048) & 8191]);
                }
                if constexpr (PART == 2 || PART == 3) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        r[i] += __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)(tid * 8 + i * 4096 + ((it * 8 + k) & 15) * 32768), 0, 0));
                }
            }
        }
        for (int i = 0; i < 8; ++i) res += r[i][0] + r[i][1];
    }
    if (res == 12345.678f) out[blockIdx.x * 512 + tid] = res;
}

template <class K>
static float time_it(K launch, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}

template <int PART, int ACC, int NOPS = 0, int NACC = 8>
static void run(const char *name, float *out, const float *src, int nwork) {
    const int iters = 2000;
    printf("-- nops after each MFMA: %d cycles, accumulators: %d\n", NOPS, NACC);
    for (int swap = 0; swap < 2; ++swap)
        for (int prio = 0; prio < 1; prio += 2) {
            float t[4];
            for (int mode = 1; mode <= 3; ++mode)
                t[mode] = time_it([&] { hipLaunchKernelGGL((coissue<PART, ACC, NOPS, NACC>), dim3(256), dim3(512), 0, 0, out, src, iters, mode, swap, prio, nwork); });
            printf("%-10s acc=%s nwork=%2d MFMA half=%s prio(mfma=%d partner=%d): mfma alone %7.1f us | partner alone %7.1f us | both %7.1f us  (max %.1f, sum %.1f)\n",
                   name, ACC ? "AGPR" : "VGPR", nwork, swap ? "younger" : "older  ", prio & 1, (prio >> 1) & 1, t[1], t[2], t[3], t[1] > t[2] ? t[1] : t[2], t[1] + t[2]);
        }
}

int main() {
    float *out, *src;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMemset(src, 0, 1 << 20);
    run<0, 0, 0, 8>("VALU", out, src, 8);
    run<0, 0, 8, 8>("VALU", out, src, 8);
    run<0, 0, 10, 8>("VALU", out, src, 8);
    run<0, 0, 12, 8>("VALU", out, src, 8);
    run<0, 0, 13, 8>("VALU", out, src, 8);
    run<0, 0, 14, 8>("VALU", out, src, 8);
    run<0, 0, 15, 8>("VALU", out, src, 8);
    run<0, 0, 16, 8>("VALU", out, src, 8);
    run<3, 0, 12, 8>("mix", out, src, 2);
    run<3, 0, 14, 8>("mix", out, src, 2);
    run<3, 0, 15, 8>("mix", out, src, 2);
    return 0;
}
