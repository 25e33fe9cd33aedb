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

// Intra-wave interleave: K instructions of one class after every MFMA of a back-to-back stream (256-thread workgroups,
// one wave per SIMD).  CLASS: 0 none, 1 v_add_f32, 2 v_pk_add_f32, 3 ds_read_b128, 4 buffer_load_dwordx4 (L2 hits),
// 5 s_add_u32 (SALU), 6 ds_write_b64, 7 v_cndmask+v_add (address-select idiom)
template <int CLASS, int K>
__global__ __launch_bounds__(256) void intra(float *out, const float *src, int iters) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float a = tid * 1e-3f, b = blockIdx.x * 1e-3f;
    float x0 = a, x1 = b;
    f32x2 y0 = {a, b}, y1 = {b, a};
    f32x4 l = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
    unsigned sacc = blockIdx.x;
    f32x4 gq[16], lq[16];
    for (int i = 0; i < 16; ++i) gq[i] = f32x4{0.f, 0.f, 0.f, 0.f}, lq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1 << 20, 0x00020000);
    lds[tid] = a; lds[tid + 256] = b;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if constexpr (CLASS == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(x1));
                    if constexpr (CLASS == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y0) : "v"(y1));
                    if constexpr (CLASS == 3) { f32x4 t = *reinterpret_cast<f32x4 *>(&lds[(tid * 4 + k * 1024 + i * 64) & 8191]); l += t; }
                    if constexpr (CLASS == 4) g += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(tid * 16 + ((i * 4 + k) & 63) * 4096), 0, 0));
                    if constexpr (CLASS == 5) asm volatile("s_add_u32 %0, %0, 7" : "+s"(sacc));
                    if constexpr (CLASS == 6) *reinterpret_cast<f32x2 *>(&lds[(tid * 2 + k * 512 + i * 32) & 8191]) = y0;
                    // 8/9/10: loads whose results are only consumed after the 16 MFMAs of the iteration (issue cost alone)
                    if constexpr (CLASS == 8) gq[(u * 8 + i) & 15] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(tid * 16 + ((i * 4 + k) & 63) * 4096), 0, 0));
                    if constexpr (CLASS == 9) lq[(u * 8 + i) & 15] = *reinterpret_cast<f32x4 *>(&lds[(tid * 4 + k * 1024 + i * 64) & 8191]);
                    if constexpr (CLASS == 10) { f32x2 t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)(tid * 8 + ((i * 4 + k) & 63) * 4096), 0, 0)); gq[(u * 8 + i) & 15][0] = t2[0]; gq[(u * 8 + i) & 15][1] = t2[1]; }
                    if constexpr (CLASS == 7) { asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(x1), "v"(a) : "vcc"); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        if constexpr (CLASS == 8 || CLASS == 10) for (int i = 0; i < 16; ++i) g += gq[i];
        if constexpr (CLASS == 9) for (int i = 0; i < 16; ++i) l += lq[i];
    }
    float res = x0 + y0[0] + y0[1] + l[0] + l[1] + l[2] + l[3] + g[0] + g[1] + g[2] + g[3] + (float)sacc;
    for (int i = 0; i < 8; ++i) for (int v = 0; v < 16; ++v) res += acc[i][v];
    if (res == 12345.678f) out[blockIdx.x * 256 + tid] = res;
}
template <int CLASS, int K>
static void run_intra(const char *name, float *out, const float *src, float base) {
    const int iters = 2000;
    float t = time_it([&] { hipLaunchKernelGGL((intra<CLASS, K>), dim3(256), dim3(256), 0, 0, out, src, iters); });
    const double per = (t - base) * 1e-6 * 2.4e9 / (iters * 16.0 * (K ? K : 1));
    printf("intra-wave: %-28s x%d per MFMA: %8.1f us  (+%.1f cycles per inserted instruction; MFMA-only %.1f us)\n", name, K, t, K ? per : 0.0, base);
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
    {
        float base = time_it([&] { hipLaunchKernelGGL((intra<0, 0>), dim3(256), dim3(256), 0, 0, out, src, 2000); });
        run_intra<0, 0>("nothing", out, src, base);
        run_intra<1, 1>("v_add_f32", out, src, base);
        run_intra<1, 4>("v_add_f32", out, src, base);
        run_intra<1, 12>("v_add_f32", out, src, base);
        run_intra<2, 1>("v_pk_add_f32", out, src, base);
        run_intra<2, 4>("v_pk_add_f32", out, src, base);
        run_intra<3, 1>("ds_read_b128", out, src, base);
        run_intra<3, 2>("ds_read_b128", out, src, base);
        run_intra<4, 1>("buffer_load_dwordx4", out, src, base);
        run_intra<4, 2>("buffer_load_dwordx4", out, src, base);
        run_intra<5, 4>("s_add_u32", out, src, base);
        run_intra<5, 12>("s_add_u32", out, src, base);
        run_intra<6, 1>("ds_write_b64", out, src, base);
        run_intra<6, 2>("ds_write_b64", out, src, base);
        run_intra<7, 2>("v_cmp+v_cndmask", out, src, base);
        run_intra<8, 1>("buffer_load_dwordx4 (deferred use)", out, src, base);
        run_intra<10, 1>("buffer_load_dwordx2 (deferred use)", out, src, base);
        run_intra<9, 1>("ds_read_b128 (deferred use)", out, src, base);
    }
    run<0, 0, 0, 8>("VALU", out, src, 8);
    run<3, 0, 0, 8>("mix", out, src, 2);
    return 0;
}
