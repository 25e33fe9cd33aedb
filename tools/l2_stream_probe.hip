// tools/l2_stream_probe.hip -- how fast can every CU stream the SAME small weight tensor out of L2?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/l2_stream_probe.hip -o /tmp/l2p && /tmp/l2p
// One 512-thread workgroup per CU reads `chunks` 16 KB chunks (32 B per thread per chunk, as conv3x3_planes_kernel stages its
// weights), cycling through a tensor of S bytes shared by all workgroups, with D chunks in flight; the data is only XORed.
// Variants: shared tensor (L2 / MALL resident) vs a private region per workgroup (HBM stream), D = 1, 2, 3, 6, and a
// matrix-instruction stream issued between the loads (does a busy matrix pipe / throttled clock change the memory side?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int D, int MFMAS>
__global__ __launch_bounds__(512, 2) void stream_kernel(const char *w, size_t S, size_t priv_stride, int chunks, uint32_t *out) {
    const int tid = threadIdx.x;
    const char *base = w + (size_t)blockIdx.x * priv_stride + tid * 16;
    const int nper = (int)(S / 16384);
    u32x4 r[D][2];
    u32x4 acc = {0, 0, 0, 0};
    f32x16 c = {};
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const size_t o = (size_t)(d % nper) * 16384;
        r[d][0] = *reinterpret_cast<const u32x4 *>(base + o);
        r[d][1] = *reinterpret_cast<const u32x4 *>(base + o + 8192);
    }
    int next = D % nper;
    for (int cc = 0; cc < chunks; cc += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            acc ^= r[d][0] ^ r[d][1];  // waits for the oldest chunk
            const size_t o = (size_t)next * 16384;
            next = next + 1 == nper ? 0 : next + 1;
            r[d][0] = *reinterpret_cast<const u32x4 *>(base + o);
            r[d][1] = *reinterpret_cast<const u32x4 *>(base + o + 8192);
#pragma unroll
            for (int k = 0; k < MFMAS; ++k)
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, acc), __builtin_bit_cast(f16x8, acc), c, 0, 0, 0);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc ^= r[d][0] ^ r[d][1];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u || c[3] == 77.f) out[0] = 1;
}

template <int D, int MFMAS>
static float run(const char *w, size_t S, size_t priv, int chunks, uint32_t *out, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((stream_kernel<D, MFMAS>), dim3(grid), dim3(512), 0, 0, w, S, priv, chunks, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream_kernel<D, MFMAS>), dim3(grid), dim3(512), 0, 0, w, S, priv, chunks, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / 5 * 1e3f;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount;
    const size_t big = (size_t)grid * (4u << 20);
    char *w; uint32_t *out;
    CK(hipMalloc(&w, big)); CK(hipMalloc(&out, 256));
    CK(hipMemset(w, 1, big));
    const int chunks = 360;  // ~ the chunk count of ten res3 tiles
    printf("%d workgroups of 512 threads, %d chunks of 16 KB each (%.1f MB per launch)\n", grid, chunks, grid * chunks * 16384.0 / 1e6);
    for (size_t S : {(size_t)147456, (size_t)589824, (size_t)2359296}) {
        printf("== shared tensor of %zu KB (every workgroup reads the same bytes)\n", S >> 10);
        float t;
        t = run<1, 0>(w, S, 0, chunks, out, grid); printf("  D=1           %7.1f us  %6.2f TB/s\n", t, grid * chunks * 16384.0 / t / 1e6);
        t = run<2, 0>(w, S, 0, chunks, out, grid); printf("  D=2           %7.1f us  %6.2f TB/s\n", t, grid * chunks * 16384.0 / t / 1e6);
        t = run<3, 0>(w, S, 0, chunks, out, grid); printf("  D=3           %7.1f us  %6.2f TB/s\n", t, grid * chunks * 16384.0 / t / 1e6);
        t = run<6, 0>(w, S, 0, chunks, out, grid); printf("  D=6           %7.1f us  %6.2f TB/s\n", t, grid * chunks * 16384.0 / t / 1e6);
        t = run<3, 24>(w, S, 0, chunks, out, grid); printf("  D=3 + 24 MFMA %7.1f us  %6.2f TB/s  (matrix time alone: %d x 24 x 32 cycles x 2 waves/SIMD)\n", t, grid * chunks * 16384.0 / t / 1e6, chunks);
        t = run<6, 24>(w, S, 0, chunks, out, grid); printf("  D=6 + 24 MFMA %7.1f us  %6.2f TB/s\n", t, grid * chunks * 16384.0 / t / 1e6);
    }
    printf("== private 4 MB region per workgroup (HBM / MALL stream)\n");
    for (int rep = 0; rep < 1; ++rep) {
        float t;
        t = run<3, 0>(w, 4u << 20, 4u << 20, 256, out, grid); printf("  D=3           %7.1f us  %6.2f TB/s\n", t, grid * 256 * 16384.0 / t / 1e6);
        t = run<6, 0>(w, 4u << 20, 4u << 20, 256, out, grid); printf("  D=6           %7.1f us  %6.2f TB/s\n", t, grid * 256 * 16384.0 / t / 1e6);
        t = run<6, 24>(w, 4u << 20, 4u << 20, 256, out, grid); printf("  D=6 + 24 MFMA %7.1f us  %6.2f TB/s\n", t, grid * 256 * 16384.0 / t / 1e6);
    }
    return 0;
}
