// glds_probe.hip -- what gfx950's LDS-DMA loads do, as far as c3_dense.h relies on it:
//   (1) `buffer_load_dwordx4 ... offen lds`: lane L's 16 bytes land at LDS [M0 base + 16 L] (lane-linear), whatever its source offset;
//   (2) a lane whose buffer offset is out of range: does its LDS slot receive zeros, or keep its old bytes?
//   (3) several DMA instructions in flight, then s_waitcnt vmcnt(0) + s_barrier: other waves see the data.
// build: hipcc --offload-arch=gfx950 -O3 tools/glds_probe.hip -o /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((void __attribute__((address_space(3))) *)(p))

__global__ void probe(const void *g, unsigned nbytes, unsigned *out) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // poison
    for (int i = tid; i < 1024; i += 256) reinterpret_cast<unsigned *>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g), 0, nbytes, 0x00020000);
    // wave w fills LDS [1024 w, 1024 w + 1024): lane L fetches piece (L ^ 5) of chunk w; every 7th lane is out of range
    const unsigned off = (lane % 7 == 3) ? 0xffffff00u : (unsigned)(wave * 1024 + ((lane ^ 5) * 16));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + wave * 1024), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // every thread reads what ANOTHER wave wrote
    const int src = ((wave + 1) & 3) * 1024 + lane * 16;
    const u32x4 v = *reinterpret_cast<const u32x4 *>(lds + src);
    for (int e = 0; e < 4; ++e) out[(size_t)tid * 4 + e] = v[e];
}

int main() {
    const unsigned n = 4096;
    std::vector<unsigned> h(n / 4);
    for (unsigned i = 0; i < n / 4; ++i) h[i] = 0x10000u + i;
    void *g;
    unsigned *out;
    hipMalloc(&g, n);
    hipMalloc((void **)&out, 256 * 16);
    hipMemcpy(g, h.data(), n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, g, n, out);
    std::vector<unsigned> o(256 * 4);
    if (hipMemcpy(o.data(), out, 256 * 16, hipMemcpyDeviceToHost) != hipSuccess) return printf("copy failed\n"), 1;
    int ok_linear = 0, oob_zero = 0, oob_kept = 0, oob_other = 0, bad = 0;
    for (int tid = 0; tid < 256; ++tid) {
        const int wave = tid >> 6, lane = tid & 63, sw = (wave + 1) & 3;  // this thread read slot `lane` of wave sw's region
        const bool oob = lane % 7 == 3;
        const unsigned want0 = 0x10000u + (unsigned)(sw * 256 + ((lane ^ 5) * 4));
        const unsigned *v = &o[(size_t)tid * 4];
        if (oob) {
            if (v[0] == 0 && v[1] == 0 && v[2] == 0 && v[3] == 0) ++oob_zero;
            else if (v[0] == 0xdeadbeefu) ++oob_kept;
            else ++oob_other;
        } else if (v[0] == want0 && v[1] == want0 + 1 && v[2] == want0 + 2 && v[3] == want0 + 3) ++ok_linear;
        else ++bad;
    }
    printf("lane-linear destination, per-lane source: %d ok, %d wrong\n", ok_linear, bad);
    printf("out-of-range lanes: %d wrote zeros, %d left the old bytes, %d something else\n", oob_zero, oob_kept, oob_other);
    return bad ? 1 : 0;
}
