// Probe: do raw buffer loads of 8 bytes work at ANY byte offset on gfx950 (and what does a load that straddles num_records return)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const uint8_t *src, uint32_t nbytes, uint32_t base_shift, uint64_t *out, int n) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + base_shift, 0, nbytes - base_shift, 0x00020000);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (uint32_t)i, 0, 0));
    out[i] = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
}
int main() {
    const int N = 256;
    std::vector<uint8_t> h(N);
    for (int i = 0; i < N; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d;
    uint64_t *o;
    hipMalloc(&d, 4096);
    hipMemset(d, 0xEE, 4096);
    hipMalloc(&o, N * 8);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    for (uint32_t shift : {0u, 1u, 3u}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, (uint32_t)N, shift, o, N);
        std::vector<uint64_t> r(N);
        hipMemcpy(r.data(), o, N * 8, hipMemcpyDeviceToHost);
        int bad = 0, first_bad = -1;
        for (int i = 0; i + 8 + (int)shift <= N; ++i) {
            uint64_t e = 0;
            for (int j = 0; j < 8; ++j) e |= (uint64_t)h[shift + i + j] << (8 * j);
            if (r[i] != e) { if (first_bad < 0) first_bad = i; ++bad; }
        }
        printf("base shift %u: %d mismatches among fully in-range offsets (first %d)\n", shift, bad, first_bad);
        for (int i = N - (int)shift - 10; i < N - (int)shift + 2; ++i) {
            uint64_t e = 0;
            for (int j = 0; j < 8; ++j) e |= (uint64_t)((int)shift + i + j < N ? h[shift + i + j] : 0) << (8 * j);
            printf("  offset %d (in-range bytes %d): got %016llx  expected-with-zero-fill %016llx\n", i, N - (int)shift - i, (unsigned long long)r[i], (unsigned long long)e);
        }
    }
    return 0;
}
