// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal INPUTS?  A = 2^-20 (subnormal in fp16), B = 1024: a flushing
// matrix core returns 0, a preserving one 16 * 2^-10.  Also prints what v_cvt_pkrtz / the conversion builtin make of
// 2^-20 (conversion may flush independently of the MFMA).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, unsigned short sub_bits) {
    f16x8 a, b;
    const _Float16 sub = __builtin_bit_cast(_Float16, sub_bits);
    for (int i = 0; i < 8; ++i) a[i] = sub, b[i] = (_Float16)1024.0f;
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        const float tiny = 9.5367431640625e-07f;  // 2^-20
        _Float16 c = (_Float16)tiny;
        out[1] = (float)c;
        out[2] = (float)(unsigned)__builtin_bit_cast(unsigned short, c);
    }
}
int main() {
    float *d, h[3];
    hipMalloc(&d, 12);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (unsigned short)0x0010);  // 16 * 2^-24 = 2^-20
    hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("mfma(subnormal 2^-20 x 1024, K=16): %g (preserved: %g, flushed: 0)\n", h[0], 16 * 1024.0 * 9.5367431640625e-07);
    printf("float 2^-20 -> fp16 -> float: %g (bits 0x%04x)\n", h[1], (unsigned)h[2]);
    return 0;
}
