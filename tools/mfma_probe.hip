// tools/mfma_probe.hip -- ablation probe of the conv implicit-GEMM kernel (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Times the real kernel and variants with parts switched off on the res1a (M=195840,N=64,K=576) and
// res3a (M=15360,N=256,K=2304) shapes of the B=256 full-alignment batch, plus a pure-MFMA pace kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../clair3_amd/csrc/c3_gemm.h"
#include "../clair3_amd/csrc/c3_wino.h"
using namespace c3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void pure_mfma(float *out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int NB>
__global__ __launch_bounds__(512, 2) void pure_mfma16(float *out, const float *w, int iters) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float bw[NB];
    for (int i = 0; i < NB; ++i) bw[i] = w[threadIdx.x + 512 * i];   // distinct resident B registers
    float a = threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NB / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[u * NACC + i], acc[i], 0, 0, 0);
        asm volatile("" : "+v"(a));
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <class K>
static float time_it(K launch, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}

template <int BN, int ABL>
static float run_conv(const float *x, const float *zeros, const float *w, const float *bias, float *out, int B, int H, int W, int Cin, int Cout) {
    ConvLoaderParams lp{x, zeros, H, W, Cin, H, W, 1, Cin / 32};
    GemmParams gp; gp.bt = w; gp.ldb = 9 * Cin; gp.M = B * H * W; gp.N = Cout; gp.nk = 9 * Cin / 32; gp.tiles_n = Cout / BN;
    gp.tiles = ((gp.M + 127) / 128) * gp.tiles_n;
    EpilogueParams ep{out, bias, nullptr, Cout, 0};
    return time_it([&] { hipLaunchKernelGGL((gemm_mfma_kernel<ConvLoader<4>, EPI_BIAS_RELU, 128, BN, ABL>), dim3(gp.tiles), dim3(256), 0, 0, lp, gp, ep); });
}

template <int ABL>
static float run_wino(const float *x, const float *zeros, const float *v, const float *bias, float *out, int B, int H, int W, int Cin, int Cout) {
    WinoParams wp;
    wp.x = x, wp.zeros = zeros, wp.v = v, wp.bias = bias, wp.res = nullptr, wp.out = out;
    wp.B = B, wp.H = H, wp.W = W, wp.Cin = Cin, wp.Cout = Cout;
    wp.th = (H + 1) / 2, wp.tw = (W + 1) / 2, wp.P = B * wp.th * wp.tw;
    wp.tiles_n = Cout / kWinoNT, wp.tiles = ((wp.P + kWinoPT - 1) / kWinoPT) * wp.tiles_n;
    return time_it([&] { hipLaunchKernelGGL((wino_conv_kernel<false, ABL>), dim3(wp.tiles), dim3(256), 0, 0, wp); });
}

template <class K>
static void occ(const char *name, K kern, int threads) {
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, 0);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
    printf("occupancy API: %-28s blocks/CU=%d (%s) regs=%d lds=%zu scratch=%zu\n", name, nb, hipGetErrorString(e), fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
}

// census: how many workgroups of a given (LDS, VGPR-like) footprint are really co-resident on one CU
template <int LDS_BYTES>
__global__ __launch_bounds__(256, 2) void census_kernel(int *maxres, int *cur, int spin) {
    __shared__ char buf[LDS_BYTES];
    buf[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    unsigned cu = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));  // CU id bits [11:8], SE [15:13] (gfx9)
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int slot = (int)(((xcc & 0xf) << 8) | ((cu >> 8) & 0xff));
    if (threadIdx.x == 0) {
        int v = atomicAdd(&cur[slot], 1) + 1;
        atomicMax(&maxres[slot], v);
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(&cur[slot], 1);
    if (buf[(threadIdx.x * 7) & 255] == 77) maxres[4095] = 1;
}

int main() {
    const int B = 256;
    occ("wino_conv_kernel<false,0>", wino_conv_kernel<false, 0>, 256);
    occ("gemm conv 128x64", gemm_mfma_kernel<ConvLoader<4>, EPI_BIAS_RELU, 128, 64, 0>, 256);
    occ("gemm conv 128x128", gemm_mfma_kernel<ConvLoader<4>, EPI_BIAS_RELU, 128, 128, 0>, 256);
    {
        int *mx, *cur;
        hipMalloc(&mx, 4096 * 4); hipMalloc(&cur, 4096 * 4);
        auto run = [&](const char *name, auto kern) {
            hipMemset(mx, 0, 4096 * 4); hipMemset(cur, 0, 4096 * 4);
            hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, mx, cur, 200);
            hipDeviceSynchronize();
            std::vector<int> h(4096); hipMemcpy(h.data(), mx, 4096 * 4, hipMemcpyDeviceToHost);
            int hist[16] = {0}; for (int i = 0; i < 4095; ++i) if (h[i] > 0 && h[i] < 16) hist[h[i]]++;
            printf("census %-16s max co-resident WGs per (xcc,hw_id-cu) slot histogram:", name);
            for (int i = 1; i < 8; ++i) printf(" %d:%d", i, hist[i]);
            printf("\n");
        };
        run("LDS 66560", census_kernel<66560>);
        run("LDS 49152", census_kernel<49152>);
        run("LDS 65536", census_kernel<65536>);
        run("LDS 81920", census_kernel<81920>);
    }
    size_t act = (size_t)B * 45 * 17 * 64;
    float *x, *y, *w, *bias, *zeros;
    CK(hipMalloc(&x, act * 4)); CK(hipMalloc(&y, act * 4)); CK(hipMalloc(&w, (size_t)256 * 2304 * 4)); CK(hipMalloc(&bias, 1024)); CK(hipMalloc(&zeros, 256));
    std::vector<float> h(act); for (size_t i = 0; i < act; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    CK(hipMemcpy(x, h.data(), act * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, h.data(), (size_t)256 * 2304 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 1024)); CK(hipMemset(zeros, 0, 256));
    {
        float *o; CK(hipMalloc(&o, 4096 * 256 * 4));
        for (int wg : {256, 512, 768, 1024}) {
            int iters = 2000;
            float us2 = time_it([&] { hipLaunchKernelGGL(pure_mfma<2>, dim3(wg), dim3(256), 0, 0, o, iters); }, 3);
            float us4 = time_it([&] { hipLaunchKernelGGL(pure_mfma<4>, dim3(wg), dim3(256), 0, 0, o, iters / 2); }, 3);
            double fl = (double)wg * 4 * iters * 16 * 4096.0;  // waves * iters * 8*NACC(=2) MFMAs * 4096 flop
            printf("pure MFMA %4d WGs: 2 acc %.1f us %.1f TF | 4 acc %.1f us %.1f TF\n", wg, us2, fl / us2 / 1e6, us4, fl / us4 / 1e6);
        }
    }
    {
        float *o; CK(hipMalloc(&o, 4096 * 512 * 4));
        for (int wg : {128, 256}) {
            int iters = 400;
            float us = time_it([&] { hipLaunchKernelGGL((pure_mfma16<5, 200>), dim3(wg), dim3(512), 0, 0, o, x, iters); }, 3);
            double nm = (double)wg * 8 * iters * 200;  // MFMAs
            printf("pure MFMA16x16x4 %d WGs x 8 waves, 5 acc, 200 resident B regs: %.1f us  %.1f TF  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", wg, us,
                   nm * 2048 / us / 1e6, us * 2400.0 / (iters * 200.0 * 2));
            float us4 = time_it([&] { hipLaunchKernelGGL((pure_mfma16<4, 128>), dim3(wg), dim3(512), 0, 0, o, x, iters); }, 3);
            printf("pure MFMA16x16x4 %d WGs x 8 waves, 4 acc, 128 resident B regs: %.1f us  (%.1f cycles/MFMA/SIMD)\n", wg, us4, us4 * 2400.0 / (iters * 128.0 * 2));
        }
    }
    const char *names[] = {"full", "no-gload", "no-gload,no-ldswrite", "no-gload,no-ldswrite,no-barrier", "mfma+ldsread only(=7)", "all off (15)", "no barrier only (4)", "no frag reads only (8)"};
    {
        double fl = 2.0 * B * 45 * 17 * 64 * 576;
        float t[8];
        t[0] = run_conv<64, 0>(x, zeros, w, bias, y, B, 45, 17, 64, 64);
        t[1] = run_conv<64, 1>(x, zeros, w, bias, y, B, 45, 17, 64, 64);
        t[2] = run_conv<64, 3>(x, zeros, w, bias, y, B, 45, 17, 64, 64);
        t[3] = run_conv<64, 7>(x, zeros, w, bias, y, B, 45, 17, 64, 64);
        t[4] = t[3];
        t[5] = run_conv<64, 15>(x, zeros, w, bias, y, B, 45, 17, 64, 64);
        t[6] = run_conv<64, 4>(x, zeros, w, bias, y, B, 45, 17, 64, 64);
        t[7] = run_conv<64, 8>(x, zeros, w, bias, y, B, 45, 17, 64, 64);
        for (int i = 0; i < 8; ++i) printf("res1a 128x64  %-34s %.1f us %.1f TF\n", names[i], t[i], fl / t[i] / 1e6);
    }
    {
        double fl = 2.0 * B * 12 * 5 * 256 * 2304;
        float t[8];
        t[0] = run_conv<128, 0>(x, zeros, w, bias, y, B, 12, 5, 256, 256);
        t[1] = run_conv<128, 1>(x, zeros, w, bias, y, B, 12, 5, 256, 256);
        t[2] = run_conv<128, 3>(x, zeros, w, bias, y, B, 12, 5, 256, 256);
        t[3] = run_conv<128, 7>(x, zeros, w, bias, y, B, 12, 5, 256, 256);
        t[4] = t[3];
        t[5] = run_conv<128, 15>(x, zeros, w, bias, y, B, 12, 5, 256, 256);
        t[6] = run_conv<128, 4>(x, zeros, w, bias, y, B, 12, 5, 256, 256);
        t[7] = run_conv<128, 8>(x, zeros, w, bias, y, B, 12, 5, 256, 256);
        for (int i = 0; i < 8; ++i) printf("res3a 128x128 %-34s %.1f us %.1f TF\n", names[i], t[i], fl / t[i] / 1e6);
        float t64 = run_conv<64, 0>(x, zeros, w, bias, y, B, 12, 5, 256, 256);
        printf("res3a 128x64  full %.1f us %.1f TF\n", t64, fl / t64 / 1e6);
    }
    {
        const char *wn[] = {"full", "no patch loads(1)", "no patch loads,no xform/LDS wr(3)", "no V loads(4)", "no epilogue(8)", "no MFMA(16)",
                            "no loads,xform,V (7)", "only MFMA+frag reads(15)", "only loads+xform (4|8|16)"};
        struct { int H, W, C; const char *name; } shp[] = {{45, 17, 64, "res1"}, {23, 9, 128, "res2"}, {12, 5, 256, "res3"}};
        for (auto &sh : shp) {
            double fl = 2.0 * B * sh.H * sh.W * sh.C * 9.0 * sh.C;
            float t[9];
            t[0] = run_wino<0>(x, zeros, w, bias, y, B, sh.H, sh.W, sh.C, sh.C);
            t[1] = run_wino<1>(x, zeros, w, bias, y, B, sh.H, sh.W, sh.C, sh.C);
            t[2] = run_wino<3>(x, zeros, w, bias, y, B, sh.H, sh.W, sh.C, sh.C);
            t[3] = run_wino<4>(x, zeros, w, bias, y, B, sh.H, sh.W, sh.C, sh.C);
            t[4] = run_wino<8>(x, zeros, w, bias, y, B, sh.H, sh.W, sh.C, sh.C);
            t[5] = run_wino<16>(x, zeros, w, bias, y, B, sh.H, sh.W, sh.C, sh.C);
            t[6] = run_wino<7>(x, zeros, w, bias, y, B, sh.H, sh.W, sh.C, sh.C);
            t[7] = run_wino<15>(x, zeros, w, bias, y, B, sh.H, sh.W, sh.C, sh.C);
            t[8] = run_wino<28>(x, zeros, w, bias, y, B, sh.H, sh.W, sh.C, sh.C);
            for (int i = 0; i < 9; ++i) printf("wino %s %-36s %.1f us  (direct-equivalent %.1f TF)\n", sh.name, wn[i], t[i], fl / t[i] / 1e6);
        }
    }
    return 0;
}
