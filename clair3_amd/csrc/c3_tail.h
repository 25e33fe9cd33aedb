// c3_tail.h -- the fully connected tail of both networks on the matrix cores.
//
//   x    = selu(L4 a + b4)                       (the split-K partials of the L4 GEMM, summed in a fixed order)
//   h_b  = selu(L5_b x + b5_b)                   b = 0..NB-1   (clair3/model.py:137-150, 392-405)
//   y_b  = softmax(selu(head_b h_b + bh_b))      heads 21 / 3 / 33 / 33, written concatenated (the predict=True layout)
//
// fc_tail_kernel (c3_kernels.h) did all of it with scalar FMAs, two windows per workgroup: every thread walked the
// 512 KB of L5 weights through 16 dependent rounds of loads and summed its split-K partials four at a time -- 28 us
// for 256 windows, almost all of it load latency.  Here:
//   splitk_reduce_selu_kernel  one thread per (window, feature): all S partials requested at once, added in the
//                              fixed order s = 0..S-1 (a window's bits never depend on its batch), + bias, SELU;
//   fc_tail_mfma_kernel        one workgroup per (16 windows, branch b): L5_b as 16 x FC x 128 on
//                              v_mfma_f32_16x16x4_f32 (wave w owns 32 of the 128 columns; its weight fragments are ONE
//                              batch of 16-byte loads issued before the activations arrive), SELU, the head as
//                              16 x 128 x 48 on three waves, SELU, and a four-lanes-per-window soft-max.
// Rows of one window never meet rows of another: probabilities are bit-identical whatever batch a window travels in.
#pragma once
#include "c3_kernels.h"

namespace c3 {

struct ReduceParams {
    const float *part;  // [S][n][FC]
    const float *bias;  // [FC]
    float *out;         // [n][FC]
    int n, FC, S;
    const float *pre = nullptr, *post = nullptr;  // [FC]: the partials of feature k carry the power of two pre[k] (SPLIT weights,
                                                  // c3_pack.h row_scales): the bias joins the sum times pre[k], the result leaves
                                                  // times post[k] = 1 / pre[k] -- bit-identical to the unscaled sum; nullptr = 1
};

__global__ __launch_bounds__(256) void splitk_reduce_selu_kernel(ReduceParams p) {
    const int64_t total = (int64_t)p.n * p.FC;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % p.FC);
    const float *src = p.part + i;
    float v = p.pre ? p.bias[k] * p.pre[k] : p.bias[k];
    int s = 0;
    for (; s < p.S; s += 8) {  // 8 independent loads in flight (the last batch clamps its surplus), summed in order
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)(s + u < p.S ? s + u : p.S - 1) * total];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s + u < p.S) v += t[u];
    }
    p.out[i] = selu_f(p.post ? v * p.post[k] : v);
}

struct Tail2Params {
    const float *x;    // [B][FC]   selu(L4)
    const float *w5f;  // [NB][4 waves][2 cb][FC/16][64 lanes][4]: L5_b[col = 32 wave + 16 cb + (lane&15)][k = 16 q + 4 (lane>>4) + e]
    const float *b5;   // [NB][128]
    const float *whf;  // [NB][3 cb][8][64 lanes][4]: head_b[out = 16 cb + (lane&15)][k = 16 q + 4 (lane>>4) + e], 0 beyond the head
    const float *bh;   // [NB][48]
    float *y;          // [B][ldy]: the row's probabilities first (ldy > nout when decoder columns follow, c3_decode.h)
    int B, NB, ldy;
    // the split-K sum inside this kernel (then x is not read): x[b][k] = selu((bias4[k] pre[k] + sum_s part[s][b][k]) post[k]), the
    // partials added in the fixed order s = 0..S-1 -- splitk_reduce_selu_kernel's arithmetic, bit for bit -- by every workgroup
    // for its own 16 windows (the NB branch workgroups of a window group repeat it: S x 16 x FC floats each, a few hundred KB out
    // of L2); branch 0 also leaves x in l4out for c3_debug_fetch
    const float *part = nullptr;  // [S][B][FC]
    const float *bias4 = nullptr, *pre = nullptr, *post = nullptr;
    float *l4out = nullptr;
    int S = 0;
};

template <int FC>
__global__ __launch_bounds__(256) void fc_tail_mfma_kernel(Tail2Params p) {
    constexpr int NQ = FC / 16, LDX = FC + 4, LDH = 128 + 4;
    __shared__ __attribute__((aligned(16))) float xs[16][LDX];
    __shared__ __attribute__((aligned(16))) float h5[16][LDH];
    __shared__ float lg[16][48];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, s = lane >> 4;
    const int b0 = blockIdx.x * 16, br = blockIdx.y;
    // label_shape = 21, 3, 33, 33 (shared/param_p.py:37)
    const int head_n = br == 0 ? 21 : br == 1 ? 3 : 33;
    const int head_off = br == 0 ? 0 : br == 1 ? 21 : br == 2 ? 24 : 57;

    // this wave's L5 fragments: requested first, they land while the activations are staged
    f32x4v wf[2][NQ];
    {
        const f32x4v *w = reinterpret_cast<const f32x4v *>(p.w5f) + ((int64_t)(br * 4 + wave) * 2 * NQ) * 64 + lane;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < NQ; ++q) wf[cb][q] = w[(cb * NQ + q) * 64];
    }
    // the head's fragments and bias too (waves 0..2): nothing below has to wait for a load it could have had already
    f32x4v hf[8];
    float hb = 0.f;
    if (wave < 3) {
        const f32x4v *w = reinterpret_cast<const f32x4v *>(p.whf) + ((int64_t)(br * 3 + wave) * 8) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) hf[q] = w[q * 64];
        hb = p.bh[br * 48 + wave * 16 + col];
    }
    // the 16 activation rows -> LDS (rows beyond B repeat the last window; their results are never written)
    if (p.part) {
        const int64_t total = (int64_t)p.B * FC;
#pragma unroll
        for (int i = 0; i < 16 * FC / 4 / 256; ++i) {
            const int idx = tid + 256 * i;
            const int t = idx / (FC / 4), c4 = idx - t * (FC / 4);
            const int b = b0 + t < p.B ? b0 + t : p.B - 1;
            const float *src = p.part + (int64_t)b * FC + 4 * c4;
            const f32x4v b4 = *reinterpret_cast<const f32x4v *>(p.bias4 + 4 * c4);
            const f32x4v pre = p.pre ? *reinterpret_cast<const f32x4v *>(p.pre + 4 * c4) : f32x4v{1.f, 1.f, 1.f, 1.f};
            const f32x4v post = p.post ? *reinterpret_cast<const f32x4v *>(p.post + 4 * c4) : f32x4v{1.f, 1.f, 1.f, 1.f};
            f32x4v v = p.pre ? b4 * pre : b4;
            for (int s0 = 0; s0 < p.S; s0 += 8) {  // 8 independent loads in flight (the last batch clamps its surplus), summed in order
                f32x4v tt[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) tt[u] = *reinterpret_cast<const f32x4v *>(src + (int64_t)(s0 + u < p.S ? s0 + u : p.S - 1) * total);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (s0 + u < p.S) v += tt[u];
            }
            f32x4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = selu_f(p.post ? v[e] * post[e] : v[e]);
            *reinterpret_cast<f32x4v *>(&xs[t][4 * c4]) = o;
            if (br == 0 && b0 + t < p.B && p.l4out) *reinterpret_cast<f32x4v *>(p.l4out + (int64_t)b * FC + 4 * c4) = o;
        }
    } else {
#pragma unroll
    for (int i = 0; i < 16 * FC / 4 / 256; ++i) {
        const int idx = tid + 256 * i;
        const int t = idx / (FC / 4), c4 = idx - t * (FC / 4);
        const int b = b0 + t < p.B ? b0 + t : p.B - 1;
        *reinterpret_cast<f32x4v *>(&xs[t][4 * c4]) = *reinterpret_cast<const f32x4v *>(p.x + (int64_t)b * FC + 4 * c4);
    }
    }
    f32x4v acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const float b = p.b5[br * 128 + wave * 32 + cb * 16 + col];
        acc[cb] = f32x4v{b, b, b, b};
    }
    __syncthreads();
    // L5_b: acc[cb][v] = window 4s+v, column 32 wave + 16 cb + col
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const f32x4v a = *reinterpret_cast<const f32x4v *>(&xs[col][16 * q + 4 * s]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wf[cb][q][e], acc[cb], 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) h5[4 * s + v][wave * 32 + cb * 16 + col] = selu_f(acc[cb][v]);
    __syncthreads();

    // head_b on waves 0..2: 16 outputs each (48 >= 33), K = 128
    if (wave < 3) {
        f32x4v ah = f32x4v{hb, hb, hb, hb};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4v a = *reinterpret_cast<const f32x4v *>(&h5[col][16 * q + 4 * s]);
#pragma unroll
            for (int e = 0; e < 4; ++e) ah = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], hf[q][e], ah, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) lg[4 * s + v][wave * 16 + col] = selu_f(ah[v]);
    }
    __syncthreads();

    // soft-max: wave 0, four lanes per window, classes part, part+4, ...
    if (wave == 0) {
        const int t = lane >> 2, part = lane & 3;
        float l[9];
        float m = -3.0e38f;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int c = part + 4 * i;
            l[i] = c < head_n ? lg[t][c] : -3.0e38f;
            m = fmaxf(m, l[i]);
        }
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            l[i] = part + 4 * i < head_n ? expf(l[i] - m) : 0.f;
            sum += l[i];
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const int b = b0 + t;
        if (b < p.B) {
            float *y = p.y + (int64_t)b * p.ldy + head_off;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (part + 4 * i < head_n) y[part + 4 * i] = l[i] / sum;
        }
    }
}

}  // namespace c3
